#!/bin/bash
# Run on the GPU box from the repo root: the bench line, then the kernel-trace summary of the same bench command
# (--pmc passes are separate calls: tools/collect_pmc_r03.sh).  Output under gpurun_out/r03/<tag>.
set -o pipefail
TAG=${1:-s2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/$TAG
mkdir -p $O
python3 $R/bench.py > $O/bench.json.log 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_prof.json.log 2> $O/bench_prof.err || exit 1
find $O/bench_prof -name '*kernel_stats.csv' | head -3
