#!/bin/bash
# Run on the GPU box from the repo root: PMC passes of the dominant kernel (separate passes: --pmc is never combined with
# other trace domains than the kernel trace).  Output under gpurun_out/r03/<tag>/pmc_*.
set -o pipefail
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/probe_h3.py 100000 4 AntSafe-v2 2 > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/probe_h3.py 100000 4 AntSafe-v2 2 > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_sq -- python3 $R/tools/probe_h3.py 100000 6 AntSafe-v2 2 > $O/pmc_sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2 -- python3 $R/tools/probe_h3.py 100000 6 AntSafe-v2 2 > $O/pmc_sq2.log 2>&1 || exit 1
find $O -name '*counter_collection.csv' | head
