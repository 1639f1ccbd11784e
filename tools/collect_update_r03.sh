#!/bin/bash
# GPU box, from the repo root: kernel trace of four full-batch constrained updates, then two PMC passes over the Fisher-vector product
# (each pass its own run; --pmc never combined with other trace domains than the kernel trace).  Output: gpurun_out/r03/update/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/update
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/probe_update.py 3400000 1 3 > $O/trace.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/tools/probe_fvp_only.py 3400000 4 > $O/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -- python3 $R/tools/probe_fvp_only.py 3400000 4 > $O/p2.log 2>&1 || exit 1
find $O -name '*kernel_stats.csv' -o -name '*counter_collection.csv' | head
