#!/bin/bash
# GPU box, from the repo root: PMC passes over the policy kernels (each pass its own run; --pmc never combined with other traces)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02/pi_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/tools/probe_fvp_only.py 3400000 4 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -- python3 $R/tools/probe_fvp_only.py 3400000 4 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $O/p3 -- python3 $R/tools/probe_fvp_only.py 3400000 4 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/p4 -- python3 $R/tools/probe_fvp_only.py 3400000 4 > $O/p4.log 2>&1
ls -R $O | grep -c counter_collection
