#!/bin/bash
# Dev sweep (GPU box): the row count from which the critics run member after member (critic_big_kernel, CMBPO_CRITIC_BIG_MIN)
# instead of one wave per member with the next step's actor riding along (critic_pair_kernel)
for B in ${1:-8000 12000 16000 20000 30000 50000}; do for m in ${2:-4096 12288 24576 65536}; do
  echo "AntSafe-v2 B=$B BIG_MIN=$m: $(CMBPO_CRITIC_BIG_MIN=$m python bench.py --task AntSafe-v2 --branches $B --maxroll 35 --no-extras --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print("%.2f M steps/s  %.1f us/step" % (d["value"]/1e6, d["ms_per_step"]*1e3/d["config"]["sampler_steps_per_phase_rank0"]))')"
done; done
