#!/bin/bash
# Dev sweep (GPU box): the alive count up to which a rollout step takes the one-workgroup bookkeeping + look-ahead loop (CMBPO_BOOK_MAX).
#   SPECS="task B maxroll;..." BM="..." bash tools/sweep_book_max.sh
IFS=';' read -ra SP <<< "${SPECS:-AntSafe-v2 1500 35;AntSafe-v2 2500 35;AntSafe-v2 4000 35;HalfCheetahSafe-v2 4000 35}"
for spec in "${SP[@]}"; do set -- $spec
  for bm in ${BM:-512 1024 2048 4096}; do
    echo "$1 B=$2 BOOK_MAX=$bm: $(CMBPO_BOOK_MAX=$bm python bench.py --task $1 --branches $2 --maxroll $3 --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print("%.2f M steps/s  %.1f us/step" % (d["value"]/1e6, d["ms_per_step"]*1e3/d["config"]["sampler_steps_per_phase_rank0"]))')"
  done
done
