#!/bin/bash
# Dev sweep (GPU box): rows per item of the f16 ensemble forward (CMBPO_ENS_H3_RT: 0 = the cost model, 1 / 2 / 4 = 32 / 64 / 128 rows)
for B in ${1:-2000 5000 10000 20000 40000}; do for rt in 0 1 2 4; do
  echo "B=$B RT=$rt: $(CMBPO_ENS_H3_RT=$rt python tools/probe_h3.py $B 100 AntSafe-v2 2 2>/dev/null | tail -n 1)"
done; done
