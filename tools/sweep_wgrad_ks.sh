#!/bin/bash
# Dev sweep (GPU box): the grid K split of the weight-gradient GEMMs (CMBPO_WGRAD_KS: the 512 x 512 layer, CMBPO_WGRAD_KS_NARROW:
# the first / last layer) against the time of a whole training step.   bash tools/sweep_wgrad_ks.sh [batch] "<ks list>" "<narrow list>"
B=${1:-2048}
for ks in ${2:-4 5 6 7 8}; do for kn in ${3:-6 8 10 12}; do
  echo "batch=$B KS=$ks KS_NARROW=$kn: $(CMBPO_WGRAD_KS=$ks CMBPO_WGRAD_KS_NARROW=$kn python tools/probe_train.py 7 37 512 30 MSPE $B 200 2>/dev/null | head -1 | sed 's/.*batch=[0-9]*: //')"
done; done
