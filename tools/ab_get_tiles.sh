#!/bin/bash
# Dev A/B (GPU box): the flatten of get() with 16-branch tiles only (-DFLAT_WIDE_TILES=0) against 64-branch tiles after short
# rollouts (and, with WAVES4=1, at four waves per SIMD), over the bench's shapes
# (tools/probe_get.py [B] [task] [maxroll] [mode] [dkl scale]).   bash tools/ab_get_tiles.sh [short]
VARS=("old:-DFLAT_WIDE_TILES=0" "new:" "old2:-DFLAT_WIDE_TILES=0" "new2:")
[ -n "$WAVES4" ] && VARS+=("waves4:-DFLAT_WAVES=4")
CFGS=("10000 HalfCheetahSafe-v2 35 schedule" "10000 HumanoidSafe-v2 15 schedule" "100000 AntSafe-v2 35 uncertainty 0.6")
[ "$1" != "short" ] && CFGS+=("100000 AntSafe-v2 35 schedule" "10000 HopperSafe-v2 15 schedule" "100000 AntSafe-v2 20 schedule")
for v in "${VARS[@]}"; do for cfg in "${CFGS[@]}"; do
  python tools/probe_variants.py rollout_state.hip tools/probe_get.py "$cfg" "$v" 2>&1 | grep -E "variant|steps taken|flatten [0-9]"
done; done
