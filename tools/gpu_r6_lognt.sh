#!/bin/bash
# Round 6: register-ring depth of the fused kernels (-DMGX_RING=8) vs the shipped 4, alternating, at 100 000 and 125 000 grids
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
for G in 100000; do for V in ring8 ring16 ring8 ring16; do
  if [ $V = base ]; then LIBV=""; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  env $LIBV timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --grids $G --legs fused_rich --hetero-steps 0 --no-cpu-baseline --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$V', $G, 'headline', d['roofline']['frac'], {k: (v['us'], v['frac']) for k, v in d['legs'].items()})" | tee -a "$OUT/exp_ring_depth16.txt"
done; done
