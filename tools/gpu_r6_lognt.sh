#!/bin/bash
# Round 6: non-temporal LOADS of the action stream in the fused kernels (-DMGX_ACT_NT=1) vs plain loads, alternating
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
for V in base actnt base actnt; do
  if [ $V = base ]; then LIBV=""; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  env $LIBV timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --legs fused_rich,rbc --hetero-steps 0 --no-cpu-baseline --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$V', 'headline', d['roofline']['frac'], {k: (v['us'], v['frac']) for k, v in d['legs'].items()})" | tee -a "$OUT/exp_act_nt_loads.txt"
done
