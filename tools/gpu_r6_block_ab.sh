#!/bin/bash
# Round 6: workgroup size of the single-step kernels (-DMGX_BLOCK=128 / 512 vs the shipped 256), alternating
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
for V in b128 b64 b128 b64; do
  if [ $V = base ]; then LIBV=""; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  [ $V != base ] && [ ! -f "$REPO/tools/bin/libmgx_$V.so" ] && continue
  env $LIBV timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --legs step,step_env,step_env_obs,step_full --hetero-steps 1024 --no-cpu-baseline --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$V', {k.replace('single_step_launches_', ''): v['us'] for k, v in d['legs'].items()})" | tee -a "$OUT/exp_block_size.txt"
done
