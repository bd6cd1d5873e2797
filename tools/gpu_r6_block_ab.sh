#!/bin/bash
# Round 6: A/B of a library variant (tools/bin/libmgx_$1.so) against the product build on the legs named in $2, alternating
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
VAR=$1; LEGS=${2:-none}; HET=${3:-0}; EXTRA=${4:-}
for V in base $VAR base $VAR base $VAR; do
  if [ $V = base ]; then LIBV=""; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  env $LIBV timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --legs $LEGS --hetero-steps $HET --no-cpu-baseline --detail /dev/null $EXTRA 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$V', 'headline', d['roofline']['frac'], {k.replace('single_step_launches_', ''): v['us'] for k, v in d['legs'].items()})" | tee -a "$OUT/exp_ab_$VAR.txt"
done
