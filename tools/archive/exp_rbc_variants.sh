#!/bin/bash
# GPU experiment: ring depth of the rule-based rollout kernel (build variants with -DMGX_RING_ROLLOUT=n into tools/bin/libmgx_r<n>.so)
run() { python bench.py --gpus 1 --mode rbc --steps 64 --warmup 16 --no-side-modes --no-cpu-baseline --hetero-steps 0 --shards ${SH:-1} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('   %.2f us per 64-step round  frac %.3f' % (r['avg_launch_us'], r['frac']))"; }
echo "ring 8 (shipped):"; run; run
echo "ring 16:"; MGX_LIB=$PWD/tools/bin/libmgx_r16.so run; MGX_LIB=$PWD/tools/bin/libmgx_r16.so run
echo "ring 8, 2 shards:"; SH=2 run
