#!/bin/bash
# Run ON THE GPU BOX: does the length of the timed region change the config-5 figures of bench.py?  (256 timed fleet steps by default)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/exp_bench_hetero_length.txt
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
cd "$REPO"
for HS in 256 2048 256 2048; do
  timeout 600 python bench.py --no-cpu-baseline --no-side-modes --no-closed-loop --hetero-steps $HS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hetero_h24_gym_steps']
print('hetero-steps $HS:', '  '.join(f\"{k} {v['us_per_step']:.2f} us ({v['roofline']['frac']:.3f})\" for k,v in h.items() if isinstance(v,dict) and 'roofline' in v))" >> "$OUT"
done
timeout 120 python tools/exp_r4_fleet.py 32 float64 2>&1 | grep -v amdgpu.ids >> "$OUT"
timeout 120 python tools/exp_r4_fleet.py 32 float64 columns 2>&1 | grep -v amdgpu.ids >> "$OUT"
cat "$OUT"
