#!/bin/bash
# Run ON THE GPU BOX: refill workgroups per CU (MGX_WIN_MIN_LDS: 0 = what the image needs, 82944 = one workgroup per CU), K = 32, current kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/exp_fleet_refill_occupancy2.txt
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for rep in 1 2; do for LAY in rows columns; do for M in 0 82944; do
  MGX_WIN_MIN_LDS=$M timeout 120 python "$REPO/tools/exp_r4_fleet.py" 32 float64 $LAY 2>&1 | grep -v amdgpu.ids >> "$OUT"
done; done; done
cat "$OUT"
