#!/bin/bash
# Run ON THE GPU BOX: column-major against row-major ring blocks, config-5 fleet, one box, interleaved.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/exp_fleet_column_major_rings.txt
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for rep in 1 2; do for DT in float64 float32; do for K in 16 32; do for LAY in rows columns; do
  timeout 120 python "$REPO/tools/exp_r4_fleet.py" $K $DT $LAY 2>&1 | grep -v amdgpu.ids >> "$OUT"
done; done; done; done
cat "$OUT"
