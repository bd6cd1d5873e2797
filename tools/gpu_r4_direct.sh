#!/bin/bash
# Run ON THE GPU BOX: step + whole observation row in one launch (K = 0) against the rings (K = 16 / 32), config-5 fleet.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/exp_fleet_direct_rows.txt
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for rep in 1 2; do
for DT in float64 float32; do
for K in 0 16 32; do
  timeout 120 python "$REPO/tools/exp_r4_fleet.py" $K $DT 2>&1 | grep -v amdgpu.ids >> "$OUT"
done
done
done
cat "$OUT"
