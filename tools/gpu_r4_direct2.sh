#!/bin/bash
# Run ON THE GPU BOX: the direct row kernel (K = 0) with the reciprocal division against the rings, after its parity tests.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/exp_fleet_direct_rows_recip.txt
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_direct_rows.py tests/test_ring_layout.py -m gpu -x -q 2>&1 | tail -5 >> "$OUT"
for DT in float64 float32; do
for K in 0 32; do
  timeout 120 python "$REPO/tools/exp_r4_fleet.py" $K $DT 2>&1 | grep -v amdgpu.ids >> "$OUT"
done
done
cat "$OUT"
