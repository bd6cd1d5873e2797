#!/bin/bash
# Run ON THE GPU BOX: where the direct row kernel (K = 0) spends its time -- variants with a part left out (wrong rows, timing only).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/exp_fleet_direct_rows_parts.txt
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
cd "$REPO"
for DT in float64; do
  echo -n "as built            " >> "$OUT"; timeout 120 python tools/exp_r4_fleet.py 0 $DT 2>&1 | grep -v amdgpu.ids >> "$OUT"
  for v in 1 2 4 3; do
    echo -n "MGX_EXP_ROWS=$v      " >> "$OUT"; MGX_LIB=$REPO/tools/bin/libmgx_rows$v.so timeout 120 python tools/exp_r4_fleet.py 0 $DT 2>&1 | grep -v amdgpu.ids >> "$OUT"
  done
done
cat "$OUT"
