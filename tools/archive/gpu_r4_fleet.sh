#!/bin/bash
# Run ON THE GPU BOX: the refill-scheduling variants of tools/exp_r4_fleet.py, one process each.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/exp_fleet_refill_occupancy.txt
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
run() { env "$@" timeout 120 python "$REPO/tools/exp_r4_fleet.py" $ARGS 2>&1 | grep -v amdgpu.ids >> "$OUT"; }
for K in 16 32 48; do
  ARGS="$K float64"
  run MGX_WIN_MIN_LDS=0
  run MGX_WIN_MIN_LDS=82944
  run MGX_WIN_MIN_LDS=0 MGX_PREFETCH_POOL=1
  run MGX_WIN_MIN_LDS=82944 MGX_PREFETCH_POOL=1
done
ARGS="24 float64"; run MGX_WIN_MIN_LDS=82944; run MGX_WIN_MIN_LDS=82944 MGX_PREFETCH_POOL=1
ARGS="64 float64"; run MGX_WIN_MIN_LDS=82944 MGX_PREFETCH_POOL=1
ARGS="16 float32"; run MGX_WIN_MIN_LDS=0; run MGX_WIN_MIN_LDS=82944; run MGX_WIN_MIN_LDS=82944 MGX_PREFETCH_POOL=1
ARGS="32 float32"; run MGX_WIN_MIN_LDS=82944; run MGX_WIN_MIN_LDS=82944 MGX_PREFETCH_POOL=1
cat "$OUT"
