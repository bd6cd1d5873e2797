#!/bin/bash
# Run ON THE GPU BOX: A/B on ONE box, interleaved twice.  usage: gpu_r4_fleet_ab.sh <variant .so under tools/bin> <out name>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$REPO/tools/bin/$1
OUT=$REPO/gpurun_out/r04/$2
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
run() { echo -n "$TAG " >> "$OUT"; env "$@" timeout 120 python "$REPO/tools/exp_r4_fleet.py" $ARGS 2>&1 | grep -v amdgpu.ids >> "$OUT"; }
for rep in 1 2; do
for K in ${KS:-16 32}; do
  ARGS="$K ${DT:-float64}"
  TAG="base   "; run MGX_DUMMY=1
  TAG="variant"; run MGX_LIB=$VAR
done
done
cat "$OUT"
