#!/bin/bash
# Round 5, fourth GPU call: whole H = 0 observation rows through a wave-private LDS tile (1-KB wave stores) against the per-lane
# stores (tools/bin/libmgx_notile.so) and against rows that never leave the chip (libmgx_rowsdiag.so); ring depth K = 32 / 40 / 48
# for the config-5 fleet; the GPU tests.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu4.log" 2>&1
tail -6 "$OUT/pytest_gpu4.log"
: > "$OUT/exp_rows_h0.txt"
for rep in 1 2; do
for V in base notile rowsdiag; do
  if [ $V = base ]; then LIBV="MGX_DUMMY=1"; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  echo "== $V (rep $rep)" >> "$OUT/exp_rows_h0.txt"
  env $LIBV timeout 200 python tools/exp_r5_env_host.py 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_rows_h0.txt"
done
done
cat "$OUT/exp_rows_h0.txt"
: > "$OUT/exp_fleet_K.txt"
for K in 32 40 48; do
  for CFG in "float32 columns" "float64 columns"; do
    timeout 200 python tools/exp_r4_fleet.py $K $CFG 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_fleet_K.txt"
  done
done
cat "$OUT/exp_fleet_K.txt"
