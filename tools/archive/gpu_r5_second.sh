#!/bin/bash
# Round 5, second GPU call: the GPU tests again, then the refill kernel with phase 1 on 256 / 512 / 1024 threads (MGX_WIN_THREADS) x the
# phase-2 unroll variants (tools/bin/libmgx_u4.so, _u8.so), alone and inside the config-5 fleet.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu2.log" 2>&1
tail -12 "$OUT/pytest_gpu2.log"
: > "$OUT/exp_refill_threads.txt"; : > "$OUT/exp_fleet_threads.txt"
for V in base u4 u8; do
  if [ $V = base ]; then LIBV="MGX_DUMMY=1"; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  [ $V != base ] && [ ! -f "$REPO/tools/bin/libmgx_$V.so" ] && continue
  for T in 256 512 1024; do
    echo "== $V threads=$T" >> "$OUT/exp_refill_threads.txt"
    env $LIBV MGX_WIN_THREADS=$T timeout 200 python tools/exp_r4_refill_alone.py 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_refill_threads.txt"
  done
done
cat "$OUT/exp_refill_threads.txt"
for V in base u4 u8; do
  if [ $V = base ]; then LIBV="MGX_DUMMY=1"; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  [ $V != base ] && [ ! -f "$REPO/tools/bin/libmgx_$V.so" ] && continue
  for T in 256 1024; do
    for CFG in "32 float32 columns" "32 float64 columns" "32 float32 rows"; do
      echo -n "$V threads=$T  " >> "$OUT/exp_fleet_threads.txt"
      env $LIBV MGX_WIN_THREADS=$T timeout 200 python tools/exp_r4_fleet.py $CFG 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_fleet_threads.txt"
    done
  done
done
cat "$OUT/exp_fleet_threads.txt"
