#!/bin/bash
# Round 5, first GPU call: the GPU tests, the driver's bench command (does the compact line parse?), the refill variants
# alone and inside the config-5 fleet, the host cost of the bound Gym step.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
python -c "from pymgrid_amd import _lib; print('csrc_hash', _lib.built_hash() or _lib.source_hash())" > "$OUT/csrc_hash.txt"
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
tail -15 "$OUT/pytest_gpu.log"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail.json" > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"
echo "bench rc $? line length $(wc -c < "$OUT/bench_driver_cmd.json")"
cat "$OUT/bench_driver_cmd.json"
timeout 300 python tools/exp_r5_env_host.py > "$OUT/exp_env_host.txt" 2>&1
cat "$OUT/exp_env_host.txt"
for V in base u8 u4 kj8 u8kj8 nostore noload; do
  if [ $V = base ]; then LIBV=""; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  [ $V != base ] && [ ! -f "$REPO/tools/bin/libmgx_$V.so" ] && continue
  echo "== $V" >> "$OUT/exp_refill_variants.txt"
  env $LIBV timeout 200 python tools/exp_r4_refill_alone.py 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_refill_variants.txt"
done
cat "$OUT/exp_refill_variants.txt"
for V in base u8 kj8 u8kj8; do
  if [ $V = base ]; then LIBV=""; else LIBV="MGX_LIB=$REPO/tools/bin/libmgx_$V.so"; fi
  [ $V != base ] && [ ! -f "$REPO/tools/bin/libmgx_$V.so" ] && continue
  for CFG in "32 float32 columns" "32 float32 rows" "32 float64 columns" "32 float64 rows"; do
    echo -n "$V  " >> "$OUT/exp_fleet_variants.txt"
    env $LIBV timeout 200 python tools/exp_r4_fleet.py $CFG 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_fleet_variants.txt"
  done
done
cat "$OUT/exp_fleet_variants.txt"
