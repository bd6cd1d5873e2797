#!/bin/bash
# Round 5, third GPU call: lock-step base rows without the dependent gather + H = 0 row inputs requested ahead of the step:
# the GPU tests (parity), the driver's bench command, the host cost of the Gym step, the config-5 fleet.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu3.log" 2>&1
tail -12 "$OUT/pytest_gpu3.log"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail3.json" > "$OUT/bench_driver_cmd3.json" 2> "$OUT/bench_driver_cmd3.err"
echo "bench rc $? line length $(wc -c < "$OUT/bench_driver_cmd3.json")"
cat "$OUT/bench_driver_cmd3.json"
timeout 300 python tools/exp_r5_env_host.py > "$OUT/exp_env_host3.txt" 2>&1
cat "$OUT/exp_env_host3.txt"
: > "$OUT/exp_fleet3.txt"
for CFG in "32 float32 columns" "32 float64 columns" "32 float32 rows" "32 float64 rows"; do
  timeout 200 python tools/exp_r4_fleet.py $CFG 2>&1 | grep -v amdgpu.ids >> "$OUT/exp_fleet3.txt"
done
cat "$OUT/exp_fleet3.txt"
