#!/bin/bash
# Round 6: the GPU test suite (+ optionally the driver's bench command: BENCH=1)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider ${PYTEST_ARGS:-} > "$OUT/pytest_gpu.log" 2>&1
tail -40 "$OUT/pytest_gpu.log"
if [ "${BENCH:-0}" = 1 ]; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail.json" > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"
  echo "bench rc $? line length $(wc -c < "$OUT/bench_driver_cmd.json")"
  cat "$OUT/bench_driver_cmd.json"
fi
