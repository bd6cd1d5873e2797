#!/bin/bash
# Round 6: the driver's bench command (+ config presets when PRESETS=1)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail.json" > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"
echo "bench rc $? line length $(wc -c < "$OUT/bench_driver_cmd.json")"
cat "$OUT/bench_driver_cmd.json"
grep -v "^bench_detail\|amdgpu.ids" "$OUT/bench_driver_cmd.err" | tail -5
if [ "${PRESETS:-0}" = 1 ]; then
  for C in 3 4; do
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --config $C --no-cpu-baseline --legs none --detail "$OUT/bench_config${C}_detail.json" > "$OUT/bench_config$C.json" 2> "$OUT/bench_config$C.err"
    echo "config $C rc $?"; cat "$OUT/bench_config$C.json"; grep -v "^bench_detail\|amdgpu.ids" "$OUT/bench_config$C.err" | tail -5
  done
fi
