#!/bin/bash
# Round 6: bench.py --config 3 | 4 (BASELINE configs[3] / [4] at 125 000 grids per GPU) on one GPU
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
for C in 3 4; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --config $C --no-cpu-baseline --legs none --detail "$OUT/bench_config${C}_detail.json" > "$OUT/bench_config${C}.json" 2> "$OUT/bench_config${C}.err"
  echo "config $C rc $?"; cat "$OUT/bench_config${C}.json"; grep -v "^bench_detail\|amdgpu.ids" "$OUT/bench_config${C}.err" | tail -5
done
