#!/bin/bash
# Round 5: the driver's bench command, twice on one box (the second run on a warm box), with the wall time of each.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
for j in a b; do
  SECONDS=0
  python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail_final_$j.json" > "$OUT/bench_driver_cmd_final_$j.json" 2> "$OUT/bench_driver_cmd_final_$j.err"
  echo "rc $? length $(wc -c < "$OUT/bench_driver_cmd_final_$j.json") wall $SECONDS s"
  cat "$OUT/bench_driver_cmd_final_$j.json"
done
