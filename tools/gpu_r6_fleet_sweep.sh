#!/bin/bash
# Round 6: config-5 fleet, ring depth K = 32 / 64 / 96 (and refill threads 512), alternating, three rounds
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
run() {
  timeout 600 python bench.py --gpus 1 --steps 4 --warmup 1 --no-side-modes --no-cpu-baseline --detail /dev/null "$@" 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$*', {k: (v['us'], v['frac']) for k, v in d['legs'].items() if k.startswith('config5')})" | tee -a "$OUT/exp_fleet_resweep2.txt"
}
for r in 1 2 3; do
run --fleet-ring 32
run --fleet-ring 64
run --fleet-ring 96
run --fleet-ring 64 --tunable win_threads=512
done
