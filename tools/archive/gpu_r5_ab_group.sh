#!/bin/bash
# Round 5: the config-5 fleet legs with 16- vs 32-grid refill workgroups, alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
: > "$OUT/exp_fleet_group_ab.txt"
for i in 1 2 3; do
for g in 0 16; do
MGX_WIN_GROUP=$g timeout 600 python bench.py --gpus 1 --no-cpu-baseline --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MGX_WIN_GROUP=$g', {k: v for k, v in d['legs'].items() if k.startswith('config5') or k.startswith('general_gym')})" | tee -a "$OUT/exp_fleet_group_ab.txt"
done
done
