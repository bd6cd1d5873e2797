#!/bin/bash
# Round 5: column-major refill stores as pairs of adjacent grids per lane (MGX_WIN_PAIRS) x grids per workgroup (MGX_WIN_GROUP).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
: > "$OUT/exp_refill_col_pairs_matrix.txt"
for cfg in "0 16" "1 16" "0 32" "1 32" "0 16" "1 32"; do
set -- $cfg
export MGX_WIN_PAIRS=$1 MGX_WIN_GROUP=$2
timeout 600 python bench.py --gpus 1 --no-cpu-baseline --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs=$1 group=$2', {k: v['us'] for k, v in d['legs'].items() if k.startswith('config5') or k.startswith('general_gym')})" | tee -a "$OUT/exp_refill_col_pairs_matrix.txt"
done
for cfg in "0 16" "1 16" "0 32" "1 32"; do
set -- $cfg
export MGX_WIN_PAIRS=$1 MGX_WIN_GROUP=$2
echo "== pairs=$1 group=$2" | tee -a "$OUT/exp_refill_col_pairs_matrix.txt"
timeout 600 python tools/exp_r5_multi_layout.py 2>&1 | grep columns | tee -a "$OUT/exp_refill_col_pairs_matrix.txt"
done
