#!/bin/bash
# Round 5: column-major refill stores as 16-byte packs of adjacent grids per lane (MGX_WIN_PAIRS=1, default) vs a word per lane (0).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_multi_windows.py tests/test_ring_layout.py tests/test_fleet_stagger.py tests/test_true_shape.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
: > "$OUT/exp_refill_col_packs.txt"
for cfg in 0 1 0 1 1; do
export MGX_WIN_PAIRS=$cfg
timeout 600 python bench.py --gpus 1 --no-cpu-baseline --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('packs=$cfg', {k: v['us'] for k, v in d['legs'].items() if k.startswith('config5') or k.startswith('general_gym')})" | tee -a "$OUT/exp_refill_col_packs.txt"
done
for cfg in 0 1; do
export MGX_WIN_PAIRS=$cfg
echo "== packs=$cfg" | tee -a "$OUT/exp_refill_col_packs.txt"
timeout 600 python tools/exp_r5_multi_layout.py 2>&1 | grep columns | tee -a "$OUT/exp_refill_col_packs.txt"
done
