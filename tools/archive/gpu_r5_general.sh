#!/bin/bash
# Round 5: the general path (tests + the bench's general legs), after a kernel change there.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_multi_small.py tests/test_multiplicity.py tests/test_multi_module.py tests/test_multi_windows.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu_general.log" 2>&1
tail -3 "$OUT/pytest_gpu_general.log"
for v in 1 1; do
MGX_MULTI_SMALL_OWN=$v timeout 600 python bench.py --gpus 1 --no-cpu-baseline --hetero-steps 0 --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('own_kernel=$v', {k: v for k, v in d['legs'].items() if k.startswith('general')})"
done
