#!/bin/bash
# Round 6: the general path's K-step launch with compile-time instance counts (mgx_fused.hip part 5) vs run-time counts
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_multi_small.py tests/test_multiplicity.py tests/test_multi_module.py tests/test_round6_goldens.py tests/test_env_step.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for ST in 1 0 1 0; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --legs none --hetero-steps 0 --no-cpu-baseline --tunable multi_static=$ST --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('multi_static=$ST', {k: v for k, v in d['legs'].items() if k.startswith('general')})" | tee -a "$OUT/exp_general_static_counts.txt"
done
