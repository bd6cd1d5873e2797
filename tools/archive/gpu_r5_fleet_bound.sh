#!/bin/bash
# Round 5: the bound fleet step (mgx_fleet_env_step): fleet tests, host cost per fleet step, the fleet legs.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_ring_layout.py tests/test_fleet_stagger.py tests/test_true_shape.py tests/test_abi_v3.py tests/test_factorised.py tests/test_multi_windows.py tests/test_gpu_parity.py tests/test_bench_contract.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8
timeout 300 python tools/exp_r5_fleet_host.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/exp_fleet_host_bound.txt"
timeout 600 python bench.py --gpus 1 --no-cpu-baseline --all-legs --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: v for k, v in d['legs'].items() if k.startswith('config5')})" | tee -a "$OUT/exp_fleet_host_bound.txt"
