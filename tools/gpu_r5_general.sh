#!/bin/bash
# Round 5: the general path (tests + the bench's general legs), after a kernel change there.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_multi_small.py tests/test_multiplicity.py tests/test_multi_module.py tests/test_multi_windows.py tests/test_env_step.py tests/test_host_logic.py tests/test_ring_layout.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu_general.log" 2>&1
tail -6 "$OUT/pytest_gpu_general.log"
timeout 600 python tools/exp_r5_multi_layout.py 2>&1 | tee "$OUT/exp_multi_layout.txt"
