#!/bin/bash
# Round 5, the run whose numbers are to be judged: the GPU tests, then everything bench.py quotes from profiles/ (tools/gpu_profile_r05.sh).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu_final.log" 2>&1
tail -5 "$OUT/pytest_gpu_final.log"
bash tools/gpu_profile_r05.sh r05 > "$OUT/profile_r05.log" 2>&1
tail -40 "$OUT/profile_r05.log"
ls -la gpurun_out/prof_r05 | head -40
