#!/bin/bash
# Round 6, first GPU call: the two-chains experiment (Gym single step from two host threads), the GPU tests, the driver's bench command.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd "$REPO"
python -c "from pymgrid_amd import _lib; print('csrc_hash', _lib.built_hash() or _lib.source_hash())" > "$OUT/csrc_hash.txt"
nproc > "$OUT/nproc.txt"
timeout 600 python tools/exp_r6_two_chains.py > "$OUT/exp_two_chains.txt" 2>&1
cat "$OUT/exp_two_chains.txt" | grep -v amdgpu.ids
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$OUT/bench_detail.json" > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"
echo "bench rc $? line length $(wc -c < "$OUT/bench_driver_cmd.json")"
cat "$OUT/bench_driver_cmd.json"
grep -v "^bench_detail\|amdgpu.ids" "$OUT/bench_driver_cmd.err" | tail -20
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
tail -15 "$OUT/pytest_gpu.log"
