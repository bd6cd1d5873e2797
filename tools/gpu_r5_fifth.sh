#!/bin/bash
# Round 5, fifth GPU call: the general path's K-step kernel with LDS-resident parameter / state columns: parity tests of the general
# path, then the bench's general-path legs.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_multi_small.py tests/test_multiplicity.py tests/test_multi_module.py tests/test_multi_windows.py tests/test_rbc.py tests/test_host_logic.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu5.log" 2>&1
tail -6 "$OUT/pytest_gpu5.log"
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --hetero-steps 0 --no-cpu-baseline --detail "$OUT/bench_detail5.json" > "$OUT/bench5.json" 2> "$OUT/bench5.err"
python - "$OUT/bench5.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["legs"].items():
    print(f"{k:48s} {v}")
PY
