#!/bin/bash
# Run ON THE GPU BOX: HBM traffic / L2 hit counters of the observation kernel inside a step loop (separate --pmc passes).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_obs
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $pass --kernel-trace -d "$OUT/$tag" -o b --output-format csv -- python "$REPO/tools/exp_obs_steps.py" > "$OUT/$tag.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for short in ("obs_rows_wave_kernel<3", "obs_rows_wave_kernel<7", "step_kernel<3", "step_kernel<7"):
            if f"mgx::{short}" in k:
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for short, d in acc.items():
        print(short, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
PY
find "$OUT" -name "*.csv" -size +1M -delete
