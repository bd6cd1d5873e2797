#!/bin/bash
# Run ON THE GPU BOX: SQ / TCC counter passes for the bench kernels (separate --pmc passes, kernel-trace only).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$1
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 32 --warmup 8 --no-cpu-baseline --hetero-steps 0 --shards 1"     # rounds of 64 env-steps
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAVES --kernel-trace -d "$OUT/sq" -o b --output-format csv -- python "$REPO/bench.py" $ARGS > "$OUT/sq.log" 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -d "$OUT/tcc" -o b --output-format csv -- python "$REPO/bench.py" $ARGS > "$OUT/tcc.log" 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d "$OUT/grbm" -o b --output-format csv -- python "$REPO/bench.py" $ARGS > "$OUT/grbm.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for short in ("step_k_kernel", "step_kernel", "rollout_kernel"):
            if f"mgx::{short}<" in k:
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for short, d in acc.items():
        print(short, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
find "$OUT" -name "*.csv" -size +1M -delete
