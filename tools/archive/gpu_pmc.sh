#!/bin/bash
# Run ON THE GPU BOX: SQ / TCC counter passes for the headline kernel of a bench mode (separate --pmc passes, kernel-trace only).
# Usage: tools/gpu_pmc.sh <tag> [bench args: e.g. --mode rbc --series materialised]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 32 --warmup 8 --no-cpu-baseline --hetero-steps 0 --no-side-modes $*"     # rounds of 64 env-steps, the headline mode only
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAVES --kernel-trace -d "$OUT/sq" -o b --output-format csv -- python "$REPO/bench.py" $ARGS > "$OUT/sq.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace -d "$OUT/sq2" -o b --output-format csv -- python "$REPO/bench.py" $ARGS > "$OUT/sq2.log" 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace -d "$OUT/tcc" -o b --output-format csv -- python "$REPO/bench.py" $ARGS > "$OUT/tcc.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
def spec(k):
    m = re.search(r"mgx::([a-z_0-9]+)(<[^(]*>)?\(", k)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else None
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        sp = spec(r["Kernel_Name"])
        if sp and sp.split("<")[0] in ("step_k_kernel", "step_kernel", "rollout_kernel", "fleet_step_kernel"):
            acc[sp][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for sp, d in acc.items():
        print(sp, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
find "$OUT" -name "*.csv" -size +1M -delete
