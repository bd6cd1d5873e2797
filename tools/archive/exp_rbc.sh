#!/bin/bash
# GPU: rule-based rollout timing (bench --mode rbc) + SQ instruction counters of rollout_kernel
for rep in 1 2; do
python bench.py --gpus 1 --mode rbc --steps 64 --warmup 16 --no-side-modes --no-cpu-baseline --hetero-steps 0 --shards 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('rbc one stream: %.2f us per 64-step launch  frac %.3f  %.2f G env-steps/s' % (r['avg_launch_us'], r['frac'], d['value']/1e9))"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAVES --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/sq -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --mode rbc --steps 16 --warmup 4 --no-side-modes --no-cpu-baseline --hetero-steps 0 --shards 1 --prewarm 0.05 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
from collections import defaultdict
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/sq"
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mgx::rollout_kernel<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {c: sum(v) / len(v) for c, v in acc.items()}
    print({c: round(v, 1) for c, v in m.items()})
    if m:
        print("VALU instructions per wave and step: %.1f   VALU busy share of wave cycles: %.2f   waiting: %.2f" %
              (m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / 64, m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/sq
