#!/bin/bash
# Round 5: instruction / wait counters of the general K-step kernel (one pass per counter group)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/kstep_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/$tag" -o b --output-format csv -- python "$REPO/tools/exp_r5_general_prof.py" kstep 256 > "$OUT/$tag.log" 2>&1
done
python - "$OUT" <<'PY' | tee "$OUT/../exp_kstep_pmc.txt"
import csv, glob, os, sys
from collections import defaultdict
top = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(os.path.join(top, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_k_multi_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:24s} launches={len(v)} mean={sum(v)/len(v):.4g}")
PY
rm -rf "$OUT"
