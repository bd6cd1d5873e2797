#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/exp_fleet_vs_env.txt"
for m in env fleet fleet3; do
  rm -rf /tmp/fv_$m
  (cd "$REPO" && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$m -o t --output-format csv -- python tools/exp_r5_fleet_vs_env.py $m > /tmp/fv_$m.log 2>&1)
  echo "== $m" | tee -a "$OUT/exp_fleet_vs_env.txt"
  python - /tmp/fv_$m <<'PY' | tee -a "$OUT/exp_fleet_vs_env.txt"
import csv, glob, os, sys
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:3]:
    v2 = sorted(v)
    print(f"{k[:70]:70s} calls={len(v)} avg_us={sum(v)/len(v)/1e3:.2f} med_us={v2[len(v2)//2]/1e3:.2f} p10={v2[len(v2)//10]/1e3:.2f} p90={v2[9*len(v2)//10]/1e3:.2f}")
PY
done
