#!/bin/bash
# Round 5: fleet_step_kernel_v (KArgs by value, bucket = blockIdx.y) vs the pointer form (MGX_FLEET_BYVALUE=0).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_ring_layout.py tests/test_fleet_stagger.py tests/test_true_shape.py tests/test_gpu_parity.py tests/test_factorised.py -m gpu -q -x -p no:cacheprovider -k "fleet or hetero or true_shape or Fleet" 2>&1 | tail -3
: > "$OUT/exp_fleet_byvalue.txt"
for cfg in 0 1 0 1; do
export MGX_FLEET_BYVALUE=$cfg
timeout 600 python bench.py --gpus 1 --no-cpu-baseline --all-legs --detail /dev/null 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('byvalue=$cfg', {k: v['us'] for k, v in d['legs'].items() if k.startswith('config5')})" | tee -a "$OUT/exp_fleet_byvalue.txt"
done
cd /tmp && export TMPDIR=/tmp
for cfg in 0 1; do
export MGX_FLEET_BYVALUE=$cfg
for m in fleet fleet3; do
  rm -rf /tmp/fv_$m
  (cd "$REPO" && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fv_$m -o t --output-format csv -- python tools/exp_r5_fleet_vs_env.py $m > /tmp/fv_$m.log 2>&1)
  echo "== byvalue=$cfg $m" | tee -a "$OUT/exp_fleet_byvalue.txt"
  python - /tmp/fv_$m <<'PY' | tee -a "$OUT/exp_fleet_byvalue.txt"
import csv, glob, os, sys
from collections import defaultdict
d = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:1]:
    v2 = sorted(v)
    print(f"{k[:70]:70s} calls={len(v)} avg_us={sum(v)/len(v)/1e3:.2f} med_us={v2[len(v2)//2]/1e3:.2f} p10={v2[len(v2)//10]/1e3:.2f} p90={v2[9*len(v2)//10]/1e3:.2f}")
PY
done
done
