#!/bin/bash
# Run ON THE GPU BOX: kernel-trace each variant of tools/exp_step_variants.py, print the distribution of step_kernel durations.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/stepvar; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for v in ${VARIANTS:-rotate same many paced}; do
  for s in ${SERIES:-factorised}; do
    python $REPO/tools/exp_step_variants.py $v $s
    timeout 300 rocprofv3 --kernel-trace -d $OUT/$v.$s -o t --output-format csv -- python $REPO/tools/exp_step_variants.py $v $s > /dev/null 2>&1
    python - $OUT/$v.$s $v $s <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                      # drop the warm-up third
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
gap = sorted(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:]))
cad = sorted(int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(rows, rows[1:]))
q = lambda v, p: v[int(p * (len(v) - 1))] / 1e3
print(f"   trace {sys.argv[2]:7s} {sys.argv[3]:12s}: kernel us min {q(d,0):.2f} p10 {q(d,.1):.2f} med {q(d,.5):.2f} p90 {q(d,.9):.2f} | "
      f"gap end->start med {q(gap,.5):.2f} p10 {q(gap,.1):.2f} | start->start med {q(cad,.5):.2f}   ({len(d)} dispatches)")
PY
    rm -rf $OUT/$v.$s
  done
done
