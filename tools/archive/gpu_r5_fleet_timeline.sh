#!/bin/bash
# Round 5: kernel durations of the config-5 fleet with rows (rocprofv3 --kernel-trace): the step launches alone and beside a ring refill.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/fleet_timeline_r05.txt"
for dt in float64 float32; do
  rm -rf /tmp/ft
  (cd "$REPO" && timeout 300 rocprofv3 --kernel-trace -d /tmp/ft -o t --output-format csv -- python tools/exp_fleet_prof.py rows $dt 2000 32 > /tmp/ft.log 2>&1)
  python - /tmp/ft $dt <<'PY' | tee -a "$OUT/fleet_timeline_r05.txt"
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
steps = [(s, e) for s, e, n in rows if "fleet_step_kernel" in n][-1024:]
refills = [(s, e, n) for s, e, n in rows if "obs_windows_k_kernel" in n and s >= steps[0][0]]
def overlapped(s, e):
    return any(rs < e and re_ > s for rs, re_, _ in refills)
beside = [(e - s) / 1e3 for s, e in steps if overlapped(s, e)]
alone = [(e - s) / 1e3 for s, e in steps if not overlapped(s, e)]
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
span = (steps[-1][1] - steps[0][0]) / 1e3 / len(steps)
print(f"== rows {sys.argv[2]}: last {len(steps)} fleet steps, {span:.2f} us per step (first start to last end)")
print(f"   fleet_step_kernel_v alone:  {len(alone):4d} launches, median {med(alone):6.2f} us")
print(f"   ... beside a refill:        {len(beside):4d} launches, median {med(beside):6.2f} us")
by = {}
for s, e, n in refills:
    by.setdefault(n.split("mgx::")[1].split("(")[0], []).append((e - s) / 1e3)
for n, v in sorted(by.items()):
    print(f"   {n:44s} {len(v):3d} launches, median {med(v):7.1f} us")
PY
done
