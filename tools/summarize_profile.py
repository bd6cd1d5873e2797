#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})

# per-kernel average duration from the raw trace (ns)
for f in find("*kernel_trace.csv"):
    if "stats" not in f:
        continue
    dur = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== per-kernel durations from the trace ==")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
        v2 = sorted(v)
        print(f"{k[:90]:90s} calls={len(v)} avg_us={sum(v)/len(v)/1e3:.2f} med_us={v2[len(v2)//2]/1e3:.2f} "
              f"min_us={v2[0]/1e3:.2f} max_us={v2[-1]/1e3:.2f}")

for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find("*counter_collection.csv"):
        if name not in f:
            continue
        acc = defaultdict(list)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        print(f"== {counter} per dispatch (raw counter units as reported: KiB) ==")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
            print(f"{k[:90]:90s} dispatches={len(v)} avg={sum(v)/len(v):.1f} total={sum(v):.1f}")
