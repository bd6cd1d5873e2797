#!/usr/bin/env python3
"""Print the last `n` kernel dispatches of a rocprofv3 kernel trace as a timeline (queue, start offset, duration) and, if a
HIP API trace is present, the mean duration of each API call."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
for f in sorted(glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "mgx::" in r["Kernel_Name"]][-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("mgx::")[1].split("(")[0][:40]
        print(f"q{r.get('Queue_Id'):>3s} +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  {name}")
for f in sorted(glob.glob(os.path.join(out, "**", "*hip_api_trace.csv"), recursive=True)):
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Function"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== HIP API calls: count, mean us ==")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"{k:40s} {len(v):8d} {sum(v) / len(v) / 1e3:8.2f}")
