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

# the fused kernel per HIP queue (bench.py steps two independent shards on two streams): kernel durations AND the cadence
# of each queue = what bench.py's HIP events on that stream measure as "duration of a round"
for f in find("*kernel_trace.csv"):
    if os.sep + "stats" + os.sep not in f:
        continue
    per_q = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "mgx::step_k_kernel<" in r["Kernel_Name"]:
                per_q[(r.get("Queue_Id"), r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    print("== step_k_kernel per queue / launch size (threads): kernel duration vs queue cadence ==")
    for (q, size), v in sorted(per_q.items()):
        v.sort()
        # the longest burst of back-to-back launches (bench.py's timed region): a new burst starts after an idle gap
        bursts, cur = [], [v[0]]
        for a, b in zip(v, v[1:]):
            if b[0] - a[1] > 40_000:
                bursts.append(cur); cur = []
            cur.append(b)
        bursts.append(cur)
        body = max(bursts, key=len)
        dur = [(e - s) / 1e3 for s, e in body]
        cad = (body[-1][1] - body[0][0]) / 1e3 / len(body)
        print(f"queue {q} threads {size}: launches={len(body)} kernel avg_us={sum(dur) / len(dur):.2f} min_us={min(dur):.2f} "
              f"max_us={max(dur):.2f}  cadence_us={cad:.2f}")

traffic = defaultdict(dict)
sized = defaultdict(dict)
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find("*counter_collection.csv"):
        if name not in f:
            continue
        acc = defaultdict(list)
        by_size = defaultdict(list)                    # (kernel, launch size in threads) -> values
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                    by_size[(r["Kernel_Name"], int(r.get("Grid_Size") or 0))].append(float(r["Counter_Value"]))
        print(f"== {counter} per dispatch (raw counter units as reported: KiB) ==")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
            print(f"{k[:90]:90s} dispatches={len(v)} avg={sum(v)/len(v):.1f} total={sum(v):.1f}")
        for k, v in acc.items():
            for short in ("step_k_kernel", "step_kernel", "rollout_kernel", "observe_kernel", "expand_kernel"):
                if f"mgx::{short}<" in k:
                    v2 = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]     # inter-quartile mean
                    traffic[short][counter] = sum(v2) / len(v2)
        for (k, threads), v in by_size.items():        # the same per launch size (sharded runs launch half-size grids)
            for short in ("step_k_kernel", "step_kernel", "rollout_kernel"):
                if f"mgx::{short}<" in k and len(v) >= 4:
                    v2 = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]
                    sized[short].setdefault(str(threads), {})[counter] = sum(v2) / len(v2)

# HBM bytes per launch: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B
# requests as 64 B for wide coalesced streaming reads, so it is doubled (MI355X_MICROARCH.md, section HBM).
import json
dur = {}
for f in find("*kernel_trace.csv"):
    if os.sep + "stats" + os.sep not in f:
        continue
    d = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        for short in traffic:
            if f"mgx::{short}<" in k:
                v2 = sorted(v)
                dur[short] = v2[len(v2) // 2] / 1e3
res = {}
for short, c in traffic.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res[short] = {"fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
                      "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
                      "median_launch_us": dur.get(short)}
by_threads = {}
for short, d in sized.items():
    for threads, c in d.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            by_threads.setdefault(short, {})[threads] = {
                "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
                "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024}
with open(os.path.join(out, "traffic.json"), "w") as fh:
    json.dump({"by_launch_threads": by_threads, "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads 1/2 of wide "
                             "coalesced streams; separate --pmc passes)", "kernels": res}, fh, indent=1)
print("== traffic.json ==")
print(json.dumps(res, indent=1))
