#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]

import re


def spec(kernel_name):
    """'void mgx::step_k_kernel<3, 4, double, false, true>(mgx::KArgs, ...)' -> 'step_k_kernel<3,4,double,false,true>' (None for
    kernels that are not the engine's)."""
    m = re.search(r"mgx::([a-z_0-9]+)(<[^(]*>)?\(", kernel_name)
    if not m:
        return None
    return (m.group(1) + (m.group(2) or "")).replace(" ", "")




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

# The headline of bench.py in this trace.  The bench line (stats.log) says which launches of the headline kernel were the
# timed ones: roofline.timed_rounds = [first, last) counts the rounds the runner had issued, and the headline mode runs
# first, so on every queue that carries its launches the timed ones are launches [first, last) of that kernel in time order.
import json
bench = None
for name in ("bench_detail.json", "stats.log"):      # round 5: stdout carries the compact record, the full one sits beside it
    try:
        for ln in open(os.path.join(out, name)):
            if ln.startswith("{"):
                bench = json.loads(ln)
    except OSError:
        pass
    if bench is not None and "timed_rounds" in bench.get("roofline", {}):
        break
for f in find("*kernel_trace.csv"):
    if os.sep + "stats" + os.sep not in f or bench is None:
        continue
    rf = bench["roofline"]
    kspec = rf["kernel"].replace(" ", "")                             # the headline's kernel specialisation
    first, last = rf.get("timed_rounds", [0, 0])
    lpr = rf["launches"] // max(1, last - first)                      # launches per round and stream
    grids = rf.get("grids_per_kernel_launch", bench["config"]["grids_per_gpu"])
    per_q = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if spec(r["Kernel_Name"]) == kspec:
                threads = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
                if grids <= threads < 1.45 * grids:
                    per_q[r.get("Queue_Id")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    print(f"== headline: {rf['kernel']} launches of the timed region (rounds [{first}, {last}) of the bench line) per queue ==")
    lo, hi, n = None, None, 0
    for q, v in sorted(per_q.items()):
        v.sort()
        body = v[first * lpr: last * lpr]
        if len(body) != (last - first) * lpr:
            print(f"queue {q}: {len(v)} launches of that size, not the headline's queue")
            continue
        dur = [(e - s) / 1e3 for s, e in body]
        cad = (body[-1][1] - body[0][0]) / 1e3 / len(body)
        lo = body[0][0] if lo is None else min(lo, body[0][0]); hi = body[-1][1] if hi is None else max(hi, body[-1][1]); n = len(body)
        print(f"queue {q}: launches={len(body)} kernel avg_us={sum(dur) / len(dur):.2f} min_us={min(dur):.2f} "
              f"max_us={max(dur):.2f}  queue cadence_us={cad:.2f}")
    if n:
        cad = (hi - lo) / 1e3 / n
        alg = rf["algorithmic_bytes_per_launch"]
        frac = alg / (cad * 1e-6) / 1e9 / rf["peak"]
        print(f"timed region: {n} rounds in {(hi - lo) / 1e3:.1f} us -> {cad:.2f} us per round (one launch per queue); "
              f"algorithmic {alg / 1e6:.1f} MB per round -> {alg / (cad * 1e-6) / 1e9:.0f} GB/s = frac {frac:.3f} of {rf['peak']:.0f} GB/s"
              f"   [bench line: avg_launch_us {rf['avg_launch_us']:.2f}, frac {rf['frac']:.3f}]")

WATCHED = ("step_k_kernel", "step_kernel", "rollout_kernel", "observe_kernel", "expand_kernel", "obs_windows_k_kernel", "fleet_step_kernel", "fleet_step_kernel_v",
           "step_multi_kernel", "step_k_multi_kernel", "step_k_multi_small_kernel", "obs_windows_k_multi_kernel")
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
        for k, v in acc.items():                       # per kernel SPECIALISATION (the factorised / materialised forms differ)
            sp = spec(k)
            if sp and sp.split("<")[0] in WATCHED:
                v2 = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]     # inter-quartile mean
                traffic[sp][counter] = sum(v2) / len(v2)
        for (k, threads), v in by_size.items():        # the same per launch size (sharded runs launch half-size grids)
            sp = spec(k)
            if sp and sp.split("<")[0] in WATCHED and len(v) >= 4:
                v2 = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]
                sized[sp].setdefault(str(threads), {})[counter] = sum(v2) / len(v2)

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
        sp = spec(k)
        if sp in traffic:
            v2 = sorted(v)
            dur[sp] = v2[len(v2) // 2] / 1e3
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
