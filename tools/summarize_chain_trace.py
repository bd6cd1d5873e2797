#!/usr/bin/env python3
"""Kernel trace of tools/exp_r6_two_chains.py ... trace (rocprofv3 --kernel-trace): per launch size of step_kernel -- 100 096 /
50 176 / 25 088 threads = 1 / 2 / 4 chains -- the kernel's own duration, each queue's start-to-start cadence, the idle gap between
a queue's consecutive kernels and how much of a kernel's duration another chain's kernel was running beside it."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "mgx::step_kernel<" in r["Kernel_Name"]:
                rows.append((int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0), r.get("Queue_Id"), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
by = defaultdict(lambda: defaultdict(list))
for size, q, s, e in rows:
    by[size][q].append((s, e))
for size in sorted(by, reverse=True):
    qs = by[size]
    allk = sorted(k for v in qs.values() for k in v)
    tail = {q: sorted(v)[len(v) // 2:] for q, v in qs.items()}          # the second half of every queue's launches: steady state
    print(f"== step_kernel launches of {size} threads: {len(allk)} on {len(qs)} queue(s)")
    for q, v in sorted(tail.items()):
        dur = sorted((e - s) / 1e3 for s, e in v)
        cad = (v[-1][0] - v[0][0]) / 1e3 / (len(v) - 1)
        gaps = sorted((v[j + 1][0] - v[j][1]) / 1e3 for j in range(len(v) - 1))
        print(f"  queue {q}: {len(v)} launches  duration med {dur[len(dur) // 2]:.2f} us (p10 {dur[len(dur) // 10]:.2f}, p90 {dur[9 * len(dur) // 10]:.2f})  "
              f"start-to-start {cad:.2f} us  idle gap med {gaps[len(gaps) // 2]:.2f} us (p90 {gaps[9 * len(gaps) // 10]:.2f})")
    if len(qs) > 1:                                                        # overlap: time inside a kernel of queue A during which any other queue ran one
        names = sorted(tail)
        a = tail[names[0]]
        others = sorted(k for q in names[1:] for k in tail[q])
        j, cov, tot = 0, 0, 0
        for s, e in a:
            tot += e - s
            while j < len(others) and others[j][1] <= s:
                j += 1
            k = j
            while k < len(others) and others[k][0] < e:
                cov += max(0, min(e, others[k][1]) - max(s, others[k][0]))
                k += 1
        print(f"  overlap: {100.0 * cov / max(1, tot):.0f} % of queue {names[0]}'s kernel time had another chain's kernel running beside it")
        lo = min(v[0][0] for v in tail.values()); hi = max(v[-1][1] for v in tail.values())
        n = min(len(v) for v in tail.values())
        print(f"  all chains: {(hi - lo) / 1e3 / n:.2f} us per env-step of the whole batch")
