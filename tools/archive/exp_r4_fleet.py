#!/usr/bin/env python3
"""Round 4: config-5 fleet (99 999 mixed grids, H = 24, T = 8 760, factorised series), rows contract -- one variant per process
(the library reads its experiment knobs from the environment once): ring depth K, refill workgroups per CU (MGX_WIN_MIN_LDS),
one pooled prefetch stream (MGX_PREFETCH_POOL), staggered ring phases.
usage: exp_r4_fleet.py K [float64|float32] [stagger|columns]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import BucketedFleet  # noqa: E402

dev = torch.device("cuda:0")
per = 33333
archs = ("genset+battery", "battery+grid", "genset+battery+grid")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16          # 0: no rings (step + whole row in one launch)
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "float32") else torch.float64
stagger = len(sys.argv) > 3 and sys.argv[3] == "stagger"
layout = "columns" if (len(sys.argv) > 3 and sys.argv[3] == "columns") else "rows"


def timeit(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3


batches = [generate(per, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised") for k, a in enumerate(archs)]
fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=K, reuse_outputs=3 * K if K else 8, stagger=stagger, obs_layout=layout)
acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
fleet.reset()
res = [timeit(lambda: fleet.step(acts), 2048, 1536 if rep == 0 else 0) for rep in range(3)]
knobs = " ".join(f"{k}={os.environ[k]}" for k in ("MGX_WIN_MIN_LDS", "MGX_PREFETCH_POOL") if k in os.environ)
print(f"{str(dt):14s} K={K:2d} stagger={int(stagger)} layout={layout:7s} {knobs:30s}: " + "  ".join(f"{w:5.1f}/{g:5.1f}" for w, g in res) + "  us wall/gpu per fleet step",
      flush=True)
fleet.close()
