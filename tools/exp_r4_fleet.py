#!/usr/bin/env python3
"""Round 4: config-5 fleet (99 999 mixed grids, H = 24, T = 8 760, factorised series), rows contract -- ring refill scheduling:
staggered ring phases per bucket (the three refills start at different fleet steps) vs all at once; ring depth K."""
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


def timeit(fn, n=1600, warm=1200):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3


variants = [(torch.float64, 16, False), (torch.float64, 16, True), (torch.float64, 32, False), (torch.float64, 32, True),
            (torch.float64, 24, True), (torch.float32, 16, False), (torch.float32, 16, True)]
if len(sys.argv) > 1:
    variants = variants[:int(sys.argv[1])]
for rep in range(2):
    for dt, K, stagger in variants:
        batches = [generate(per, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised") for k, a in enumerate(archs)]
        fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=K, reuse_outputs=3 * K, stagger=stagger)
        acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
        fleet.reset()
        wall, gpu = timeit(lambda: fleet.step(acts))
        print(f"rep {rep} {str(dt):14s} K={K:2d} stagger={int(stagger)}: {wall:6.1f} us wall  {gpu:6.1f} us gpu per fleet step", flush=True)
        fleet.close()
        del fleet, batches
        torch.cuda.empty_cache()
