#!/usr/bin/env python3
"""Config-5 fleet (99 999 mixed grids, H = 24, T = 8 760, factorised series), rows contract: how the observation rings are
renewed -- ahead on the prefetch streams (default) or as chunks inside the step launches -- and the ring depth K; float64 / float32."""
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


def timeit(fn, n=1500, warm=1500):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3


for dt in (torch.float64, torch.float32):
    for refill, K in (("ahead", 16), ("chunks", 16), ("chunks", 8), ("ahead", 8), ("ahead", 32), ("chunks", 32)):
        batches = [generate(per, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised") for k, a in enumerate(archs)]
        fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=K, reuse_outputs=3 * K, refill=refill)
        acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
        fleet.reset()
        wall, gpu = timeit(lambda: fleet.step(acts))
        print(f"{str(dt):14s} refill={refill:6s} K={K:2d}: {wall:6.1f} us wall  {gpu:6.1f} us gpu per fleet step", flush=True)
        fleet.close()
        del fleet, batches
        torch.cuda.empty_cache()
