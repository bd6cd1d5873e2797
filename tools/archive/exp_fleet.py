#!/usr/bin/env python3
"""GPU experiment: where does a heterogeneous fleet step spend its time (per bucket, streams vs sequential)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import BucketedFleet  # noqa: E402

dev = torch.device("cuda:0")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 33333


def timeit(fn, n=256):
    for _ in range(32):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for dt in (torch.float64, torch.float32):
    for K in (0, 8):
        batches = [generate(per, n_steps=1500, seed=43 + k, arch=arch, horizon=24, device=dev)
                   for k, arch in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
        fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=K)
        acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
        fleet.reset()
        us_fleet = timeit(lambda: fleet.step(acts))
        each = [timeit(lambda e=e, a=a: e.step(a)) for e, a in zip(fleet.envs, acts)]
        seq = timeit(lambda: [e.step(a) for e, a in zip(fleet.envs, acts)])
        print(f"{str(dt):14s} K={K}: fleet.step {us_fleet:7.1f} us   sequential {seq:7.1f} us   buckets alone "
              + " / ".join(f"{u:6.1f}" for u in each))
        fleet.close()
        del fleet, batches
        torch.cuda.empty_cache()
