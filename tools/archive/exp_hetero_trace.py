#!/usr/bin/env python3
"""Heterogeneous H=24 fleet stepping alone (bench.py's hetero leg), for rocprofv3 timelines.
usage: exp_hetero_trace.py [steps] [prefetch K] [float64|float32] [chunks|ahead]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import BucketedFleet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "float32") else torch.float64
refill = sys.argv[4] if len(sys.argv) > 4 else "chunks"
dev = torch.device("cuda:0")
per = int(os.environ.get("PER", 33333))
series = os.environ.get("SERIES", "factorised")
batches = [generate(per, n_steps=8760, seed=43 + k, arch=arch, horizon=24, device=dev, series=series)
           for k, arch in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=K, reuse_outputs=3 * K, refill=refill)
gen = torch.Generator(device=dev); gen.manual_seed(11)
acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev, generator=gen) for e in fleet.envs]
fleet.reset()
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    fleet.reset()
    for _ in range(1000):
        fleet.step(acts)
    torch.cuda.synchronize()
fleet.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fleet.step(acts)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"hetero K={K} {dt} refill={refill}: host issue {1e6 * (t1 - t0) / steps:.1f} us/step, wall {1e6 * (t2 - t0) / steps:.1f} us/step")
