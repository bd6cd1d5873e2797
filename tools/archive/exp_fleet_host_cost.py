#!/usr/bin/env python3
"""Where the host time of a fused fleet step goes: the C call alone (mgx_fleet_step with a fixed item array, no Python
bookkeeping) vs BucketedFleet.step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import _raw_stream  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import BucketedFleet  # noqa: E402

dev = torch.device("cuda:0")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 2000          # small buckets: the GPU is never the bound
batches = [generate(per, n_steps=30000, seed=43 + k, arch=arch, horizon=24, device=dev)
           for k, arch in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
fleet = BucketedFleet.from_batches(batches, reuse_outputs=48)
acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
fleet.reset()
for _ in range(2000):
    fleet.step(acts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5000):
    fleet.step(acts)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"BucketedFleet.step: {1e6 * (t1 - t0) / 5000:.2f} us per call (3 buckets of {per} grids)")
key = next(k for k, v in fleet._plans.items() if all(it.refill_ring is None and not it.wait_prefetch for it in v[0]))
items = fleet._plans[key][0]
for it, a in zip(items, acts):
    it.actions = a.data_ptr()
lib, st = fleet.envs[0].engine._lib, _raw_stream(0)
t0 = time.perf_counter()
for _ in range(5000):
    lib.mgx_fleet_step(items, 3, 1, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"mgx_fleet_step alone (ctypes call, one fleet_step_kernel launch): {1e6 * (t1 - t0) / 5000:.2f} us per call")
