#!/usr/bin/env python3
"""Config-5 fleet on rings (K = 32, float64 rows): GPU time per fleet step in consecutive windows of 256 steps after a reset --
is there a start-up transient?  (bench.py times 256 steps starting 64 steps after a reset.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import BucketedFleet  # noqa: E402

dev = torch.device("cuda:0")
per, K = 33333, 32
layout = sys.argv[1] if len(sys.argv) > 1 else "rows"
batches = [generate(per, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised")
           for k, a in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
fleet = BucketedFleet.from_batches(batches, obs_prefetch=K, reuse_outputs=3 * K, obs_layout=layout)
acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
fleet.reset()
for _ in range(3000):
    fleet.step(acts)
torch.cuda.synchronize()
for rep in range(3):
    fleet.reset()
    for _ in range(64):
        fleet.step(acts)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    ev[0].record()
    for w in range(12):
        for _ in range(256):
            fleet.step(acts)
        ev[w + 1].record()
    torch.cuda.synchronize()
    print(f"{layout} after reset + 64 steps, windows of 256 steps: " + " ".join(f"{ev[w].elapsed_time(ev[w + 1]) / 256 * 1e3:5.1f}" for w in range(12)), flush=True)
fleet.close()
