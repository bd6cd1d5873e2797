#!/usr/bin/env python3
"""bench.py's config-5 rows leg, taken apart: the same call sequence (warm-up blocks of 1000 steps each behind a reset, then reset + 64
steps + the timed steps), with an event every 64 steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import BucketedFleet  # noqa: E402

dev = torch.device("cuda:0")
per, K, rows = 33333, 32, 8760
mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
batches = [generate(per, n_steps=rows, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised")
           for k, a in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
fleet = BucketedFleet.from_batches(batches, obs_prefetch=K, reuse_outputs=3 * K)
gen = torch.Generator(device=dev); gen.manual_seed(11)
if mode == "plainrand":
    acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
else:
    acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev, generator=gen) for e in fleet.envs]


def fstep():
    if fleet.envs[0].current_step >= rows - 1:
        fleet.reset()
    return fleet.step(acts)


prev, t_end = None, time.perf_counter() + 4.0
blocks = 0
while time.perf_counter() < t_end:
    fleet.reset()
    t0 = time.perf_counter()
    for _ in range(1000):
        fstep()
    torch.cuda.synchronize(dev)
    cur = time.perf_counter() - t0
    blocks += 1
    if prev is not None and abs(cur - prev) < 0.03 * prev:
        break
    prev = cur
for rep in range(3):
    fleet.reset()
    for _ in range(64):
        fstep()
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
    ev[0].record()
    for w in range(16):
        for _ in range(64):
            fstep()
        ev[w + 1].record()
    torch.cuda.synchronize(dev)
    print(f"{mode}: {blocks} warm-up blocks; after reset + 64 steps, per 64 steps: " + " ".join(f"{ev[w].elapsed_time(ev[w + 1]) / 64 * 1e3:5.1f}" for w in range(16)), flush=True)
fleet.close()
