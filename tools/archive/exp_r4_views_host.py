#!/usr/bin/env python3
"""Round 4: where does the host time of a views-contract fleet step go?  cProfile of 3000 fleet steps (config-5 fleet, obs_views) +
wall / GPU time per step with and without taking the window views."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import BucketedFleet  # noqa: E402

dev = torch.device("cuda:0")
per = 33333
batches = [generate(per, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised")
           for k, a in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
fleet = BucketedFleet.from_batches(batches, obs_views=True, reuse_outputs=48)
acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]


def loop(n, take):
    for _ in range(n):
        obs = fleet.step(acts)[0]
        if take:
            for o in obs:
                o.load; o.pv; o.grid


for take in (False, True):
    fleet.reset(); loop(1500, take); torch.cuda.synchronize()
    fleet.reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); loop(3000, take); e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"take views={take}: host issue {1e6 * (t1 - t0) / 3000:.2f} us/step, gpu {e0.elapsed_time(e1) / 3000 * 1e3:.2f} us/step", flush=True)
fleet.reset()
pr = cProfile.Profile()
pr.enable(); loop(3000, True); pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
