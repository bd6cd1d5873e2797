#!/usr/bin/env python3
"""Where does the HOST time of a fleet step go?  A small fleet (GPU time negligible) stepped under cProfile, rows and views
contracts; then the same fleets at 99 999 grids timed (wall per step).
   python tools/exp_fleet_host.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import BucketedFleet

dev = torch.device("cuda:0")
archs = ("genset+battery", "battery+grid", "genset+battery+grid")


def build(per, contract):
    batches = [generate(per, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised") for k, a in enumerate(archs)]
    kw = dict(obs_views=True) if contract == "views" else dict(obs_prefetch=16)
    fleet = BucketedFleet.from_batches(batches, reuse_outputs=48, **kw)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev, generator=gen) for e in fleet.envs]
    return fleet, acts


for contract in ("views", "rows"):
    fleet, acts = build(1024, contract)
    fleet.reset()
    for _ in range(500):
        fleet.step(acts)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5000):
        o = fleet.step(acts)[0]
        if contract == "views":
            for v in o:
                v.load; v.pv; v.grid
    pr.disable()
    torch.cuda.synchronize()
    print(f"==== {contract}: host profile of 5000 fleet steps (3 buckets of 1024 grids) ====")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    fleet.close()

for contract in ("views", "rows"):
    fleet, acts = build(33333, contract)
    fleet.reset()
    for _ in range(1000):
        fleet.step(acts)
    torch.cuda.synchronize()
    for consume in (False, True):
        t0 = time.perf_counter()
        for _ in range(4000):
            o = fleet.step(acts)[0]
            if consume and contract == "views":
                for v in o:
                    v.load; v.pv; v.grid
        torch.cuda.synchronize()
        print(f"{contract} 99 999 grids, consume={consume}: {(time.perf_counter() - t0) / 4000 * 1e6:.2f} us per fleet step (wall)")
    fleet.close()
