#!/usr/bin/env python3
"""Round 5: what the HOST spends per BucketedFleet.step (three buckets, rings of 32 blocks): a fleet so small that its kernels take
nothing (3 x 256 grids), wall time per step = the interpreter + ctypes + driver launch path."""
import sys, time
import torch
sys.path.insert(0, ".")
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import BucketedFleet

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32        # ring depth
for per in ((256, 33333) if len(sys.argv) < 3 else (33333,)):
    for dt in (torch.float64, torch.float32):
        batches = [generate(per, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised")
                   for k, a in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
        fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=K, reuse_outputs=3 * K)
        acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
        fleet.reset()
        for _ in range(2000):
            fleet.step(acts)
        torch.cuda.synchronize()
        n = 6000
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(n):
            fleet.step(acts)
        t_issue = time.perf_counter() - t0
        e1.record(); torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"K {K:3d} per-bucket grids {per:6d} {str(dt):14s} issue {t_issue / n * 1e6:6.2f} us/step  wall {wall / n * 1e6:6.2f}  gpu {e0.elapsed_time(e1) * 1e3 / n:6.2f}", flush=True)
        fleet.close()
