#!/usr/bin/env python3
"""Round 5: Gym steps of the general path (2 gensets + 2 batteries + grid, H = 24, 162 columns) with row-major vs column-major
ring blocks, N = 100 000, ring depth 32."""
import sys, time
import torch
sys.path.insert(0, ".")
from pymgrid_amd import BatchedMicrogridEnv
from pymgrid_amd.generator import generate, widen

dev = torch.device("cuda:0")
N, T = 100_000, 600
import os
K = int(os.environ.get("EXP_K", "32"))
import itertools
for multi, dt, layout in itertools.product((True, False), (torch.float64, torch.float32), ("rows", "columns")):
    if True:
        base = generate(N, n_steps=T, seed=42, arch="genset+battery+grid", horizon=24, device=dev)
        env = BatchedMicrogridEnv(widen(base, n_genset=2, n_battery=2, n_grid=1) if multi else base, obs_prefetch=K, reuse_outputs=96,
                                  obs_layout=layout, obs_dtype=dt)
        del base
        L = env.layout
        a = torch.rand(N, L.action_dim, dtype=torch.float64, device=dev)
        env.reset()
        for _ in range(40):
            env.step(a)
        torch.cuda.synchronize()
        n = T - 24 - 40 - 8
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(n):
            env.step(a)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        esz = 4 if dt == torch.float32 else 8
        b = (L.bytes_per_step() - 1 + esz * L.obs_dim) * N
        print(f"{'multi ' if multi else 'single'} {str(dt):14s} {layout:8s} {us:7.2f} us/step  wall {(time.perf_counter() - t0) / n * 1e6:7.2f}  alg {b / 1e6:.1f} MB  "
              f"frac {b / (us * 1e-6) / 8e12:.3f}", flush=True)
        env.close()
