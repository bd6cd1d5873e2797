#!/usr/bin/env python3
"""General path (2 gensets + 2 batteries + 1 grid per microgrid, H = 24, float64 rows): a Gym step with whole observation rows,
per-step rows (observe_row_multi inside step_multi_kernel: one lane writes its grid's row) against rings refilled by
obs_windows_k_multi_kernel.   python tools/exp_r4_multi_rings.py [N] [rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv
from pymgrid_amd.generator import generate, widen

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda:0")
for dt in (torch.float64, torch.float32):
    for K in (0, 16, 32):
        base = generate(N, n_steps=T, seed=42, arch="genset+battery+grid", horizon=24, device=dev)
        env = BatchedMicrogridEnv(widen(base, n_genset=2, n_battery=2, n_grid=1), obs_prefetch=K, obs_dtype=dt, reuse_outputs=3 * max(K, 1))
        del base
        L = env.layout
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        a = torch.rand(N, L.action_dim, dtype=torch.float64, device=dev, generator=gen)
        res = []
        for rep in range(3):
            env.reset()
            for _ in range(40):
                env.step(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = T - 24 - 40 - 8
            t0 = time.perf_counter(); e0.record()
            for _ in range(n):
                env.step(a)
            e1.record(); torch.cuda.synchronize()
            res.append((1e6 * (time.perf_counter() - t0) / n, 1e3 * e0.elapsed_time(e1) / n))
        row_b = L.obs_dim * (8 if dt == torch.float64 else 4)
        print(f"{str(dt):14s} D={L.obs_dim} K={K:2d}: " + "  ".join(f"{w:6.1f}/{g:6.1f}" for w, g in res) + f"  us wall/gpu per step   "
              f"({N * row_b / (min(g for _, g in res) * 1e-6) / 1e12:.2f} TB/s of rows)", flush=True)
        env.close()
        del env
        torch.cuda.empty_cache()
