#!/usr/bin/env python3
"""GPU experiment: env.step with per-step observation rows vs obs_prefetch=K (window prefetch), H = 24."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")
for arch in ("genset+battery", "genset+battery+grid"):
    for dt in (torch.float64, torch.float32):
        for K in (0, 4, 8, 16, 32):
            env = BatchedMicrogridEnv(generate(N, n_steps=1500, seed=1, arch=arch, horizon=24, device=dev), obs_dtype=dt,
                                      obs_prefetch=K)
            a = torch.rand(N, env.layout.action_dim, dtype=torch.float64, device=dev)
            env.reset()
            for _ in range(64):
                env.step(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(1024):
                env.step(a)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 1024 * 1e3
            print(f"{arch:20s} D={env.layout.obs_dim:4d} {str(dt):14s} prefetch K={K:3d}  {us:7.2f} us/step  {N / us / 1e3:6.2f} G env-steps/s")
            env.close()
