#!/usr/bin/env python3
"""Ring refills ALONE (no steps beside them): obs_windows_k_kernel of one 33 333-grid genset+battery+grid bucket (D = 156, H = 24), K = 32 blocks,
float64 and float32 rows, row- and column-major blocks -- is the float refill byte-bound?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N, K = 33333, 32
for layout in ("rows", "columns"):
    for dt in (torch.float64, torch.float32):
        env = BatchedMicrogridEnv(generate(N, n_steps=8760, seed=45, arch="genset+battery+grid", horizon=24, device=dev, series="factorised"),
                                  obs_prefetch=K, obs_dtype=dt, obs_layout=layout)
        env.reset()
        e = env.engine
        ring = env._rings[2]
        for _ in range(5):
            e.observe_windows_ahead(K, out=ring); e.prefetch_wait()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for _ in range(n):
            e.observe_windows_ahead(K, out=ring); e.prefetch_wait()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        mb = K * N * env.layout.obs_dim * (8 if dt == torch.float64 else 4) / 1e6
        print(f"{layout:8s} {str(dt):14s}: {us:7.1f} us per refill of {mb:7.1f} MB  = {mb / us:.2f} TB/s", flush=True)
        env.close()
