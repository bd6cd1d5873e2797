#!/usr/bin/env python3
"""Round 4: observation rows WITHOUT rings on factorised series (the window sources are cache-resident base tables): per-step
obs_rows_wave_kernel behind the step kernel vs the prefetched rings, one 100 000-grid genset+battery+grid batch, H = 24."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 100000))


def timeit(fn, n=1024, warm=768):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3


for arch in ("genset+battery+grid", "genset+battery"):
    for series in ("factorised", "materialised"):
        for K in (0, 16, 32):
            b = generate(N, n_steps=8760 if series == "factorised" else 2400, seed=43, arch=arch, horizon=24, device=dev, series=series)
            env = BatchedMicrogridEnv(b, obs_prefetch=K, reuse_outputs=4)
            a = torch.rand(N, b.layout.action_dim, dtype=torch.float64, device=dev)
            env.reset()
            w, g = timeit(lambda: env.step(a))
            D = b.layout.obs_dim
            print(f"{arch:20s} {series:12s} D={D:3d} obs_prefetch={K:2d}: {w:6.1f} us wall {g:6.1f} us gpu per step  ({N * D * 8 / g / 1e6:5.2f} TB/s of rows)", flush=True)
            env.close()
            del env, b
            torch.cuda.empty_cache()
