#!/usr/bin/env python3
"""GPU experiment: does splitting the batch over S streams (independent halves whose launch tails / heads overlap) beat one
launch sequence over all N grids?  Fused kernel, K = 64 per launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N, K, L = 100_000, 64, 96
for S in (1, 2, 3, 2, 1):
    n = N // S // 16 * 16 if S == 3 else N // S
    engs = [StepEngine(generate(n, n_steps=K * L + 8, seed=1 + j, device=dev)) for j in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    acts = [torch.rand(2, K, n, 3, dtype=torch.float64, device=dev) for _ in range(S)]
    outs = [dict(reward=torch.empty(K, n, dtype=torch.float64, device=dev), done=torch.empty(K, n, dtype=torch.uint8, device=dev),
                 soc_trace=torch.empty(K, n, dtype=torch.float64, device=dev)) for _ in range(S)]

    def run(launches):
        for j in range(launches):
            for e, st, a, o in zip(engs, streams, acts, outs):
                with torch.cuda.stream(st):
                    e.step_k(a[j & 1], reward=True, done=True, soc_trace=True, out=o)
    run(8)
    torch.cuda.synchronize()
    for e in engs:
        e.reset(want_obs=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(L - 8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = (L - 8) * K
    print(f"{S} stream(s) x {n} grids: {dt / (L - 8) * 1e6:7.1f} us per 64-step round of all {N} grids   {N * steps / dt / 1e9:6.2f} G env-steps/s")
    for e in engs:
        e.close()
