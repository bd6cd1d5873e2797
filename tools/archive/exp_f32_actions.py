#!/usr/bin/env python3
"""GPU experiment: fused kernel with float32 actions, [K,N,3] vs padded [K,N,4] (MGX_LIB=...pad4 build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N, K = 100_000, 64
pad = "pad4" in os.environ.get("MGX_LIB", "")
b = generate(N, n_steps=2200, seed=1, device=dev)
for dt in (torch.float64, torch.float32):
    eng = StepEngine(b, action_dtype=dt)
    A = 4 if (pad and dt == torch.float32) else 3
    pool = torch.rand(4, K, N, A, dtype=dt, device=dev)
    out = dict(reward=torch.empty(K, N, dtype=torch.float64, device=dev), done=torch.empty(K, N, dtype=torch.uint8, device=dev),
               soc_trace=torch.empty(K, N, dtype=torch.float64, device=dev))
    eng._check_actions = lambda a, lead: a          # experiment: the padded layout has 4 columns

    def run(n):
        for j in range(n):
            if eng.current_step + K > 2200:
                eng.reset(want_obs=False)
            eng.step_k(pool[j % 4], reward=True, done=True, soc_trace=True, out=out)
    run(16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(128); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 128 * 1e3
    print(f"{os.path.basename(os.environ.get('MGX_LIB', 'shipped')):18s} actions {str(dt):14s} [K,N,{A}]  {us:6.1f} us/launch  {N * K / us / 1e3:6.1f} G env-steps/s")
    eng.close()
