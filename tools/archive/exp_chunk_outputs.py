#!/usr/bin/env python3
"""GPU experiment: is the advantage of 48-step launches over 64-step launches an Infinity-Cache effect of re-using the same
output buffers every launch?  One launch sequence over 100 000 grids; outputs written to 1 set of buffers (re-used every
launch) or cycled over 8 sets (each set is rewritten only after > 650 MB of other traffic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N = 100_000
eng = StepEngine(generate(N, n_steps=8760, seed=1, device=dev))
for K in (32, 48, 64):
    for sets in (1, 8):
        a = torch.rand(4, K, N, 3, dtype=torch.float64, device=dev)
        outs = [dict(reward=torch.empty(K, N, dtype=torch.float64, device=dev), done=torch.empty(K, N, dtype=torch.uint8, device=dev),
                     soc_trace=torch.empty(K, N, dtype=torch.float64, device=dev)) for _ in range(sets)]

        def run(n):
            for j in range(n):
                if eng.current_step + K > eng.layout.final_step:
                    eng.reset(want_obs=False)
                eng.step_k(a[j & 3], reward=True, done=True, soc_trace=True, out=outs[j % sets])
        run(20000 // K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L = 49152 // K
        e0.record(); run(L); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / L * 1e3
        print(f"K={K:3d} output sets={sets}: {us:6.1f} us per launch = {us * 64 / K:6.2f} us per 64 steps   {N * K / us / 1e3:6.2f} G env-steps/s")
        del a, outs
        torch.cuda.empty_cache()
