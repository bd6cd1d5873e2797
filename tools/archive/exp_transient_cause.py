#!/usr/bin/env python3
"""GPU experiment: is the start-up transient of the fused kernel tied to the FIRST PASS over the series (cold address
translation) or to elapsed time (clocks)?  Same kernel, series of 2 048 rows (a pass = 32 launches) vs 8 760 rows (136)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N, K, B, NB = 100_000, 64, 16, 24
for T in (2048, 8760, 2048):
    eng = StepEngine(generate(N, n_steps=T, seed=1, device=dev))
    a = torch.rand(4, K, N, 3, dtype=torch.float64, device=dev)
    outs = [dict(reward=torch.empty(K, N, dtype=torch.float64, device=dev), done=torch.empty(K, N, dtype=torch.uint8, device=dev),
                 soc_trace=torch.empty(K, N, dtype=torch.float64, device=dev)) for _ in range(4)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(NB + 1)]
    torch.cuda.synchronize()
    import time; time.sleep(0.5)                      # let the GPU idle: every series starts from the same clock state
    ev[0].record()
    n = 0
    for b in range(NB):
        for j in range(B):
            if eng.current_step + K > eng.layout.final_step:
                eng.reset(want_obs=False)
            eng.step_k(a[n & 3], reward=True, done=True, soc_trace=True, out=outs[n & 3]); n += 1
        ev[b + 1].record()
    torch.cuda.synchronize()
    print(f"T={T:5d} (pass = {T // K:3d} launches), us per launch per block of {B}:", " ".join(f"{ev[b].elapsed_time(ev[b + 1]) / B * 1e3:5.1f}" for b in range(NB)))
    eng.close()
    del a, outs
    torch.cuda.empty_cache()
