#!/usr/bin/env python3
"""GPU experiment: eager single-step launches vs a replayed HIP graph of 64 steps (device-resident counter)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import StepEngine
from pymgrid_amd.generator import generate
dev = torch.device("cuda:0")
S = 64
for N in (1000, 10000, 100000):
    eng = StepEngine(generate(N, n_steps=8000, seed=1, device=dev))
    a = torch.rand(S, N, 3, dtype=torch.float64, device=dev)
    out = dict(reward=torch.empty(N, dtype=torch.float64, device=dev), done=torch.empty(N, dtype=torch.uint8, device=dev))
    def eager(n):
        for k in range(n):
            eng.step(a[k % S], want_obs=False, out=out)
    eager(64); torch.cuda.synchronize()
    t0 = time.perf_counter(); eager(1024); torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 1024 * 1e6
    eng.reset(want_obs=False)
    eng.use_device_counter(True)
    side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for k in range(S):
                eng.step(a[k], want_obs=False, out=out)
    torch.cuda.current_stream(dev).wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / (16 * S) * 1e6
    print(f"N={N:7d}: eager {te:6.2f} us/step ({N/te/1e3:6.2f} G env-steps/s)   graph replay {tg:6.2f} us/step ({N/tg/1e3:6.2f} G env-steps/s)")
    eng.close()
