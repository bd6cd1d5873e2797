#!/usr/bin/env python3
"""GPU experiment: observation kernels inside a real step loop (t advances, windows slide) for the H=24 layouts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
DT = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
dev = torch.device("cuda:0")
for arch, H in (("genset+battery", 24), ("genset+battery+grid", 24)):
    b = generate(N, n_steps=1200, seed=1, arch=arch, horizon=H, device=dev)
    eng = StepEngine(b, obs_dtype=DT)
    L = eng.layout
    a = torch.rand(N, L.action_dim, dtype=torch.float64, device=dev)
    out = dict(reward=torch.empty(N, dtype=torch.float64, device=dev), done=torch.empty(N, dtype=torch.uint8, device=dev),
               obs=torch.empty(N, L.obs_dim, dtype=DT, device=dev))
    for _ in range(50):
        eng.step(a, want_obs=True, want_log=False, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000):
        eng.step(a, want_obs=True, want_log=False, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 1000 * 1e3
    print(f"{os.path.basename(os.environ.get('MGX_LIB', 'shipped')):22s} {arch:20s} D={L.obs_dim:4d} step+obs {us:7.2f} us/step  {N / us / 1e3:6.2f} G env-steps/s ({DT})")
    eng.close()
