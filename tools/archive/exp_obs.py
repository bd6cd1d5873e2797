#!/usr/bin/env python3
"""GPU experiment: time mgx_observe for the H=24 layouts (which MGX_LIB variant is loaded is up to the caller)."""
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
    b = generate(N, n_steps=600, seed=1, arch=arch, horizon=H, device=dev)
    eng = StepEngine(b, obs_dtype=DT)
    obs = torch.empty(N, eng.layout.obs_dim, dtype=DT, device=dev)
    for _ in range(20):
        eng.observe(out=obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        eng.observe(out=obs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    D = eng.layout.obs_dim
    print(f"{os.path.basename(os.environ.get('MGX_LIB', 'shipped')):22s} {arch:20s} D={D:4d} observe {us:7.2f} us  {obs.element_size() * D * N / us / 1e3:7.1f} GB/s written ({DT})")
    eng.close()
