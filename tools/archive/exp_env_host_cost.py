#!/usr/bin/env python3
"""Host time per Gym step of the batched envs at a size where the GPU is never the bound (N = 1 000)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N = 1000
for name, cls, H, kw in (("continuous H=0", BatchedMicrogridEnv, 0, {}), ("continuous H=24 ring", BatchedMicrogridEnv, 24, {}),
                         ("continuous H=24 per-step rows", BatchedMicrogridEnv, 24, dict(obs_prefetch=0)),
                         ("discrete H=0", DiscreteBatchedMicrogridEnv, 0, {}), ("discrete H=24 ring", DiscreteBatchedMicrogridEnv, 24, {}),
                         ("continuous H=0 no obs", BatchedMicrogridEnv, 0, dict(observations=False))):
    env = cls(generate(N, n_steps=30000, seed=1, arch="genset+battery+grid", horizon=H, device=dev), **kw)
    a = env.sample_action()
    env.reset()
    for _ in range(2000):
        env.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10000):
        env.step(a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:32s} {1e6 * (t1 - t0) / 10000:6.2f} us of host time per env.step (N = {N})")
    env.close()
