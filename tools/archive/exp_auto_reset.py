#!/usr/bin/env python3
"""Per-grid auto-reset (rolling windows): time per Gym step at N = 100 000, H = 24 (argv[2]), next to the lock-step env."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import PerGridWindowEnv  # noqa: E402

dev = torch.device("cuda:0")
N = 100_000


def timed(step, n=300):
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


series = sys.argv[1] if len(sys.argv) > 1 else "materialised"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 24
print(f"series = {series}, horizon = {H}")
for arch in ("genset+battery", "genset+battery+grid"):
    make = lambda: generate(N, n_steps=8760, seed=1, arch=arch, horizon=H, device=dev, series=series)
    env = BatchedMicrogridEnv(make())
    a = env.sample_action(); env.reset()
    print(f"{arch:20s} lock-step env, ring prefetch K = 16      {timed(lambda: env.step(a)):7.1f} us/step")
    env.close()
    env = BatchedMicrogridEnv(make(), obs_prefetch=0)
    env.reset()
    print(f"{arch:20s} lock-step env, per-step rows             {timed(lambda: env.step(a)):7.1f} us/step")
    env.close()
    for native in (False, True):
        for fo in (False, True):
            w = PerGridWindowEnv(make(), trajectory_length=168, auto_reset=True, final_observation=fo, native=native)
            w.reset()
            how = "in place, restart inside the step kernel" if native else "rolling windows (restart gather + observe)"
            print(f"{arch:20s} auto-reset (168-step episodes), final_observation={int(fo)}, {how:42s} {timed(lambda: w.step(a)):7.1f} us/step")
            w.close()
