#!/usr/bin/env python3
"""GPU experiment: host-side cost per call of engine.step / env.step / fleet.step (tiny N: the GPU is never the bound)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N = 256


def rate(fn, n=5000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for H, pre in ((0, 0), (24, 0), (24, 8)):
    env = BatchedMicrogridEnv(generate(N, n_steps=8000, seed=1, arch="genset+battery+grid", horizon=H, device=dev), obs_prefetch=pre)
    a = torch.rand(N, env.layout.action_dim, dtype=torch.float64, device=dev)
    env.reset()
    out = dict(reward=torch.empty(N, dtype=torch.float64, device=dev), done=torch.empty(N, dtype=torch.uint8, device=dev))

    def eng_step():
        if env.engine.current_step > 7900:
            env.engine.reset(want_obs=False)
        env.engine.step(a, want_obs=False, out=out)

    def env_step():
        if env.engine.current_step > 7900:
            env.reset()
        env.step(a)
    print(f"H={H} prefetch={pre}: engine.step(no obs) {rate(eng_step):6.2f} us   env.step {rate(env_step):6.2f} us")
    if H == 24 and pre == 8:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3000):
            env_step()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    env.close()
denv = DiscreteBatchedMicrogridEnv(generate(N, n_steps=8000, seed=1, arch="genset+battery+grid", horizon=24, device=dev), obs_prefetch=8)
ids = denv.sample_action()
denv.reset()


def d_step():
    if denv.engine.current_step > 7900:
        denv.reset()
    denv.step(ids)
print(f"discrete env.step (H=24, prefetch 8) {rate(d_step):6.2f} us")
