#!/usr/bin/env python3
"""Where the host time of a closed Gym loop goes (N = 100 000, H = 0): policy ops alone, env.step alone, both; cProfile of env.step."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
env = BatchedMicrogridEnv(generate(N, n_steps=8760, seed=42, arch="genset+battery", device=dev, series="factorised"))
g = torch.Generator(device=dev); g.manual_seed(5)
W = torch.randn(env.layout.obs_dim, env.layout.action_dim, dtype=torch.float64, device=dev, generator=g)
obs = env.reset()
a = torch.sigmoid(obs @ W)
sync = lambda: torch.cuda.synchronize(dev)


def timed(fn, n=2000, warm=200):
    for _ in range(warm):
        fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = time.perf_counter() - t0
    sync()
    return (time.perf_counter() - t0) / n * 1e6, t_issue / n * 1e6


state = {"obs": obs}


def loop():
    state["obs"] = env.step(torch.sigmoid(state["obs"] @ W))[0]


def room():
    if env.engine.current_step > 6000:
        env.reset()


print("policy ops alone (matmul + sigmoid)      wall %.2f us  host issue %.2f us" % timed(lambda: torch.sigmoid(obs @ W)))
env.reset()
print("env.step alone (same action tensor)      wall %.2f us  host issue %.2f us" % timed(lambda: env.step(a)))
env.reset()
print("closed loop                              wall %.2f us  host issue %.2f us" % timed(loop))
env.reset()
e = env.engine
out = {}
print("engine.step (reward only, no obs)        wall %.2f us  host issue %.2f us" % timed(lambda: e.step(a, want_obs=False)))
env.reset()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    env.step(a)
pr.disable()
sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
