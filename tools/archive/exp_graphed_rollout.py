#!/usr/bin/env python3
"""GPU experiment: policy-in-the-loop stepping, eager Python loop vs GraphedRollout (one HIP graph per 16 steps)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv, GraphedRollout  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
S = 16
for N in (1_000, 10_000, 100_000):
    for H in (0, 24):
        def make():
            return BatchedMicrogridEnv(generate(N, n_steps=4000, seed=6, arch="genset+battery", horizon=H, device=dev),
                                       obs_dtype=torch.float32, action_dtype=torch.float32, obs_prefetch=0)
        env = make()
        D, A = env.layout.obs_dim, env.layout.action_dim
        W1 = torch.randn(D, 64, device=dev) * 0.2
        W2 = torch.randn(64, A, device=dev) * 0.2

        def policy(obs):
            return torch.sigmoid(torch.relu(obs @ W1) @ W2)
        obs = env.reset()
        for _ in range(64):
            obs, _, _, _ = env.step(policy(obs))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1024):
            obs, _, _, _ = env.step(policy(obs))
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 1024 * 1e6
        env.close()
        env = make()
        roll = GraphedRollout(env, policy, S)
        for _ in range(8):
            roll.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(64):
            roll.run()
        torch.cuda.synchronize()
        graphed = (time.perf_counter() - t0) / (64 * S) * 1e6
        print(f"N={N:7d} H={H:2d} D={D:3d}: policy(2-layer fp32)+env.step  eager {eager:7.2f} us/step   graphed {graphed:7.2f} us/step   "
              f"x{eager / graphed:4.1f}   {N / graphed / 1e3:6.2f} G env-steps/s")
        roll.close(); env.close()
