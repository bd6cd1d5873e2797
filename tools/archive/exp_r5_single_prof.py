#!/usr/bin/env python3
"""Round 5: 4 000 bound Gym steps (Template-4, H = 0, factorised, N grids) without / with rows -- for rocprofv3 --kernel-trace."""
import sys
import torch
sys.path.insert(0, ".")
from pymgrid_amd import BatchedMicrogridEnv
from pymgrid_amd.generator import generate
N = int(sys.argv[1]); rows = sys.argv[2] == "rows"
dev = torch.device("cuda:0")
env = BatchedMicrogridEnv(generate(N, n_steps=8760, seed=42, arch="genset+battery", device=dev, series="factorised"), reuse_outputs=4, observations=rows)
a = torch.rand(N, env.layout.action_dim, dtype=torch.float64, device=dev)
env.reset()
for _ in range(4000):
    env.step(a)
torch.cuda.synchronize()
