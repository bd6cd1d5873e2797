#!/usr/bin/env python3
"""GPU experiment: bench.py's heterogeneous Gym-step leg in isolation, repeated (order / warm-up effects)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for rep in range(3):
    out = bench.hetero_gym_steps(100_000, dev, 0, 1, 256)
    print(rep, {k: round(v["us_per_step"], 1) for k, v in out.items() if isinstance(v, dict)})
