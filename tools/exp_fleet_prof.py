#!/usr/bin/env python3
"""Step the config-5 fleet (99 999 grids, H = 24, factorised series) under one observation contract -- for rocprofv3:
   rocprofv3 --kernel-trace --stats -d out -- python tools/exp_fleet_prof.py views|rows|rows_rowmajor [float64|float32] [steps] [ring depth K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import BucketedFleet

contract = sys.argv[1] if len(sys.argv) > 1 else "views"
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "float32" else torch.float64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0")
batches = [generate(33333, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised")
           for k, a in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
# rows: the fleet's default ring layout (column-major blocks); rows_rowmajor: obs_layout="rows" (contiguous [N, D] rows)
kw = dict(obs_views=True) if contract == "views" else dict(obs_prefetch=K, obs_layout="rows" if contract == "rows_rowmajor" else None)
fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, reuse_outputs=3 * K, **kw)
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = [torch.rand(33333, e.layout.action_dim, dtype=torch.float64, device=dev, generator=gen) for e in fleet.envs]
fleet.reset()
for _ in range(steps):
    fleet.step(acts)
torch.cuda.synchronize()
fleet.close()
