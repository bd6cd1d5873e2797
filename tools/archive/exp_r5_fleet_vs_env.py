#!/usr/bin/env python3
"""Round 5: the same 99 999-grid batch of ONE layout stepped (views contract: no refills, the step writes its 6 state columns)
   env    BatchedMicrogridEnv.step          -> step_kernel<7> (KArgs by value in the kernarg segment)
   fleet  BucketedFleet of that one bucket  -> fleet_step_kernel (KArgs behind a pointer, run-time flags)
   fleet3 the config-5 mix (three buckets)  -> fleet_step_kernel
for rocprofv3 --kernel-trace --stats: kernel durations are what is compared (both loops are host-paced)."""
import sys
import torch
sys.path.insert(0, ".")
from pymgrid_amd import BatchedMicrogridEnv
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import BucketedFleet

mode = sys.argv[1]
dev = torch.device("cuda:0")
n = 3000
if mode == "env":
    b = generate(99999, n_steps=8760, seed=45, arch="genset+battery+grid", horizon=24, device=dev, series="factorised")
    env = BatchedMicrogridEnv(b, obs_views=True, reuse_outputs=96)
    a = torch.rand(99999, env.layout.action_dim, dtype=torch.float64, device=dev)
    env.reset()
    for _ in range(n):
        env.step(a)
else:
    if mode == "fleet":
        batches = [generate(99999, n_steps=8760, seed=45, arch="genset+battery+grid", horizon=24, device=dev, series="factorised")]
    else:
        batches = [generate(33333, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised")
                   for k, a in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
    fleet = BucketedFleet.from_batches(batches, obs_views=True, reuse_outputs=96)
    acts = [torch.rand(e.layout.n_grids, e.layout.action_dim, dtype=torch.float64, device=dev) for e in fleet.envs]
    fleet.reset()
    for _ in range(n):
        fleet.step(acts)
torch.cuda.synchronize()
