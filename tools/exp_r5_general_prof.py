#!/usr/bin/env python3
"""The general path (2 gensets + 2 batteries + 1 grid per microgrid, 100 000 grids) one leg at a time -- for rocprofv3 counter passes:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- python tools/exp_r5_general_prof.py single|kstep|kstep3|rbc|gymrows [steps]
The same shapes as bench.py's general_path_leg: single Gym steps (step_multi_kernel), K = 32 fused steps per launch
(step_k_multi_kernel), Gym steps with 24-hour rows off rings of 32 blocks (step_multi_kernel + obs_windows_k_multi_kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv, StepEngine  # noqa: E402
from pymgrid_amd.generator import generate, widen  # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else "single"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
N = 100_000
gen = torch.Generator(device=dev); gen.manual_seed(3)
if leg in ("single", "kstep", "kstep3", "rbc"):
    n_kind = 3 if leg == "kstep3" else 2            # kstep3: three gensets + three batteries (the run-time-count K-step kernel)
    gb = widen(generate(N, n_steps=1200, seed=42, arch="genset+battery+grid", device=dev), n_genset=n_kind, n_battery=n_kind, n_grid=1)
    ge = StepEngine(gb)
    A = ge.layout.action_dim
    if leg == "single":
        a1 = torch.rand(N, A, dtype=torch.float64, device=dev, generator=gen)
        r1 = torch.empty(N, dtype=torch.float64, device=dev)
        for _ in range(steps):
            ge.step(a1, want_obs=False, want_log=False, out=dict(reward=r1), want_done=False)
    elif leg == "rbc":               # RuleBasedControl's fixed lists through mgx_rollout_lists (rollout_multi_small_kernel), 32 steps per launch
        from pymgrid_amd.priority_list import get_instance_priority_lists, lists_array
        from pymgrid_amd.rbc import default_instance_priority_ids
        L = ge.layout
        pls = get_instance_priority_lists(L.n_genset, L.n_battery, L.n_grid, (), L.grid_before_battery)
        tab = torch.as_tensor(lists_array(pls), device=dev).contiguous()
        ids = torch.from_numpy(default_instance_priority_ids(gb, pls)).to(dev).to(torch.int32).contiguous()
        out = {"reward": torch.empty(32, N, dtype=torch.float64, device=dev)}
        for _ in range(max(2, steps // 32)):
            ge.rollout_lists(ids, tab, 32, reward=True, out=out)
    else:
        aK = torch.rand(32, N, A, dtype=torch.float64, device=dev, generator=gen)
        for _ in range(max(2, steps // 32)):
            ge.step_k(aK, normalized=True, reward=True, soc_trace=False)
    torch.cuda.synchronize()
    ge.close()
else:
    base = generate(N, n_steps=700, seed=42, arch="genset+battery+grid", horizon=24, device=dev)
    env = BatchedMicrogridEnv(widen(base, n_genset=2, n_battery=2, n_grid=1), obs_prefetch=32, reuse_outputs=96)
    ao = torch.rand(N, env.layout.action_dim, dtype=torch.float64, device=dev, generator=gen)
    env.reset()
    for _ in range(min(steps, 700 - 24 - 8)):
        env.step(ao)
    torch.cuda.synchronize()
    env.close()
