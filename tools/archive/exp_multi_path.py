#!/usr/bin/env python3
"""General kernels (several modules of a kind per microgrid): time per single step at N = 100 000 next to the
single-instance kernel, for a few module mixes.  Columns of the extra instances are scaled copies of the generator's."""
import os
import sys
import dataclasses

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import MicrogridBatch, StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")
base = generate(N, n_steps=600, seed=3, arch="genset+battery+grid", device=dev)


def widen(n_gen, n_bat, n_grid, n_load, n_pv):
    L = dataclasses.replace(base.layout, n_genset=n_gen, n_battery=n_bat, n_grid=n_grid, n_load=n_load, n_pv=n_pv)
    cols = {}
    for k, v in base.cols.items():
        n = n_gen if k.startswith("gen_") else n_bat if (k.startswith("bat_") or k in ("charge", "soc")) else \
            n_grid if k.startswith("grid_m") or k.startswith("grid_c") else 0
        if k in ("grid_ts", "grid_lo", "grid_hi"):
            reps = [v * (1.0 + 0.0 * j) for j in range(n_grid)]
            cols[k] = (torch.stack(reps, dim=1 if k == "grid_ts" else 0) if n_grid > 1 else v).contiguous()
        elif k in ("load_ts", "pv_ts"):
            m = n_load if k == "load_ts" else n_pv
            cols[k] = (torch.stack([v / m] * m, dim=1) if m > 1 else v).contiguous()
        elif k in ("load_lo", "load_hi", "pv_lo", "pv_hi"):
            m = n_load if k.startswith("load") else n_pv
            cols[k] = (torch.stack([v / m] * m, dim=0) if m > 1 else v).contiguous()
        elif n > 1:
            cols[k] = torch.stack([v if v.dtype != torch.float64 else v * (1.0 + 0.05 * j) for j in range(n)], dim=0).contiguous()
        else:
            cols[k] = v.clone()
    return MicrogridBatch(L, cols)


for mix in ((1, 1, 1, 1, 1), (1, 1, 1, 2, 2), (2, 2, 1, 1, 1), (2, 2, 2, 2, 2), (4, 4, 2, 3, 3), (8, 8, 8, 8, 8)):
    b = widen(*mix)
    eng = StepEngine(b)
    a = torch.rand(N, eng.action_dim, dtype=torch.float64, device=dev)
    for want_log in (False, True):
        for _ in range(20):
            eng.reset(0, want_obs=False); eng.step(a, want_obs=False, want_log=want_log)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.reset(0, want_obs=False)
        e0.record()
        for _ in range(500):
            eng.step(a, want_obs=False, want_log=want_log)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 500 * 1e3
        kind = "general" if b.layout.multi else "single-instance"
        print(f"gensets {mix[0]} batteries {mix[1]} grids {mix[2]} loads {mix[3]} pvs {mix[4]}  log={int(want_log)}  {kind:15s} "
              f"{us:8.2f} us/step  {N / us / 1e3:6.2f} G env-steps/s  (A = {eng.action_dim}, L = {eng.log_dim})")
    # the K-step loop of the general path (mgx_step_k on such layouts) / the fused kernel of the single-instance layout
    K = 64
    acts = torch.rand(K, N, eng.action_dim, dtype=torch.float64, device=dev)
    out = dict(reward=torch.empty(K, N, dtype=torch.float64, device=dev))
    for _ in range(3):
        eng.reset(0, want_obs=False); eng.step_k(acts, out=out, reward=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.reset(0, want_obs=False)
    e0.record()
    for _ in range(8):
        eng.step_k(acts, out=out, reward=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (8 * K) * 1e3
    print(f"gensets {mix[0]} batteries {mix[1]} grids {mix[2]} loads {mix[3]} pvs {mix[4]}  fused K={K}      "
          f"{us:8.2f} us/step  {N / us / 1e3:6.2f} G env-steps/s")
    eng.close()
