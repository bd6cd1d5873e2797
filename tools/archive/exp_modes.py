#!/usr/bin/env python3
"""GPU experiment: time the single-step kernel in its output modes (core / +obs / +log / H=24 observations) and the
observe kernel; report achieved algorithmic GB/s.  Usage: python tools/exp_modes.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dev = torch.device("cuda:0")


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


for arch, H in (("genset+battery", 0), ("genset+battery", 24), ("genset+battery+grid", 24)):
    b = generate(N, n_steps=600, seed=1, arch=arch, horizon=H, device=dev)
    eng = StepEngine(b)
    L = eng.layout
    a = torch.rand(N, L.action_dim, dtype=torch.float64, device=dev)
    out = dict(reward=torch.empty(N, dtype=torch.float64, device=dev), done=torch.empty(N, dtype=torch.uint8, device=dev),
               obs=torch.empty(N, L.obs_dim, dtype=torch.float64, device=dev),
               log=torch.empty(eng.log_dim, N, dtype=torch.float64, device=dev))

    def run(obs, log):
        def f():
            if eng.current_step >= 590:
                eng.reset(want_obs=False)
            eng.step(a, want_obs=obs, want_log=log, out=out)
        return f
    print(f"--- {arch} H={H} N={N} D={L.obs_dim} L={eng.log_dim}")
    for name, obs, log in (("core", False, False), ("obs", True, False), ("log", False, True), ("obs+log", True, True)):
        us = timeit(run(obs, log))
        B = L.bytes_per_step(log=log, obs=obs)
        print(f"step {name:8s} {us:8.2f} us/step   {B:5d} B/grid  {B * N / us / 1e3:8.1f} GB/s   {N / us / 1e3:6.2f} G env-steps/s")
    us = timeit(lambda: eng.observe(out=out["obs"]))
    print(f"observe        {us:8.2f} us          {8 * L.obs_dim:5d} B/grid written {8 * L.obs_dim * N / us / 1e3:8.1f} GB/s")
    eng.close()
