#!/usr/bin/env python3
"""What sets the duration of ONE single-step kernel (step_kernel<3>, N = 100 000)?  Runs one variant per process so that a
rocprofv3 kernel trace of the process holds that variant's dispatches only (tools/exp_step_variants.sh).
   python tools/exp_step_variants.py <variant> [series]
variants: rotate  = action / reward rows rotate through 4 x 64 rows (what bench.py does), Python loop of mgx_step
          same    = the same action row and reward row every step, Python loop
          many    = rotate, but one mgx_step_many call per 64 steps (deep queue)
          paced   = same, with the host waiting for the GPU after every launch (kernel alone on an idle chip)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

variant = sys.argv[1]
series = sys.argv[2] if len(sys.argv) > 2 else "factorised"
dev = torch.device("cuda:0")
N, T, K = 100_000, 8760, 64
b = generate(N, n_steps=T, seed=42, arch="genset+battery", device=dev, series=series)
eng = StepEngine(b)
gen = torch.Generator(device=dev); gen.manual_seed(7)
pool = torch.rand(4, K, N, 3, dtype=torch.float64, device=dev, generator=gen)
rew = torch.empty(4, K, N, dtype=torch.float64, device=dev)
don = torch.empty(4, K, N, dtype=torch.uint8, device=dev)
eng.reset(want_obs=False)
sync = lambda: torch.cuda.synchronize(dev)


def run(rounds):
    for r in range(rounds):
        if eng.current_step + K > eng.layout.final_step:
            eng.reset(want_obs=False)
        p = r % 4
        if variant == "many":
            eng.step_many(pool[p], out=dict(reward=rew[p], done=don[p]))
            continue
        for k in range(K):
            kk = (p, k) if variant == "rotate" else (0, 0)
            eng.step(pool[kk], out=dict(reward=rew[kk], done=don[kk]), want_obs=False)
            if variant == "paced":
                sync()


run(30); sync()
t0 = time.perf_counter()
R = 60
run(R); sync()
print(f"{variant:7s} {series:12s}: {(time.perf_counter() - t0) / R / K * 1e6:6.2f} us per env-step (wall)")
