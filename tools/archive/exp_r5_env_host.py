#!/usr/bin/env python3
"""Round 5: what one Gym step costs from a Python loop at N = 100 000 (Template-4, H = 0, factorised series) -- the bound step
(mgx_env_bind / mgx_env_step: one C call, outputs in rotating buffers) against the per-call bookkeeping of rounds 1-4, without and
with observation rows, and the C-side cadence (mgx_step_many) beside them.  Wall time per step (the loop is host- or kernel-paced,
whichever is slower) and GPU time per step (HIP events)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000


def timeit(fn, n=4000, warm=1500):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, e0.elapsed_time(e1) / n * 1e3


def host_only(fn, n=20000):
    """the host's share: the same calls with the GPU idle-waited out of the picture is not possible; instead time the call
    stream while the queue is short (sync every 16 calls)"""
    t = 0.0
    for _ in range(n // 16):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(16):
            fn()
        t += time.perf_counter() - t0
    return t / (n // 16 * 16) * 1e6


for label, kw, cls in (("bound, no rows", dict(reuse_outputs=4, observations=False), BatchedMicrogridEnv),
                       ("per-call, no rows", dict(reuse_outputs=0, observations=False), BatchedMicrogridEnv),
                       ("bound, rows H=0", dict(reuse_outputs=4), BatchedMicrogridEnv),
                       ("per-call, rows H=0", dict(reuse_outputs=0), BatchedMicrogridEnv),
                       ("bound discrete, rows H=0", dict(reuse_outputs=4, remove_redundant_gensets=False), DiscreteBatchedMicrogridEnv)):
    env = cls(generate(N, n_steps=8760, seed=42, arch="genset+battery", device=dev, series="factorised"), **kw)
    a = env.sample_action()
    env.reset()

    def step():
        if env.current_step >= 8700:
            env.reset()
        env.step(a)
    w, g = timeit(step)
    h = host_only(step)
    print(f"{label:28s} bound={env._fp is not None!s:5s}: {w:6.2f} us wall  {g:6.2f} us gpu per env.step   host alone {h:5.2f} us", flush=True)
    env.close()

# the C-side cadence: 64 single-step launches per call
from pymgrid_amd import StepEngine  # noqa: E402
e = StepEngine(generate(N, n_steps=8760, seed=42, arch="genset+battery", device=dev, series="factorised"))
acts = torch.rand(64, N, 3, dtype=torch.float64, device=dev)
out = dict(reward=torch.empty(64, N, dtype=torch.float64, device=dev))


def many():
    if e.current_step + 64 > 8700:
        e.reset(want_obs=False)
    e.step_many(acts, out=out, done=False)
w, g = timeit(many, n=200, warm=60)
print(f"{'mgx_step_many (64 per call)':28s}            : {w / 64:6.2f} us wall  {g / 64:6.2f} us gpu per env-step", flush=True)
e.close()
