#!/usr/bin/env python3
"""Does the relative placement of the fused kernel's three streams (actions read, reward / SoC written) in memory change its
rate?  All buffers are carved out of ONE allocation with controlled byte offsets between them; N = 100 000 factorised grids,
K = 64, two shards, 4 rotating buffer sets (as bench.py).   python tools/exp_buffer_skew.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N, T, K, SETS = 100_000, 8760, 64, 4
b = generate(N, n_steps=T, seed=42, arch="genset+battery", device=dev, series="factorised")
eng = StepEngine(b)
eng.set_shards(2)
A = 3
act_bytes, out_bytes = K * N * A * 8, K * N * 8
MB2 = 2 << 20
slot = ((max(act_bytes, out_bytes) + MB2 - 1) // MB2 + 1) * MB2          # every buffer gets a 2 MiB-aligned slot + room for the skew
arena = torch.empty(SETS * 3 * slot + (64 << 20), dtype=torch.uint8, device=dev)
base = (-arena.data_ptr()) % MB2                                          # first 2 MiB boundary inside the arena
gen = torch.Generator(device=dev); gen.manual_seed(7)
src = torch.rand(K, N, A, dtype=torch.float64, device=dev, generator=gen)


def carve(off, shape):
    n = 1
    for s in shape:
        n *= s
    return arena[off: off + n * 8].view(torch.float64).view(*shape)


def run(skew_r, skew_s, skew_set, rounds=400, warm=150):
    sets = []
    for j in range(SETS):
        o = base + j * 3 * slot + j * skew_set
        acts = carve(o, (K, N, A)); acts.copy_(src)
        rew = carve(o + slot + skew_r, (K, N))
        soc = carve(o + 2 * slot + skew_s, (K, N))
        sets.append((acts, dict(reward=rew, soc_trace=soc)))
    eng.reset(want_obs=False)
    state = {"r": 0}

    def fn():
        if eng.current_step + K > eng.layout.final_step:
            eng.reset(want_obs=False)
        a, o = sets[state["r"] % SETS]
        eng.step_k(a, out=o, reward=True, done=False, soc_trace=True)
        state["r"] += 1
    eng.fork()
    for _ in range(warm):
        fn()
    eng.join(); torch.cuda.synchronize(dev); eng.fork()
    t0 = time.perf_counter()
    for _ in range(rounds):
        fn()
    eng.join(); torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / rounds * 1e6


print(f"arena base mod 2 MiB = 0; slot = {slot >> 20} MiB; us per 64-step round of {N} grids (two shards), 400 rounds each")
for _ in range(2):
    for skew_r, skew_s, skew_set in ((0, 0, 0), (0, 128, 0), (0, 256, 0), (0, 512, 0), (0, 1024, 0), (0, 2048, 0), (0, 4096, 0),
                                     (0, 8192, 0), (0, 65536, 0), (0, 1 << 20, 0), (4096, 8192, 0), (1024, 2048, 0),
                                     (0, 0, 4096), (0, 0, 65536), (2048, 4096, 1024), (256, 512, 128)):
        us = run(skew_r, skew_s, skew_set)
        print(f"reward +{skew_r:7d} B  soc +{skew_s:7d} B  per set +{skew_set:6d} B : {us:6.2f} us  "
              f"{N * K / us / 1e3:6.1f} G env-steps/s", flush=True)
