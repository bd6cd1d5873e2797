#!/usr/bin/env python3
"""GPU experiment: does the fused kernel's time per launch drift over a long run (clock / power management)?
Blocks of 32 launches, each timed with HIP events, for one launch sequence and for two shards on two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import StreamShards  # noqa: E402

dev = torch.device("cuda:0")
N, K, B, NB = 100_000, 64, 32, 24
T = K * B * NB + 8


def outs(n):
    return dict(reward=torch.empty(K, n, dtype=torch.float64, device=dev), done=torch.empty(K, n, dtype=torch.uint8, device=dev),
                soc_trace=torch.empty(K, n, dtype=torch.float64, device=dev))


eng = StepEngine(generate(N, n_steps=min(T, 8760), seed=1, device=dev))
a = torch.rand(2, K, N, 3, dtype=torch.float64, device=dev)
o = outs(N)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(NB + 1)]
torch.cuda.synchronize()
ev[0].record()
for b in range(NB):
    for j in range(B):
        if eng.current_step + K > eng.layout.final_step:
            eng.reset(want_obs=False)
        eng.step_k(a[j & 1], reward=True, done=True, soc_trace=True, out=o)
    ev[b + 1].record()
torch.cuda.synchronize()
print("one stream, us per launch per block of 32:", " ".join(f"{ev[b].elapsed_time(ev[b + 1]) / B * 1e3:5.1f}" for b in range(NB)))
eng.close()
sh = StreamShards([generate(N, n_steps=min(T, 8760), seed=1, device=dev, rank=j, world=2) for j in range(2)])
aa = [torch.rand(2, K, N // 2, 3, dtype=torch.float64, device=dev) for _ in range(2)]
oo = [outs(N // 2) for _ in range(2)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(NB + 1)]
torch.cuda.synchronize()
sh.fork()
ev[0].record(sh.streams[0])
for b in range(NB):
    for j in range(B):
        if sh.engines[0].current_step + K > sh.engines[0].layout.final_step:
            sh.reset()
        sh.step_k([x[j & 1] for x in aa], outs=oo, reward=True, done=True, soc_trace=True)
    ev[b + 1].record(sh.streams[0])
sh.join()
torch.cuda.synchronize()
print("two shards, us per round per block of 32: ", " ".join(f"{ev[b].elapsed_time(ev[b + 1]) / B * 1e3:5.1f}" for b in range(NB)))
