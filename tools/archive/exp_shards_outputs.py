#!/usr/bin/env python3
"""GPU experiment: two shards on two streams, 64-step launches -- is their advantage over one launch sequence still there
when the outputs are NOT re-written in place every launch (8 sets of output buffers per shard instead of 1)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import StreamShards  # noqa: E402

dev = torch.device("cuda:0")
N, K, S = 100_000, 64, 2
sh = StreamShards([generate(N, n_steps=8760, seed=1, device=dev, rank=j, world=S) for j in range(S)])
n = N // S
for sets in (1, 8, 1, 8):
    a = [torch.rand(4, K, n, 3, dtype=torch.float64, device=dev) for _ in range(S)]
    outs = [[dict(reward=torch.empty(K, n, dtype=torch.float64, device=dev), done=torch.empty(K, n, dtype=torch.uint8, device=dev),
                  soc_trace=torch.empty(K, n, dtype=torch.float64, device=dev)) for _ in range(S)] for _ in range(sets)]

    def run(rounds):
        for j in range(rounds):
            if sh.engines[0].current_step + K > sh.engines[0].layout.final_step:
                sh.reset()
            sh.step_k([x[j & 3] for x in a], outs=outs[j % sets], reward=True, done=True, soc_trace=True)
    sh.fork()
    run(400)
    sh.join(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sh.fork()
    e0.record(sh.streams[0]); run(768); e1.record(sh.streams[0])
    sh.join(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 768 * 1e3
    print(f"2 shards K=64 output sets={sets}: {us:6.1f} us per round   {N * K / us / 1e3:6.2f} G env-steps/s   frac {378.8e6 / us / 1e6 / 8000:.3f}")
    del a, outs
    torch.cuda.empty_cache()
