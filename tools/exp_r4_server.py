#!/usr/bin/env python3
"""Round 4: the resident step server against per-step launches, N = 100 000 Template-4 grids (float32 observations / actions).
  (a) env.step ALONE: controls already on the device, steps released by host stores (immediate) in bursts of the ring depth
  (b) stream-ordered posts (hipStreamWriteValue32 per step), no consumer
  (c) the dependent loop a policy makes: write controls -> post -> wait (hipStreamWaitValue32) -> dependent kernel
  (d) the same loop on per-step launches (engine.step)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 100000))
series = os.environ.get("SERIES", "factorised")
st = torch.cuda.current_stream(dev)


def engine():
    b = generate(N, n_steps=8760, seed=42, arch="genset+battery", device=dev, series=series)
    return StepEngine(b, obs_dtype=torch.float32, action_dtype=torch.float32)


def bench_server(mode, steps=4000, R=8):
    e = engine()
    slots = e.server_start(n_slots=R, max_steps=steps + 64, want_obs=True, immediate=(mode == "immediate"), idle_timeout_ms=500)
    for sl in slots:
        sl["actions"].uniform_()
    scratch = torch.zeros(N, dtype=torch.float32, device=dev)
    st.synchronize()
    t0 = time.perf_counter()
    if mode == "immediate":                       # bursts of R steps, one wait per burst
        for k0 in range(0, steps, R):
            for _ in range(R):
                e.server_post()
            e.server_wait()
    elif mode == "stream":                        # a stream write op per step, one wait per R steps
        for k in range(steps):
            e.server_post()
            if k % R == R - 1:
                e.server_wait()
    else:                                         # dependent loop: consume the slot's observation, produce the next controls
        for k in range(steps):
            sl, nx = slots[k % R], slots[(k + 1) % R]
            e.server_post()
            e.server_wait()
            torch.sigmoid(sl["obs"][:, :3], out=nx["actions"])
    st.synchronize()
    wall = time.perf_counter() - t0
    n = e.server_stop()
    e.close()
    return wall / steps * 1e6, n


def bench_launches(dependent, steps=4000):
    e = engine()
    from pymgrid_amd import BatchedMicrogridEnv
    env = BatchedMicrogridEnv(e.batch, obs_dtype=torch.float32, action_dtype=torch.float32, reuse_outputs=8)
    obs = env.reset()
    a = torch.rand(N, 3, dtype=torch.float32, device=dev)
    for _ in range(200):
        env.step(a)
    st.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if dependent:
            a = torch.sigmoid(obs[:, :3])
        obs = env.step(a)[0]
    st.synchronize()
    wall = time.perf_counter() - t0
    env.close()
    return wall / steps * 1e6


print(f"N = {N}, series = {series}, float32 observations / actions")
for rep in range(2):
    for mode in ("immediate", "stream", "dependent"):
        us, n = bench_server(mode)
        print(f"server  {mode:10s}: {us:6.2f} us per env-step  ({n} steps served)", flush=True)
    print(f"launches alone     : {bench_launches(False):6.2f} us per env-step", flush=True)
    print(f"launches dependent : {bench_launches(True):6.2f} us per env-step (sigmoid policy kernel between steps)", flush=True)
