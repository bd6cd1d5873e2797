#!/usr/bin/env python3
"""Which part of an auto-reset step costs what (N = 100 000, D = 156, rolling windows with rings K = 16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymgrid_amd.generator import generate  # noqa: E402
from pymgrid_amd.hetero import PerGridWindowEnv  # noqa: E402

dev = torch.device("cuda:0")
N = 100_000
w = PerGridWindowEnv(generate(N, n_steps=8760, seed=1, arch="genset+battery+grid", horizon=24, device=dev), trajectory_length=168,
                     auto_reset=True)
a = w.env.sample_action()
env, e = w.env, w.env.engine
none = torch.zeros(N, dtype=torch.uint8, device=dev)
lengths = torch.full((N,), 168, dtype=torch.int32, device=dev)


def timed(step, n=320):
    w.reset()
    for _ in range(64):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def full():
    w.step(a)


def step_only():
    env.step(a)


def step_gather():
    _, _, done, _ = env.step(a)
    e._call(e._lib.mgx_reset_grids_random, done.data_ptr(), 0, 168, e._window_start.data_ptr(), lengths.data_ptr(), e._window_t0.data_ptr())


def step_patch():
    _, _, done, _ = env.step(a)
    e._call(e._lib.mgx_patch_windows, done.data_ptr(), env.obs_prefetch, env._ring.data_ptr(), env._ring_pos, 0, env._restart_acc.data_ptr())


def step_gather_nobody():
    env.step(a)
    e._call(e._lib.mgx_reset_grids_random, none.data_ptr(), 0, 168, e._window_start.data_ptr(), lengths.data_ptr(), e._window_t0.data_ptr())


for name, fn in (("env.step in rolling mode with rings (no restarts)", step_only), ("+ restart kernel, nobody restarts", step_gather_nobody),
                 ("+ restart kernel (finished grids restart)", step_gather), ("+ patch kernel only (no restart)", step_patch),
                 ("full auto-reset step", full)):
    print(f"{name:55s} {timed(fn):7.1f} us/step")
