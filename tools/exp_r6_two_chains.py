"""Round 6: the Gym single step as TWO dependent launch chains (mgx_set_shards(2)) issued from two host threads (launch workers,
include/mgx.h mgx_set_launch_threads) against one chain from one thread.  N = 100 000 template-4 grids, factorised series.
  python tools/exp_r6_two_chains.py [N] [steps]
Legs: mgx_step_many (64 steps per call) and env.step from a Python loop (bound step), x {1 shard, 2 shards caller-issued,
2 shards with launch threads, 4 shards with launch threads}, with and without H = 0 observation rows; every leg's rewards ==
the one-chain leg's."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from pymgrid_amd import BatchedMicrogridEnv  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
CH = 64


def make(observations):
    b = generate(N, n_steps=8760, seed=42, arch="genset+battery", device=dev, series="factorised")
    env = BatchedMicrogridEnv(b, observations=observations, reuse_outputs=4)
    env._state0 = {k: b.cols[k].clone() for k in ("charge", "soc", "gen_status")}
    return env


def bytes_per_step(env, rows):
    L = env.layout
    return (L.bytes_per_step() - 1 + 2 + (8 * L.obs_dim if rows else 0)) * N


gen = torch.Generator(device=dev); gen.manual_seed(7)
pool = torch.rand(4, CH, N, 3, dtype=torch.float64, device=dev, generator=gen)
views = [[pool[j][k] for k in range(CH)] for j in range(4)]


def run_leg(env, mode, shards, threads, rows):
    eng = env.engine
    for k, v in env._state0.items():           # reset() moves the counter only (base_module.py:65-77): put the dynamic state back
        env.batch.cols[k].copy_(v)
    env.reset(0)
    env.set_shards(shards)
    eng.set_launch_threads(2 if threads else 0)
    outs = [dict(reward=torch.empty(CH, N, dtype=torch.float64, device=dev)) for _ in range(4)]
    if rows and mode == "many":
        for o in outs:
            o["obs"] = torch.empty(CH, N, env.layout.obs_dim, dtype=torch.float64, device=dev)
    rounds = STEPS // CH

    def go(n):
        for r in range(n):
            if mode == "many":
                eng.step_many(pool[r % 4], normalized=True, out=outs[r % 4], done=False, want_obs=rows)
            else:
                step = env.step
                for a in views[r % 4]:
                    step(a)

    eng.fork()
    go(8)
    eng.join(); torch.cuda.synchronize(); eng.fork()
    streams = eng.shard_streams() or [torch.cuda.current_stream(dev)]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in streams]
    t0 = time.perf_counter()
    for (e0, _), s in zip(ev, streams):
        e0.record(s)
    go(rounds)
    t_issue = time.perf_counter() - t0
    for (_, e1), s in zip(ev, streams):
        e1.record(s)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    eng.join()
    gpu = max(a.elapsed_time(b) for a, b in ev) * 1e-3
    n = rounds * CH
    # a checksum of the last round's rewards (many) / the final state (env): the legs must agree bit for bit
    torch.cuda.synchronize()
    chk = env.batch.cols["charge"].clone()
    env.set_shards(1)
    us = gpu / n * 1e6
    frac = bytes_per_step(env, rows) / (gpu / n) / 8e12
    print(f"{mode:5s} rows={int(rows)} shards={shards} threads={int(threads)}: {us:6.2f} us/env-step (gpu)  {wall / n * 1e6:6.2f} (wall)  "
          f"host issue {t_issue / n * 1e6:5.2f} us  frac {frac:.3f}", flush=True)
    return chk


TRACE = len(sys.argv) > 3 and sys.argv[3] == "trace"       # under rocprofv3: the one-call legs only (kernels tell apart by launch size)
for rows in ((False,) if TRACE else (False, True)):
    env = make(rows)
    for mode in (("many",) if TRACE else ("many", "env")):
        ref = None
        for shards, threads in (((1, False), (2, True), (4, True)) if TRACE else ((1, False), (2, False), (2, True), (4, True), (1, False))):
            chk = run_leg(env, mode, shards, threads, rows)
            if ref is None:
                ref = chk
            assert torch.equal(ref, chk), "state differs between launch shapes"
    env.close()
print("all legs left the same battery charge as the one-chain leg")
