#!/usr/bin/env python3
"""bench.py -- microgrid env-steps/s of the batched step engine (BASELINE.json metric).

Workload (BASELINE.json configs[2]): N = 100 000 generated Template-4 grids (genset + battery + load + pv) per GPU,
T = 8760 hourly rows, synthetic data drawn with the MicrogridGenerator sizing rules (pymgrid_amd/generator.py),
normalised U[0,1) actions.  A "step" is ONE env-step of all N grids of a rank (one pass of the hot path over the
batch).  Everything the timed region reads (columns, series, actions) is resident in HBM before timing starts.

Modes
  fused (default)  K steps are issued as ceil(K / chunk) launches of the K-step kernel (mgx_step_k): parameters and
                   state stay in registers, actions / series rows / per-step outputs (reward, done, SoC) stream.
                   The rank's N grids are stepped as --shards (2) independent shards of N / 2 grids, each with its own
                   engine and HIP stream and never joined between launches (pymgrid_amd.hetero.StreamShards): grids do
                   not interact, and two launch sequences out of phase fill each other's ramp-up / tail gaps (+5..10 %).
                   A roofline "launch" is then one ROUND = one kernel launch per shard stream (all N grids, 64 steps);
                   its duration is a shard stream's cadence (HIP events on that stream over the region / launches).
                   The same kernel as ONE launch sequence over all N grids is reported under "other".
  step             one launch of the single-step kernel (mgx_step) per env-step -- the Gym cadence.
  rbc              rule-based control rolled out on device (mgx_rollout_discrete, one fixed priority list per grid):
                   the control is expanded in-kernel, so there is no action stream at all.
All are timed in every run; --mode picks which one is the headline `value`; the others are reported under "other".

Every mode is preceded by PREWARM_S = 0.1 s of its own launches (untimed set-up: code-object load and the ~15 ms the clocks
need to settle under this load), then the W warm-up steps, then EXACTLY K timed steps between barrier + synchronize.

Launch:  python bench.py [--gpus N --steps K --warmup W]          (N>1: torchrun, one rank per GPU, RCCL only
                                                                    for the final metrics all-reduce)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pymgrid_amd import _lib  # noqa: E402
from pymgrid_amd import distributed as mdist  # noqa: E402
from pymgrid_amd.engine import StepEngine  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402

PREWARM_S = 0.1            # seconds of untimed device pre-warm before each mode's W warm-up steps (see measure())
OUT_SETS = 4               # sets of output buffers each runner cycles through (see Runner)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the GPU needs ~10-20 ms of this load before the time per launch settles (81 -> 75 us over the first ~160
    # launches of one stream, 77 -> 67.7 us over ~250 rounds of two shards: profiles/r01/exp_time_dependence.txt), so the
    # default warm-up is 384 launches (~27 ms) and the timed region 1024 launches (~70 ms)
    ap.add_argument("--steps", type=int, default=65536)
    ap.add_argument("--warmup", type=int, default=24576)
    ap.add_argument("--grids", type=int, default=100_000, help="microgrids PER GPU (weak scaling)")
    ap.add_argument("--rows", type=int, default=8760, help="time-series rows T")
    ap.add_argument("--mode", choices=["fused", "step", "rbc"], default="fused")
    ap.add_argument("--chunk", type=int, default=64, help="env-steps per fused launch")
    ap.add_argument("--shards", type=int, default=2,
                    help="fused mode: independent shards per GPU, one HIP stream each (1: one launch sequence over all grids)")
    ap.add_argument("--arch", default="genset+battery")
    ap.add_argument("--hetero-steps", type=int, default=256, help="timed Gym steps of the heterogeneous H=24 fleet (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


class Runner:
    """Issues env-steps on the current stream; wraps around the series with reset() (a host counter)."""

    def __init__(self, eng, chunk, pool):
        self.eng, self.chunk, self.pool = eng, chunk, pool          # pool: [P, chunk, N, A] actions
        L = eng.layout
        N = L.n_grids
        dev = eng.device
        # OUT_SETS sets of [chunk, N] output buffers, cycled: a launch never rewrites what the previous three wrote, so no
        # output line can still sit in the 256 MB Infinity Cache when it is written again (re-using ONE set makes 48-step
        # launches look 9 % faster than they are from HBM: profiles/r01/exp_chunk_outputs.txt)
        self.out_sets = [dict(reward=torch.empty(chunk, N, dtype=torch.float64, device=dev),
                              done=torch.empty(chunk, N, dtype=torch.uint8, device=dev),
                              soc_trace=torch.empty(chunk, N, dtype=torch.float64, device=dev)) for _ in range(OUT_SETS)]
        self.reward_k = self.out_sets[0]["reward"]
        self.out1 = dict(reward=torch.empty(N, dtype=torch.float64, device=dev),
                         done=torch.empty(N, dtype=torch.uint8, device=dev))
        self.launches = 0
        self.i = 0

    def _room(self, k):
        if self.eng.current_step + k > self.eng.layout.final_step:
            self.eng.reset(want_obs=False)

    def fused(self, steps):
        done = 0
        while done < steps:
            k = min(self.chunk, steps - done)
            self._room(k)
            a = self.pool[self.i % self.pool.shape[0]]
            o = self.out_sets[self.launches % OUT_SETS]
            self.eng.step_k(a[:k], normalized=True, out={name: t[:k] for name, t in o.items()},
                            reward=True, done=True, soc_trace=True)
            self.i += 1; self.launches += 1; done += k

    def rbc(self, steps):
        done = 0
        while done < steps:
            k = min(self.chunk, steps - done)
            self._room(k)
            o = self.out_sets[self.launches % OUT_SETS]
            self.eng.rollout_discrete(self.rbc_ids, self.rbc_table, k, reward=True, done=True, soc_trace=True,
                                      out={name: t[:k] for name, t in o.items()})
            self.launches += 1; done += k

    def single(self, steps):
        for s in range(steps):
            self._room(1)
            a = self.pool[self.i % self.pool.shape[0]][s % self.chunk]
            self.eng.step(a, normalized=True, want_obs=False, want_log=False, out=self.out1)
            self.launches += 1
        self.i += 1


class ShardRunner:
    """The fused modes over S independent shards of the rank's grids (pymgrid_amd.hetero.StreamShards): every shard has
    its own engine, action pool, output buffers and HIP stream; the launch sequences are not joined between launches."""

    def __init__(self, shards, chunk, seed):
        from pymgrid_amd.priority_list import get_priority_lists, table_array
        from pymgrid_amd.rbc import default_priority_ids
        self.shards, self.chunk = shards, chunk
        dev = shards.device
        self.pools, self.outs, self.rbc_ids, self.rbc_tables = [], [], [], []
        for j, eng in enumerate(shards.engines):
            L, n = eng.layout, eng.N
            gen = torch.Generator(device=dev); gen.manual_seed(seed + j)
            self.pools.append(torch.rand(4, chunk, n, L.action_dim, dtype=torch.float64, device=dev, generator=gen))
            self.outs.append([dict(reward=torch.empty(chunk, n, dtype=torch.float64, device=dev),
                                   done=torch.empty(chunk, n, dtype=torch.uint8, device=dev),
                                   soc_trace=torch.empty(chunk, n, dtype=torch.float64, device=dev)) for _ in range(OUT_SETS)])
            lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, False)
            self.rbc_tables.append(table_array(lists))
            self.rbc_ids.append(torch.from_numpy(default_priority_ids(eng.batch, lists, remove_redundant_gensets=False)).to(dev))
        self.launches = 0            # per stream
        self.i = 0

    def _room(self, k):
        e = self.shards.engines[0]
        if e.current_step + k > e.layout.final_step:
            self.shards.reset()

    def fused(self, steps):
        done = 0
        while done < steps:
            k = min(self.chunk, steps - done)
            self._room(k)
            self.shards.step_k([p[self.i % 4][:k] for p in self.pools],
                               outs=[{name: t[:k] for name, t in o[self.launches % OUT_SETS].items()} for o in self.outs],
                               normalized=True, reward=True, done=True, soc_trace=True)
            self.i += 1; self.launches += 1; done += k

    def rbc(self, steps):
        done = 0
        while done < steps:
            k = min(self.chunk, steps - done)
            self._room(k)
            self.shards.rollout_discrete(self.rbc_ids, self.rbc_tables, k,
                                         outs=[{name: t[:k] for name, t in o[self.launches % OUT_SETS].items()} for o in self.outs],
                                         reward=True, done=True, soc_trace=True)
            self.launches += 1; done += k


def _kernel_durations_us(self, rounds):
    """Mean start-to-end time of the shard kernels themselves (HIP events around every launch, a short extra pass after the
    timed region): what rocprofv3 reports as the kernel's duration.  Shorter than the cadence of a round because the S
    concurrent kernels overlap only partly."""
    ev = []
    self.shards.fork()
    for _ in range(rounds):
        k = self.chunk
        self._room(k)
        for j, (eng, st) in enumerate(zip(self.shards.engines, self.shards.streams)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                e0.record(st)
                eng.step_k(self.pools[j][self.i % 4][:k], out=self.outs[j][self.i % OUT_SETS], normalized=True, reward=True,
                           done=True, soc_trace=True)
                e1.record(st)
            ev.append((e0, e1))
        self.i += 1
    self.shards.join()
    torch.cuda.synchronize(self.shards.device)
    return sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev) * 1e3


ShardRunner.kernel_durations_us = _kernel_durations_us


def timed_shards(fn, steps, shards, device):
    """barrier + sync | K steps on every shard stream | join + sync + barrier; returns (wall seconds, GPU seconds =
    the longest shard stream's elapsed time between its own start and stop events)."""
    mdist.barrier()
    torch.cuda.synchronize(device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in shards.streams]
    t0 = time.perf_counter()
    shards.fork()
    for (e0, _), st in zip(ev, shards.streams):
        e0.record(st)
    fn(steps)
    for (_, e1), st in zip(ev, shards.streams):
        e1.record(st)
    shards.join()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    mdist.barrier()
    return t1 - t0, max(e0.elapsed_time(e1) for e0, e1 in ev) * 1e-3


def timed(fn, steps, device):
    """barrier + sync | K steps | sync + barrier; returns (wall seconds, GPU-event seconds)."""
    mdist.barrier()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    fn(steps)
    e1.record()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    mdist.barrier()
    return t1 - t0, e0.elapsed_time(e1) * 1e-3


def hetero_gym_steps(N, dev, rank, world, steps):
    from pymgrid_amd.hetero import BucketedFleet
    per = N // 3
    out = {}
    for name, dt in (("float64_rows", torch.float64), ("float32_rows", torch.float32)):
        batches = [generate(per * world, n_steps=steps + 1100, seed=43 + k, arch=arch, horizon=24, device=dev, rank=rank,
                            world=world) for k, arch in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
        fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=8)
        gen = torch.Generator(device=dev); gen.manual_seed(11 + rank)
        acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev, generator=gen) for e in fleet.envs]
        # warm-up by wall time: the fleet is built on the host while the GPU idles and clocks down, and under this
        # host-paced load the clocks take a few tenths of a second to come back (first leg measured 5x slow with a
        # 512-step warm-up, 1.6x slow with 0.6 s)
        prev, t_end = None, time.perf_counter() + 4.0
        while time.perf_counter() < t_end:                # until two consecutive 1000-step blocks agree within 3 %
            fleet.reset()
            t0 = time.perf_counter()
            for _ in range(1000):
                fleet.step(acts)
            torch.cuda.synchronize(dev)
            cur = time.perf_counter() - t0
            if prev is not None and abs(cur - prev) < 0.03 * prev:
                break
            prev = cur
        fleet.reset()
        for _ in range(64):
            fleet.step(acts)
        wall, _ = timed(lambda k: [fleet.step(acts) for _ in range(k)], steps, dev)
        wall = mdist.max_over_ranks(wall, dev)
        out[name] = {"value": 3 * per * world * steps / wall, "us_per_step": wall / steps * 1e6}
        fleet.close()
        del fleet, batches
        torch.cuda.empty_cache()
    out.update({"grids_per_gpu": 3 * per, "obs_dims": [56, 106, 156], "horizon": 24, "obs_prefetch": 8, "steps": steps,
                "workload": "BASELINE configs[4] mix per GPU: 1/3 genset+battery, 1/3 battery+grid, 1/3 genset+battery+grid; "
                            "Gym step() with observation rows"})
    return out


def cpu_baseline(eng, pool, seconds):
    """The CPU oracle (C restatement of the reference loop, oracle/mgx_oracle.c) timed on this host: a bounded
    sample of the SAME workload (first n grids, their real columns / series / actions)."""
    from oracle import oracle as orc
    orc.build()
    L = eng.layout
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = min(L.n_grids, 65536)
    K = min(pool.shape[1], L.final_step)
    cols = {}
    for k, v in eng.batch.cols.items():
        if k in ("load_ts", "pv_ts"):
            cols[k] = v[:K + 1, :n].contiguous().cpu().numpy()
        elif k == "grid_ts":
            cols[k] = v[:K + 1, :, :n].contiguous().cpu().numpy()
        elif v.dim() == 1:
            a = v[:n].cpu().numpy()
            cols[k] = a.view(np.uint32) if v.dtype == torch.int32 else a
    cols["layout"] = dict(N=n, T=K + 1, horizon=0, final_step=K + 1, has_genset=int(L.has_genset),
                          has_battery=int(L.has_battery), has_grid=int(L.has_grid))
    acts = pool[0][:K, :n].contiguous().cpu().numpy()

    def run(nthreads, budget):
        st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            orc.run_batch(cols, st, 0, K, acts, normalized=True, want_reward=False, nthreads=nthreads)
            done += n * K
        return done / (time.perf_counter() - t0), done
    # per-instance cadence: ONE microgrid object stepped from a Python loop, as a reference user would
    # step `env.step(a)` -- the C restatement does the arithmetic, Python only the call: ~1 s
    one = {k: (v[:, 0] if v.ndim == 2 else (v[:, :, 0] if v.ndim == 3 else v[0])) for k, v in cols.items()
           if isinstance(v, np.ndarray)}
    p1 = dict(load_ts=-one["load_ts"], pv_ts=one["pv_ts"], horizon=0, final_step=K + 1, initial_step=0,
              unbalanced=dict(loss_load_cost=float(one["loss_load_cost"]), overgeneration_cost=float(one["overgeneration_cost"])))
    if L.has_battery:
        p1["battery"] = dict(min_capacity=float(one["bat_min_capacity"]), max_capacity=float(one["bat_max_capacity"]),
                             max_charge=float(one["bat_max_charge"]), max_discharge=float(one["bat_max_discharge"]),
                             efficiency=float(one["bat_efficiency"]), battery_cost_cycle=float(one["bat_cost_cycle"]),
                             init_soc=float(one["soc"]))
    if L.has_genset:
        p1["genset"] = dict(running_min_production=float(one["gen_running_min"]), running_max_production=float(one["gen_running_max"]),
                            genset_cost=float(one["gen_cost"]), co2_per_unit=float(one["gen_co2_per_unit"]),
                            cost_per_unit_co2=float(one["gen_cost_per_unit_co2"]), start_up_time=0, wind_down_time=0)
    per_instance = None
    if not L.has_grid:
        om = orc.OracleMicrogrid(p1)
        a1 = acts[:, 0]
        n_py, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            om.reset()
            for k in range(K - 1):
                om.run(dict(genset=a1[k, :2], battery=a1[k, 2]) if L.has_genset and L.has_battery else
                       (dict(genset=a1[k, :2]) if L.has_genset else dict(battery=a1[k, 0])), True)
            n_py += K - 1
        per_instance = n_py / (time.perf_counter() - t0)
    v1, n1 = run(1, seconds * 0.3)
    # pick the OpenMP thread count that is fastest on this host (containers often expose more logical CPUs
    # than they may use), then spend the rest of the budget on it
    cands = sorted({c for c in (4, 8, 16, 32, 64, 128, cores) if c <= cores})
    probe = {c: run(c, seconds * 0.05)[0] for c in cands}
    best = max(probe, key=probe.get)
    vall, nall = run(best, seconds * 0.4)
    return {"value": vall, "unit": "env-steps/s", "cores": best, "kind": "port",
            "value_1thread": v1, "host_logical_cpus": cores,
            "per_instance_python_loop_1core": per_instance,
            "thread_probe": {str(c): round(v) for c, v in probe.items()},
            "sample": f"first {n} grids x {K} steps of the benchmark batch, repeated for ~{seconds:.0f} s "
                      f"({n1 + nall} env-steps on 1 and {best} threads): oracle/mgx_oracle.c (scalar C restatement of "
                      f"the reference loop, OpenMP over tiles of 64 grids); the Python reference itself runs ~2e3 "
                      f"env-steps/s/core (BASELINE.md)"}


def measured_traffic(kernel, grids, chunk):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS command
    (tools/gpu_profile.sh -> profiles/<round>/traffic.json); None when no matching profile is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("chunk") != chunk:
            continue
        # launches are recorded by size (threads = workgroups * 256; a workgroup owns 192..256 grids)
        for threads, e in sorted(d.get("by_launch_threads", {}).get(kernel, {}).items(), key=lambda kv: int(kv[0])):
            if grids <= int(threads) < 1.45 * grids:
                return e["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
        if d.get("grids") == grids and kernel in d.get("kernels", {}):
            return d["kernels"][kernel]["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, None


def main():
    args = parse()
    rank, world, local = mdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    _lib.build()
    local = int(os.environ.get("MGX_FORCE_LOCAL_RANK", local))     # tests: several ranks on one GPU (with a gloo backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    N, chunk = args.grids, args.chunk
    n_total = N * world

    batch = generate(n_total, n_steps=args.rows, seed=42, arch=args.arch, device=dev, rank=rank, world=world)
    eng = StepEngine(batch)
    L = eng.layout
    gen = torch.Generator(device=dev); gen.manual_seed(7 + rank)
    pool = torch.rand(4, chunk, N, L.action_dim, dtype=torch.float64, device=dev, generator=gen)
    run = Runner(eng, chunk, pool)
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    from pymgrid_amd.rbc import default_priority_ids
    lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, False)
    run.rbc_table = table_array(lists)
    run.rbc_ids = torch.from_numpy(default_priority_ids(batch, lists, remove_redundant_gensets=False)).to(dev)

    # extra measurement: S independent shards of the rank's grids, one HIP stream each (the launch sequences of shards
    # owe each other nothing and fill each other's ramp-up / tail gaps)
    S = max(1, args.shards)
    if N % S:
        raise SystemExit(f"--grids {N} is not divisible by --shards {S}")
    shards = srun = None
    if S > 1:
        from pymgrid_amd.hetero import StreamShards
        shards = StreamShards([generate(n_total, n_steps=args.rows, seed=42, arch=args.arch, device=dev,
                                        rank=rank * S + j, world=world * S) for j in range(S)])
        srun = ShardRunner(shards, chunk, 7 + 1000 * rank)

    def measure(mode, sharded, steps, warmup):
        n_launch = N // S if sharded else N                      # grids per kernel launch
        r = srun if sharded else run
        fn = {"fused": r.fused, "step": getattr(r, "single", None), "rbc": r.rbc}[mode]
        (shards.reset() if sharded else eng.reset(want_obs=False))
        # device pre-warm (untimed, part of set-up like the data generation): the first launches after start-up or after
        # an idle phase run on a cold device -- code-object load, and ~15 ms until the clocks settle under this load
        # (profiles/r01/exp_transient_cause.txt) -- so PREWARM_S seconds of the same launches precede the W warm-up steps
        t_end = time.perf_counter() + PREWARM_S
        while time.perf_counter() < t_end:
            fn(chunk if mode != "step" else 64)
            torch.cuda.synchronize(dev)
        (shards.reset() if sharded else eng.reset(want_obs=False))
        fn(warmup)
        r.launches = 0
        wall, gpu = timed_shards(fn, steps, shards, dev) if sharded else timed(fn, steps, dev)
        wall = mdist.max_over_ranks(wall, dev)
        gpu = mdist.max_over_ranks(gpu, dev)
        launches = r.launches                                    # per stream
        if mode in ("fused", "rbc"):
            A8 = 8 * L.action_dim if mode == "rbc" else 0        # rbc: no action stream; + 1 id byte per grid, once
            once = 1 if mode == "rbc" else 0
            per_launch = sum(L.bytes_fused(min(chunk, steps - k0)) - A8 * min(chunk, steps - k0) + once
                             for k0 in range(0, steps, chunk)) / launches
            unit_bytes = (L.bytes_fused(chunk) - A8 * chunk + once) / chunk
        else:
            unit_bytes = L.bytes_per_step()
            per_launch = unit_bytes
        # sharded: a "launch" is one ROUND = S kernel launches issued together, one per shard stream, covering all N grids;
        # its duration is the cadence of a shard stream (HIP events on that stream over the region / its launches)
        per_launch_bytes = per_launch * N
        avg_launch_s = gpu / launches
        achieved = per_launch_bytes / avg_launch_s / 1e9
        kname = {"fused": "step_k_kernel", "step": "step_kernel", "rbc": "rollout_kernel"}[mode]
        traffic, traffic_src = measured_traffic(kname, n_launch, chunk)
        if traffic is not None and sharded:
            traffic *= S
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": per_launch_bytes,
                "kernel": {"fused": "step_k_kernel<3,4,double>", "step": "step_kernel<3>",
                           "rbc": "rollout_kernel<3,4,false>"}[mode],
                "bytes_per_env_step": unit_bytes, "launches": launches, "avg_launch_us": avg_launch_s * 1e6}
        if sharded:
            roof.update({"launch": f"one round = {S} concurrent kernel launches, one per shard stream, {n_launch} grids each",
                         "concurrent_streams": S, "grids_per_kernel_launch": n_launch,
                         "kernel_avg_duration_us": r.kernel_durations_us(16)})
        return {"value": n_total * steps / wall, "steps": steps, "ms_per_step": wall / steps * 1e3, "roofline": roof}

    results = {}
    side = {"fused": (32768, 16384), "step": (8192, 2048), "rbc": (16384, 8192)}     # (steps, warm-up) when not the headline
    for mode in ("fused", "step", "rbc"):
        main_mode = mode == args.mode
        results[mode] = measure(mode, sharded=(S > 1 and mode == "fused"),
                                steps=args.steps if main_mode else min(args.steps, side[mode][0]),
                                warmup=args.warmup if main_mode else min(args.warmup, side[mode][1]))
    if S > 1:        # the same fused kernel as ONE launch sequence over all N grids (reported under "other")
        results["fused_one_stream"] = measure("fused", sharded=False, steps=min(args.steps, side["fused"][0]),
                                              warmup=min(args.warmup, side["fused"][1]))

    # BASELINE configs[4] in miniature, reported under "other": a heterogeneous fleet (1/3 genset+battery, 1/3
    # battery+grid, 1/3 genset+battery+grid; forecast_horizon = 24) stepped through the Gym surface WITH observations
    # (window prefetch K = 8), one bucket per HIP stream.
    hetero = None
    if args.hetero_steps > 0:
        hetero = hetero_gym_steps(N, dev, rank, world, args.hetero_steps)

    # metrics vector: episode-return sum + mean SoC, all-reduced over ranks (the ONLY collective; RCCL over xGMI)
    sums = eng.metrics(torch.stack([run.reward_k[-1], batch.cols["soc"]]))
    mdist.all_reduce_metrics(sums)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(eng, pool, args.cpu_seconds)

    if rank == 0:
        main_r = results[args.mode]
        names = {"fused": "fused_launches", "step": "single_step_launches", "rbc": "rbc_rollout_on_device",
                 "fused_one_stream": "fused_launches_one_stream"}
        line = {
            "metric": "microgrid env-steps/sec", "value": main_r["value"], "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_r["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N} generated 4-module grids (genset+battery+load+pv) per GPU, T={args.rows}, "
                                   f"H=0, normalised random actions (BASELINE configs[2])",
                       "grids_per_gpu": N, "grids_total": n_total, "mode": args.mode,
                       "steps_per_launch": 1 if args.mode == "step" else chunk,
                       "outputs": "reward+done+soc per step" + ("" if args.mode == "step" else " (streamed [K,N])"),
                       "parallelism": f"grids sharded x{world} ranks, no data-path collective"
                                      + (f"; fused mode: {S} independent shards per GPU on {S} HIP streams" if S > 1 else "")},
            "roofline": main_r["roofline"],
            "cpu_baseline": cpu,
            "other": {names[m]: {"value": r["value"], "steps": r["steps"], "ms_per_step": r["ms_per_step"],
                                 "roofline": r["roofline"]} for m, r in results.items() if m != args.mode},
            "metrics_allreduce": {"sum_last_reward": float(sums[0]), "mean_soc": float(sums[1]) / n_total},
            "hetero_h24_gym_steps": hetero,
            "prewarm_seconds_per_mode": PREWARM_S,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if shards is not None:
        shards.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
