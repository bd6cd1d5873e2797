#!/usr/bin/env python3
"""bench.py -- microgrid env-steps/s of the batched step engine (BASELINE.json metric).

Workload (BASELINE.json configs[2]): N = 100 000 generated Template-4 grids (genset + battery + load + pv) per GPU,
T = 8760 hourly rows, the series of MicrogridGenerator (pymgrid_amd/generator.py: the reference's base profiles x the per-grid
sizing ratio), normalised U[0,1) actions.  Everything the timed region reads (columns, series, actions) is resident in HBM
before timing starts.

Series layout (--series): "factorised" (default, the headline) keeps the series as the generator defines them -- base profile
id + ratio per grid, the product formed in the kernels with the generator's own multiply, bit-identical to the arrays -- so
no [T, N] series exist or are streamed; "materialised" streams [T, N] arrays written once by mgx_synthesize_series.  Both are
timed in every run (the other one under "other"), each with its own algorithmic byte count.  `done` is not written by the
headline launch: in lock-step it is the same for every grid and follows from the step counter (engine.done_steps).

A bench STEP is one pass of the hot path over one [1024, N, A] batch of actions: `--launches-per-step` (16) fused launches
(mgx_step_k) of `--chunk` (64) consecutive env-steps of ALL grids of the rank, per shard stream -- 1 024 env-steps of every grid,
~0.85 ms of GPU time at N = 100 000 (a single 64-step launch is 52 us: twenty of them were a 1-ms timed region and the clock
noise of the box decided the figure).  `--steps K --warmup W` time exactly K such steps after W untimed ones; `value` = grids x K
x 1024 / wall time (env-steps/s, whole job), `ms_per_step` = wall time per step.  A roofline "launch" is one 64-step round (one
kernel per shard stream, all N grids).

OUTPUT: stdout carries exactly ONE line -- a compact JSON record (< 4 KB: the headline fields, `roofline`, `cpu_baseline`, and one
{us, frac} pair per side leg under "legs").  Everything else (full roofline blocks of the legs, clocks, spreads, notes) goes to
`bench_detail.json` beside this script and to stderr.

Modes (--mode picks the headline `value`; the Gym-cadence legs run by default, the rest with --all-legs)
  fused (default)  parameters and state stay in registers for the 64 steps of a round, actions / series rows / per-step
                   outputs (reward, done, SoC) stream.  The rank's grids are stepped as --shards (2) contiguous ranges on
                   the engine's internal HIP streams (mgx_set_shards): ranges are not joined between rounds, so one range's
                   launch ramp-up / tail overlaps the other's steady state.  A roofline "launch" is one round (one kernel per
                   shard stream, all N grids); its duration is the cadence of a shard stream (HIP events on that stream over
                   the timed region / rounds).  The same kernel as ONE launch sequence is reported under "other".
  step             the Gym cadence: one launch of the single-step kernel per env-step, 64 of them per round issued by ONE
                   call (mgx_step_many); the same from a Python loop around env.step is reported beside it.
  rbc              rule-based control rolled out on device (mgx_rollout_discrete, one fixed priority list per grid).

Every mode is preceded by PREWARM_S seconds of its own rounds (untimed set-up: code-object load and the ~15 ms the clocks need
to settle under this load; it runs straight into the W warm-up steps), then EXACTLY K timed steps between barrier + synchronize.

Launch:  python bench.py [--gpus N --steps K --warmup W]      N > 1 without WORLD_SIZE in the environment: the script
         re-launches itself under torch.distributed.run, one rank per GPU (RCCL only for the final metrics all-reduce).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PREWARM_S = 0.3            # seconds of untimed device pre-warm before each mode's W warm-up rounds
OUT_SETS = 4               # sets of output buffers each runner cycles through (no Infinity-Cache absorption of rewrites)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
SIDE_ROUNDS = (192, 64)    # (timed, warm-up) rounds of the modes that are not the headline
LAT_FLOOR_US = 1.7         # launch boundary of a kernel that cannot overlap its predecessor (MI355X_MICROARCH.md: 1.5-1.9 us)
LAT_STREAM_GBS = 6000.0    # ... and the rate at which its one dependent round trip then moves (what the chip sustains)
MAX_LINE = 4096            # the compact stdout record stays below this (tests/test_bench_contract.py)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed bench STEPS (one step = --launches-per-step rounds of --chunk env-steps of all grids)")
    ap.add_argument("--warmup", type=int, default=16, help="untimed steps before the timed ones")
    ap.add_argument("--grids", type=int, default=None,
                    help="microgrids PER GPU (weak scaling); default 100 000 (--config 2) / 125 000 (--config 3, 4: 1 M over 8 GPUs)")
    ap.add_argument("--rows", type=int, default=8760, help="time-series rows T")
    ap.add_argument("--mode", choices=["fused", "step", "rbc"], default="fused")
    ap.add_argument("--chunk", type=int, default=64, help="env-steps per round (= per fused launch)")
    ap.add_argument("--launches-per-step", type=int, default=16,
                    help="rounds (fused launches per shard stream) that make up one bench step: 16 x 64 = 1 024 env-steps of every grid")
    ap.add_argument("--shards", type=int, default=None,
                    help="grid ranges stepped on internal HIP streams (mgx_set_shards); 1: one launch sequence.  Default: 2")
    ap.add_argument("--arch", default="genset+battery")
    ap.add_argument("--series", choices=["factorised", "materialised"], default="factorised",
                    help="series layout of the headline batch (the other layout is timed with --all-legs)")
    ap.add_argument("--uniform-columns", action="store_true",
                    help="hold the parameters MicrogridGenerator gives every microgrid (battery efficiency / cycle cost, genset cost "
                         "and co2 figures, unbalanced-energy costs, zero genset timers) once instead of as [N] columns "
                         "(mgx_columns.uniform_mask).  Off by default: 60 fewer bytes per grid and single step buy no time "
                         "(profiles/r03/exp_uniform_columns.txt)")
    ap.add_argument("--hetero-steps", type=int, default=1024,
                    help="timed Gym steps of the heterogeneous H=24 fleet (0: skip); the region starts 256 steps after a reset: the "
                         "first ~100 steps behind a reset + device synchronisation run 2-5 us slower (profiles/r04/exp_fleet_transient2.txt)")
    ap.add_argument("--no-side-modes", action="store_true", help="time the headline mode only (+ the fleet unless --hetero-steps 0)")
    ap.add_argument("--all-legs", action="store_true",
                    help="also time the legs that are not part of the default record: the rule-based rollout, the other series layout, "
                         "one launch sequence, the other ring layout / the views contract of the fleet, the closed policy loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-closed-loop", action="store_true",
                    help="skip the Gym legs that write observation rows per step (the PMC passes use it: single steps WITH rows would be "
                         "averaged into the single-step kernel's counters)")
    ap.add_argument("--legs", default=None,
                    help="comma-separated side legs to time INSTEAD of the default set (step, step_env, step_env_obs, step_2chains, "
                         "step_env_2chains, step_env_obs_2chains, fused_rich, step_full, rbc): the PMC passes of one launch shape")
    ap.add_argument("--config", type=int, choices=[2, 3, 4], default=2,
                    help="BASELINE.json configs[] index the job lands on: 2 = 100 000 template-4 grids per GPU (the metric's N = 100k), "
                         "3 = 1 M template-4 grids over 8 GPUs (125 000 per GPU), 4 = 1 M heterogeneous H = 24 grids over 8 GPUs (the fleet "
                         "leg at 125 000 per GPU becomes what `value` reports)")
    ap.add_argument("--fleet-ring", type=int, default=32, help="observation-ring depth K of the config-5 fleet leg (experiments; default 32)")
    ap.add_argument("--tunable", action="append", default=[], metavar="NAME=VALUE",
                    help="mgx_set_tunable before anything runs (A/B of launch shapes: e.g. multi_static=0); repeatable")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU time of the baseline sample")
    ap.add_argument("--prewarm", type=float, default=PREWARM_S)
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (the stdout line is the compact one); '' = nowhere")
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring the N ranks up, check the process group and the metrics all-reduce, print {n_gpus: N} (no GPU work)")
    args = ap.parse_args(argv)
    if args.grids is None:
        args.grids = 100_000 if args.config == 2 else 125_000
    if args.config == 4 and args.hetero_steps <= 0:
        ap.error("--config 4 reports the heterogeneous fleet: --hetero-steps must be > 0")
    return args


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) under torch.distributed.run and relay the
    result.  Returns the exit code, or None when this process is already a rank / a single-GPU run."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.run(cmd, env=env).returncode


class Runner:
    """Issues rounds on one engine.  Sharded: the engine steps in S grid ranges on its internal streams."""

    def __init__(self, eng, chunk, seed, shards, pool=None, rank=0, world=1):
        from pymgrid_amd.priority_list import get_priority_lists, table_array
        from pymgrid_amd.rbc import default_priority_ids
        self.eng, self.chunk, self.S = eng, chunk, shards
        L, N, dev = eng.layout, eng.N, eng.device
        if pool is None:
            # normalised U[0, 1) actions of the GLOBAL batch (seed 7; one [chunk, N x world, A] draw per pool entry, of which this
            # rank keeps its own grid range): the job's results do not depend on how many GPUs it is spread over
            gen = torch.Generator(device=dev); gen.manual_seed(seed)
            pool = torch.empty(4, chunk, N, L.action_dim, dtype=torch.float64, device=dev)
            for j in range(4):
                full = torch.rand(chunk, N * world, L.action_dim, dtype=torch.float64, device=dev, generator=gen)
                pool[j] = full[:, rank * N:(rank + 1) * N]
                del full
        self.pool = pool
        # OUT_SETS sets of [chunk, N] output buffers, cycled: a launch never rewrites what the previous three wrote, so no
        # output line can still sit in the 256 MB Infinity Cache when it is written again
        self.outs = [dict(reward=torch.empty(chunk, N, dtype=torch.float64, device=dev),
                          done=torch.empty(chunk, N, dtype=torch.uint8, device=dev),
                          soc_trace=torch.empty(chunk, N, dtype=torch.float64, device=dev)) for _ in range(OUT_SETS)]
        lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, False)
        self.rbc_table = table_array(lists)
        self.rbc_ids = torch.from_numpy(default_priority_ids(eng.batch, lists, remove_redundant_gensets=False)).to(dev)
        self.rounds = 0            # rounds issued since the process started (per mode runner)
        self.env = None            # a BatchedMicrogridEnv around self.eng (the step_env leg)
        self.act_views = None
        self.streams = []
        self.done_stream = False

    def shard(self, on):
        if self.env is not None:                   # (the env re-binds its step to the handle around the change)
            self.env.set_shards(self.S if on else 1)
        else:
            self.eng.set_shards(self.S if on else 1)
        self.streams = self.eng.shard_streams() if on and self.S > 1 else []

    def reset(self):
        if self.env is not None:
            self.env.reset()
        else:
            self.eng.reset(want_obs=False)

    def _room(self):
        if self.eng.current_step + self.chunk > self.eng.layout.final_step:
            self.reset()

    # `done`: in lock-step it is the same for every grid, done(k) = (t + k >= final_step - 1); the launches below do not write
    # it per grid (self.done_stream = False) -- engine.done_steps(K) derives the [K, N] view from the step counter

    def fused(self, rounds):
        for _ in range(rounds):
            self._room()
            self.eng.step_k(self.pool[self.rounds % 4], normalized=True, out=self.outs[self.rounds % OUT_SETS],
                            reward=True, done=self.done_stream, soc_trace=True)
            self.rounds += 1

    def rbc(self, rounds):
        for _ in range(rounds):
            self._room()
            self.eng.rollout_discrete(self.rbc_ids, self.rbc_table, self.chunk, reward=True, done=self.done_stream, soc_trace=True,
                                      out=self.outs[self.rounds % OUT_SETS])
            self.rounds += 1

    def _rich_outs(self):
        """Two sets of the FULL outputs of a fused round (Microgrid.run's per-step returns and log, microgrid.py:305-319,
        base_module.py:276-290): reward, done, SoC, the packed genset status word and the L log columns [K, L, N]."""
        if getattr(self, "rich_outs", None) is None:
            N, dev, Ld = self.eng.N, self.eng.device, self.eng.log_dim
            self.rich_outs = [dict(reward=torch.empty(self.chunk, N, dtype=torch.float64, device=dev),
                                   done=torch.empty(self.chunk, N, dtype=torch.uint8, device=dev),
                                   soc_trace=torch.empty(self.chunk, N, dtype=torch.float64, device=dev),
                                   status_trace=torch.empty(self.chunk, N, dtype=torch.int32, device=dev),
                                   log=torch.empty(self.chunk, Ld, N, dtype=torch.float64, device=dev)) for _ in range(2)]
        return self.rich_outs

    def fused_rich(self, rounds):
        """The fused launch writing EVERYTHING the reference's step returns and logs: reward, per-grid done, SoC, genset status and
        the balance / module log columns of every step (SURVEY 8(d) "full total")."""
        outs = self._rich_outs()
        for _ in range(rounds):
            self._room()
            self.eng.step_k(self.pool[self.rounds % 4], normalized=True, out=outs[self.rounds % 2], reward=True, done=True,
                            soc_trace=True, status_trace=True, log=True)
            self.rounds += 1

    def step_full(self, rounds):
        """Gym cadence with the full outputs: every single-step launch writes reward, per-grid done, the H = 0 observation row and
        the L log columns."""
        if getattr(self, "full_outs", None) is None:
            N, dev, Ld, D = self.eng.N, self.eng.device, self.eng.log_dim, self.eng.obs_dim
            self.full_outs = [dict(reward=torch.empty(self.chunk, N, dtype=torch.float64, device=dev),
                                   done=torch.empty(self.chunk, N, dtype=torch.uint8, device=dev),
                                   obs=torch.empty(self.chunk, N, D, dtype=torch.float64, device=dev),
                                   log=torch.empty(self.chunk, Ld, N, dtype=torch.float64, device=dev)) for _ in range(2)]
        for _ in range(rounds):
            self._room()
            self.eng.step_many(self.pool[self.rounds % 4], normalized=True, out=self.full_outs[self.rounds % 2], done=True,
                               want_obs=True, want_log=True)
            self.rounds += 1

    def step(self, rounds):
        """Gym cadence: `chunk` single-step launches per round, issued by one call of the C ABI."""
        for _ in range(rounds):
            self._room()
            o = self.outs[self.rounds % OUT_SETS]
            self.eng.step_many(self.pool[self.rounds % 4], normalized=True, out=dict(reward=o["reward"], done=o["done"]),
                               done=self.done_stream)
            self.rounds += 1

    def step_env(self, rounds):
        """The Gym surface from a Python loop: `for a in actions: env.step(a)` on a BatchedMicrogridEnv whose step is bound to the
        handle (mgx_env_bind: one C call per env-step, outputs in rotating buffers)."""
        if self.act_views is None:                  # the [N, A] views of the pooled actions, built once (a view costs ~1 us)
            self.act_views = [[self.pool[j][k] for k in range(self.chunk)] for j in range(self.pool.shape[0])]
        step = self.env.step
        for _ in range(rounds):
            self._room()
            for ak in self.act_views[self.rounds % 4]:
                step(ak)
            self.rounds += 1

    def kernel_durations_us(self, fn, rounds=16):
        """Mean start-to-end time of the kernels of a round (HIP events around every launch on its own stream, a short
        extra pass after the timed region; launches outside [0.5, 2] x the median are left out): what rocprofv3 reports as the
        kernel's duration."""
        dev = self.eng.device
        streams = self.streams or [torch.cuda.current_stream(dev)]
        ev = []
        self.eng.fork()
        for _ in range(rounds):
            pair = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in streams]
            for (e0, _), st in zip(pair, streams):
                e0.record(st)
            fn(1)
            for (_, e1), st in zip(pair, streams):
                e1.record(st)
            ev += pair
        self.eng.join()
        torch.cuda.synchronize(dev)
        d = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        med = d[len(d) // 2]
        kept = [x for x in d if 0.5 * med <= x <= 2.0 * med]      # a stalled launch (a 70 ms hiccup was seen once) is not the kernel
        return sum(kept) / len(kept) * 1e3


def sample_device_state(out, delay_s=0.12):
    """Clocks / power / temperature of the GPU as rocm-smi reports them, sampled once from a helper thread while the caller keeps
    the device busy (the untimed pre-warm of the headline): box-to-box differences of the headline show up here or nowhere."""
    import threading

    def work():
        time.sleep(delay_s)
        try:
            res = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True,
                                 timeout=10)
            txt = res.stdout[res.stdout.index("{"):]
            cards = json.loads(txt)
            card = next(iter(cards.values()))              # (a rank sees its own GPU only when the launcher isolates devices)
            out.update({k.rstrip(":"): v for k, v in card.items()})
        except Exception as e:          # noqa: BLE001  (diagnostics only)
            out["error"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=work, daemon=True)
    th.start()
    return th


def timed(run, fn, rounds, device, mdist, per_round=False):
    """barrier + sync | K rounds | sync + barrier; returns (wall seconds, GPU seconds = the longest launch stream's elapsed
    time between its own start and stop events, per-round times).  per_round: an event behind every round on every launch stream
    (a timestamp packet each: no GPU work) -> the K round durations in us (per round the slowest stream's cadence), else None."""
    streams = run.streams or [torch.cuda.current_stream(device)]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in streams]
    marks = [[torch.cuda.Event(enable_timing=True) for _ in range(rounds)] for _ in streams] if per_round else None
    mdist.barrier()
    run.eng.fork()                      # stream ordering only (no GPU work): the shard streams line up behind the caller's stream
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for (e0, _), st in zip(ev, streams):
        e0.record(st)
    if per_round:
        for r in range(rounds):
            fn(1)
            for m, st in zip(marks, streams):
                m[r].record(st)
    else:
        fn(rounds)
    for (_, e1), st in zip(ev, streams):
        e1.record(st)
    torch.cuda.synchronize(device)      # the whole device: every shard stream has drained
    t1 = time.perf_counter()
    run.eng.join()
    mdist.barrier()
    round_us = None
    if per_round:
        round_us = []
        for r in range(rounds):
            # one sample per (stream, round): a stream's own cadence (the slower stream of a round would bias the level upward)
            round_us.extend(((ev[j][0] if r == 0 else marks[j][r - 1]).elapsed_time(marks[j][r])) * 1e3 for j in range(len(streams)))
    return t1 - t0, max(e0.elapsed_time(e1) for e0, e1 in ev) * 1e-3, round_us


def hetero_gym_steps(N, dev, rank, world, steps, mdist, rows, series, uniform, all_legs=False, K_ring=32):
    """BASELINE configs[4] per GPU: a heterogeneous fleet (1/3 genset+battery, 1/3 battery+grid, 1/3 genset+battery+grid;
    forecast_horizon = 24, T = `rows`) stepped through the Gym surface WITH observations, under both observation contracts:
      rows   step() returns the [N, D] rows (D = 56 / 152 / 156): written ahead in rings of 16 row blocks, the step adds the
             state columns (float64 and float32 rows);
      views  step() returns ObsViews: strided views into the series normalised once (mgx_normalise_series) + the 6 state
             columns the step writes -- the windows repeat 24 / 25 of the previous step's, so nothing else needs to move."""
    from pymgrid_amd.generator import generate
    from pymgrid_amd.hetero import BucketedFleet
    per = N // 3
    # K_ring: ring depth, 32 by default: 25 vs 27.5 us per fleet step against K = 16 (profiles/r04/exp_fleet_refill_occupancy.txt)
    out = {}
    archs = ("genset+battery", "battery+grid", "genset+battery+grid")
    #   rows            step() returns the [N, D] observation, the fleet's DEFAULT ring layout: column-major blocks (obs = the view with
    #                   strides (1, pitch): the state columns a step adds are coalesced runs instead of 48 bytes per row)
    #   rows_rowmajor   the same with obs_layout="rows": contiguous [N, D] rows              (--all-legs)
    #   views           the zero-copy contract                                               (--all-legs)
    for contract_name in (("rows", "rows_rowmajor", "views") if all_legs else ("rows",)):
        contract = "rows" if contract_name.startswith("rows") else "views"
        for dt_name, dt in (("float64", torch.float64), ("float32", torch.float32)):
            name = f"{dt_name}_{contract_name}"
            batches = [generate(per * world, n_steps=rows, seed=43 + k, arch=arch, horizon=24, device=dev, rank=rank,
                                world=world, series=series, uniform_columns=uniform) for k, arch in enumerate(archs)]
            if contract == "rows":
                fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=K_ring, reuse_outputs=3 * K_ring,
                                                   obs_layout="rows" if contract_name == "rows_rowmajor" else None)
            else:
                fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_views=True, reuse_outputs=3 * K_ring)
            gen = torch.Generator(device=dev); gen.manual_seed(11 + rank)
            acts = [torch.rand(per, e.layout.action_dim, dtype=torch.float64, device=dev, generator=gen) for e in fleet.envs]

            def consume(obs):          # what a policy would do first: take the window views (host work of the views contract)
                if contract == "views":
                    for o in obs:
                        o.load; o.pv; o.grid

            def fstep():               # one fleet step; short series (tests) wrap around through a reset
                if fleet.envs[0].current_step >= rows - 1:
                    fleet.reset()
                return fleet.step(acts)
            # warm-up by wall time: the fleet is built on the host while the GPU idles and clocks down
            prev, t_end = None, time.perf_counter() + 4.0
            n_warm = max(1000, 256 + steps)                   # (covers the rows of the timed region: their views are built here)
            while time.perf_counter() < t_end:                # until two consecutive blocks of n_warm steps agree within 3 %
                fleet.reset()
                t0 = time.perf_counter()
                for _ in range(n_warm):
                    consume(fstep()[0])
                torch.cuda.synchronize(dev)
                cur = time.perf_counter() - t0
                if prev is not None and abs(cur - prev) < 0.03 * prev:
                    break
                prev = cur
            fleet.reset()
            for _ in range(256):
                fstep()
            mdist.barrier()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(steps):
                consume(fstep()[0])
            e1.record()
            torch.cuda.synchronize(dev)
            wall = mdist.max_over_ranks(time.perf_counter() - t0, dev)
            gpu = mdist.max_over_ranks(e0.elapsed_time(e1) * 1e-3, dev)
            mdist.barrier()
            esz = 8 if dt == torch.float64 else 4
            # algorithmic bytes of one fleet step (SURVEY 8(d) formula): the core step of every bucket (no done byte: lock-step;
            # factorised: the grid's factors instead of its row values) + rows: the observation row written (esz * D) and the
            # window source read once per ring refill; views: the state columns written (esz * S)
            alg = 0
            for e in fleet.envs:
                L = e.layout
                fact = e.batch.factorised
                c_ts = L.n_load + L.n_pv + 4 * int(L.has_grid)
                core = L.bytes_per_step() - 1 + ((18 + 2 * int(L.has_grid)) - 8 * c_ts if fact else 0) - e.batch.uniform_param_bytes()
                if contract == "rows":
                    src = (18 + 2 * int(L.has_grid)) / K_ring if fact else 8 * c_ts * (K_ring + L.horizon) / K_ring
                    alg += L.n_grids * (core + esz * L.obs_dim + src)
                else:
                    alg += L.n_grids * (core + esz * e.engine.state_dim)
            ach = alg / (gpu / steps) / 1e9
            # PMC bytes of the same fleet shape, if a profile of it -- taken on the kernels now running -- is committed
            traffic = None
            tf, traffic_src = profile_json(f"traffic_fleet_{contract_name}_{dt_name}.json")
            if tf is not None and (tf.get("grids_per_gpu"), tf.get("series"), tf.get("contract"), tf.get("dtype"),
                                   tf.get("obs_prefetch")) == (3 * per, series, contract_name, dt_name, K_ring if contract == "rows" else tf.get("obs_prefetch")):
                traffic = tf["hbm_bytes_per_fleet_step"]
            elif tf is not None:
                traffic_src += ": another fleet shape"
            launch = ("one fleet step = one mgx_fleet_step call: ONE fleet_step_kernel_v launch over the three buckets (KArgs by value, bucket = blockIdx.y)"
                      + (f" + every {K_ring}th step the observation ring after next ({K_ring} row blocks: obs_windows_k_kernel per "
                         f"bucket on the engines' prefetch streams, beside the following step launches); bytes and time are per "
                         f"fleet step, refills included" if contract == "rows" else
                         "; the window columns are views of the once-normalised series (no bytes per step), the step writes the "
                         "6 state columns.  From Python this contract is HOST-paced (the kernel itself takes 8.7 us per fleet step, "
                         "profiles/r03/fleet_views_kernel_stats.csv): avg_launch_us is the host's pace with the caller taking all "
                         "window views at every step (memoised per step index: a loop over the same rows builds each view once)"))
            out[name] = {"value": 3 * per * world * steps / wall, "us_per_step": wall / steps * 1e6,
                         "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                      "frac_wall": alg / (wall / steps) / 1e9 / HBM_PEAK_GBS,
                                      "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg,
                                      "avg_launch_us": gpu / steps * 1e6, "launch": launch,
                                      "kernel": "fleet_step_kernel_v" + (f" + obs_windows_k_kernel<F> x3 / {K_ring}" if contract == "rows" else ""),
                                      "refill": fleet.refill if contract == "rows" else None,
                                      "ring_blocks": ("column-major [D, pitch]: obs = the [N, D] view with strides (1, pitch) (the default)" if contract_name == "rows"
                                                      else ("row-major [N, D]" if contract == "rows" else None))}}
            if contract == "views":
                # ONE launch per fleet step that writes 6 state columns: a dependent round trip behind a launch boundary, not a stream
                rfv = out[name]["roofline"]
                model_us = LAT_FLOOR_US + alg / (LAT_STREAM_GBS * 1e3)
                rfv.update({"bound": "latency", "model_us": model_us, "frac_of_latency_model": model_us / (gpu / steps * 1e6)})
            obs_dims = [e.layout.obs_dim for e in fleet.envs]
            fleet.close()
            del fleet, batches
            torch.cuda.empty_cache()
    out.update({"grids_per_gpu": 3 * per, "obs_dims": obs_dims, "horizon": 24, "obs_prefetch": K_ring, "steps": steps, "rows": rows,
                "series": series,
                "workload": "BASELINE configs[4] mix per GPU: 1/3 genset+battery, 1/3 battery+grid, 1/3 genset+battery+grid; "
                            f"T = {rows}, forecast_horizon = 24; Gym step() with observations (rows / zero-copy views)"})
    return out


def closed_loop_leg(args, n_total, dev, rank, world, mdist):
    """An agent IN the loop: obs -> a small on-device policy -> env.step -> obs, from Python (--all-legs)."""
    from pymgrid_amd.generator import generate
    from pymgrid_amd import BatchedMicrogridEnv

    def loop(dtype, reuse, light=False, auto_reset=False):
        b = generate(n_total, n_steps=args.rows, seed=42, arch=args.arch, device=dev, rank=rank, world=world, series=args.series)
        if auto_reset:           # every grid on its own random 168-step episodes, restarted by the step that ends them
            from pymgrid_amd.hetero import PerGridWindowEnv
            env = PerGridWindowEnv(b, trajectory_length=168, auto_reset=True, seed=3, obs_dtype=dtype, action_dtype=dtype,
                                   reuse_outputs=reuse)
        else:
            env = BatchedMicrogridEnv(b, obs_dtype=dtype, action_dtype=dtype, reuse_outputs=reuse)
        g = torch.Generator(device=dev); g.manual_seed(5)
        A = env.layout.action_dim
        W = torch.randn(env.layout.obs_dim, A, dtype=dtype, device=dev, generator=g)
        w = torch.randn(A, dtype=dtype, device=dev, generator=g)
        # light: a per-feature policy (two elementwise kernels) -- torch's GEMM for a [N, 8] x [8, 3] product takes 23 us in
        # float64 and 84 us in float32 on this stack, which would hide the env behind the policy
        policy = (lambda o: torch.sigmoid(o[:, :A] * w)) if light else (lambda o: torch.sigmoid(o @ W))
        obs = env.reset()
        n = min(2000, args.rows - 200)

        def timed_loop(fn, sync_ranks):
            for _ in range(100):
                fn()
            if sync_ranks:
                mdist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(dev)
            wall = time.perf_counter() - t0
            if sync_ranks:
                wall = mdist.max_over_ranks(wall, dev)
                mdist.barrier()
            return wall
        st = {"obs": obs}

        def both():
            st["obs"] = env.step(policy(st["obs"]))[0]
        wall = timed_loop(both, True)
        policy_us = timed_loop(lambda: policy(obs), False) / n * 1e6                  # the two torch kernels alone
        env.reset()
        a = policy(obs)
        env_us = timed_loop(lambda: env.step(a), False) / n * 1e6                     # env.step alone (host-paced)
        env.close()
        return {"value": n_total * n / wall, "us_per_step": wall / n * 1e6, "steps": n,
                "policy_kernels_alone_us": policy_us, "env_step_alone_us": env_us}
    out = {"float64": loop(torch.float64, 0), "float32_io_rotating_outputs": loop(torch.float32, 4, light=True)}
    if args.series == "factorised":
        out["float32_io_auto_reset_168_step_episodes"] = loop(torch.float32, 4, light=True, auto_reset=True)
    out["loop"] = ("obs [N, 8] -> sigmoid(obs @ W) -> BatchedMicrogridEnv.step (one launch) -> obs, issued from Python: three "
                   "kernels per env-step (matmul, sigmoid, step), each waiting for the one before -- the env's share is "
                   "env_step_alone_us.  float32_io: observations and actions cross the boundary as floats (what a policy "
                   "network consumes / emits; the step itself stays float64), reward and rows in 4 rotating buffers "
                   "(reuse_outputs), and the policy is sigmoid(obs[:, :A] * w): two elementwise kernels instead of torch's "
                   "GEMM, which takes 23 us (float64) / 84 us (float32) for this [N, 8] x [8, 3] product.  auto_reset: PerGridWindowEnv -- "
                   "every grid walks its own random 168-step episodes and is restarted by the step that ends its episode "
                   "(in-place episodes, mgx_set_auto_reset: still one launch per env-step)")
    out["value"], out["us_per_step"] = out["float64"]["value"], out["float64"]["us_per_step"]
    return out


def general_path_leg(args, N, n_total, chunk, dev, rank, world, mdist):
    """The GENERAL path on the board: 2 gensets + 2 batteries + 1 grid per microgrid (module_container.py:355-413 allows any
    multiplicity; such layouts run on the general kernels: columns [n, N], MicrogridStep's lists in LDS).  Single Gym steps, the
    K-step launch and Gym steps with whole 24-hour rows, each against its own algorithmic bytes (SURVEY 8(d) with per-instance
    parameter / state counts) and, where a counter pass of this hash is committed, its PMC traffic (profiles/r*/traffic_general.json)."""
    from pymgrid_amd.engine import StepEngine
    from pymgrid_amd.generator import generate
    from pymgrid_amd.generator import widen
    rows_g = min(args.rows, 1200)                   # materialised series [T, n, N]: 1 200 rows are 10 GB at N = 100 000
    base = generate(n_total, n_steps=rows_g, seed=42, arch="genset+battery+grid", device=dev, rank=rank, world=world)
    gb = widen(base, n_genset=2, n_battery=2, n_grid=1)
    gb3 = widen(base, n_genset=3, n_battery=3, n_grid=1)     # THREE of a kind: M = 3 register slots, counts at compile time
    del base
    ge = StepEngine(gb)
    Lg = ge.layout
    gen = torch.Generator(device=dev); gen.manual_seed(3 + rank)
    a1 = torch.rand(N, Lg.action_dim, dtype=torch.float64, device=dev, generator=gen)
    Kg = min(chunk, 32)
    aK = torch.rand(Kg, N, Lg.action_dim, dtype=torch.float64, device=dev, generator=gen)
    out = {}
    reward1 = torch.empty(N, dtype=torch.float64, device=dev)

    def run(fn, n, per_call_steps):
        ge.reset(0, want_obs=False)
        for _ in range(max(8, n // 8)):
            fn()
        ge.reset(0, want_obs=False)
        mdist.barrier(); torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize(dev)
        wall = mdist.max_over_ranks(time.perf_counter() - t0, dev)
        gpu = mdist.max_over_ranks(e0.elapsed_time(e1) * 1e-3, dev)
        mdist.barrier()
        return wall, gpu, n * per_call_steps
    n1 = min(400, rows_g - 64)
    wall, gpu, steps1 = run(lambda: ge.step(a1, want_obs=False, want_log=False, out=dict(reward=reward1), want_done=False), n1, 1)
    b1 = (Lg.bytes_per_step() - 1) * N              # (no done byte: lock-step)
    out["single_steps"] = {"value": n_total * steps1 / wall, "us_per_step": gpu / steps1 * 1e6, "us_per_step_wall": wall / steps1 * 1e6,
                           "roofline": {"bound": "hbm", "achieved": b1 / (gpu / steps1) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": b1 / (gpu / steps1) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                        "algorithmic_bytes_per_launch": b1, "kernel": "step_multi_kernel<7>",
                                        "bytes_per_env_step": Lg.bytes_per_step() - 1}}
    nK = max(2, min(12, (rows_g - 64) // Kg))
    wall, gpu, stepsK = run(lambda: ge.step_k(aK, normalized=True, reward=True, soc_trace=False), nK, Kg)
    # Round 5: the K-step kernel of the general path keeps parameters and state in LDS for the whole launch, so -- like the headline's
    # fused kernel -- it is charged what it MUST move: controls + series rows + reward per step, parameters / state once per launch
    # (SURVEY 8(d) would charge B per step whatever the kernel re-reads: that figure is kept beside it as frac_charged_per_step)
    c_ts = Lg.n_load + Lg.n_pv + 4 * Lg.n_grid
    stream_b = 8 * (Lg.action_dim + c_ts) + 8                      # per env-step: controls, series rows read; reward written
    once_b = Lg.bytes_per_step() - 1 - stream_b                    # per launch: parameter columns + state read and written
    bK_launch = (stream_b * Kg + once_b) * N
    bK = (Lg.bytes_per_step() - 1) * N
    out["k_step_launches"] = {"value": n_total * stepsK / wall, "us_per_step": gpu / stepsK * 1e6, "steps_per_launch": Kg,
                              "roofline": {"bound": "hbm", "achieved": bK_launch / Kg / (gpu / stepsK) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": bK_launch / Kg / (gpu / stepsK) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                           "frac_charged_per_step": bK / (gpu / stepsK) / 1e9 / HBM_PEAK_GBS,
                                           "algorithmic_bytes_per_launch": bK_launch, "kernel": "step_k_multi_small_kernel<7, CountsCT<2,2,1,1,1>, 2>",
                                           "bytes_per_env_step": bK_launch / Kg / N, "avg_launch_us": gpu / stepsK * Kg * 1e6,
                                           "note": "parameters and state live in LDS for the launch (round 5): charged once per launch; "
                                                   "per step the controls, the series rows and the reward stream"}}
    out["k_step_launches"]["roofline_valu"] = valu_roofline("step_k_multi_small_kernel<7,mgx::CountsCT<2,2,1,1,1>,2>", gpu / stepsK * Kg, N, 1, dev)
    # RuleBasedControl on the same layout (f1 on the general path, round 6): one fixed priority list per grid (sorted by marginal cost,
    # rbc.py:26-62), the controls expanded in registers (rollout_multi_small_kernel); per step only the series rows are read
    from pymgrid_amd.priority_list import get_instance_priority_lists, lists_array
    from pymgrid_amd.rbc import default_instance_priority_ids
    pls = get_instance_priority_lists(Lg.n_genset, Lg.n_battery, Lg.n_grid, (), Lg.grid_before_battery)
    pl_tab = torch.as_tensor(lists_array(pls), device=dev).contiguous()
    pl_ids = torch.from_numpy(default_instance_priority_ids(gb, pls)).to(dev).to(torch.int32).contiguous()
    rbc_out = {"reward": torch.empty(Kg, N, dtype=torch.float64, device=dev)}
    wall, gpu, stepsR = run(lambda: ge.rollout_lists(pl_ids, pl_tab, Kg, reward=True, out=rbc_out), nK, Kg)
    streamR = 8 * c_ts + 8
    bR_launch = (streamR * Kg + (Lg.bytes_per_step() - 1 - stream_b)) * N
    out["rbc_rollout"] = {"value": n_total * stepsR / wall, "us_per_step": gpu / stepsR * 1e6, "steps_per_launch": Kg,
                          "roofline": {"bound": "hbm", "achieved": bR_launch / Kg / (gpu / stepsR) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": bR_launch / Kg / (gpu / stepsR) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                       "algorithmic_bytes_per_launch": bR_launch, "kernel": "rollout_multi_small_kernel<7, CountsCT<2,2,1,1,1>, 2>",
                                       "bytes_per_env_step": bR_launch / Kg / N, "avg_launch_us": gpu / stepsR * Kg * 1e6,
                                       "note": "no action stream: the priority-list walk (5 modules) and the step are arithmetic -- VALU / latency bound; "
                                               "the run-time-count kernel with its walk over the batch's columns took 14.8 us per env-step"}}
    out["rbc_rollout"]["roofline_valu"] = valu_roofline("rollout_multi_small_kernel<7,mgx::CountsCT<2,2,1,1,1>,2>", gpu / stepsR * Kg, N, 1, dev)
    out["layout"] = "2 gensets + 2 batteries + 1 grid + load + pv per microgrid (general kernels), materialised series"
    out["grids_per_gpu"], out["rows"] = N, rows_g
    ge.close()
    del ge, gb
    # three gensets + three batteries + a grid: the K-step launch (step_k_multi_small_kernel<7, CountsCT<3,3,1,1,1>, 3>)
    ge = StepEngine(gb3)
    L3 = ge.layout
    aK3 = torch.rand(Kg, N, L3.action_dim, dtype=torch.float64, device=dev, generator=gen)
    wall, gpu, stepsK = run(lambda: ge.step_k(aK3, normalized=True, reward=True, soc_trace=False), nK, Kg)
    c_ts3 = L3.n_load + L3.n_pv + 4 * L3.n_grid
    stream3 = 8 * (L3.action_dim + c_ts3) + 8
    b3_step = (L3.bytes_per_step() - 1) * N              # this form re-reads parameters and state every step (through the caches)
    b3_launch = (stream3 * Kg + (L3.bytes_per_step() - 1 - stream3)) * N
    out["k_step_3_of_a_kind"] = {"value": n_total * stepsK / wall, "us_per_step": gpu / stepsK * 1e6, "steps_per_launch": Kg,
                                 "roofline": {"bound": "hbm", "achieved": b3_launch / Kg / (gpu / stepsK) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": b3_launch / Kg / (gpu / stepsK) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                              "frac_charged_per_step": b3_step / (gpu / stepsK) / 1e9 / HBM_PEAK_GBS,
                                              "algorithmic_bytes_per_launch": b3_launch, "kernel": "step_k_multi_small_kernel<7, CountsCT<3,3,1,1,1>, 3>",
                                              "bytes_per_env_step": b3_launch / Kg / N, "avg_launch_us": gpu / stepsK * Kg * 1e6,
                                              "layout": "3 gensets + 3 batteries + 1 grid + load + pv per microgrid",
                                              "note": "three instance slots per kind in registers, counts at compile time (round 6; the run-time-count "
                                                      "kernel with its LDS lists took 11.5 us per env-step: 0.16)"}}
    out["k_step_3_of_a_kind"]["roofline_valu"] = valu_roofline("step_k_multi_small_kernel<7,mgx::CountsCT<3,3,1,1,1>,3>", gpu / stepsK * Kg, N, 1, dev)
    ge.close()
    del ge, gb3
    torch.cuda.empty_cache()
    # Gym steps with whole observation rows (H = 24): rings refilled by the general window kernel (obs_windows_k_multi_kernel;
    # the step adds its 12 state columns) -- per-step rows by one lane per grid were 261 us (profiles/r04/exp_multi_rings.txt)
    from pymgrid_amd import BatchedMicrogridEnv
    rows_o = min(args.rows, 600)
    base = generate(n_total, n_steps=rows_o, seed=42, arch="genset+battery+grid", horizon=24, device=dev, rank=rank, world=world)
    env = BatchedMicrogridEnv(widen(base, n_genset=2, n_battery=2, n_grid=1), obs_prefetch=32, reuse_outputs=96)
    del base
    Lo = env.layout
    ao = torch.rand(N, Lo.action_dim, dtype=torch.float64, device=dev, generator=gen)
    env.reset()
    for _ in range(40):
        env.step(ao)
    mdist.barrier(); torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    no = rows_o - 24 - 40 - 8
    t0 = time.perf_counter(); e0.record()
    for _ in range(no):
        env.step(ao)
    e1.record(); torch.cuda.synchronize(dev)
    wall = mdist.max_over_ranks(time.perf_counter() - t0, dev)
    gpu = mdist.max_over_ranks(e0.elapsed_time(e1) * 1e-3, dev)
    bo = (Lo.bytes_per_step() - 1 + 8 * Lo.obs_dim) * N
    out["gym_steps_rows_h24"] = {"value": n_total * no / wall, "us_per_step": gpu / no * 1e6, "ring_depth": 32, "obs_dim": Lo.obs_dim,
                                 "roofline": {"bound": "hbm", "achieved": bo / (gpu / no) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": bo / (gpu / no) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                              "algorithmic_bytes_per_launch": bo, "kernel": "step_multi_kernel<7> + obs_windows_k_multi_kernel<7, double>",
                                              "bytes_per_env_step": bo // N}}
    env.close()
    # PMC traffic of the three legs, where a counter pass taken on the kernels now running is committed (tools/gpu_profile_r06.sh)
    tg, tsrc = profile_json("traffic_general.json")
    for key in ("single_steps", "k_step_launches", "gym_steps_rows_h24", "k_step_3_of_a_kind"):
        rf = out[key]["roofline"]
        rf["traffic_source"] = tsrc
        e = (tg or {}).get(key)
        if e is not None and e.get("grids_per_gpu") == N:
            rf["traffic"] = e["hbm_bytes_per_launch"]
    # a single general step is one dependent round trip behind a launch boundary, like the single-instance one
    rf = out["single_steps"]["roofline"]
    model_us = LAT_FLOOR_US + rf["algorithmic_bytes_per_launch"] / (LAT_STREAM_GBS * 1e3)
    rf.update({"bound": "latency", "model_us": model_us, "frac_of_latency_model": model_us / out["single_steps"]["us_per_step"]})
    return out


def compact_line(detail, mode, detail_name):
    """The ONE stdout line: the headline fields, the headline's roofline + cpu_baseline, one {us, frac[, lat, t/a]} entry per side leg
    -- derived from the full record (`detail`, which goes to bench_detail.json / stderr) and kept under MAX_LINE characters: the
    driver keeps a bounded tail of stdout, and a record it cannot parse is a round without a measurement (round 4)."""
    def r4(x):
        return None if x is None else float(f"{x:.4g}")

    def leg(r, us_key=None):
        """{us per env-step (GPU time per launch or fleet step), HBM-roofline fraction[, fraction of the latency model, traffic / algorithmic]}"""
        if not r or "error" in r:
            return {"error": str((r or {}).get("error", "not run"))[:80]}
        q = r["roofline"]
        out = {"us": r4(r[us_key] if us_key else q.get("avg_launch_us")), "frac": r4(q["frac"])}
        if q.get("frac_of_latency_model") is not None:
            out["lat"] = r4(q["frac_of_latency_model"])
        if q.get("traffic") is not None and q.get("algorithmic_bytes_per_launch"):
            out["t/a"] = r4(q["traffic"] / q["algorithmic_bytes_per_launch"])
        rv = r.get("roofline_valu")
        if rv and rv.get("frac") is not None:       # the issue-side fraction; the leg is bound by whichever ceiling it is closer to
            out["valu"] = r4(rv["frac"])
            out["bound"] = "valu" if rv["frac"] > q["frac"] else "hbm"
        return out
    legs = {}
    for name, r in (detail.get("other") or {}).items():
        legs[name] = leg(r)
    hetero, general, closed = detail.get("hetero_h24_gym_steps"), detail.get("general_path_2g2b1grid"), detail.get("closed_loop_policy_gym_steps")
    if isinstance(hetero, dict):
        for k, v in hetero.items():
            if isinstance(v, dict) and "roofline" in v:
                legs["config5_" + k] = leg(v)
        if "error" in hetero:
            legs["config5"] = {"error": str(hetero["error"])[:80]}
    if isinstance(general, dict):
        for k, short in (("single_steps", "general_single_step"), ("k_step_launches", "general_k_step"), ("gym_steps_rows_h24", "general_gym_rows_h24"),
                         ("k_step_3_of_a_kind", "general_k_step_3_of_a_kind"), ("rbc_rollout", "general_rbc_rollout")):
            if k in general:
                legs[short] = leg(general[k], "us_per_step")
        if "error" in general:
            legs["general"] = {"error": str(general["error"])[:80]}
    if isinstance(closed, dict) and "us_per_step" in closed:
        legs["closed_loop_policy"] = {"us": r4(closed["us_per_step"])}
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: detail[k] for k in keep}
    cfg = detail["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "env_steps_per_step", "steps_per_launch", "grids_per_gpu", "grids_total", "mode", "series",
                                          "parallelism", "backend")}
    rf = detail["roofline"]
    line["roofline"] = {"bound": rf["bound"], "achieved": r4(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"], "frac": r4(rf["frac"]),
                        "traffic": rf["traffic"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                        "kernel": rf["kernel"], "avg_launch_us": r4(rf["avg_launch_us"]), "launches": rf["launches"],
                        "bytes_per_env_step": r4(rf["bytes_per_env_step"]), "frac_wall": r4(rf["frac_wall"]),
                        "traffic_source": rf.get("traffic_source")}
    rv = detail.get("roofline_valu")
    if rv and rv.get("frac") is not None:
        line["roofline"]["valu_frac"] = r4(rv["frac"])
    cpu = detail.get("cpu_baseline")
    if cpu is not None and "error" not in cpu:
        line["cpu_baseline"] = {"value": r4(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                "value_1thread": r4(cpu["value_1thread"]), "host_logical_cpus": cpu["host_logical_cpus"],
                                "sample": cpu["sample_short"]}
    else:
        line["cpu_baseline"] = cpu
    line["csrc_hash"] = detail["csrc_hash"]
    line["per_rank_env_steps_per_s"] = [r4(v) for v in detail["per_rank_env_steps_per_s"]]
    ma = detail["metrics_allreduce"]
    line["metrics"] = {"sum_last_reward": ma["sum_last_reward"], "mean_soc": ma["mean_soc"], "backend": ma["collective_backend"]}
    line["legs"] = legs
    line["detail"] = detail_name
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= MAX_LINE:                      # never let the record outgrow what the driver parses: the optional legs go first
        line["legs"] = {k: v for k, v in legs.items() if "error" not in v and k.startswith(("single_step", "config5", "general"))}
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= MAX_LINE:
        line["legs"] = {"dropped": "record too long: see " + str(detail_name)}
        text = json.dumps(line, separators=(",", ":"))
    return text


def _lib_factor_columns():
    from pymgrid_amd import _lib
    return _lib.FACTOR_COLUMNS


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
    src = dict(eng.batch.cols)
    if eng.batch.factorised:            # the sample's series, formed from the factors as the kernels form them
        from pymgrid_amd.generator import materialise_series
        src.update(materialise_series(eng.batch, n_rows=K + 1, n_grids=n))
        src = {k: v for k, v in src.items() if k not in _lib_factor_columns()}
    for k, v in src.items():
        if k in ("load_ts", "pv_ts"):
            cols[k] = v[:K + 1, :n].contiguous().cpu().numpy()
        elif k == "grid_ts":
            cols[k] = v[:K + 1, :, :n].contiguous().cpu().numpy()
        elif v.dim() == 1:
            a = np.ascontiguousarray(v[:n].cpu().numpy())          # (uniform columns are stride-0 views)
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
    # pick the OpenMP thread count that is fastest on this host (the container exposes every logical CPU of the box, but
    # its cgroup quota is smaller: above the quota the threads time-share and the rate collapses), then spend the rest of
    # the budget on it
    cands = sorted({c for c in (4, 8, 16, 32, 64, 128, cores) if c <= cores})
    probe = {c: run(c, seconds * 0.05)[0] for c in cands}
    best = max(probe, key=probe.get)
    vall, nall = run(best, seconds * 0.4)
    quota = None
    try:                                                       # cgroup v2 CPU quota of this container ("max" = none)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    return {"value": vall, "unit": "env-steps/s", "cores": best, "kind": "port",
            "value_1thread": v1, "host_logical_cpus": cores, "cgroup_cpu_quota": quota,
            "per_instance_python_loop_1core": per_instance,
            "thread_probe": {str(c): round(v) for c, v in probe.items()},
            "sample_short": f"first {n} grids x {K} steps of the bench batch, ~{seconds:.0f} s, oracle/mgx_oracle.c (C restatement, OpenMP)",
            "sample": f"first {n} grids x {K} steps of the benchmark batch, repeated for ~{seconds:.0f} s "
                      f"({n1 + nall} env-steps on 1 and {best} threads): oracle/mgx_oracle.c (scalar C restatement of "
                      f"the reference loop, OpenMP over tiles of 64 grids); the Python reference itself runs ~2e3 "
                      f"env-steps/s/core (BASELINE.md)"}


def valu_roofline(kname, launch_s, grids_per_launch, concurrent, dev):
    """The issue-side ceiling of a kernel: cycles its waves spent EXECUTING vector-ALU instructions (SQ_ACTIVE_INST_VALU x 4: the
    counter ticks in quad-cycles; summed over the waves of a launch; a committed PMC pass of this command, profiles/r*/valu.json,
    taken on the kernels now running) over the VALU cycles the chip had in the launch's duration (SIMDs x shader clock x time,
    measured live).  1.0 = every SIMD issued a VALU instruction in every cycle.  `concurrent`: launches running side by side."""
    vj, vsrc = profile_json("valu.json")
    e = (vj or {}).get("kernels", {}).get(kname.replace(" ", ""))
    if e is None:
        return {"bound": "valu", "achieved": None, "peak": None, "unit": "Gcycle/s", "frac": None, "source": vsrc}
    props = torch.cuda.get_device_properties(dev)
    simds = 4 * props.multi_processor_count
    ghz = float(vj.get("sclk_mhz") or 2400.0) * 1e-3
    per_round = e["valu_active_cycles_per_launch"] * concurrent * (grids_per_launch / e["grids_per_launch"])
    ach = per_round / launch_s / 1e9
    return {"bound": "valu", "achieved": ach, "peak": simds * ghz, "unit": "Gcycle/s", "frac": ach / (simds * ghz),
            "valu_instructions_per_wave_and_env_step": e.get("valu_insts_per_wave_step"), "source": vsrc,
            "what": "wave-cycles spent executing VALU instructions per second over SIMDs x shader clock"}


def measured_traffic(kernel, grids, chunk):
    """HBM bytes per launch of the kernel specialisation `kernel` ("step_k_kernel<3,8,double,false,true>"; a launch over `grids`
    grids and `chunk` steps) from the committed rocprofv3 PMC passes of THIS command (tools/gpu_profile_r06.sh ->
    profiles/<round>/traffic.json); None when no profile of that specialisation and launch shape is committed."""
    import glob
    want = kernel.replace(" ", "")
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("chunk") != chunk:
            continue
        # launches are recorded by size (threads = workgroups * 256; a workgroup owns 192..256 grids)
        for threads, e in sorted(d.get("by_launch_threads", {}).get(want, {}).items(), key=lambda kv: int(kv[0])):
            if grids <= int(threads) < 1.45 * grids:
                if d.get("csrc_hash") != CSRC_HASH:      # counters of OTHER kernels say nothing about these: no figure
                    stale = stale or f"{os.path.relpath(f, ROOT)}: STALE (taken on kernels {d.get('csrc_hash')}, running {CSRC_HASH})"
                    break
                return e["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, stale


CSRC_HASH = None      # source hash of the libmgx.so this process runs (set in main): profiles are only quoted when they carry it


def profile_json(name):
    """A committed counter summary (profiles/r*/<name>) taken on the kernels that are running, else (None, why not)."""
    import glob
    why = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("csrc_hash") == CSRC_HASH:
            return d, os.path.relpath(f, ROOT)
        why = why or f"{os.path.relpath(f, ROOT)}: STALE (taken on kernels {d.get('csrc_hash')}, running {CSRC_HASH})"
    return None, why


def main():
    args = parse()
    rc = self_launch(args)
    if rc is not None:
        raise SystemExit(rc)
    from pymgrid_amd import _lib
    from pymgrid_amd import distributed as mdist
    from pymgrid_amd.engine import StepEngine
    from pymgrid_amd.generator import generate
    rank, world, local = mdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the job has {world} rank(s): launch with `python bench.py --gpus N` "
                         f"(self-launching) or torch.distributed.run --nproc-per-node N")
    if args.launch_check:          # the N-rank launch path alone (runs on a CPU box with MGX_DIST_BACKEND=gloo)
        if world > 1:
            import torch.distributed as dist
            assert dist.get_world_size() == args.gpus
            on_gpu = dist.get_backend() != "gloo"
            if on_gpu:
                local = int(os.environ.get("MGX_FORCE_LOCAL_RANK", local))
                torch.cuda.set_device(local)
            cdev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
            ranks = mdist.gather_over_ranks(rank, cdev)
            assert ranks == list(range(world)), ranks
            # THE collective of the engine, for real: every rank contributes (rank + 1, 1) over the default backend (RCCL on a GPU
            # node; bounded, with the gloo fallback reported) -- the sums every rank must see are (W (W + 1) / 2, W)
            sums = mdist.all_reduce_metrics(torch.tensor([rank + 1.0, 1.0], dtype=torch.float64, device=cdev))
            ok = sums.cpu().tolist() == [world * (world + 1) / 2, float(world)]
            oks = mdist.gather_over_ranks(1.0 if ok else 0.0, cdev)
            if rank == 0:
                print(json.dumps({"launch_check": True, "n_gpus": world, "backend": dist.get_backend(), "ranks": ranks,
                                  "metrics_allreduce": {"sums": sums.cpu().tolist(), "ok_on_every_rank": all(v == 1.0 for v in oks),
                                                        "collective_backend": mdist.last_collective.get("backend"),
                                                        "error": mdist.last_collective.get("error")}}), flush=True)
            mdist.barrier()
            if mdist.last_collective.get("hung"):
                sys.stdout.flush()
                os._exit(0 if ok else 1)              # a thread is still stuck inside the backend
            dist.destroy_process_group()
        else:
            print(json.dumps({"launch_check": True, "n_gpus": 1, "backend": None, "ranks": [0]}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    _lib.build()
    for kv in args.tunable:
        name, _, value = kv.partition("=")
        _lib.set_tunable(name, int(value))
    global CSRC_HASH
    CSRC_HASH = _lib.built_hash(_lib.LIB_PATH) or _lib.source_hash()     # which kernels this run measures (profiles are quoted only if they match)
    local = int(os.environ.get("MGX_FORCE_LOCAL_RANK", local))     # tests: several ranks on one GPU (with a gloo backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == args.gpus
        want = os.environ.get("MGX_DIST_BACKEND", "nccl")
        if dist.get_backend() != want:
            raise SystemExit(f"backend {dist.get_backend()} != {want}")
    N, chunk = args.grids, args.chunk
    n_total = N * world
    other_series = "materialised" if args.series == "factorised" else "factorised"
    shards_of = {"factorised": 2, "materialised": 2}
    if args.shards is not None:
        shards_of = {k: max(1, args.shards) for k in shards_of}

    runs = {}

    def runner(series):
        """Engine + runner of one series layout (built on first use; both share the action pool and the parameter draw)."""
        if series not in runs:
            b = generate(n_total, n_steps=args.rows, seed=42, arch=args.arch, device=dev, rank=rank, world=world, series=series,
                         uniform_columns=(series == "factorised" and args.uniform_columns))
            pool = next(iter(runs.values())).pool if runs else None
            runs[series] = Runner(StepEngine(b), chunk, 7, shards_of[series], pool=pool, rank=rank, world=world)
        return runs[series]

    def env_runner(observations):
        """The Gym surface itself: BatchedMicrogridEnv.step from a Python loop (the bound step: one mgx_env_step call per env-step),
        rewards (and rows) in 4 rotating buffers.  Its own batch of the same draw (the env owns an engine)."""
        key = "env_obs" if observations else "env"
        if key not in runs:
            from pymgrid_amd import BatchedMicrogridEnv
            b = generate(n_total, n_steps=args.rows, seed=42, arch=args.arch, device=dev, rank=rank, world=world, series=args.series)
            env = BatchedMicrogridEnv(b, observations=observations, reuse_outputs=4)
            assert env._fp is not None, "the env did not bind its step (mgx_env_bind)"
            r = Runner(env.engine, chunk, 7, 2 if args.shards is None else max(1, args.shards), pool=runner(args.series).pool)
            r.env = env
            runs[key] = r
        return runs[key]

    run = runner(args.series)
    eng, batch = run.eng, run.eng.batch
    L = eng.layout
    LPS = max(1, args.launches_per_step)             # rounds per bench step

    device_state = {}

    def measure(mode, sharded, rounds, warmup, series=None, run=None, spread=False):
        run = run or runner(series or args.series)
        eng, S = run.eng, run.S
        sharded = sharded and S > 1
        fact = eng.batch.factorised
        ub = eng.batch.uniform_param_bytes()         # parameter bytes per grid NOT read: batch-uniform columns are held once
        run.shard(sharded)
        fn = getattr(run, mode)
        run.reset()
        # device pre-warm (untimed, part of set-up like the data generation): the first launches after start-up or after
        # an idle phase run on a cold device -- code-object load, and ~15 ms until the clocks settle under this load
        # (profiles/r01/exp_transient_cause.txt) -- so PREWARM_S seconds of the same rounds precede the W warm-up rounds,
        # with no idle gap between them
        r0 = run.rounds
        run.eng.fork()
        sampler = None
        if mode == args.mode and not device_state and rank == 0:      # the headline's pre-warm: what the device runs at under this load
            sampler = sample_device_state(device_state)
        t_end = time.perf_counter() + args.prewarm
        while time.perf_counter() < t_end:
            fn(8)
            if run.rounds - r0 > 64:                             # keep the launch queues short: ~64 rounds ahead at most
                run.eng.join(); torch.cuda.synchronize(dev); run.eng.fork(); r0 = run.rounds
        if sampler is not None:
            sampler.join(2.0)
        fn(warmup)
        first = run.rounds
        wall, gpu, _ = timed(run, fn, rounds, dev, mdist)
        # the spread of a timed region: a second pass directly behind it, with an event behind every round on every launch stream
        # (inside the timed region itself the extra host calls would be part of what is measured)
        round_us = timed(run, fn, min(rounds, 256), dev, mdist, per_round=True)[2] if spread else None
        walls = mdist.gather_over_ranks(wall, dev)
        wall, gpu = max(walls), mdist.max_over_ranks(gpu, dev)
        n_launch = (N + S - 1) // S if sharded else N             # grids per kernel launch
        obs_rows = mode == "step_env" and run.env._observations
        rich, full = mode == "fused_rich", mode == "step_full"     # every output of the reference's step: + done, status, log [, row]
        if mode in ("fused", "rbc", "fused_rich"):
            A8 = 8 * L.action_dim * chunk if mode == "rbc" else 0  # rbc: no action stream; + 1 id byte per grid, once
            per_launch = L.bytes_fused(chunk, done=run.done_stream or rich, status_trace=rich, log=rich, factorised=fact) - A8 \
                + (1 if mode == "rbc" else 0) - ub
            launches_per_round = 1
        else:                                                      # `chunk` single-step launches per round
            # single steps of a factorised batch read the grid's factors (2 ratios + 2 profile ids: 18 B) where the
            # materialised one reads 2 row values (16 B); no done byte.  With observation rows (H = 0): + 8 D written per grid;
            # with the log: + 8 L written and the pre-step SoC read
            # (rows: + the series' observation bounds the normalisation reads, lo and hi per component: 16 C_ts -- parameters like
            #  the module columns; until round 6 they were left out of the count and showed up as t/a = 1.14)
            c_ts = L.n_load + L.n_pv + 4 * int(L.has_grid)
            per_launch = L.bytes_per_step(log=full) - (0 if (run.done_stream or full) else 1) + (2 if fact else 0) - ub \
                + ((8 * L.obs_dim + 16 * c_ts) if (obs_rows or full) else 0)
            launches_per_round = chunk
        launches = rounds * launches_per_round                     # per stream
        per_launch_bytes = per_launch * N                          # one launch on every shard stream = all N grids
        avg_launch_s = gpu / launches
        achieved = per_launch_bytes / avg_launch_s / 1e9
        wall_launch_s = wall / launches
        ft = "true" if fact else "false"
        kname = {"fused": f"step_k_kernel<3,8,double,false,{ft}>", "step": "step_kernel<3,false>", "step_env": "step_kernel<3,false>",
                 "rbc": f"rollout_kernel<3,8,false,false,{ft}>", "fused_rich": f"step_k_kernel<3,16,double,true,{ft}>",
                 "step_full": "step_kernel<3,false>"}[mode]
        tkey = kname + ("+rows" if obs_rows else "") + ("+rows+log" if full else "")     # the key of this launch shape in traffic.json
        traffic, traffic_src = measured_traffic(tkey, n_launch, chunk)
        if traffic is not None and sharded:
            traffic *= S
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the same bytes over the WALL time per launch (what ms_per_step reports: host + queue included); `frac` is from
                # HIP events on the launch stream(s) over the same timed region
                "frac_wall": per_launch_bytes / wall_launch_s / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": per_launch_bytes, "kernel": kname, "series": "factorised" if fact else "materialised",
                "uniform_parameter_bytes_per_grid": ub,
                "bytes_per_env_step": per_launch / (chunk if mode in ("fused", "rbc", "fused_rich") else 1),
                "launches": launches, "avg_launch_us": avg_launch_s * 1e6, "avg_launch_us_wall": wall_launch_s * 1e6,
                "timed_rounds": [first, first + rounds]}
        if mode in ("step", "step_env", "step_full") and not sharded:
            # A single-step launch cannot overlap its predecessor (the next step needs this one's state): it is ONE dependent round
            # trip behind a launch boundary, not a stream.  Its ceiling is the guide's latency model, not the HBM peak:
            # t_model = launch boundary (1.5-1.9 us: LAT_FLOOR_US) + bytes / the rate the chip sustains (LAT_STREAM_GBS)
            model_us = LAT_FLOOR_US + per_launch_bytes / (LAT_STREAM_GBS * 1e3)
            roof.update({"bound": "latency", "model_us": model_us, "frac_of_latency_model": model_us / (avg_launch_s * 1e6),
                         "model": f"{LAT_FLOOR_US} us launch boundary + bytes / {LAT_STREAM_GBS / 1e3:.0f} TB/s (MI355X_MICROARCH.md); "
                                  f"`frac` stays the HBM-peak fraction for comparison with the streaming kernels"})
        if round_us:                                               # the spread inside the timed region (the headline only)
            srt = sorted(round_us)
            roof["round_us"] = {"n": len(srt), "min": srt[0], "median": srt[len(srt) // 2], "max": srt[-1], "mean": sum(srt) / len(srt),
                                "frac_at_median": per_launch_bytes * launches_per_round / (srt[len(srt) // 2] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "what": "cadence of every launch stream in each round of a SECOND pass (<= 256 rounds) directly behind the "
                                        "timed region (a HIP event behind every round on each launch stream: n = streams x rounds samples), "
                                        "rank 0.  The events themselves cost a few us per round (compare `mean` with avg_launch_us of the "
                                        "timed region, which has none): read this block as the SPREAD of the rounds, not as their level"}
        if sharded:
            roof.update({"launch": f"one round = {S} concurrent kernel launches, one per internal shard stream "
                                   f"(mgx_set_shards), {n_launch} grids each, never joined between rounds",
                         "concurrent_streams": S, "grids_per_kernel_launch": n_launch})
            if mode in ("step", "step_env", "step_full"):
                roof["launch"] = (f"every env-step = {S} launches of {n_launch} grids, one per shard stream, each shard's dependent chain "
                                  f"issued by a host thread of its own (mgx_set_launch_threads): a range's step k + 1 waits for its own "
                                  f"step k only; open loop (actions staged ahead), joined once behind the timed region")
        if mode in ("fused", "rbc", "fused_rich"):
            roof["kernel_avg_duration_us"] = run.kernel_durations_us(fn)
        if mode == "rbc" and fact:
            roof["note"] = ("with factorised series this kernel reads nothing per step (18.5 B/env-step are its reward / SoC writes): "
                            "it is bound by fp64 VALU issue, not by HBM -- see roofline_valu; the HBM fraction is reported for "
                            "completeness, the materialised form is the bandwidth-bound one")
        if mode in ("fused", "rbc", "fused_rich"):
            # the issue-side ceiling (counters are per KERNEL launch of n_launch grids; a round runs S of them side by side)
            roof_valu = valu_roofline(kname, avg_launch_s, n_launch, S if sharded else 1, dev)
        else:
            roof_valu = None
        run.shard(False)
        if rich:
            run.rich_outs = None                     # 2 x 1.2 GB of log buffers at N = 100 000
        if full:
            run.full_outs = None
        return {"value": n_total * rounds * chunk / wall, "rounds": rounds, "warmup_rounds": warmup, "wall_s": wall,
                "us_per_env_step": wall / (rounds * chunk) * 1e6, "ms_per_round": wall / rounds * 1e3,
                "roofline": roof, "roofline_valu": roof_valu, "per_rank_env_steps_per_s": [N * rounds * chunk / w for w in walls]}

    # the headline first (its launch indices in a rocprofv3 trace are then [prewarm + W x LPS, prewarm + (W + K) x LPS) per queue)
    results = {args.mode: measure(args.mode, sharded=args.mode in ("fused", "rbc"), rounds=args.steps * LPS, warmup=args.warmup * LPS,
                                  spread=True)}

    def guarded(what, fn):
        """The side legs must not take the headline down with them: a failure is reported in the record instead.  With several
        ranks a failure ends the job loudly (barriers inside the leg would hang the other ranks)."""
        err = None
        try:
            out = fn()
        except Exception as e:          # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
            print(f"bench.py[rank {rank}]: {what} failed: {err}", file=sys.stderr)
            if world > 1:
                raise
            out = None
        return out if err is None else {"error": err}

    side = (min(args.steps * LPS, SIDE_ROUNDS[0]), min(args.warmup * LPS, SIDE_ROUNDS[1]))
    if not args.no_side_modes:
        # the Gym cadence: single-step launches issued by one C call, and the Gym surface itself from a Python loop
        legs_only = [x for x in args.legs.split(",") if x] if args.legs is not None else None

        def want(name, default=True):
            return (name in legs_only) if legs_only is not None else default

        if args.mode != "step" and want("step"):
            results["step"] = guarded("step", lambda: measure("step", False, side[0], side[1]))
        if want("step_env"):
            results["step_env"] = guarded("step_env", lambda: measure("step_env", False, min(side[0], 64), min(side[1], 16), run=env_runner(False)))
        if want("step_env_obs", not args.no_closed_loop):
            results["step_env_obs"] = guarded("step_env_obs", lambda: measure("step_env", False, min(side[0], 64), min(side[1], 16),
                                                                              run=env_runner(True)))
        # the same Gym cadence as TWO dependent launch chains (grid ranges on two shard streams, each issued by its own host thread)
        if want("step_2chains"):
            results["step_2chains"] = guarded("step_2chains", lambda: measure("step", True, side[0], side[1]))
        if want("step_env_2chains", args.all_legs):     # (the Python-loop forms of the two-chain step: --all-legs; the one-call form above stays
                                                        #  in the default line as the record of the negative result)
            results["step_env_2chains"] = guarded("step_env_2chains", lambda: measure("step_env", True, min(side[0], 64), min(side[1], 16),
                                                                                      run=env_runner(False)))
        if want("step_env_obs_2chains", args.all_legs and not args.no_closed_loop):
            results["step_env_obs_2chains"] = guarded("step_env_obs_2chains", lambda: measure("step_env", True, min(side[0], 64), min(side[1], 16),
                                                                                              run=env_runner(True)))
        # every output of the reference's step inside the timed region (SURVEY 8(d) "full total"): per-grid done, genset status, the
        # balance / module log columns of every step -- fused and at the Gym cadence (there with the H = 0 observation row)
        if args.mode == "fused" and want("fused_rich"):
            results["fused_rich"] = guarded("fused_rich", lambda: measure("fused_rich", True, min(side[0], 64), min(side[1], 16)))
        if want("step_full", not args.no_closed_loop):
            results["step_full"] = guarded("step_full", lambda: measure("step_full", False, min(side[0], 64), min(side[1], 16)))
        if args.mode != "rbc" and want("rbc"):       # f1: the RuleBasedControl year on device (no host, no action stream)
            results["rbc"] = guarded("rbc", lambda: measure("rbc", True, side[0], side[1]))
        for key in ("env", "env_obs"):
            if key in runs:
                runs.pop(key).env.close()
    if not args.no_side_modes and args.all_legs:
        for mode in ("fused", "rbc"):
            if mode != args.mode and mode not in results:
                results[mode] = guarded(mode, lambda mode=mode: measure(mode, True, side[0], side[1]))
        # the other series layout: the fused kernel (as sharded as that layout likes it, and as ONE launch sequence) + the rollout
        if shards_of[args.series] > 1 and args.mode == "fused":      # the headline kernel as ONE launch sequence
            results["fused_one_stream"] = guarded("fused_one_stream", lambda: measure("fused", False, side[0], side[1]))
        results["fused_other_series"] = guarded("fused_other_series", lambda: measure("fused", True, side[0], side[1], other_series))
        if shards_of[other_series] > 1:
            results["fused_other_series_one_stream"] = guarded("fused_other_series_one_stream",
                                                               lambda: measure("fused", False, side[0], side[1], other_series))
        results["rbc_other_series"] = guarded("rbc_other_series", lambda: measure("rbc", True, side[0], side[1], other_series))
        if other_series in runs:       # 14 GB of [T, N] series: free them before the fleet leg
            runs.pop(other_series).eng.close()
            torch.cuda.empty_cache()

    # an agent IN the loop (the fused modes replay pre-staged actions): obs -> a small on-device policy -> env.step -> obs, from
    # Python, observation rows written every step (H = 0: 8 values per grid)
    closed = None
    if not args.no_side_modes and args.all_legs and not args.no_closed_loop:
        closed = guarded("closed_loop_policy_gym_steps", lambda: closed_loop_leg(args, n_total, dev, rank, world, mdist))

    # The GENERAL path: 2 gensets + 2 batteries + 1 grid per microgrid (module_container.py:355-413 allows any multiplicity)
    general = None
    if not args.no_side_modes and (args.legs is None or "general" in args.legs.split(",")):
        general = guarded("general_path_2g2b1grid", lambda: general_path_leg(args, N, n_total, chunk, dev, rank, world, mdist))
        torch.cuda.empty_cache()

    hetero = None
    if args.hetero_steps > 0:
        # --config 4: the fleet IS the job -- `--steps` bench steps of `chunk` fleet Gym steps each are what the head reports
        fleet_steps = args.steps * chunk if args.config == 4 else args.hetero_steps
        hetero = guarded("hetero_h24_gym_steps", lambda: hetero_gym_steps(N, dev, rank, world, fleet_steps, mdist, args.rows,
                                                                          args.series, args.series == "factorised" and args.uniform_columns,
                                                                          all_legs=args.all_legs, K_ring=args.fleet_ring))

    # metrics vector: episode-return sum + mean SoC, all-reduced over ranks (the ONLY collective; RCCL over xGMI)
    local_sums = eng.metrics(torch.stack([run.outs[0]["reward"][-1], batch.cols["soc"]]))
    per_rank_sums = mdist.gather_over_ranks(float(local_sums[0]), dev)
    sums = mdist.all_reduce_metrics(local_sums.clone())     # bounded: RCCL raising or hanging ends in a reported gloo fallback, not in a lost line
    mdist.barrier()

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:      # rank 0 only; for N > 1 a shorter sample while the other ranks wait
        try:
            cpu = cpu_baseline(eng, run.pool, args.cpu_seconds if world == 1 else min(args.cpu_seconds, 4.0))
        except Exception as e:          # noqa: BLE001  (a host without gcc / OpenMP must not cost the GPU numbers)
            print(f"bench.py: cpu_baseline failed: {type(e).__name__}: {e}", file=sys.stderr)
            cpu = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        main_r = results[args.mode]
        S = run.S if args.mode in ("fused", "rbc") else 1
        names = {"fused": "fused_launches", "step": "single_step_launches_one_call", "rbc": "rbc_rollout_on_device",
                 "fused_one_stream": "fused_launches_one_stream",
                 "fused_other_series": f"fused_launches_{other_series}",
                 "fused_other_series_one_stream": f"fused_launches_{other_series}_one_stream",
                 "rbc_other_series": f"rbc_rollout_{other_series}", "step_env": "single_step_launches_python_loop",
                 "step_env_obs": "single_step_launches_python_loop_with_rows",
                 "step_2chains": "single_step_launches_one_call_2chains", "step_env_2chains": "single_step_launches_python_loop_2chains",
                 "step_env_obs_2chains": "single_step_launches_python_loop_with_rows_2chains",
                 "fused_rich": "fused_launches_full_outputs", "step_full": "single_step_launches_full_outputs"}
        backend = None
        if world > 1:
            import torch.distributed as dist
            backend = dist.get_backend()
        wall_steps = main_r["wall_s"]
        rf = main_r["roofline"]
        head = {
            "metric": "microgrid env-steps/sec", "value": main_r["value"], "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall_steps / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N} generated 4-module grids (genset+battery+load+pv) per GPU, T={args.rows}, H=0, "
                                   f"normalised random actions (BASELINE configs[{args.config if args.config != 4 else 2}]"
                                   + (f": {n_total} grids over {world} GPUs" if args.config == 3 else "") + ")",
                       "env_steps_per_step": chunk * LPS, "steps_per_launch": 1 if args.mode == "step" else chunk,
                       "grids_per_gpu": N, "grids_total": n_total, "mode": args.mode, "series": args.series,
                       "parallelism": f"grids sharded x{world} ranks, no data-path collective"
                                      + (f"; {S} shard streams per GPU" if S > 1 else ""),
                       "backend": backend},
        }
        detail = dict(head)
        detail["config"] = dict(head["config"], **{
            "step": f"one bench step = {LPS} rounds of {chunk} consecutive env-steps of all grids of a rank",
            "uniform_parameter_columns": batch.uniform_columns(),
            "series_note": ("base profile id + ratio per grid; the product is formed in the kernel, bit-identical to the [T, N] arrays"
                            if args.series == "factorised" else "[T, N] float64 arrays"),
            "outputs": "reward + SoC per grid and step" + ("" if args.mode == "step" else " (streamed [K, N])")
                       + "; done is derived from the step counter (lock-step: the same for every grid), not streamed",
            "prewarm_seconds_per_mode": args.prewarm, "world_size": world})
        detail.update({
            "roofline": rf, "roofline_valu": main_r.get("roofline_valu"), "csrc_hash": CSRC_HASH, "cpu_baseline": cpu,
            "per_rank_env_steps_per_s": main_r["per_rank_env_steps_per_s"],
            "other": {names[m]: r for m, r in results.items() if m != args.mode},
            "metrics_allreduce": {"sum_last_reward": float(sums[0]), "mean_soc": float(sums[1]) / n_total,
                                  "per_rank_sum_last_reward": per_rank_sums,
                                  "collective_backend": mdist.last_collective["backend"], "collective_error": mdist.last_collective["error"],
                                  "collective_hung": mdist.last_collective.get("hung", False)},
            "closed_loop_policy_gym_steps": closed, "hetero_h24_gym_steps": hetero, "general_path_2g2b1grid": general,
            "device_state_under_load": device_state or None})

        if args.config == 4 and isinstance(hetero, dict) and "float64_rows" in hetero:
            # BASELINE configs[4]: the heterogeneous H = 24 fleet through the Gym surface WITH observation rows is what the job is;
            # a bench step = `chunk` fleet Gym steps of all grids of a rank (the Template-4 headline above stays in `other`)
            fl = hetero["float64_rows"]
            detail["other"] = dict(detail["other"], **{names[args.mode]: dict(main_r)})
            detail.update({"value": fl["value"], "ms_per_step": fl["us_per_step"] * chunk * 1e-3, "roofline": fl["roofline"],
                           "roofline_valu": None, "per_rank_env_steps_per_s": [fl["value"] / world] * world})
            detail["config"] = dict(detail["config"], workload=hetero["workload"] + f" (BASELINE configs[4]: {3 * (N // 3) * world} grids over "
                                    f"{world} GPUs), float64 rows", mode="fleet_gym_steps_rows", env_steps_per_step=chunk,
                                    steps_per_launch=1, grids_per_gpu=3 * (N // 3), grids_total=3 * (N // 3) * world)
            for k, v in (("bytes_per_env_step", fl["roofline"]["algorithmic_bytes_per_launch"] / (3 * (N // 3))), ("launches", fleet_steps),
                         ("timed_rounds", None)):
                detail["roofline"].setdefault(k, v)
        detail["leg_names"] = names
        text = compact_line(detail, args.mode, os.path.basename(args.detail) if args.detail else None)
        dtext = json.dumps(detail)
        if args.detail:
            try:
                with open(args.detail, "w") as fh:
                    fh.write(dtext + "\n")
            except OSError as e:
                print(f"bench.py: could not write {args.detail}: {e}", file=sys.stderr)
        print("bench_detail " + dtext, file=sys.stderr, flush=True)
        print(text, flush=True)
    if world > 1:
        import torch.distributed as dist
        mdist.barrier()                 # (control plane: gloo; a dead rank is named after MGX_CTRL_TIMEOUT_S instead of hanging RCCL)
        if mdist.last_collective.get("hung"):       # a helper thread is still stuck inside RCCL: tearing the group down would hang too
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
