#!/usr/bin/env python3
"""Round-3 A/B on the GPU box: materialised vs factorised series, done as bytes / bits / derived, for the fused kernel, the
rule-based rollout and the single-step (Gym) cadence; hipGraph replays of single steps on one and two streams.
   python tools/exp_r3_modes.py [--grids 100000] [--what fused,rbc,step,graph]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, sync, rounds, warm):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(rounds):
        fn()
    sync()
    return (time.perf_counter() - t0) / rounds * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", type=int, default=100_000)
    ap.add_argument("--rows", type=int, default=8760)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--what", default="fused,rbc,step,graph")
    ap.add_argument("--series", default="materialised,factorised")
    args = ap.parse_args()
    from pymgrid_amd import StepEngine
    from pymgrid_amd.generator import generate
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    from pymgrid_amd.rbc import default_priority_ids
    dev = torch.device("cuda:0")
    N, T, K = args.grids, args.rows, args.chunk
    what = set(args.what.split(","))
    sync = lambda: torch.cuda.synchronize(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    pool = torch.rand(4, K, N, 3, dtype=torch.float64, device=dev, generator=gen)
    lists = get_priority_lists(True, True, False, False)
    table = table_array(lists)
    ids = None
    for name in args.series.split(","):
        b = generate(N, n_steps=T, seed=42, arch="genset+battery", device=dev, series=name)
        if ids is None:
            ids = torch.from_numpy(default_priority_ids(b, lists, remove_redundant_gensets=False)).to(dev)
        eng = StepEngine(b)
        L = eng.layout
        outs = [dict(reward=torch.empty(K, N, dtype=torch.float64, device=dev),
                     done=torch.empty(K, N, dtype=torch.uint8, device=dev),
                     soc_trace=torch.empty(K, N, dtype=torch.float64, device=dev)) for _ in range(4)]
        state = {"r": 0}

        def room():
            if eng.current_step + K > L.final_step:
                eng.reset(want_obs=False)

        for shards in (1, 2):
            if "fused" in what:
                for done in ("u8", "bits", "none"):
                    eng.set_shards(shards)
                    eng.set_done_format(done == "bits")
                    o = [dict(x) for x in outs]
                    if done == "bits":
                        for x in o:
                            x["done"] = torch.empty(K, (N + 15) // 16, dtype=torch.int16, device=dev)

                    def fn():
                        room()
                        eng.step_k(pool[state["r"] % 4], out=o[state["r"] % 4], reward=True, done=done != "none", soc_trace=True)
                        state["r"] += 1
                    eng.reset(want_obs=False)
                    eng.fork()
                    us = timeit(fn, lambda: (eng.join(), sync(), eng.fork()), 600, 300)
                    eng.join(); sync()
                    B = L.bytes_fused(K, done=done != "none", done_bits=done == "bits", factorised=b.factorised) * N
                    print(f"{name:13s} fused  shards={shards} done={done:5s}: {us:7.2f} us / {K} steps  {N * K / us / 1e3:7.2f} G env-steps/s  "
                          f"{B / K / N:6.2f} B/step  frac {B / us / 1e3 / 8000:.3f}", flush=True)
                eng.set_done_format(False)
            if "rbc" in what:
                for done in (True, False):
                    eng.set_shards(shards)

                    def fn():
                        room()
                        eng.rollout_discrete(ids, table, K, out=outs[state["r"] % 4], reward=True, done=done, soc_trace=True)
                        state["r"] += 1
                    eng.reset(want_obs=False)
                    eng.fork()
                    us = timeit(fn, lambda: (eng.join(), sync(), eng.fork()), 600, 300)
                    eng.join(); sync()
                    B = (L.bytes_fused(K, done=done, factorised=b.factorised) - 24 * K + 1) * N
                    print(f"{name:13s} rbc    shards={shards} done={str(done):5s}: {us:7.2f} us / {K} steps  {N * K / us / 1e3:7.2f} G env-steps/s  "
                          f"{B / K / N:6.2f} B/step  frac {B / us / 1e3 / 8000:.3f}", flush=True)
        eng.set_shards(1)
        if "step" in what:
            for done in (True, False):
                def fn():
                    room()
                    o = outs[state["r"] % 4]
                    eng.step_many(pool[state["r"] % 4], out=dict(reward=o["reward"], done=o["done"]), done=done)
                    state["r"] += 1
                eng.reset(want_obs=False)
                us = timeit(fn, sync, 200, 100) / K
                B = (L.bytes_per_step() - (0 if done else 1)) * N
                print(f"{name:13s} step_many done={str(done):5s}: {us:6.2f} us per env-step  {N / us / 1e3:6.2f} G env-steps/s  frac {B / us / 1e3 / 8000:.3f}",
                      flush=True)
        if "graph" in what:
            # K single steps captured in one hipGraph (device counter), replayed: the launch path without the host
            eng.reset(want_obs=False)
            eng.use_device_counter(True)
            o = outs[0]
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                eng.step_many(pool[0], out=dict(reward=o["reward"], done=o["done"]))          # warm-up on the capture stream
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    eng.step_many(pool[0], out=dict(reward=o["reward"], done=o["done"]))
            torch.cuda.current_stream(dev).wait_stream(side)
            sync()
            reps = (T - 3 * K) // K
            for _ in range(2):
                eng.use_device_counter(False); eng.reset(want_obs=False); eng.use_device_counter(True)
                sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    gr.replay()
                sync()
                us = (time.perf_counter() - t0) / reps / K * 1e6
            print(f"{name:13s} graph replay of {K} single steps, one stream: {us:6.2f} us per env-step  frac {L.bytes_per_step() * N / us / 1e3 / 8000:.3f}",
                  flush=True)
            eng.use_device_counter(False)
            del gr
        eng.close()
    if "graph" in what:
        # two half batches on two streams, captured in ONE graph: the launch latency of one chain under the other's data phase
        for S in (2, 4):
            per = N // S
            engs = [StepEngine(generate(N, n_steps=T, seed=42, arch="genset+battery", device=dev, rank=r, world=S, series="factorised"))
                    for r in range(S)]
            acts = [torch.rand(K, per, 3, dtype=torch.float64, device=dev, generator=gen) for _ in range(S)]
            rew = [torch.empty(K, per, dtype=torch.float64, device=dev) for _ in range(S)]
            don = [torch.empty(K, per, dtype=torch.uint8, device=dev) for _ in range(S)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
            for e in engs:
                e.reset(want_obs=False)
                e.use_device_counter(True)
            cur = torch.cuda.current_stream(dev)

            def body():
                for e, a, r, d, st in zip(engs, acts, rew, don, streams):
                    st.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(st):
                        e.step_many(a, out=dict(reward=r, done=d))
                for st in streams:
                    torch.cuda.current_stream(dev).wait_stream(st)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                body()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    body()
            cur.wait_stream(side)
            sync()
            reps = (T - 3 * K) // K
            for _ in range(2):
                for e in engs:
                    e.use_device_counter(False); e.reset(want_obs=False); e.use_device_counter(True)
                sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    gr.replay()
                sync()
                us = (time.perf_counter() - t0) / reps / K * 1e6
            print(f"factorised    graph replay of {K} single steps, {S} chains of {per} grids: {us:6.2f} us per env-step of {N} grids  "
                  f"frac {engs[0].layout.bytes_per_step() * N / us / 1e3 / 8000:.3f}", flush=True)
            for e in engs:
                e.use_device_counter(False)
                e.close()
            del gr


if __name__ == "__main__":
    main()
