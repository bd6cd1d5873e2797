"""Heterogeneous fleets (BASELINE config 5): microgrids with different module sets / horizons cannot share one SoA
batch (a batch has one layout = one kernel specialisation), so a fleet is bucketed by layout
(``scenario.bucket_by_layout``) and every bucket gets its own ``MicrogridBatch`` + engine.  Buckets are independent.
They are issued back to back on the caller's stream by default: at 10^4..10^5 grids per bucket the kernels fill the chip
and the step is bound by the host's launch rate, where per-bucket HIP streams (``streams=True``: fork / join events
around every bucket) were measured 2.7x SLOWER (103 vs 39 us per fleet step, 3 buckets of 33k grids, H = 24);
streams only pay for many tiny buckets.
"""
import numpy as np
import torch

from .batch import MicrogridBatch
from .envs import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv
from .scenario import bucket_by_layout


class BucketedFleet:
    """N microgrids of mixed layouts behind one step()/reset() surface.

    ``step`` takes / returns lists with one entry per bucket (tensors of that bucket's shape); ``scatter`` puts a
    per-grid quantity (reward, done, ...) back into fleet order.
    """

    def __init__(self, grids, device="cuda", discrete=False, streams=False, **env_kwargs):
        self.n_grids = len(grids)
        self.device = torch.device(device)
        self.buckets = list(bucket_by_layout(grids).items())          # [(key, [indices])]
        cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
        self.envs, self.index = [], []
        for _, idx in self.buckets:
            self.envs.append(cls(MicrogridBatch.from_grids([grids[i] for i in idx], device=device), **env_kwargs))
            self.index.append(torch.as_tensor(np.asarray(idx), device=self.device))
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.envs] \
            if (streams and self.device.type == "cuda") else []

    @classmethod
    def from_batches(cls, batches, discrete=False, streams=False, **env_kwargs):
        """Fleet over ready-made ``MicrogridBatch`` objects (e.g. ``generator.generate`` per architecture): bucket k owns
        fleet positions [sum(n_0..n_{k-1}), ... + n_k)."""
        self = cls.__new__(cls)
        self.device = batches[0].device
        env_cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
        self.envs, self.index, self.buckets, start = [], [], [], 0
        for b in batches:
            n = b.layout.n_grids
            self.envs.append(env_cls(b, **env_kwargs))
            self.index.append(torch.arange(start, start + n, device=self.device))
            self.buckets.append((b.layout, range(start, start + n)))
            start += n
        self.n_grids = start
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.envs] \
            if (streams and self.device.type == "cuda") else []
        return self

    def __len__(self):
        return self.n_grids

    def _each(self, fn):
        """Run fn(env, k) for every bucket: back to back on the caller's stream, or (``streams=True``) each on its own
        stream with the caller's stream waiting for all of them."""
        if not self.streams:
            return [fn(env, k) for k, env in enumerate(self.envs)]
        cur = torch.cuda.current_stream(self.device)
        out = []
        for k, (env, st) in enumerate(zip(self.envs, self.streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out.append(fn(env, k))
        for st in self.streams:
            cur.wait_stream(st)
        return out

    def reset(self):
        return self._each(lambda env, k: env.reset())

    def step(self, actions, **kw):
        """actions: list with one tensor per bucket.  Returns (obs_list, reward_list, done_list, info_list)."""
        res = self._each(lambda env, k: env.step(actions[k], **kw))
        return tuple(list(x) for x in zip(*res))

    def sample_action(self, generator=None):
        return [env.sample_action(generator=generator) for env in self.envs]

    def scatter(self, per_bucket):
        """[tensor [n_b, ...] per bucket] -> one tensor [N, ...] in the order the grids were given."""
        first = per_bucket[0]
        out = torch.empty((self.n_grids,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
        for idx, v in zip(self.index, per_bucket):
            out[idx] = v
        return out

    def close(self):
        for env in self.envs:
            env.close()


class PerGridWindowEnv:
    """Per-grid random episode windows (the per-microgrid ``FixedLengthStochasticTrajectory`` of the reference,
    microgrid/trajectory/stochastic.py:15-30, for a batch): every grid gets its OWN start row, all episodes have the
    same length, so the batch still advances in lock-step and the hot path is unchanged.

    How: at ``reset()`` the rows ``[start_i, start_i + length + H]`` of every grid's series are gathered ONCE into a
    short window buffer ``[length + H + 1, N]`` (a few MB) and the engine steps over that buffer from row 0.  The
    observation bounds stay those of the full series, as in the reference.
    """

    def __init__(self, full_batch, trajectory_length, discrete=False, generator=None, **env_kwargs):
        L = full_batch.layout
        if L.n_load != 1 or L.n_pv != 1:
            raise NotImplementedError("per-grid windows need one load and one renewable module per grid")
        self.full = full_batch
        self.length = int(trajectory_length)
        self.rows = self.length + L.horizon + 1
        if self.rows > L.final_step - L.initial_step:
            raise ValueError(f'Cannot create a trajectory of length {self.length}'
                             f'between initial_step ({L.initial_step}) and final_step ({L.final_step})')
        self.generator = generator
        dev = full_batch.device
        N = L.n_grids
        cols = dict(full_batch.cols)                   # parameters / state / bounds are shared with the full batch
        cols["load_ts"] = torch.empty(self.rows, N, dtype=torch.float64, device=dev)
        cols["pv_ts"] = torch.empty(self.rows, N, dtype=torch.float64, device=dev)
        if L.has_grid:
            cols["grid_ts"] = torch.empty(self.rows, 4, N, dtype=torch.float64, device=dev)
        from dataclasses import replace
        wl = replace(L, n_steps=self.rows, initial_step=0, final_step=self.length)
        self.window = MicrogridBatch(wl, cols, forecast_noise=full_batch.forecast_noise)
        cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
        self.env = cls(self.window, **env_kwargs)
        self.starts = None
        self._k = torch.arange(self.rows, device=dev).unsqueeze(1)            # [rows, 1]

    def draw_starts(self):
        """initial_i ~ U{initial_step, ..., final_step - length - H - 1} (so that the forecast window of the last
        step still lies inside the series)."""
        L = self.full.layout
        hi = L.final_step - self.rows + 1
        return torch.randint(L.initial_step, hi, (L.n_grids,), device=self.full.device, generator=self.generator)

    def reset(self, starts=None):
        self.starts = self.draw_starts() if starts is None else torch.as_tensor(starts, device=self.full.device)
        idx = self._k + self.starts.unsqueeze(0)                               # [rows, N] absolute rows
        self.window.cols["load_ts"].copy_(torch.gather(self.full.cols["load_ts"], 0, idx))
        self.window.cols["pv_ts"].copy_(torch.gather(self.full.cols["pv_ts"], 0, idx))
        if self.full.layout.has_grid:
            idx4 = idx.unsqueeze(1).expand(-1, 4, -1)
            self.window.cols["grid_ts"].copy_(torch.gather(self.full.cols["grid_ts"], 0, idx4))
        return self.env.reset()

    def step(self, action, **kw):
        return self.env.step(action, **kw)

    def __getattr__(self, name):              # everything else (engine, action_space, sample_action, ...) as the env
        return getattr(self.env, name)


class StreamShards:
    """Independent shards of a fleet -- each a ``MicrogridBatch`` with its own engine -- driven on one HIP stream each and
    NOT joined between calls.

    Why: grids never interact, so the launch sequence of one shard owes nothing to another's.  On one stream every
    fused launch has a ramp-up (parameters, ring fill) and a tail (the last waves) during which the chip is not full;
    two shard sequences on two streams run out of phase and fill each other's gaps: measured 70-74 us instead of
    76-78 us per 64-step round of 100 000 grids (2 x 50 000; three shards 73 us, four 76 us:
    ``profiles/r01/exp_two_streams.txt``).  Outputs are valid on the caller's stream after ``join()``.
    """

    def __init__(self, batches, **engine_kwargs):
        from .engine import StepEngine
        self.engines = [StepEngine(b, **engine_kwargs) for b in batches]
        self.device = self.engines[0].device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.engines]
        self.n_grids = sum(e.N for e in self.engines)

    def __len__(self):
        return len(self.engines)

    def fork(self):
        """Shard streams wait for the caller's stream (inputs produced there are then safe to read)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(cur)

    def join(self):
        """The caller's stream waits for every shard stream (outputs are then safe to read there)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def each(self, fn):
        """fn(engine, shard_index) on every shard's stream; returns the results, does not join."""
        out = []
        for k, (eng, st) in enumerate(zip(self.engines, self.streams)):
            with torch.cuda.stream(st):
                out.append(fn(eng, k))
        return out

    def _guard(self, k, *tensors):
        """Tell the caching allocator that shard k's stream uses these tensors (they were allocated on another stream and
        must not be recycled while a launch that reads / writes them is still queued here)."""
        for t in tensors:
            if isinstance(t, dict):
                self._guard(k, *t.values())
            elif torch.is_tensor(t):
                t.record_stream(self.streams[k])

    def step_k(self, actions, outs=None, **kw):
        """``StepEngine.step_k`` per shard: actions / outs are lists with one entry per shard."""
        def one(eng, k):
            self._guard(k, actions[k], None if outs is None else outs[k])
            return eng.step_k(actions[k], out=None if outs is None else outs[k], **kw)
        return self.each(one)

    def rollout_discrete(self, ids, tables, K, outs=None, **kw):
        def one(eng, k):
            self._guard(k, ids[k], None if outs is None else outs[k])
            return eng.rollout_discrete(ids[k], tables[k], K, out=None if outs is None else outs[k], **kw)
        return self.each(one)

    def reset(self, initial_step=None):
        self.each(lambda eng, k: eng.reset(initial_step, want_obs=False))

    def close(self):
        for eng in self.engines:
            eng.close()
