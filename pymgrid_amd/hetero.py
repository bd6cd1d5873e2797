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


def L_multi(layout):
    return layout.n_load != 1 or layout.n_pv != 1


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
        """actions: list with one tensor per bucket.  Returns (obs_list, reward_list, done_list, info_list).

        Buckets on the caller's stream go through ONE call of the C ABI (``mgx_fleet_step``): every bucket's step launch
        and, where a bucket's observation ring is used up, its window prefetch are issued from C back to back -- three
        Python-level ``env.step`` calls cost ~30 us, more than the kernels of a 100 000-grid fleet step take."""
        if self.streams or any(env.raise_errors for env in self.envs) or any(isinstance(a, dict) for a in actions) \
                or any(isinstance(env, DiscreteBatchedMicrogridEnv) and L_multi(env.layout) for env in self.envs):
            res = self._each(lambda env, k: env.step(actions[k], **kw))
            return tuple(list(x) for x in zip(*res))
        return self._step_fused(actions, **kw)

    def _step_fused(self, actions, normalized=True):
        import ctypes as C
        from . import _lib
        from .engine import _raw_stream
        items = getattr(self, "_items", None)
        if items is None:
            items = self._items = (_lib.FleetItem * len(self.envs))()
            for it, env in zip(items, self.envs):
                it.struct_size = C.sizeof(_lib.FleetItem)
                it.handle = env.engine._h.value
                if isinstance(env, DiscreteBatchedMicrogridEnv):
                    it.table, it.n_actions = env.engine._table_ptr(env._table)
        obs_l, reward_l, done_l, info_l, refills = [], [], [], [], []
        for it, env, a in zip(items, self.envs, actions):
            e = env.engine
            discrete = isinstance(env, DiscreteBatchedMicrogridEnv)
            if discrete:
                if not (torch.is_tensor(a) and a.dtype == torch.int32 and a.is_contiguous() and a.device == e.device
                        and tuple(a.shape) == (e.N,)):
                    a = torch.as_tensor(np.asarray(a.cpu() if torch.is_tensor(a) else a), device=e.device).to(torch.int32).contiguous()
                it.action_id = a.data_ptr()
            else:
                a = e._check_actions(a, ())
                it.actions = None if a is None else a.data_ptr()
            want_obs, out = env._obs_target()
            reward = e._empty(e.N)
            done = e._empty(e.N, dtype=torch.uint8)
            obs = (out["obs"] if out else e._obs_buf(None)) if want_obs else None
            log = e._empty(e.log_dim, e.N) if env._keep_log else None
            refill = env._ring is not None and not want_obs
            it.reward, it.done = reward.data_ptr(), done.data_ptr()
            it.obs = None if obs is None else obs.data_ptr()
            it.log = None if log is None else log.data_ptr()
            it.refill_ring = env._ring.data_ptr() if refill else None
            it.refill_K = env.obs_prefetch if refill else 0
            refills.append(refill)
            obs_l.append(obs); reward_l.append(reward); done_l.append(done.view(torch.bool)); info_l.append({} if log is None else {"log": log})
        e0 = self.envs[0].engine
        idx = e0._dev_index
        if e0._only_device or torch.cuda.current_device() == idx:
            rc = e0._lib.mgx_fleet_step(items, len(self.envs), 1 if normalized else 0, _raw_stream(idx))
        else:
            with torch.cuda.device(idx):
                rc = e0._lib.mgx_fleet_step(items, len(self.envs), 1 if normalized else 0, _raw_stream(idx))
        _lib.check(rc)
        for k, (env, refill) in enumerate(zip(self.envs, refills)):
            if env._ring is not None:
                if refill:
                    env._ring_pos = 0
                    obs_l[k] = env._ring[0]
                else:
                    env._ring_pos += 1
            obs_l[k] = env._select_obs(obs_l[k])
            if info_l[k]:
                env._log_rows.append(info_l[k]["log"])
                env._shaped_rows.append(reward_l[k].clone())
        return obs_l, reward_l, done_l, info_l

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
    """Per-grid random episodes: what a population of reference microgrids does when each has its own
    ``trajectory_func`` (microgrid.py:205-225) -- ``FixedLengthStochasticTrajectory(trajectory_length)`` per grid
    (trajectory/stochastic.py:15-30: own start row, common length) or, with ``trajectory_length=None``,
    ``StochasticTrajectory`` per grid (trajectory/stochastic.py:9-12: own start row AND own final step, so the grids
    report ``done`` at different steps).

    The batch keeps one step counter: at ``reset()`` a HIP kernel (``mgx_reset_windows``) gathers every grid's rows
    ``[start_i, start_i + max length + H]`` of the full series into window buffers the step kernels walk from row 0, and
    ``done`` is per grid.  Observation bounds stay those of the full series, as in the reference.  Draws come from a
    torch generator on the device (the reference draws from numpy's global stream, one microgrid at a time).
    """

    def __init__(self, full_batch, trajectory_length=None, discrete=False, generator=None, **env_kwargs):
        L = full_batch.layout
        if L.n_load != 1 or L.n_pv != 1:
            raise NotImplementedError("per-grid windows need one load and one renewable module per grid")
        self.full = full_batch
        self.length = None if trajectory_length is None else int(trajectory_length)
        if self.length is not None and L.final_step - L.initial_step < self.length:
            raise ValueError(f'Cannot create a trajectory of length {self.length}'
                             f'between initial_step ({L.initial_step}) and final_step ({L.final_step})')
        self.generator = generator
        cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
        self.env = cls(full_batch, **env_kwargs)
        self.starts = self.lengths = None

    def draw(self):
        """(starts, lengths): FixedLengthStochasticTrajectory / StochasticTrajectory draws, one per grid."""
        L, dev, N = self.full.layout, self.full.device, self.full.layout.n_grids
        lo, hi = L.initial_step, L.final_step

        def randint(low, high):              # elementwise np.random.randint(low, high): high exclusive, tensors allowed
            u = torch.rand(N, device=dev, generator=self.generator, dtype=torch.float64)
            span = torch.as_tensor(high, device=dev) - torch.as_tensor(low, device=dev)
            return (torch.as_tensor(low, device=dev) + torch.clamp((u * span).floor().long(), max=span - 1)).to(torch.int32)
        if self.length is not None:
            if hi - self.length <= lo:       # np.random.randint(initial, final - length) needs a non-empty range
                return torch.full((N,), lo, dtype=torch.int32, device=dev), None
            return randint(lo, hi - self.length), None
        starts = randint(lo, hi - 2)
        finals = randint(starts.long(), hi)
        lengths = torch.clamp(finals - starts, min=1).to(torch.int32)       # a zero-length draw steps once (done at once)
        return starts, lengths

    def reset(self, starts=None, lengths=None):
        if starts is None:
            starts, lengths = self.draw()
        dev = self.full.device
        self.starts = torch.as_tensor(np.asarray(starts.cpu() if torch.is_tensor(starts) else starts), device=dev).to(torch.int32)
        self.lengths = None if lengths is None else \
            torch.as_tensor(np.asarray(lengths.cpu() if torch.is_tensor(lengths) else lengths), device=dev).to(torch.int32)
        max_len = self.length if self.lengths is None else int(self.lengths.max().item())
        if max_len is None:
            raise ValueError("lengths are required when the env was built without a trajectory_length")
        return self.env.reset_windows(self.starts, self.lengths, max_len)

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
