"""Heterogeneous fleets (BASELINE config 5): microgrids with different module sets / horizons cannot share one SoA
batch (a batch has one layout = one kernel specialisation), so a fleet is bucketed by layout
(``scenario.bucket_by_layout``) and every bucket gets its own ``MicrogridBatch`` + engine.  Buckets are independent.

A fleet step is ONE call of the C ABI (``mgx_fleet_step``) and one ``fleet_step_kernel_v`` launch for all layouts (up to 5 per launch); the
observation rings of the buckets are renewed ahead of time on the engines' prefetch streams (``refill="ahead"``, K = 32 by default) or as
chunks inside the step launches (``refill="chunks"``): 29.5-32 / 33-35 us per 100 000-grid step at H = 24.  Per-bucket HIP
streams (``streams=True``: fork / join events around every bucket, one ``env.step`` each) were measured 2.7x slower than
back-to-back launches at 33k grids per bucket (host-bound) and only pay for many tiny buckets.
"""
import os

import numpy as np
import torch

from . import _lib
from .batch import MicrogridBatch
from .engine import _raw_stream
from .envs import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv, ObsViews
from .scenario import bucket_by_layout


def L_multi(layout):
    return layout.multi


class _BoundFleet:
    """What ``BucketedFleet.step`` needs once every bucket's handle carries an env plan (``mgx_env_bind``) and the fleet steps through
    ``mgx_fleet_env_step``: the handle / action-pointer arrays of the C call and the pre-built tensors a step returns.  The handles
    walk the rotating output slots and the observation rings themselves; ``slot`` / ``p`` mirror their positions."""
    __slots__ = ("fn", "hs", "ptrs", "n", "dev", "guard", "adt", "ashape", "R", "slot", "p", "nring", "obs", "rew", "keep", "VB")


class BucketedFleet:
    """N microgrids of mixed layouts behind one step()/reset() surface.

    ``step`` takes / returns lists with one entry per bucket (tensors of that bucket's shape); ``scatter`` puts a
    per-grid quantity (reward, done, ...) back into fleet order.
    """

    def __init__(self, grids, device="cuda", discrete=False, streams=False, reuse_outputs=0, fused=True, refill="ahead",
                 stagger=None, **env_kwargs):
        self._init_fused(reuse_outputs, fused, refill, stagger)
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
        self._decide_fused()

    def _decide_fused(self):
        """Fused mode (one ``mgx_fleet_step`` call per fleet step, ring refills in chunks inside the step launches) unless
        the fleet was asked for per-bucket streams / per-env steps or holds envs that need host work per step."""
        self.fused = bool(self._want_fused and not self.streams and not any(env.raise_errors for env in self.envs)
                          and not any(isinstance(env, DiscreteBatchedMicrogridEnv) and L_multi(env.layout) for env in self.envs))
        for env in self.envs:
            env._chunked = self.fused and self.refill == "chunks" and not L_multi(env.layout)
            env._fleet_owned = self.fused
            env._fleet_ref = self if self.fused else None
            env._fast_ok = not self.fused       # a fused fleet steps its envs through mgx_fleet_step: no bound env steps
            env._rebind_fast()
        # (stagger=True keeps the fleet on its per-step plans: the bound step, mgx_fleet_env_step, walks rings from phase 0 only --
        #  _bind_eligible -- which costs ~3 us of host time per fleet step; staggering measured no gain, it is off by default)
        # stagger: bucket j's rings change j * K / n_buckets steps before bucket 0's, so that the buckets' ring refills -- each a
        # burst of K row blocks on a prefetch stream -- start at different fleet steps instead of all at once
        ringed = [env for env in self.envs if env.obs_prefetch]
        for j, env in enumerate(ringed):
            env._ring_phase = (j * env.obs_prefetch) // len(ringed) if (self.stagger and self.refill == "ahead") else 0

    def _init_fused(self, reuse_outputs, fused=True, refill="ahead", stagger=None):
        # refill: how a fused fleet renews its observation rings -- "ahead" (default): the whole ring after next as one launch
        # per bucket on the engines' prefetch streams, running beside the following K step launches (what single envs do;
        # 29.5 us per 100k-grid step at K = 16); "chunks": 1/(K-1) of the next ring inside every step launch (33-34 us at
        # K = 8, its best depth; the even cadence suits hosts that issue slower than ~11 us per step)
        if refill not in ("chunks", "ahead"):
            raise ValueError("refill must be 'chunks' or 'ahead'")
        self.refill = refill
        self.stagger = bool(stagger)
        # reuse_outputs = R > 0: step() returns reward / done as views into R rotating buffers per bucket (valid for R - 1
        # further steps) instead of fresh tensors -- no allocation on the hot path
        self.reuse_outputs = int(reuse_outputs)
        self._want_fused = bool(fused)
        self._plans, self._n_steps, self._out_reward, self._out_done, self._out_obs = {}, 0, None, None, None
        self._bound, self._bind_ok = None, True     # the bound step (mgx_fleet_env_step): see _try_bind

    @classmethod
    def from_batches(cls, batches, discrete=False, streams=False, reuse_outputs=0, fused=True, refill="ahead", stagger=None,
                     **env_kwargs):
        """Fleet over ready-made ``MicrogridBatch`` objects (e.g. ``generator.generate`` per architecture): bucket k owns
        fleet positions [sum(n_0..n_{k-1}), ... + n_k)."""
        self = cls.__new__(cls)
        self._init_fused(reuse_outputs, fused, refill, stagger)
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
        self._decide_fused()
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

        def guard(x):          # allocated on a bucket stream, read on the caller's: the caching allocator must know (else
            if torch.is_tensor(x):                          # it may recycle the block while the caller's reads are pending)
                x.record_stream(cur)
            elif isinstance(x, dict):
                for v in x.values():
                    guard(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    guard(v)
        guard(out)
        return out

    def reset(self):
        return self._each(lambda env, k: env.reset())

    def step(self, actions, **kw):
        """actions: list with one tensor per bucket.  Returns (obs_list, reward_list, done_list, info_list).

        Buckets on the caller's stream go through ONE call of the C ABI (``mgx_fleet_step``): every bucket's step launch
        and, where a bucket's observation ring is used up, its window prefetch are issued from C back to back -- three
        Python-level ``env.step`` calls cost ~30 us, more than the kernels of a 100 000-grid fleet step take."""
        if not self.fused:
            res = self._each(lambda env, k: env.step(actions[k], **kw))
            return tuple(list(x) for x in zip(*res))
        actions = [env.control_to_tensor(a).to(env.engine.action_dtype) if isinstance(a, dict) else a
                   for env, a in zip(self.envs, actions)]
        return self._step_fused(actions, **kw)

    def _plan(self, key):
        """Everything about a fleet step that depends only on where the envs stand in their observation rings (and, with
        ``reuse_outputs``, on the output slot): the filled-in ``mgx_fleet_item`` array (all but the action pointers), the
        observation views the step returns and the ring state after it.  A fleet that walks its rings in lock-step meets
        3 K such situations, so after the first lap a step costs a dictionary look-up instead of ~25 us of bookkeeping."""
        import ctypes as C
        states, slot = key
        n = len(self.envs)
        items = (_lib.FleetItem * n)()
        obs_l, next_states, fixed_out = [], [], []
        for k, (it, env, st) in enumerate(zip(items, self.envs, states)):
            e = env.engine
            it.struct_size = C.sizeof(_lib.FleetItem)
            it.handle = e._h.value
            if isinstance(env, DiscreteBatchedMicrogridEnv):
                it.table, it.n_actions = e._table_ptr(env._table)
            env._set_plan_state(st)
            want_obs, target, wait = env._obs_plan()
            chunk = env._chunk_plan()             # chunked envs: this step's share of the next ring
            refill = env._obs_commit()            # other envs: (ring, ahead) when this step moves on to the prefetched ring
            next_states.append(env._plan_state())
            obs = target if want_obs else None
            if obs is None and want_obs and slot is not None and self._out_obs[k] is not None:
                obs = self._out_obs[k][slot]          # no rings (whole rows written per step): rotating row buffers
            it.obs = None if obs is None else obs.data_ptr()
            it.wait_prefetch = int(wait)
            if chunk:
                it.refill_ring, it.refill_K = chunk[0].data_ptr(), env.obs_prefetch
                it.refill_ahead, it.refill_chunk, it.refill_chunks = chunk[1], chunk[2], chunk[3]
            else:
                it.refill_ring = refill[0].data_ptr() if refill else None
                it.refill_K, it.refill_ahead = (env.obs_prefetch, refill[1]) if refill else (0, 0)
            obs_l.append((obs, want_obs and obs is None))          # (ring view, needs a fresh row buffer per step)
            env._set_plan_state(st)           # planning must not move the env: the step applies next_states once
                                              # mgx_fleet_step has succeeded
            if slot is not None:                                     # rotating output buffers: pointers are part of the plan
                r, d = self._out_reward[k][slot], self._out_done[k][slot]
                it.reward, it.done = r.data_ptr(), d.data_ptr()
                fixed_out.append((r, d.view(torch.bool)))
        # fast path: everything a step returns is known in advance (rotating output buffers, observations out of the rings or
        # none at all, no log rows): the step only has to hand in the action pointers
        fast = None
        if slot is not None and not any(fresh for _, fresh in obs_l) \
                and not any(env._keep_log or env._obs_index is not None for env in self.envs):
            sync = [(env, st) for env, st in zip(self.envs, next_states) if st is not None]
            fast = ([o for o, _ in obs_l], [r for r, _ in fixed_out], [d for _, d in fixed_out], sync,
                    (tuple(next_states), (slot + 1) % self.reuse_outputs))
        return items, obs_l, next_states, fixed_out, fast

    def _step_fused(self, actions, normalized=True):
        envs = self.envs
        R = self.reuse_outputs
        if R and self._out_reward is None:
            self._out_reward = [e.engine._empty(R, e.engine.N) for e in envs]
            self._out_done = [e.engine._empty(R, e.engine.N, dtype=torch.uint8) for e in envs]
            # envs that write whole observation rows per step (no rings, no views): R rotating [N, D] buffers per bucket
            self._out_obs = [torch.empty(R, e.engine.N, e.engine.obs_dim, dtype=e.engine.obs_dtype, device=e.engine.device)
                             if (e._observations and e._ring is None and not e._views) else None for e in envs]
        b = self._bound
        if b is None and R and self._bind_ok:
            b = self._try_bind()
        if b is not None:
            return self._step_bound(b, actions, normalized)
        slot = (self._n_steps % R) if R else None
        key = (tuple(e._plan_state() for e in envs), slot)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._plan(key)
        items, obs_plan, next_states, fixed_out, fast = plan
        if fast is not None:
            return self._step_fast(items, fast, actions, normalized)
        obs_l, reward_l, done_l, info_l = [], [], [], []
        for k, (it, env, a) in enumerate(zip(items, envs, actions)):
            e = env.engine
            if it.table:                          # discrete bucket: priority-list ids
                if not (torch.is_tensor(a) and a.dtype == torch.int32 and a.is_contiguous() and a.device == e.device
                        and a.shape == (e.N,)):
                    a = torch.as_tensor(np.asarray(a.cpu() if torch.is_tensor(a) else a), device=e.device).to(torch.int32).contiguous()
                it.action_id = a.data_ptr()
            else:
                a = e._check_actions(a, ())
                it.actions = None if a is None else a.data_ptr()
            obs, fresh = obs_plan[k]
            if fresh:
                obs = e._obs_buf(None)
                it.obs = obs.data_ptr()
            dconst = env._lockstep_done()        # lock-step: `done` is a constant tensor, the kernel writes no flags
            if R:
                reward, done = fixed_out[k]
                it.done = None if dconst is not None else done.data_ptr()
            else:
                reward = e._empty(e.N)
                it.reward = reward.data_ptr()
                if dconst is None:
                    d8 = e._empty(e.N, dtype=torch.uint8)
                    it.done = d8.data_ptr()
                    done = d8.view(torch.bool)
                else:
                    it.done = None
            if dconst is not None:
                done = dconst
            if env._keep_log:
                log = e._empty(e.log_dim, e.N)
                it.log = log.data_ptr()
                info_l.append({"log": log})
            else:
                info_l.append({})
            obs_l.append(obs); reward_l.append(reward); done_l.append(done)
        e0 = envs[0].engine
        idx = e0._dev_index
        if e0._only_device or torch.cuda.current_device() == idx:
            rc = e0._lib.mgx_fleet_step(items, len(envs), 1 if normalized else 0, _raw_stream(idx))
        else:
            with torch.cuda.device(idx):
                rc = e0._lib.mgx_fleet_step(items, len(envs), 1 if normalized else 0, _raw_stream(idx))
        if rc:
            _lib.check(rc)
        self._n_steps += 1
        for k, env in enumerate(envs):
            if env.engine._t is not None:
                env.engine._t += 1                # the host mirror of the handle's counter (mgx_fleet_step moved it)
            env._set_plan_state(next_states[k])
            if env._views:
                obs_l[k] = env._view_now()
            elif env._obs_index is not None:
                obs_l[k] = env._select_obs(obs_l[k])
            if env._keep_log:
                env._log_rows.append(info_l[k]["log"])
                env._shaped_rows.append(reward_l[k].clone())
        return obs_l, reward_l, done_l, info_l

    def _step_fast(self, items, fast, actions, normalized):
        """The per-step path of a fleet whose returns are all known in advance (``_plan``): hand in the action pointers, make ONE
        C call.  Kept lean -- at 100 000 grids a views-contract fleet step is 8.7 us of GPU time, so every microsecond of Python
        here is a microsecond per env-step (profiles/r04/exp_views_host_profile.txt)."""
        obs_plan, reward_l, done_plan, sync, _ = fast
        envs = self.envs
        n = len(envs)
        obs_l, done_l = list(obs_plan), list(done_plan)
        keep = None                               # converted ids stay alive until the launch has been issued
        for k in range(n):
            env, a, it = envs[k], actions[k], items[k]
            e = env.engine
            # lock-step: `done` is one of two constant tensors, the kernel writes no flags (BatchedMicrogridEnv._lockstep_done)
            if e._window_start is None and not e._dev_counter:
                it.done = None
                done_l[k] = env._done_const[e._t >= e.window[1] - 1]
            else:
                it.done = done_l[k].data_ptr()
            if it.table:                          # discrete bucket: priority-list ids
                if not (torch.is_tensor(a) and a.dtype == torch.int32 and a.is_contiguous() and a.is_cuda and a.shape == (env.n_grids,)):
                    a = torch.as_tensor(np.asarray(a.cpu() if torch.is_tensor(a) else a), device=env.batch.device).to(torch.int32).contiguous()
                    keep = (keep or []) + [a]
                it.action_id = a.data_ptr()
            else:
                if not (a.dtype == e.action_dtype and a.shape == e._action_shape and a.is_contiguous() and a.is_cuda):
                    a = e._check_actions(a, ())   # raises with the full message
                it.actions = a.data_ptr()
        e0 = envs[0].engine
        idx = e0._dev_index
        if e0._only_device or torch.cuda.current_device() == idx:
            rc = e0._lib.mgx_fleet_step(items, n, 1 if normalized else 0, _raw_stream(idx))
        else:
            with torch.cuda.device(idx):
                rc = e0._lib.mgx_fleet_step(items, n, 1 if normalized else 0, _raw_stream(idx))
        if rc:
            _lib.check(rc)
        self._n_steps += 1
        for env, st in sync:                      # (the ring / state-buffer position after the step: BatchedMicrogridEnv._set_plan_state)
            if st[0] == "v":
                env._state_pos = st[1]
            else:
                env._ring_idx, env._ring_pos = st
                env._ring = env._rings[st[0]]
        for k in range(n):
            env = envs[k]
            e = env.engine
            if e._t is not None:
                e._t += 1                         # the host mirror of the handle's counter (mgx_fleet_step moved it)
            if env._views:
                obs_l[k] = env._view_now()
        return obs_l, list(reward_l), done_l, [{} for _ in range(n)]      # (fresh dicts: Gym wrappers write into info)

    # ---- the bound step: one C call per fleet step, the handles walk slots and rings -------------------------------------------------
    def _bind_eligible(self):
        R = self.reuse_outputs
        if not (self.fused and R and R <= _lib.ENV_MAX_SLOTS and self.refill == "ahead" and len(self.envs) <= 64):
            return False
        for env in self.envs:
            e = env.engine
            if env._keep_log or env._obs_index is not None or env.raise_errors or env._chunked or env._sync_rings or env._ring_phase \
                    or e._t is None or e._dev_counter or e.n_shards != 1 or e._window_start is not None \
                    or getattr(env, "check_asserts", False) or (env._views and R % env.VIEW_BUFFERS):
                return False
        return True

    def _try_bind(self):
        """Bind every bucket's handle at the fleet's CURRENT position (mgx_env_bind + mgx_env_seek).  Returns the _BoundFleet or None
        (then the per-step plans of _step_fused stay in charge)."""
        import ctypes as C
        if not self._bind_eligible():
            self._bind_ok = False                         # (until something moves an env: _invalidate_bound)
            return None
        R, envs, n = self.reuse_outputs, self.envs, len(self.envs)
        slot0 = self._n_steps % R
        b = _BoundFleet()
        keep, obs_l, nring, p0 = [], [], [], []
        done_ok = []
        for k, env in enumerate(envs):
            e = env.engine
            K = env.obs_prefetch if env._ring is not None else 0
            slots = (_lib.EnvSlot * R)()
            for j in range(R):
                slots[j].reward = self._out_reward[k][j].data_ptr()
                slots[j].done = None                      # lock-step: `done` is one of two constant tensors
                if env._views:                            # the state buffer the step of slot j writes (R is a multiple of VIEW_BUFFERS)
                    slots[j].obs = env._state_bufs[(env._state_pos + 1 + ((j - slot0) % R)) % env.VIEW_BUFFERS].data_ptr()
                elif self._out_obs[k] is not None:
                    slots[j].obs = self._out_obs[k][j].data_ptr()
                else:
                    slots[j].obs = None
            plan = _lib.EnvPlan()
            plan.struct_size = C.sizeof(_lib.EnvPlan)
            plan.n_slots, plan.slots, plan.ring_K = R, slots, K
            if K:
                for r in range(3):
                    plan.rings[r] = env._rings[r].data_ptr()
            table = getattr(env, "_table", None)
            if table is not None:
                plan.table, plan.n_actions = e._table_ptr(table)
            keep.append((slots, plan))
            if e._lib.mgx_env_bind(e._h, C.byref(plan)) or \
                    e._lib.mgx_env_seek(e._h, slot0, env._ring_idx if K else 0, env._ring_pos if K else 0):
                for env2 in envs[:k + 1]:
                    env2.engine._lib.mgx_env_bind(env2.engine._h, None)
                self._bind_ok = False                     # (a mode the handles walk no rings in: do not try again every step)
                return None
            done_ok.append(True)
            nring.append(3 * K)
            p0.append((env._ring_idx * K + env._ring_pos) if K else 0)
            if K:
                obs_l.append([env._rings[r][kk] for r in range(3) for kk in range(K)])
            elif env._views:
                obs_l.append(None)
            elif self._out_obs[k] is not None:
                obs_l.append([self._out_obs[k][j] for j in range(R)])
            else:
                obs_l.append([None] * R)
        e0 = envs[0].engine
        b.fn, b.n, b.dev, b.guard = e0._lib.mgx_fleet_env_step, n, e0._dev_index, not e0._only_device
        b.hs = (C.c_void_p * n)(*[env.engine._h.value for env in envs])
        b.ptrs = (C.c_void_p * n)()
        b.adt = [torch.int32 if getattr(env, "_table", None) is not None else env.engine.action_dtype for env in envs]
        b.ashape = [torch.Size((env.n_grids,)) if getattr(env, "_table", None) is not None else torch.Size(env.engine._action_shape)
                    for env in envs]
        b.R, b.slot, b.p, b.nring, b.obs = R, slot0, p0, nring, obs_l
        b.VB = BatchedMicrogridEnv.VIEW_BUFFERS
        b.rew = [[self._out_reward[k][j] for k in range(n)] for j in range(R)]
        b.keep = keep
        self._bound = b
        return b

    def _invalidate_bound(self):
        """Something is about to move an env (reset, per-grid episodes, a new ring depth ...): positions back to the envs, and the next
        fleet step looks again whether it can bind."""
        self._unbind_bound()
        self._bind_ok = True

    def _unbind_bound(self):
        """Hand the positions back to the envs' Python state (called before anything else moves an env: BatchedMicrogridEnv._unbind_fast)."""
        b, self._bound = self._bound, None
        if b is None:
            return
        for k, env in enumerate(self.envs):
            if b.nring[k]:
                K = b.nring[k] // 3
                env._ring_idx, env._ring_pos = divmod(b.p[k], K)
                env._ring = env._rings[env._ring_idx]
            if env.engine._h.value:
                env.engine._lib.mgx_env_bind(env.engine._h, None)

    def _step_bound(self, b, actions, normalized):
        """One bound fleet step: the controls' addresses in, the pre-built views of the slots / ring blocks the handles used out."""
        envs, n = self.envs, b.n
        ptrs, adt, ashape = b.ptrs, b.adt, b.ashape
        keep = None
        for k in range(n):
            a = actions[k]
            if not (torch.is_tensor(a) and a.dtype == adt[k] and a.shape == ashape[k] and a.is_contiguous() and a.is_cuda):
                if adt[k] == torch.int32 and getattr(envs[k], "_table", None) is not None:
                    a = torch.as_tensor(np.asarray(a.cpu() if torch.is_tensor(a) else a), device=envs[k].batch.device).to(torch.int32).contiguous()
                else:
                    a = envs[k].engine._check_actions(a, ())      # converts, or raises with the full message
                keep = (keep or []) + [a]
            ptrs[k] = a.data_ptr()
        if b.guard and torch.cuda.current_device() != b.dev:
            with torch.cuda.device(b.dev):
                rc = b.fn(b.hs, ptrs, n, 1 if normalized else 0, _raw_stream(b.dev))
        else:
            rc = b.fn(b.hs, ptrs, n, 1 if normalized else 0, _raw_stream(b.dev))
        if rc:
            if rc == _lib.MGX_ERR_DEVICE:             # the launch sequence broke off somewhere: the handles' positions are not ours any more
                self._bound = None
            _lib.check(rc)                            # (range / argument errors are raised before anything moves: still bound)
        self._n_steps += 1
        slot = b.slot
        b.slot = slot + 1 if slot + 1 < b.R else 0
        obs_l, done_l = [None] * n, [None] * n
        p, nring, obs = b.p, b.nring, b.obs
        for k in range(n):
            env = envs[k]
            e = env.engine
            t = e._t
            done_l[k] = env._done_const[t >= e.window[1] - 1]
            e._t = t + 1
            if nring[k]:
                q = p[k] + 1
                if q == nring[k]:
                    q = 0
                p[k] = q
                obs_l[k] = obs[k][q]
            elif env._views:
                env._state_pos = (env._state_pos + 1) % b.VB
                obs_l[k] = env._view_now()
            else:
                obs_l[k] = obs[k][slot]
        return obs_l, list(b.rew[slot]), done_l, [{} for _ in range(n)]      # (fresh dicts: Gym wrappers write into info)

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
        self._unbind_bound()
        for env in self.envs:
            env._fleet_ref = None
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

    ``auto_reset=True``: every grid restarts ON ITS OWN the moment its episode is over (rolling windows,
    ``mgx_reset_windows_rolling`` / ``mgx_reset_grids``) -- N reference microgrids that are each reset when they report
    ``done``, as a vectorised Gym env does.  ``step`` then returns the first observation of the new episode for the grids that
    just finished and the batch never needs a global ``reset()`` again.  ``final_observation=True`` also returns the last rows
    of the finished episodes in ``info["final_observation"]`` ([N, D]; only the rows of the grids with ``done`` set are
    meaningful -- what a vectorised Gym env reports -- the others may hold older rows).  With a forecast horizon the
    observation rings stay in use: they are refilled on the caller's stream and the restarted grids' rows are patched into them
    (``mgx_patch_windows``).
    """

    FINAL_BUFFERS = 4

    def __init__(self, full_batch, trajectory_length=None, discrete=False, generator=None, auto_reset=False,
                 final_observation=False, seed=0, native=None, **env_kwargs):
        L = full_batch.layout
        # Several modules of a kind per grid (round 6): equal-length windows are gathered per reset (mgx_reset_windows on the general
        # path); auto_reset runs IN PLACE only (mgx_reset_episodes: the grid reads its own rows of the [T, n, N] series, the step kernel
        # restarts it) with device draws and rows per step -- the rolling window buffers and the ring patches are single-instance
        if L.multi:
            if auto_reset and (generator is not None or final_observation or native is False or discrete):
                raise NotImplementedError("several modules of a kind per grid: auto_reset runs in place with device draws "
                                          "(generator=None, native, continuous controls, no final_observation)")
            if auto_reset:
                env_kwargs = dict(env_kwargs, obs_prefetch=0)
                native = True
        self.full = full_batch
        self.length = None if trajectory_length is None else int(trajectory_length)
        if self.length is not None and L.final_step - L.initial_step < self.length:
            raise ValueError(f"the batch's window [{L.initial_step}, {L.final_step}) holds {L.final_step - L.initial_step} steps: "
                             f"too short for episodes of {self.length}")
        self.generator = generator
        cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
        self.auto_reset = bool(auto_reset)
        self.final_observation = bool(final_observation)
        # auto_reset without a torch generator: the restarts' episodes are drawn inside the restart kernel (Philox of
        # (seed; grid, counter)): one launch per step instead of a dozen small torch kernels
        self.seed = int(seed)
        self._device_draws = self.auto_reset and generator is None
        # native: in-place episodes (mgx_reset_episodes).  Nothing is gathered at a (re)start, and with device
        # draws the step kernel restarts finished grids ITSELF (mgx_set_auto_reset): an auto-reset step is ONE launch, as in
        # lock-step (9.8 instead of 18.4 us per 100 000-grid step without a forecast horizon); with observation rings one more
        # (the restarted grids' window columns are patched into the ring: 22 / 43 instead of 27 / 50 us at D = 56 / 156).
        # [T, N] series run in place too: the handle keeps a grid-major copy [N, T, 2 or 6] of them (as much memory again) in which a
        # grid's own row is one 16- or 48-byte read and consecutive rows share lines -- 9.6 / 21.5 us per 100 000-grid auto-reset
        # step without / with a GridModule against 20.7 / 37.3 us on rolling window buffers, 25 / 48 against 33 / 72 us with 24-hour
        # observation rows (profiles/r04/exp_auto_reset_materialised_grid_major.txt)
        if native is None:
            native = self.auto_reset and not env_kwargs.get("obs_views") and not full_batch.layout.multi
        if native and not self.auto_reset:
            raise ValueError("native=True is the auto_reset=True path (equal-length windows are gathered once per reset)")
        self.native = bool(native)
        env_kwargs.setdefault("obs_layout", "rows")       # restarted grids are patched into row-major rings (mgx_patch_windows)
        self.env = cls(full_batch, **env_kwargs)
        self.starts = self.lengths = None
        self._final_bufs = None
        self._final_pos = 0

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
        validate = starts is not None          # the env's own draws lie inside the window by construction: no host check
        if starts is None:
            starts, lengths = self.draw()
        dev = self.full.device
        self.starts = torch.as_tensor(np.asarray(starts.cpu() if torch.is_tensor(starts) else starts), device=dev).to(torch.int32)
        self.lengths = None if lengths is None else \
            torch.as_tensor(np.asarray(lengths.cpu() if torch.is_tensor(lengths) else lengths), device=dev).to(torch.int32)
        max_len = self.length if self.lengths is None else int(self.lengths.max().item())
        if max_len is None:
            raise ValueError("lengths are required when the env was built without a trajectory_length")
        if self.auto_reset:      # the rings must hold the longest episode any LATER restart can draw
            L = self.full.layout
            max_len = self.length if self.length is not None else L.final_step - L.initial_step
            if self.native:
                obs = self.env.reset_windows(self.starts, self.lengths, max_len, rolling="inplace", validate=validate)
                if self._device_draws:
                    e = self.env.engine
                    if self.lengths is None:
                        self.lengths = torch.full_like(self.starts, self.length if self.length is not None else 0)
                    e.set_auto_reset(True, self.seed, self.length or 0, lengths_out=self.lengths)
                    self.starts = e._window_start                   # updated in place by the step kernels
                return obs
            return self.env.reset_windows(self.starts, self.lengths, max_len, rolling=True, validate=validate)
        return self.env.reset_windows(self.starts, self.lengths, max_len, validate=validate)

    def step(self, action, **kw):
        if not self.auto_reset:
            return self.env.step(action, **kw)
        env = self.env
        if self.native and self._device_draws and env._ring is not None:
            # rings: the step kernel restarts the grids it finishes and adds the state columns to the ring's row; the window columns
            # of the restarted grids in the rest of the ring are then patched (mgx_patch_windows): two launches
            final = None
            if self.final_observation:       # the patch saves the rows it is about to replace (the finished grids' only)
                final = self._next_final_buf()
                env.engine.set_final_obs(final)
            obs, reward, done, info = env.step(action, **kw)
            obs = env._rows_after_restart(done.view(torch.uint8), True)
            if final is not None:
                info = dict(info, final_observation=env._select_obs(final))
            return obs, reward, done, info
        if self.native and self._device_draws:            # one launch: the step kernel restarts the grids it finishes
            final = None
            if self.final_observation and env._observations:
                final = self._next_final_buf()
                env.engine.set_final_obs(final)
            obs, reward, done, info = env.step(action, **kw)
            if final is not None:
                info = dict(info, final_observation=env._select_obs(final))
            return obs, reward, done, info
        want_rows = env._observations
        if want_rows and not self.final_observation and env._ring is None:   # the rows come from the observe pass behind the restarts
            env._observations = False
        try:
            obs, reward, done, info = env.step(action, **kw)
        finally:
            env._observations = want_rows
        final = obs.clone() if (self.final_observation and obs is not None) else None     # (a ring view: patched below)
        if self._device_draws:
            if self.lengths is None:
                self.lengths = torch.full_like(self.starts, self.length if self.length is not None else 0)
            new_obs = env.reset_grids_random(done, self.seed, self.length or 0, lengths_out=self.lengths)
            self.starts = env.engine._window_start          # updated in place by the kernel
        else:
            starts, lengths = self.draw()                  # a draw per grid; only the finished grids take theirs
            new_obs = env.reset_grids(done, starts, lengths, validate=False)
            self.starts = torch.where(done, starts, self.starts)
            if lengths is not None:
                self.lengths = torch.where(done, lengths, self.lengths)
        if self.final_observation:
            info = dict(info, final_observation=final)
        return (new_obs if new_obs is not None else obs), reward, done, info

    def _next_final_buf(self):
        """One of FINAL_BUFFERS rotating [N, D] buffers for ``info["final_observation"]`` (valid for FINAL_BUFFERS - 1 further steps)."""
        if self._final_bufs is None:
            e = self.env.engine
            self._final_bufs = torch.zeros(self.FINAL_BUFFERS, e.N, e.obs_dim, dtype=e.obs_dtype, device=e.device)
        self._final_pos = (self._final_pos + 1) % self.FINAL_BUFFERS
        return self._final_bufs[self._final_pos]

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
