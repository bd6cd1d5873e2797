"""The bound Gym step (mgx_env_bind / mgx_env_step / mgx_env_step_discrete, ABI v8): the handle itself picks the rotating output
buffers and walks the observation rings, `env.step` is one C call.  Pinned three ways: against the env with per-call bookkeeping
(reuse_outputs=0: the path every other test pins against the oracle and the reference-made fixtures) value for value through ring
changes, resets in the middle of a ring and the end of the series; against the CPU oracle directly; and the handle's own position
report against the Python mirror.  Reference loop being replaced: `while not done: env.step(a)` (README.md:109-111,
envs/discrete/discrete.py:109-143, envs/base/base.py:169-209)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _position(env):
    e = env.engine
    s, r, p = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
    assert e._lib.mgx_env_position(e._h, C.byref(s), C.byref(r), C.byref(p)) == 0
    return s.value, r.value, p.value


@pytest.mark.parametrize("arch,H,K,dtype,discrete,layout", [
    ("genset+battery", 0, 0, torch.float64, False, "rows"), ("genset+battery+grid", 0, 0, torch.float32, True, "rows"),
    ("genset+battery+grid", 24, 8, torch.float64, False, "rows"), ("genset+battery", 24, 5, torch.float32, False, "columns"),
    ("battery+grid", 7, 3, torch.float64, True, "columns"), ("genset+battery+grid", 23, 16, torch.float32, True, "rows")])
def test_bound_step_equals_the_per_call_step(arch, H, K, dtype, discrete, layout, device):
    from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T = 1003, 130
    cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
    kw = dict(remove_redundant_gensets=False) if discrete else {}
    R = 3 * K if K else 4

    def make(reuse):
        b = generate(N, n_steps=T, seed=5, arch=arch, horizon=H, device=device, mixed_timers=True, series="factorised")
        return cls(b, obs_dtype=dtype, obs_prefetch=K, obs_layout=layout, reuse_outputs=reuse, **kw)
    slow, fast = make(0), make(R)
    assert slow._fp is None and fast._fp is not None
    g = torch.Generator(device=device); g.manual_seed(2)
    held = []                                   # outputs stay valid for R - 1 (ring blocks: K) further steps
    for start, n_steps in ((0, 3 * max(K, 2) + 2), (11, max(K, 2) + 1), (T - H - 5, H + 4)):
        o1, o2 = slow.reset(start), fast.reset(start)
        assert fast._fp is not None and torch.equal(o1, o2), (start, "reset")
        held = []                               # (a reset refills the rings: what was handed out before is gone)
        for k in range(min(n_steps, T - start - 1)):
            a = slow.sample_action(generator=g)
            (o1, r1, d1, i1), (o2, r2, d2, i2) = slow.step(a), fast.step(a)
            assert o2.shape == o1.shape and torch.equal(o1, o2), (start, k)
            assert torch.equal(r1, r2) and torch.equal(d1, d2) and i2 == {}
            held.append((o1.clone(), r1.clone(), o2, r2))
            for c1, c2, v1, v2 in held[-(K if K else R - 1):]:        # (a ring block stays intact for K further steps)
                assert torch.equal(c1, v1) and torch.equal(c2, v2)
            fp = fast._fp
            s, ring, pos = _position(fast)
            assert s == (fp.k - 1) % R and (not K or ring * K + pos == fp.p)
            assert fast.current_step == slow.current_step
    # past the end of the series: the same refusal as the per-call path
    slow.reset(T - 2); fast.reset(T - 2)
    a = slow.sample_action(generator=g)
    slow.step(a); fast.step(a); slow.step(a); fast.step(a)
    with pytest.raises(Exception):
        fast.step(a)
    slow.close(); fast.close()


def test_bound_step_vs_the_oracle(device, oracle):
    """400 Gym steps of a Template-4 batch through env.step (bound) == the CPU oracle's rewards and final state."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T, K = 4099, 420, 400
    b = generate(N, n_steps=T, seed=17, arch="genset+battery", device=device, mixed_timers=True, series="factorised")
    cols = b.numpy_columns()
    st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
    env = BatchedMicrogridEnv(b, reuse_outputs=4, observations=False)
    assert env._fp is not None
    g = torch.Generator(device=device); g.manual_seed(4)
    acts = torch.rand(K, N, 3, dtype=torch.float64, device=device, generator=g)
    rew = torch.empty(K, N, dtype=torch.float64, device=device)
    env.reset()
    for k in range(K):
        obs, r, d, info = env.step(acts[k])
        assert obs is None and not bool(d[0])
        rew[k] = r
    ref = oracle.run_batch(cols, st, 0, K, acts.cpu().numpy(), normalized=True, nthreads=8)
    assert np.array_equal(rew.cpu().numpy(), ref)
    assert np.array_equal(b.cols["charge"].cpu().numpy(), st["charge"])
    assert np.array_equal(b.cols["gen_status"].cpu().numpy().view(np.uint32), st["gen_status"].view(np.uint32))
    env.close()


@pytest.mark.parametrize("H,discrete", [(0, False), (0, True)])
def test_bound_step_with_auto_reset_episodes(H, discrete, device):
    """PerGridWindowEnv(auto_reset=True) on in-place episodes: per-grid `done` out of the rotating slots, restarts by the step kernel."""
    from pymgrid_amd.generator import generate
    from pymgrid_amd.hetero import PerGridWindowEnv
    N, T = 2051, 300

    def make(reuse):
        b = generate(N, n_steps=T, seed=3, arch="genset+battery+grid", horizon=H, device=device, mixed_timers=True, series="factorised")
        kw = dict(remove_redundant_gensets=False) if discrete else {}
        return PerGridWindowEnv(b, trajectory_length=17, discrete=discrete, auto_reset=True, seed=9, reuse_outputs=reuse, obs_prefetch=0, **kw)
    slow, fast = make(0), make(4)
    g = torch.Generator(device=device); g.manual_seed(7)
    starts, _ = slow.draw()
    o1, o2 = slow.reset(starts.clone()), fast.reset(starts.clone())
    assert slow.env._fp is None and fast.env._fp is not None and torch.equal(o1, o2)
    n_done = 0
    for k in range(60):
        a = slow.env.sample_action(generator=g)
        (o1, r1, d1, _), (o2, r2, d2, _) = slow.step(a), fast.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), k
        n_done += int(d2.sum())
        assert torch.equal(slow.env.current_steps, fast.env.current_steps)
    assert n_done >= 3 * N
    slow.env.close(); fast.env.close()


def test_bind_refuses_what_it_cannot_walk(device):
    from pymgrid_amd import BatchedMicrogridEnv, _lib
    from pymgrid_amd.generator import generate
    b = generate(300, n_steps=80, seed=1, arch="genset+battery", horizon=4, device=device, series="factorised")
    env = BatchedMicrogridEnv(b, obs_prefetch=4, reuse_outputs=12, log=True)
    assert env._fp is None                      # log rows are fresh tensors per step: per-call bookkeeping
    e = env.engine
    assert e._lib.mgx_env_step(e._h, None, 1, None) == _lib.MGX_ERR_INVALID and b"no plan" in e._lib.mgx_last_error()
    plan = _lib.EnvPlan()
    plan.struct_size = C.sizeof(_lib.EnvPlan) - 4
    assert e._lib.mgx_env_bind(e._h, C.byref(plan)) == _lib.MGX_ERR_INVALID
    env.close()
    # a fleet owns its envs' rings: they are never bound
    from pymgrid_amd.hetero import BucketedFleet
    fl = BucketedFleet.from_batches([generate(300, n_steps=80, seed=2, arch=a, horizon=4, device=device, series="factorised")
                                     for a in ("genset+battery", "battery+grid")], obs_prefetch=4, reuse_outputs=12)
    assert fl.fused and all(env._fp is None for env in fl.envs)
    fl.close()


@pytest.mark.parametrize("rows", [False, True])
def test_two_launch_chains_equal_the_one_stream_step(rows, device):
    """mgx_set_shards(2) + launch threads (include/mgx.h mgx_set_launch_threads): the Gym step as two dependent launch chains, shard 1
    issued by the library's resident host thread -- through env.step (bound, one hand-over per call: mode 2), through step_many
    (each thread its shard's K launches) and with the threads off: rewards, rows and the final state `==` the one-stream env."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T, K = 3001, 200, 48

    def make():
        b = generate(N, n_steps=T, seed=8, arch="genset+battery+grid", device=device, mixed_timers=True, series="factorised")
        return BatchedMicrogridEnv(b, observations=rows, reuse_outputs=4)
    g = torch.Generator(device=device); g.manual_seed(3)
    one = make()
    acts = torch.rand(3 * K, N, one.layout.action_dim, dtype=torch.float64, device=device, generator=g)
    one.reset()
    ref_rew, ref_obs = [], []
    for a in acts:
        o, r, d, _ = one.step(a)
        ref_rew.append(r.clone()); ref_obs.append(o.clone() if rows else None)
    for mode in (2, 0):
        two = make()
        two.reset()
        two.set_shards(2)
        two.engine.set_launch_threads(mode)
        assert two._fp is not None and two.engine.n_shards == 2
        two.fork()
        got_rew, got_obs = [], []
        for k in range(K):                                     # the bound step, one C call per env-step
            o, r, d, _ = two.step(acts[k])
            two.join()
            got_rew.append(r.clone()); got_obs.append(o.clone() if rows else None)
            two.fork()
        out = dict(reward=torch.empty(K, N, dtype=torch.float64, device=device))
        if rows:
            out["obs"] = torch.empty(K, N, two.layout.obs_dim, dtype=torch.float64, device=device)
        two._unbind_fast()
        two.engine.set_launch_threads(1 if mode else 0)
        o2, r2, _, _ = two.engine.step_many(acts[K:2 * K], normalized=True, out=out, done=False, want_obs=rows)   # K launches per shard and call
        two.join()
        got_rew += list(r2); got_obs += (list(o2) if rows else [None] * K)
        two.set_shards(1)                                      # (re-binds at the counter the sharded calls left)
        assert two.current_step == 2 * K
        for k in range(2 * K, 3 * K):
            o, r, d, _ = two.step(acts[k])
            got_rew.append(r.clone()); got_obs.append(o.clone() if rows else None)
        for k in range(3 * K):
            assert torch.equal(got_rew[k], ref_rew[k]), (mode, k)
            assert not rows or torch.equal(got_obs[k], ref_obs[k]), (mode, k)
        for name in ("charge", "soc", "gen_status"):
            assert torch.equal(two.batch.cols[name], one.batch.cols[name]), (mode, name)
        two.close()
    one.close()


def test_tunables_are_read_back_and_validated(device):
    """mgx_set_tunable / mgx_get_tunable: the library's launch-shape knobs (it reads no environment variables)."""
    from pymgrid_amd import MgxError, _lib
    for name in _lib.TUNABLES:
        cur, dflt = _lib.get_tunable(name)
        assert cur == dflt, name
    _lib.set_tunable("win_threads", 512)
    assert _lib.get_tunable("win_threads") == (512, 0)
    _lib.set_tunable("win_threads", 0)
    for name, bad in (("win_threads", 300), ("win_pairs", 2), ("launch_threads", 3), ("fleet_byvalue", -1)):
        with pytest.raises(MgxError):
            _lib.set_tunable(name, bad)
    assert _lib.lib().mgx_set_tunable(99, 0) == _lib.MGX_ERR_INVALID
    assert _lib.lib().mgx_abi_minor() == _lib.ABI_MINOR
