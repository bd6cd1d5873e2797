"""In-place per-grid episodes (mgx_reset_episodes / mgx_set_auto_reset / mgx_set_final_obs; ABI v6): factorised batches step per-grid
episodes on the series themselves -- no window buffers, a restart rewrites two words per grid, and with auto-reset the step kernel
restarts the grids it finishes.  Pinned three ways: against the CPU oracle on per-grid shifted series, against the rolling window
buffers (mgx_reset_windows_rolling, itself pinned against per-grid oracle microgrids in test_abi_v3.py) step by step through
restarts, and PerGridWindowEnv(native=True) against PerGridWindowEnv(native=False) with the same device draws."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SOAK = int(os.environ.get("MGX_FUZZ_SEED", "0"))          # soak runs: another draw of every batch / episode / action sequence

ARCHS = ("genset+battery", "battery+grid", "genset+battery+grid")


def _gen(n, T, arch, device, H=0, seed=9, series="factorised", **kw):
    from pymgrid_amd.generator import generate
    return generate(n, n_steps=T, seed=seed + 1000 * SOAK, arch=arch, device=device, horizon=H, mixed_timers=True,
                    series=series, **kw)


@pytest.mark.parametrize("arch", ARCHS)
def test_inplace_episodes_vs_the_oracle_on_shifted_series(arch, device, oracle):
    """Grid i steps L rows from its own start row: the same rewards and final state as the oracle stepping, from row 0, the series
    whose column i is the grid's series shifted by start_i."""
    from pymgrid_amd import StepEngine
    N, T, L = 2051, 400, 37
    b = _gen(N, T, arch, device, seed=31)
    cols = b.numpy_columns()
    st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
    rs = np.random.RandomState(3 + SOAK)
    starts = rs.randint(0, T - L + 1, size=N).astype(np.int32)
    starts[:4] = (0, T - L, 1, T - L - 1)
    rows = starts[None, :] + np.arange(L)[:, None]                                # [L, N]
    shifted = dict(cols)
    shifted["layout"] = dict(cols["layout"], T=L, final_step=L)
    shifted["load_ts"] = np.ascontiguousarray(np.take_along_axis(cols["load_ts"], rows, 0))
    shifted["pv_ts"] = np.ascontiguousarray(np.take_along_axis(cols["pv_ts"], rows, 0))
    if "grid_ts" in cols and cols["grid_ts"] is not None:
        shifted["grid_ts"] = np.ascontiguousarray(np.take_along_axis(cols["grid_ts"], rows[:, None, :].repeat(4, 1), 0))
    g = torch.Generator(device=device); g.manual_seed(5 + SOAK)
    acts = torch.rand(L, N, b.layout.action_dim, dtype=torch.float64, device=device, generator=g)
    e = StepEngine(b)
    e.reset_episodes(torch.from_numpy(starts).to(device), None, L, want_obs=False)
    rew = torch.empty(L, N, dtype=torch.float64, device=device)
    don = torch.empty(L, N, dtype=torch.uint8, device=device)
    for k in range(L):
        e.step(acts[k], want_obs=False, out=dict(reward=rew[k], done=don[k]))
    ref = oracle.run_batch(shifted, st, 0, L, acts.cpu().numpy(), normalized=True, nthreads=8)
    assert np.array_equal(rew.cpu().numpy(), ref)
    assert np.array_equal(b.cols["charge"].cpu().numpy(), st["charge"])
    d = don.cpu().numpy()
    assert not d[:-1].any() and d[-1].all()
    e.close()


@pytest.mark.parametrize("series", ["factorised", "materialised"])
@pytest.mark.parametrize("arch,H,discrete,prefetch", [("genset+battery", 0, False, 0), ("genset+battery+grid", 0, True, 0),
                                                     ("battery+grid", 5, False, 0), ("genset+battery+grid", 24, False, 0),
                                                     ("genset+battery", 3, True, 0), ("genset+battery+grid", 24, False, 4),
                                                     ("battery+grid", 5, True, 16), ("genset+battery", 7, False, 5)])
def test_inplace_episodes_equal_rolling_windows_through_restarts(arch, H, discrete, prefetch, series, device):
    """The same episodes on window rings (gathered rows) and in place (row offsets): observations, rewards, per-grid done flags
    and per-grid step counters agree at every step, through individual restarts with new starts and lengths, windows that reach
    the end of the series, and many more steps than the longest episode.  Factorised series (the rows formed from the base
    tables) and [T, N] arrays (every lane gathers its own row)."""
    from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv
    N, T, max_len = 1500, 300, 14
    cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
    kw = dict(remove_redundant_gensets=False) if discrete else {}
    ring = cls(_gen(N, T, arch, device, H, series=series), obs_prefetch=0, **kw)               # window buffers, per-step rows: the plain path
    inpl = cls(_gen(N, T, arch, device, H, series=series), obs_prefetch=prefetch, **kw)        # in place; prefetch > 0: rings + mgx_patch_windows
    rs = np.random.RandomState(8 + SOAK)
    lengths = rs.randint(1, max_len + 1, size=N).astype(np.int32)
    starts = np.array([rs.randint(0, T - n + 1) for n in lengths], dtype=np.int32)
    starts[:3] = T - lengths[:3]                                                # ... ending at the very end of the series
    o1 = ring.reset_windows(starts, lengths, max_length=max_len, rolling=True)
    o2 = inpl.reset_windows(starts, lengths, max_length=max_len, rolling="inplace")
    assert torch.equal(o1, o2) and inpl.obs_prefetch == prefetch and inpl._sync_rings == (prefetch > 0)
    g = torch.Generator(device=device); g.manual_seed(1 + SOAK)
    for k in range(3 * max_len + 5):
        a = (torch.randint(0, ring.action_space.n, (N,), dtype=torch.int32, device=device, generator=g) if discrete
             else torch.rand(N, ring.layout.action_dim, dtype=torch.float64, device=device, generator=g))
        o1, r1, d1, _ = ring.step(a)
        o2, r2, d2, _ = inpl.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), k
        assert torch.equal(ring.current_steps, inpl.current_steps), k
        if bool(d1.any()):
            new_len = rs.randint(1, max_len + 1, size=N).astype(np.int32)
            new_start = np.array([rs.randint(0, T - n + 1) for n in new_len], dtype=np.int32)
            if k % 2:
                new_start = (T - new_len).astype(np.int32)
            assert torch.equal(ring.reset_grids(d1, new_start, new_len), inpl.reset_grids(d2, new_start, new_len)), k
    from pymgrid_amd import MgxError
    with pytest.raises(MgxError):                           # single steps only
        inpl.engine.step_k(torch.rand(4, N, ring.layout.action_dim, dtype=torch.float64, device=device))
    assert torch.equal(ring.reset(), inpl.reset())          # a plain reset leaves the mode
    assert not inpl.engine._inplace
    a = (torch.zeros(N, dtype=torch.int32, device=device) if discrete
         else torch.rand(N, ring.layout.action_dim, dtype=torch.float64, device=device, generator=g))
    o1, r1, d1, _ = ring.step(a); o2, r2, d2, _ = inpl.step(a)
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    ring.close(); inpl.close()


@pytest.mark.parametrize("arch,H,discrete,length,final,prefetch",
                         [("genset+battery", 0, False, 9, False, 0), ("genset+battery", 0, False, 9, True, 0),
                          ("genset+battery+grid", 0, True, None, True, 0), ("battery+grid", 4, False, 7, True, 0),
                          ("genset+battery+grid", 24, False, 11, False, 0), ("genset+battery", 2, True, None, False, 0),
                          ("genset+battery+grid", 24, False, 11, True, 4), ("battery+grid", 6, True, None, True, 16),
                          ("genset+battery", 3, False, 5, False, 3)])
@pytest.mark.parametrize("series", ["factorised", "materialised"])
def test_native_auto_reset_equals_the_rolling_window_auto_reset(arch, H, discrete, length, final, prefetch, series, device):
    """PerGridWindowEnv(auto_reset=True) with device draws: native (one launch per step: the step kernel restarts the grids it
    finishes, mgx_set_auto_reset; the pre-restart rows through mgx_set_final_obs) against the rolling windows (step, restart
    gather, observation pass): observations, rewards, done flags, final observations, the drawn starts / lengths and the per-grid
    counters are identical step by step."""
    from pymgrid_amd.hetero import PerGridWindowEnv
    N, T = 1100, 200
    kw = dict(discrete=discrete, auto_reset=True, final_observation=final, seed=123, trajectory_length=length)
    if discrete:
        kw["remove_redundant_gensets"] = False
    roll = PerGridWindowEnv(_gen(N, T, arch, device, H, series=series), native=False, obs_prefetch=0, **kw)
    nat = PerGridWindowEnv(_gen(N, T, arch, device, H, series=series), native=True, obs_prefetch=prefetch,   # prefetch > 0: rings + patches
                           reuse_outputs=(3 if H in (0, 3) else 0), **kw)                         # (rotating reward / done / row buffers)
    assert nat.native and not roll.native
    lengths = None
    rs = np.random.RandomState(2 + SOAK)
    if length is None:
        lengths = rs.randint(1, 30, size=N).astype(np.int32)
        starts = np.array([rs.randint(0, T - n + 1) for n in lengths], dtype=np.int32)
    else:
        starts = rs.randint(0, T - length + 1, size=N).astype(np.int32)
    assert torch.equal(roll.reset(starts, lengths), nat.reset(starts, lengths))
    g = torch.Generator(device=device); g.manual_seed(1 + SOAK)
    n_done = 0
    for k in range(70):
        a = (torch.randint(0, roll.action_space.n, (N,), dtype=torch.int32, device=device, generator=g) if discrete
             else torch.rand(N, roll.layout.action_dim, dtype=torch.float64, device=device, generator=g))
        o1, r1, d1, i1 = roll.step(a)
        o2, r2, d2, i2 = nat.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2), k
        assert torch.equal(o1, o2), k
        if final:                                           # the rows of the grids that finished (the others are unspecified)
            assert torch.equal(i1["final_observation"][d1], i2["final_observation"][d2]), k
        assert torch.equal(roll.starts, nat.starts) and torch.equal(roll.lengths, nat.lengths), k
        assert torch.equal(roll.current_steps, nat.current_steps), k
        n_done += int(d1.sum())
    assert n_done > N                                       # every grid restarted at least once on average
    roll.close(); nat.close()


@pytest.fixture
def no_grid_major_copy():
    """mgx_set_tunable(MGX_TUNE_GRID_MAJOR_COPY, 0) for the test's duration (the tunables are process-wide)."""
    from pymgrid_amd import _lib
    _lib.set_tunable("grid_major_copy", 0)
    yield
    _lib.set_tunable("grid_major_copy", 1)


@pytest.mark.parametrize("arch,H,prefetch", [("genset+battery+grid", 0, 0), ("battery+grid", 5, 0), ("genset+battery+grid", 24, 4)])
def test_inplace_episodes_without_the_grid_major_copy(arch, H, prefetch, device, no_grid_major_copy):
    """[T, N] series when the handle cannot have its grid-major copy (allocation refused; here: the grid_major_copy tunable is 0): the lanes gather
    their rows out of the [T, N] arrays -- the same values as the window buffers, step by step through restarts."""
    from pymgrid_amd import BatchedMicrogridEnv
    N, T, max_len = 900, 200, 11
    ring = BatchedMicrogridEnv(_gen(N, T, arch, device, H, series="materialised"), obs_prefetch=0)
    inpl = BatchedMicrogridEnv(_gen(N, T, arch, device, H, series="materialised"), obs_prefetch=prefetch)
    rs = np.random.RandomState(4 + SOAK)
    lengths = rs.randint(1, max_len + 1, size=N).astype(np.int32)
    starts = np.array([rs.randint(0, T - n + 1) for n in lengths], dtype=np.int32)
    assert torch.equal(ring.reset_windows(starts, lengths, max_length=max_len, rolling=True),
                       inpl.reset_windows(starts, lengths, max_length=max_len, rolling="inplace"))
    g = torch.Generator(device=device); g.manual_seed(6 + SOAK)
    for k in range(2 * max_len + 3):
        a = torch.rand(N, ring.layout.action_dim, dtype=torch.float64, device=device, generator=g)
        o1, r1, d1, _ = ring.step(a)
        o2, r2, d2, _ = inpl.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), k
        if bool(d1.any()):
            new_len = rs.randint(1, max_len + 1, size=N).astype(np.int32)
            new_start = np.array([rs.randint(0, T - n + 1) for n in new_len], dtype=np.int32)
            assert torch.equal(ring.reset_grids(d1, new_start, new_len), inpl.reset_grids(d2, new_start, new_len)), k
    ring.close(); inpl.close()


def test_inplace_episodes_defaults_and_refusals(device):
    from pymgrid_amd import BatchedMicrogridEnv, MgxError, StepEngine
    from pymgrid_amd.hetero import PerGridWindowEnv
    N, T = 300, 100
    bm = _gen(N, T, "genset+battery", device, series="materialised")
    e = StepEngine(bm)
    with pytest.raises(MgxError):
        e.set_auto_reset(True)                               # not stepping in-place episodes yet
    e.reset_episodes(torch.zeros(N, dtype=torch.int32, device=device), None, 10)      # [T, N] series: offered since round 4
    e.set_auto_reset(True, 3, 10)
    e.close()
    # defaults: native (in place) for either series form; rings stay in use with a forecast horizon
    assert PerGridWindowEnv(_gen(N, T, "genset+battery", device), trajectory_length=5, auto_reset=True).native
    assert PerGridWindowEnv(_gen(N, T, "genset+battery", device, H=6), trajectory_length=5, auto_reset=True).native
    assert PerGridWindowEnv(bm, trajectory_length=5, auto_reset=True).native
    assert PerGridWindowEnv(_gen(N, T, "battery+grid", device, H=6, series="materialised"), trajectory_length=5, auto_reset=True).native
    env = BatchedMicrogridEnv(_gen(N, T, "genset+battery+grid", device, H=6), obs_prefetch=0)
    env.reset_windows(np.zeros(N, dtype=np.int32), None, max_length=10, rolling="inplace")
    with pytest.raises(MgxError):
        env.engine.set_final_obs(torch.empty(N, env.layout.obs_dim, dtype=torch.float64, device=device)) or env.engine.step(
            env.sample_action(), want_obs=False)            # final rows need a step that writes observations
    env.close()


@pytest.mark.parametrize("series", ["factorised", "materialised"])
def test_native_auto_reset_at_the_true_shape_of_configs2(series, device):
    """BASELINE configs[2] (100 000 generated Template-4 grids x 8 760 rows), every grid on its own random 168-step episodes for
    400 steps (every grid restarts at least twice): in-place episodes == rolling windows, rewards / done / observations / draws.
    [T, N] series: the in-place env reads the handle's 14 GB grid-major copy of them."""
    from pymgrid_amd.hetero import PerGridWindowEnv
    N, T = 100_000, 8760
    kw = dict(auto_reset=True, seed=5, trajectory_length=168)
    roll = PerGridWindowEnv(_gen(N, T, "genset+battery", device, seed=42, series=series), native=False, **kw)
    nat = PerGridWindowEnv(_gen(N, T, "genset+battery", device, seed=42, series=series), **kw)
    assert nat.native
    g = torch.Generator(device=device); g.manual_seed(3 + SOAK)
    st = torch.randint(0, T - 168, (N,), dtype=torch.int32, device=device, generator=g)
    st[:3] = torch.tensor([0, T - 168, T - 169], dtype=torch.int32)
    # staggered first episodes, so that restarts happen at every step
    ln = torch.randint(1, 169, (N,), dtype=torch.int32, device=device, generator=g)
    assert torch.equal(roll.reset(st, ln), nat.reset(st, ln))
    bad = torch.zeros((), dtype=torch.int64, device=device)
    n_done = torch.zeros((), dtype=torch.int64, device=device)
    for k in range(400):
        a = torch.rand(N, 3, dtype=torch.float64, device=device, generator=g)
        o1, r1, d1, _ = roll.step(a)
        o2, r2, d2, _ = nat.step(a)
        bad += (o1 != o2).sum() + (r1 != r2).sum() + (d1 != d2).sum()
        n_done += d1.sum()
    assert int(bad) == 0
    assert torch.equal(roll.starts, nat.starts) and torch.equal(roll.lengths, nat.lengths)
    assert torch.equal(roll.current_steps, nat.current_steps)
    assert int(n_done) > 2 * N
    roll.close(); nat.close()


def test_mode_changes_leave_nothing_behind(device):
    """One env walks through every episode mode in turn -- in-place episodes with automatic restarts, a plain reset, lock-step
    steps, gathered per-grid windows, in-place again with another maximum length and manual restarts, rolling windows -- next to
    an env that only ever uses window buffers; a mode must not leak offsets, restart switches or final-row pointers into the next."""
    from pymgrid_amd import BatchedMicrogridEnv
    N, T, H = 900, 160, 3
    a_env = BatchedMicrogridEnv(_gen(N, T, "genset+battery+grid", device, H), obs_prefetch=0)
    b_env = BatchedMicrogridEnv(_gen(N, T, "genset+battery+grid", device, H), obs_prefetch=0)
    rs = np.random.RandomState(4 + SOAK)
    g = torch.Generator(device=device); g.manual_seed(6 + SOAK)

    def steps(n, restart=None):
        for k in range(n):
            a = torch.rand(N, 4, dtype=torch.float64, device=device, generator=g)
            o1, r1, d1, _ = a_env.step(a)
            o2, r2, d2, _ = b_env.step(a)
            assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(o1, o2), k
            if restart is not None and bool(d1.any()):
                restart(d1)

    def draws(max_len):
        ln = rs.randint(1, max_len + 1, size=N).astype(np.int32)
        st = np.array([rs.randint(0, T - n + 1) for n in ln], dtype=np.int32)
        return st, ln

    def manual(max_len):
        def f(d):
            st, ln = draws(max_len)
            assert torch.equal(a_env.reset_grids(d, st, ln), b_env.reset_grids(d, st, ln))
        return f

    # 1. in place with automatic restarts (a) vs rolling windows restarted by mgx_reset_grids_random with the same seed (b)
    st, ln = draws(9)
    lens_a = torch.zeros(N, dtype=torch.int32, device=device)
    lens_b = torch.zeros(N, dtype=torch.int32, device=device)
    assert torch.equal(a_env.reset_windows(st, ln, max_length=9, rolling="inplace"), b_env.reset_windows(st, ln, max_length=9, rolling=True))
    a_env.engine.set_auto_reset(True, seed=11, fixed_length=9, lengths_out=lens_a)
    final = torch.empty(N, a_env.layout.obs_dim, dtype=torch.float64, device=device)
    a_env.engine.set_final_obs(final)
    for k in range(25):
        a = torch.rand(N, 4, dtype=torch.float64, device=device, generator=g)
        o1, r1, d1, _ = a_env.step(a)
        o2, r2, d2, _ = b_env.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(final, o2), k         # final = the row before the restart
        o2 = b_env.reset_grids_random(d2, 11, 9, lengths_out=lens_b)
        assert torch.equal(o1, o2), k
    assert torch.equal(a_env.current_steps, b_env.current_steps)
    # 2. a plain reset: lock-step again (no offsets, no restarts, no final rows)
    assert torch.equal(a_env.reset(5), b_env.reset(5))
    final.fill_(-1.0)
    steps(6)
    assert bool((final == -1.0).all()) and a_env.current_step == 11
    # 3. gathered per-grid windows (equal lengths), fused launch inside them
    st = rs.randint(0, T - 7 + 1, size=N).astype(np.int32)
    assert torch.equal(a_env.reset_windows(st, None, max_length=7), b_env.reset_windows(st, None, max_length=7))
    steps(3)
    acts = torch.rand(4, N, 4, dtype=torch.float64, device=device, generator=g)
    ra, rb = a_env.engine.step_k(acts, reward=True)["reward"], b_env.engine.step_k(acts, reward=True)["reward"]
    assert torch.equal(ra, rb)
    # 4. in place again, another maximum length, manual restarts, auto-reset OFF (finished grids keep reporting done)
    st, ln = draws(13)
    assert torch.equal(a_env.reset_windows(st, ln, max_length=13, rolling="inplace"), b_env.reset_windows(st, ln, max_length=13, rolling=True))
    steps(20, manual(13))
    # 5. auto-reset on, then off again in the same episode mode
    a_env.engine.set_auto_reset(True, seed=3, fixed_length=5, lengths_out=lens_a)
    for k in range(8):
        a = torch.rand(N, 4, dtype=torch.float64, device=device, generator=g)
        o1, r1, d1, _ = a_env.step(a)
        o2, r2, d2, _ = b_env.step(a)
        o2 = b_env.reset_grids_random(d2, 3, 5, lengths_out=lens_b)
        assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(o1, o2), k
    a_env.engine.set_auto_reset(False)
    steps(12, manual(13))
    # 6. rolling windows on the env that was in place, in place on the other: the roles swapped
    st, ln = draws(6)
    assert torch.equal(a_env.reset_windows(st, ln, max_length=6, rolling=True), b_env.reset_windows(st, ln, max_length=6, rolling="inplace"))
    steps(15, manual(6))
    assert torch.equal(a_env.reset(), b_env.reset())
    steps(4)
    a_env.close(); b_env.close()


@pytest.mark.parametrize("length,H,prefetch", [(1, 0, 0), (1, 45, 0), (None, 0, 0), (None, 45, 4), (40, 3, 2), (39, 0, 0)])
def test_native_auto_reset_at_the_edges(length, H, prefetch, device):
    """Episodes of one step (every grid restarts at every step), of the whole series (one possible start row), stochastic lengths
    over the whole window, forecast windows longer than the series: in place == rolling windows."""
    from pymgrid_amd.hetero import PerGridWindowEnv
    N, T = 130, 40
    kw = dict(auto_reset=True, final_observation=True, seed=8, trajectory_length=length)
    roll = PerGridWindowEnv(_gen(N, T, "genset+battery+grid", device, H), native=False, obs_prefetch=0, **kw)
    nat = PerGridWindowEnv(_gen(N, T, "genset+battery+grid", device, H), native=True, obs_prefetch=prefetch, **kw)
    rs = np.random.RandomState(1 + SOAK)
    if length is None:
        lengths = rs.randint(1, T + 1, size=N).astype(np.int32)
        starts = np.array([rs.randint(0, T - n + 1) for n in lengths], dtype=np.int32)
    else:
        lengths, starts = None, rs.randint(0, T - length + 1, size=N).astype(np.int32)
    assert torch.equal(roll.reset(starts, lengths), nat.reset(starts, lengths))
    g = torch.Generator(device=device); g.manual_seed(2 + SOAK)
    for k in range(3 * T):
        a = torch.rand(N, 4, dtype=torch.float64, device=device, generator=g)
        o1, r1, d1, i1 = roll.step(a)
        o2, r2, d2, i2 = nat.step(a)
        assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(o1, o2), k
        assert torch.equal(i1["final_observation"][d1], i2["final_observation"][d2]), k
        assert torch.equal(roll.starts, nat.starts) and torch.equal(roll.lengths, nat.lengths), k
        if length == 1:
            assert bool(d1.all())
    roll.close(); nat.close()


@pytest.mark.parametrize("arch,discrete", [("genset+battery", False), ("genset+battery+grid", False), ("battery+grid", True)])
def test_done_grids_stepped_past_the_series_stay_defined(arch, discrete, device, oracle):
    """A grid whose episode is over and that is never restarted keeps stepping on its own series; the shared counter has no end, so
    nothing refuses the step that leaves the series.  From row T on the kernels re-read the LAST row (clamped: no read beyond the
    base tables / outage words): rewards and state == the oracle stepping a series whose rows past T repeat row T - 1."""
    from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv
    N, T, L, extra = 1027, 70, 6, 150                       # 150 steps past episodes that end at the very end of the series
    cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
    kw = dict(remove_redundant_gensets=False) if discrete else {}
    b = _gen(N, T, arch, device, seed=77)
    cols = b.numpy_columns()
    st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
    env = cls(b, obs_prefetch=0, **kw)
    starts = np.full(N, T - L, dtype=np.int32)
    starts[::3] = T - L - 5
    env.reset_windows(starts, None, max_length=L, rolling="inplace")
    K = L + extra
    rows = np.minimum(starts[None, :] + np.arange(K)[:, None], T - 1)              # [K, N]: clamped at the last row
    walked = dict(cols)
    walked["layout"] = dict(cols["layout"], T=K, final_step=K)
    walked["load_ts"] = np.ascontiguousarray(np.take_along_axis(cols["load_ts"], rows, 0))
    walked["pv_ts"] = np.ascontiguousarray(np.take_along_axis(cols["pv_ts"], rows, 0))
    if cols.get("grid_ts") is not None:
        walked["grid_ts"] = np.ascontiguousarray(np.take_along_axis(cols["grid_ts"], rows[:, None, :].repeat(4, 1), 0))
    g = torch.Generator(device=device); g.manual_seed(2 + SOAK)
    rew = torch.empty(K, N, dtype=torch.float64, device=device)
    if discrete:
        ids = torch.randint(0, env.action_space.n, (K, N), dtype=torch.int32, device=device, generator=g)
        for k in range(K):
            obs, rew[k], done, _ = env.step(ids[k])
            assert torch.isfinite(obs).all()
        from pymgrid_amd.priority_list import table_array
        ref = oracle.rollout_batch(walked, st, 0, K, ids.cpu().numpy().astype(np.uint8), table_array(env.actions_list), nthreads=8)
    else:
        acts = torch.rand(K, N, b.layout.action_dim, dtype=torch.float64, device=device, generator=g)
        for k in range(K):
            obs, rew[k], done, _ = env.step(acts[k])
            assert torch.isfinite(obs).all()
        ref = oracle.run_batch(walked, st, 0, K, acts.cpu().numpy(), normalized=True, nthreads=8)
    assert bool(done.all())
    assert np.array_equal(rew.cpu().numpy(), ref)
    assert np.array_equal(b.cols["charge"].cpu().numpy(), st["charge"]) and np.isfinite(st["charge"]).all()
    env.close()


@pytest.mark.parametrize("counts,H", [((2, 2, 1, 1, 1), 0), ((2, 2, 1, 1, 1), 3), ((3, 1, 2, 2, 1), 2)])
def test_inplace_episodes_with_several_modules_of_a_kind(counts, H, device):
    """Round 6: per-grid episodes IN PLACE on the general path (several gensets / batteries / grids per microgrid; the register form
    and the run-time-count form): grid i reads its own rows of the [T, n, N] series -- against the gathered window buffers
    (mgx_reset_windows on the general path, pinned in tests/test_multi_windows.py) step by step: first episode, a restart of every
    grid with new starts (mgx_reset_grids), and PerGridWindowEnv(auto_reset=True) whose step kernel restarts the grids it finishes
    with device draws (the draws are read back and replayed on the gathered windows)."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import widen
    from pymgrid_amd.hetero import PerGridWindowEnv
    ng, nb, nr, nl, npv = counts
    N, T, Lg = 700, 160, 9

    def batch():
        return widen(_gen(N, T, "genset+battery+grid", device, H=H, seed=41, series="materialised"), n_genset=ng, n_battery=nb, n_grid=nr,
                     n_load=nl, n_pv=npv)
    inpl, gath = BatchedMicrogridEnv(batch(), obs_prefetch=0), BatchedMicrogridEnv(batch(), obs_prefetch=0)
    rs = np.random.RandomState(8 + SOAK)
    g = torch.Generator(device=device); g.manual_seed(2 + SOAK)
    A = inpl.layout.action_dim

    def episode(starts, first_a, first_b):
        assert torch.equal(first_a, first_b)
        for k in range(Lg):
            a = torch.rand(N, A, dtype=torch.float64, device=device, generator=g)
            (o1, r1, d1, _), (o2, r2, d2, _) = inpl.step(a), gath.step(a)
            assert torch.equal(r1, r2) and torch.equal(d1, d2), k
            assert torch.equal(o1, o2), k
            assert bool(d1.all()) == (k == Lg - 1) and bool(d1.any()) == (k == Lg - 1)
    starts = rs.randint(0, T - Lg + 1, size=N).astype(np.int32)
    starts[:3] = (0, T - Lg, 1)
    episode(starts, inpl.reset_windows(starts, None, max_length=Lg, rolling="inplace"), gath.reset_windows(starts, None, max_length=Lg))
    starts2 = rs.randint(0, T - Lg + 1, size=N).astype(np.int32)
    every = torch.ones(N, dtype=torch.uint8, device=device)
    episode(starts2, inpl.reset_grids(every, starts2, None), gath.reset_windows(starts2, None, max_length=Lg))
    for name in ("charge", "soc", "gen_status"):
        assert torch.equal(inpl.batch.cols[name], gath.batch.cols[name]), name
    inpl.close()
    # auto-reset: the step kernel restarts every grid at the end of its episode with its own draw
    auto = PerGridWindowEnv(batch(), trajectory_length=Lg, auto_reset=True, seed=13 + SOAK)
    assert auto.native and auto.env._ring is None
    for name in ("charge", "soc", "gen_status"):                   # carry on from the state the gathered env stands at
        auto.env.batch.cols[name].copy_(gath.batch.cols[name])
    starts3 = rs.randint(0, T - Lg + 1, size=N).astype(np.int32)
    o_a, o_g = auto.reset(starts3), gath.reset_windows(starts3, None, max_length=Lg)
    for episode_no in range(3):
        assert torch.equal(o_a, o_g), episode_no
        for k in range(Lg):
            a = torch.rand(N, A, dtype=torch.float64, device=device, generator=g)
            (o_a, r1, d1, _), (o_g, r2, d2, _) = auto.step(a), gath.step(a)
            assert torch.equal(r1, r2) and torch.equal(d1, d2), (episode_no, k)
            if k < Lg - 1:
                assert torch.equal(o_a, o_g), (episode_no, k)
        new_starts = auto.starts.clone()                           # what the kernel drew for the grids it restarted (all of them)
        assert int(new_starts.min()) >= 0 and int(new_starts.max()) <= T - Lg and not torch.equal(new_starts.cpu(), torch.as_tensor(starts3))
        o_g = gath.reset_windows(new_starts, None, max_length=Lg)  # the gathered windows replay the draw: o_a is the new episode's first row
    with pytest.raises(NotImplementedError):
        PerGridWindowEnv(batch(), trajectory_length=Lg, auto_reset=True, final_observation=True)
    auto.env.close(); gath.close()
