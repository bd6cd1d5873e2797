"""Factorised series (mgx_columns.base_load: base profile x per-grid ratio, formed in the kernels) == the materialised [T, N]
series, bit for bit, on every kernel that reads a series; `done` as bit sets / derived from the counter; the zero-copy
observation contract (ObsViews) == the rows contract.

Reference being matched: MicrogridGenerator.py:137-147 (_scale_ts), :205-212 (co2), :253-285 (tariff), :321-340 (weak grid);
base_timeseries_module.py:68-79,103-140,162-170 (stored sign, windows); forecaster.py:120-149 (padding, clip)."""
import numpy as np
import pytest
import torch

ARCHS = ("genset+battery", "battery+grid", "genset+battery+grid")


# ---------------------------------------------------------------------------------------------------------------------
# CPU: host packing
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arch", ARCHS)
def test_factorised_batch_materialises_to_the_generated_series_host(arch):
    """generate(series="factorised") holds base tables + factors; materialise() (the one multiply, torch on the host) gives
    exactly the arrays the materialised generator builds, bounds included."""
    from pymgrid_amd.generator import generate
    bm = generate(48, n_steps=300, seed=5, arch=arch, device="cpu", mixed_timers=True)
    bf = generate(48, n_steps=300, seed=5, arch=arch, device="cpu", mixed_timers=True, series="factorised")
    assert bf.factorised and not bm.factorised and "load_ts" not in bf.cols
    m = bf.materialise()
    assert set(m.cols) == set(bm.cols)
    for k, v in bm.cols.items():
        assert torch.equal(v, m.cols[k]), k
    assert m.cols["charge"] is bf.cols["charge"]                      # state columns are shared with the twin
    nc = bf.numpy_columns()                                           # what the oracle is fed in the GPU tests
    assert np.array_equal(nc["load_ts"], bm.cols["load_ts"].numpy())


def test_outage_bits_round_trip():
    from pymgrid_amd.generator import pack_outage_bits, unpack_outage_bits
    rs = np.random.RandomState(0)
    for T in (1, 63, 64, 65, 200):
        status = (rs.rand(T, 7) > 0.3).astype(np.float64)
        bits = pack_outage_bits(status)
        assert bits.shape == ((T + 63) // 64, 7) and bits.dtype == np.uint64
        back = unpack_outage_bits(torch.from_numpy(bits.view(np.int64).copy()), T)
        assert np.array_equal(back.numpy(), status)


def test_bytes_fused_accounting():
    from pymgrid_amd import BatchLayout
    t4 = BatchLayout(n_grids=1, n_steps=10, has_genset=True, has_battery=True, has_grid=False)
    assert t4.bytes_fused(64) == 140 + 64 * 57
    # factorised, `done` derived from the counter: 24 B of actions + reward + SoC per step; factors 18 B once
    assert t4.bytes_fused(64, factorised=True, done=False) == 140 + 18 + 64 * 40
    assert t4.bytes_fused(64, factorised=True, done=True, done_bits=True) == 140 + 18 + 64 * 40.125
    t5 = BatchLayout(n_grids=1, n_steps=10, has_genset=True, has_battery=True, has_grid=True)
    assert t5.bytes_fused(64, factorised=True, done=False) - t5.bytes_fused(64, done=False) == 20 + 64 * (0.125 - 48)


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
def _pair(n, T, arch, device, horizon=0, seed=9, **kw):
    """(materialised batch, factorised batch) of the same draw, each with its own state columns."""
    from pymgrid_amd.generator import generate
    bm = generate(n, n_steps=T, seed=seed, arch=arch, device=device, horizon=horizon, mixed_timers=True, **kw)
    bf = generate(n, n_steps=T, seed=seed, arch=arch, device=device, horizon=horizon, mixed_timers=True, series="factorised", **kw)
    return bm, bf


def _same_state(bm, bf):
    for k in ("charge", "soc", "gen_status"):
        if k in bm.cols:
            assert torch.equal(bm.cols[k], bf.cols[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
def test_device_outage_words_and_twin_equal_the_synthesised_series(arch, device):
    """The outage words the synthesis kernel packs == the grid_status column it writes; the factorised batch's materialised
    twin == the materialised batch (same Philox draws), incl. the analytically derived grid bounds."""
    bm, bf = _pair(3000, 700, arch, device)
    m = bf.materialise()
    for k, v in bm.cols.items():
        assert torch.equal(v, m.cols[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
@pytest.mark.parametrize("act_dtype", [torch.float64, torch.float32])
def test_fused_and_single_steps_factorised_vs_materialised(arch, act_dtype, device):
    """mgx_step_k (lean and rich form, across LDS chunks of 128 rows and 64-row outage words, from an odd start row), mgx_step,
    mgx_check_step, mgx_step_many: every output and the final state are equal."""
    from pymgrid_amd import StepEngine
    N, T = 5000, 700
    bm, bf = _pair(N, T, arch, device)
    em, ef = StepEngine(bm, action_dtype=act_dtype), StepEngine(bf, action_dtype=act_dtype)
    g = torch.Generator(device=device); g.manual_seed(1)
    A = bm.layout.action_dim
    for t0, K in ((37, 300), (0, 5), (T - 131, 131)):
        acts = torch.rand(K, N, A, dtype=act_dtype, device=device, generator=g)
        for e in (em, ef):
            e.reset(t0, want_obs=False)
        st = bm.state()
        om = em.step_k(acts, reward=True, done=True, soc_trace=True)
        of = ef.step_k(acts, reward=True, done=True, soc_trace=True)
        for k in om:
            assert torch.equal(om[k], of[k]), (k, t0)
        _same_state(bm, bf)
        # the HOT form of the loop (exactly reward + SoC, `done` derived from the counter) == the general form
        end = bm.state()
        for b, e in ((bm, em), (bf, ef)):
            b.load_state(st)
            e.reset(t0, want_obs=False)
            assert torch.equal(e.done_steps(K), om["done"].view(torch.bool))
            oh = e.step_k(acts, reward=True, done=False, soc_trace=True)
            assert set(oh) == {"reward", "soc_trace"}
            assert torch.equal(oh["reward"], om["reward"]) and torch.equal(oh["soc_trace"], om["soc_trace"]), t0
            for k, v in end.items():
                assert torch.equal(b.cols[k], v), k
    for e in (em, ef):
        e.reset(100, want_obs=False)
    acts = torch.rand(150, N, A, dtype=act_dtype, device=device, generator=g)
    om = em.step_k(acts, reward=True, done=True, soc_trace=True, status_trace=True, log=True)
    of = ef.step_k(acts, reward=True, done=True, soc_trace=True, status_trace=True, log=True)
    for k in om:
        assert torch.equal(om[k], of[k]), k
    _same_state(bm, bf)
    # single steps, the dry run, K launches by one call
    vm, vf = em.check_step(acts[0]), ef.check_step(acts[0])
    assert torch.equal(vm, vf)
    for k in range(3):
        rm = em.step(acts[k], want_obs=True, want_log=True)
        rf = ef.step(acts[k], want_obs=True, want_log=True)
        for x, y in zip(rm, rf):
            assert torch.equal(x, y)
    sm, sf = em.step_many(acts[3:9], want_obs=True), ef.step_many(acts[3:9], want_obs=True)
    for x, y in zip(sm[:3], sf[:3]):
        assert torch.equal(x, y)
    _same_state(bm, bf)
    em.close(); ef.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
def test_discrete_paths_factorised_vs_materialised(arch, device):
    """Priority-list expansion, the one-launch discrete step, per-step and fixed-list rollouts (RuleBasedControl)."""
    from pymgrid_amd import StepEngine
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    N, T = 4000, 520
    bm, bf = _pair(N, T, arch, device)
    em, ef = StepEngine(bm), StepEngine(bf)
    L = bm.layout
    lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, False)
    table = table_array(lists)
    g = torch.Generator(device=device); g.manual_seed(2)
    ids32 = torch.randint(0, len(lists), (N,), dtype=torch.int32, device=device, generator=g)
    for e in (em, ef):
        e.reset(11, want_obs=False)
    assert torch.equal(em.expand_discrete(ids32, table), ef.expand_discrete(ids32, table))
    rm = em.step_discrete(ids32, table, want_obs=True, want_log=True, want_control=True)
    rf = ef.step_discrete(ids32, table, want_obs=True, want_log=True, want_control=True)
    for x, y in zip(rm, rf):
        assert torch.equal(x, y)
    K = 400
    ids = torch.randint(0, len(lists), (K, N), dtype=torch.uint8, device=device, generator=g)
    st, t_st = bm.state(), em.current_step
    om = em.rollout_discrete(ids, table, K, reward=True, done=True, soc_trace=True)
    of = ef.rollout_discrete(ids, table, K, reward=True, done=True, soc_trace=True)
    for k in om:
        assert torch.equal(om[k], of[k]), k
    _same_state(bm, bf)
    end = bm.state()
    for b, e in ((bm, em), (bf, ef)):                       # the HOT form (reward + SoC only) of both rollouts == the general one
        for these in (ids, ids[0].contiguous()):
            b.load_state(st)
            e.reset(t_st, want_obs=False)
            ref = e.rollout_discrete(these, table, K, reward=True, done=True, soc_trace=True)
            b.load_state(st)
            e.reset(t_st, want_obs=False)
            oh = e.rollout_discrete(these, table, K, reward=True, done=False, soc_trace=True)
            assert torch.equal(oh["reward"], ref["reward"]) and torch.equal(oh["soc_trace"], ref["soc_trace"])
        b.load_state(st)
        e.reset(t_st, want_obs=False)
        e.rollout_discrete(ids, table, K, reward=True, done=False, soc_trace=True)
        for k, v in end.items():
            assert torch.equal(b.cols[k], v), k
    for e in (em, ef):
        e.reset(70, want_obs=False)
    fixed = ids[0].contiguous()
    om = em.rollout_discrete(fixed, table, K, reward=True, done=True, soc_trace=True, status_trace=True, log=True)
    of = ef.rollout_discrete(fixed, table, K, reward=True, done=True, soc_trace=True, status_trace=True, log=True)
    for k in om:
        assert torch.equal(om[k], of[k]), k
    _same_state(bm, bf)
    em.close(); ef.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
@pytest.mark.parametrize("obs_dtype", [torch.float64, torch.float32])
def test_observation_rows_and_rings_factorised_vs_materialised(arch, obs_dtype, device):
    """H = 24 observation rows: per-step rows (obs_rows_wave_kernel), prefetched rings (obs_windows_k_kernel), end-of-series
    padding; per-grid windows gathered out of the factors (mgx_reset_windows)."""
    from pymgrid_amd import BatchedMicrogridEnv
    N, T, H = 3001, 200, 24
    bm, bf = _pair(N, T, arch, device, horizon=H)
    g = torch.Generator(device=device); g.manual_seed(3)
    for prefetch in (0, 8):
        envm = BatchedMicrogridEnv(bm, obs_dtype=obs_dtype, obs_prefetch=prefetch)
        envf = BatchedMicrogridEnv(bf, obs_dtype=obs_dtype, obs_prefetch=prefetch)
        om, of = envm.reset(T - 40), envf.reset(T - 40)
        assert torch.equal(om, of)
        for k in range(40):                                      # to the last row: windows run into the padding
            a = envm.sample_action(generator=g)
            om, rm, dm, _ = envm.step(a)
            of, rf, df, _ = envf.step(a)
            assert torch.equal(om, of) and torch.equal(rm, rf) and torch.equal(dm, df), (prefetch, k)
        assert bool(dm.all())
        # per-grid episodes: rows gathered from the factors
        starts = torch.randint(0, T - 30, (N,), dtype=torch.int32, device=device, generator=g)
        lengths = torch.randint(1, 30, (N,), dtype=torch.int32, device=device, generator=g)
        om, of = envm.reset_windows(starts, lengths), envf.reset_windows(starts, lengths)
        assert torch.equal(om, of)
        for k in range(int(lengths.max().item())):
            a = envm.sample_action(generator=g)
            om, rm, dm, _ = envm.step(a)
            of, rf, df, _ = envf.step(a)
            assert torch.equal(om, of) and torch.equal(rm, rf) and torch.equal(dm, df), (prefetch, k)
        om, of = envm.reset(), envf.reset()                      # back to the full (factorised) series
        assert torch.equal(om, of) and envf.engine.batch.factorised
        a = envm.sample_action(generator=g)
        assert torch.equal(envm.step(a)[0], envf.step(a)[0])
        envm.close(); envf.close()
        bm.load_state(bf.state())


@pytest.mark.gpu
def test_done_bit_sets_and_lockstep_done(device):
    """MGX_DONE_BITS == the byte form (per-grid episode ends, mgx_reset_windows); lock-step `done` derived from the counter
    (engine.done_steps) == what the kernel writes."""
    from pymgrid_amd import StepEngine
    N, T, K = 5003, 300, 90
    bm, bf = _pair(N, T, "genset+battery", device)
    g = torch.Generator(device=device); g.manual_seed(4)
    acts = torch.rand(K, N, 3, dtype=torch.float64, device=device, generator=g)
    for b in (bm, bf):
        e = StepEngine(b)
        e.reset(T - K, want_obs=False)
        expect = e.done_steps(K)
        st = b.state()
        d8 = e.step_k(acts, reward=False, done=True)["done"]
        assert torch.equal(d8.view(torch.bool), expect) and bool(expect[-1].all()) and not bool(expect[:-1].any())
        # per-grid episode ends
        b.load_state(st)
        starts = torch.randint(0, T - K, (N,), dtype=torch.int32, device=device, generator=g)
        lengths = torch.randint(1, K + 1, (N,), dtype=torch.int32, device=device, generator=g)
        e.reset_windows(starts, lengths, want_obs=False)
        with pytest.raises(RuntimeError):
            e.done_steps(K)
        d8 = e.step_k(acts, reward=False, done=True)["done"].view(torch.bool)
        assert torch.equal(d8, torch.arange(K, device=device)[:, None] >= (lengths[None, :] - 1))
        b.load_state(st)
        e.reset_windows(starts, lengths, want_obs=False)
        e.set_done_format(True)
        words = e.step_k(acts, reward=False, done=True)["done"]
        assert words.shape == (K, (N + 15) // 16) and words.dtype == torch.int16
        assert torch.equal(e.unpack_done_bits(words), d8)
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
@pytest.mark.parametrize("series", ["materialised", "factorised"])
@pytest.mark.parametrize("obs_dtype", [torch.float64, torch.float32])
def test_zero_copy_views_equal_rows(arch, series, obs_dtype, device):
    """obs_views=True: the strided views into the once-normalised series + the compact state columns == the [N, D] rows of
    the default contract at every step, through the end-of-series padding and a per-grid-window episode."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T, H = 2500, 160, 24
    kw = dict(n_steps=T, seed=12, arch=arch, device=device, horizon=H, mixed_timers=True, series=series)
    br, bv = generate(N, **kw), generate(N, **kw)
    rows = BatchedMicrogridEnv(br, obs_dtype=obs_dtype)
    views = BatchedMicrogridEnv(bv, obs_dtype=obs_dtype, obs_views=True)
    g = torch.Generator(device=device); g.manual_seed(5)
    o, v = rows.reset(T - 50), views.reset(T - 50)
    assert torch.equal(o, v.flat()) and v.load.shape == (N, 1 + H) and v.state.shape == (N, views.engine.state_dim)
    assert v.load.data_ptr() == views._norm["load"].data_ptr() + (T - 50) * v.load.element_size()      # a view, not a copy
    held = []
    for k in range(49):
        a = rows.sample_action(generator=g)
        o, r, d, _ = rows.step(a)
        v, r2, d2, _ = views.step(a)
        assert torch.equal(o, v.flat()) and torch.equal(r, r2) and torch.equal(d, d2), k
        held.append((o, v))
        if len(held) >= views.VIEW_BUFFERS - 1:                  # an observation stays valid for VIEW_BUFFERS - 1 more steps
            o_old, v_old = held.pop(0)
            assert torch.equal(o_old, v_old.flat())
    n = v.nested()
    assert set(n) == {"load", "pv", "battery"} | ({"genset"} if br.layout.has_genset else set()) | ({"grid"} if br.layout.has_grid else set())
    starts = torch.randint(0, T - 20, (N,), dtype=torch.int32, device=device, generator=g)
    lengths = torch.randint(1, 20, (N,), dtype=torch.int32, device=device, generator=g)
    o, v = rows.reset_windows(starts, lengths), views.reset_windows(starts, lengths)
    assert torch.equal(o, v.flat())
    for k in range(int(lengths.max().item())):
        a = rows.sample_action(generator=g)
        o, r, d, _ = rows.step(a)
        v, r2, d2, _ = views.step(a)
        assert torch.equal(o, v.flat()) and torch.equal(r, r2) and torch.equal(d, d2), k
    o, v = rows.reset(), views.reset()
    assert torch.equal(o, v.flat())
    rows.close(); views.close()


@pytest.mark.gpu
def test_views_refuse_bounds_that_do_not_bound(device):
    """With a bound column tighter than its series the reference's forecast clip bites and windows are no longer slices of
    one normalised series: normalise_series says so."""
    from pymgrid_amd import BatchedMicrogridEnv, MgxError
    from pymgrid_amd.generator import generate
    b = generate(300, n_steps=60, seed=1, arch="genset+battery", device=device, horizon=4)
    b.cols["pv_hi"].mul_(0.5)
    with pytest.raises(MgxError):
        BatchedMicrogridEnv(b, obs_views=True)


@pytest.mark.gpu
@pytest.mark.parametrize("discrete", [False, True])
def test_fleet_with_views_and_factorised_series_vs_rows(discrete, device):
    """A heterogeneous fleet (three layouts, H = 24) stepped by mgx_fleet_step -- continuous controls, or priority-list ids
    (DiscreteMicrogridEnv.step for every grid) -- factorised + views == materialised + rows."""
    from pymgrid_amd.generator import generate_fleet
    from pymgrid_amd.hetero import BucketedFleet
    n, T, H = 9000, 100, 24
    pr = generate_fleet(n, n_steps=T, seed=17, horizon=H, device=device, mixed_timers=True)
    pv = generate_fleet(n, n_steps=T, seed=17, horizon=H, device=device, mixed_timers=True, series="factorised")
    names = list(pr)
    kw = dict(discrete=True, remove_redundant_gensets=False) if discrete else {}
    rows = BucketedFleet.from_batches([pr[k][0] for k in names], obs_prefetch=8, reuse_outputs=24, **kw)
    views = BucketedFleet.from_batches([pv[k][0] for k in names], obs_views=True, reuse_outputs=24, **kw)
    assert rows.fused and views.fused
    o, v = rows.reset(), views.reset()
    for x, y in zip(o, v):
        assert torch.equal(x, y.flat())
    g = torch.Generator(device=device); g.manual_seed(8)
    for k in range(T):
        acts = rows.sample_action(generator=g)
        o, r, d, _ = rows.step(acts)
        v, r2, d2, _ = views.step(acts)
        for b in range(len(names)):
            assert torch.equal(o[b], v[b].flat()) and torch.equal(r[b], r2[b]) and torch.equal(d[b], d2[b]), (k, names[b])
    assert all(bool(x.all()) for x in d)
    rows.close(); views.close()


@pytest.mark.gpu
def test_true_shape_config3_factorised_vs_materialised(device):
    """BASELINE configs[2] at its true shape: 100 000 Template-4 grids x 8 760 rows -- the factorised batch (no [T, N] series:
    14 GB less) steps bit-identically to the materialised one, in two shards as bench.py steps it, at the start and across
    the end of the year."""
    from pymgrid_amd import StepEngine
    from pymgrid_amd.generator import generate
    N, T, K = 100_000, 8760, 192
    bm = generate(N, n_steps=T, seed=42, arch="genset+battery", device=device)
    bf = generate(N, n_steps=T, seed=42, arch="genset+battery", device=device, series="factorised")
    assert "load_ts" not in bf.cols and bf.cols["base_load"].shape == (T, 8)
    em, ef = StepEngine(bm), StepEngine(bf)
    g = torch.Generator(device=device); g.manual_seed(7)
    acts = torch.rand(K, N, 3, dtype=torch.float64, device=device, generator=g)
    for shards in (1, 2):
        for t0 in (0, T - K):
            for e in (em, ef):
                e.set_shards(shards)
                e.reset(t0, want_obs=False)
                e.fork()
            om = em.step_k(acts, reward=True, done=True, soc_trace=True)
            of = ef.step_k(acts, reward=True, done=True, soc_trace=True)
            em.join(); ef.join()
            torch.cuda.synchronize(device)
            for k in om:
                assert torch.equal(om[k], of[k]), (k, t0, shards)
            _same_state(bm, bf)
    em.close(); ef.close()


# ---------------------------------------------------------------------------------------------------------------------
# flat_order="gym": the reference's flat observation under gym's key-sorting Dict
# ---------------------------------------------------------------------------------------------------------------------
def test_gym_flat_order_layout():
    """BatchLayout(flat_order="gym"): blocks in alphabetical module order, names and slices consistent."""
    from pymgrid_amd import BatchLayout
    L = BatchLayout(n_grids=1, n_steps=10, horizon=2, has_genset=True, has_battery=True, has_grid=True, flat_order="gym")
    sl = L.obs_slices()
    assert list(sl) == ["battery", "genset", "grid", "load", "pv"]
    assert (sl["battery"], sl["genset"], sl["grid"], sl["load"], sl["pv"]) == \
        (slice(0, 2), slice(2, 6), slice(6, 18), slice(18, 21), slice(21, 24))
    assert L.obs_names[:6] == ["soc", "current_charge", "current_status", "goal_status", "steps_until_up", "steps_until_down"]
    assert L.obs_names[18:] == ["load_current", "load_forecast_0", "load_forecast_1", "renewable_current", "renewable_forecast_0",
                                "renewable_forecast_1"]
    M = BatchLayout(n_grids=1, n_steps=10, horizon=2, has_genset=True, has_battery=True, has_grid=True)
    assert list(M.obs_slices()) == ["load", "pv", "genset", "battery", "grid"] and M.obs_dim == L.obs_dim == 24


@pytest.mark.refcheck
def test_reference_flat_observation_under_a_key_sorting_dict_is_the_gym_order():
    """The reference builds gym.spaces.Dict(obs_space) from a plain dict (envs/base/base.py:128-163) and flattens observations
    through it (:211-223); gym <= 0.26 / gymnasium SORT the keys of such a Dict.  With a Dict that does what gym does, the
    reference's own flat observation == the nested observation concatenated in BatchLayout(flat_order="gym") order."""
    import _refenv
    if not _refenv.reference_available():
        pytest.skip("reference not present")
    from collections import OrderedDict

    class SortingDict(_refenv.Dict):                       # gym/spaces/dict.py: OrderedDict(sorted(spaces.items())) for a plain dict
        def __init__(self, spaces=None, seed=None, **kw):
            super().__init__(spaces, seed, **kw)
            self.spaces = OrderedDict(sorted(self.spaces.items()))
    _refenv.import_reference()
    import gym
    from pymgrid.envs import DiscreteMicrogridEnv
    from pymgrid_amd import BatchLayout
    old = gym.spaces.Dict
    import pymgrid.envs.base.base as base_mod
    old_base = base_mod.Dict
    try:
        gym.spaces.Dict = base_mod.Dict = SortingDict
        for n in (0, 1, 2):
            nested_env = DiscreteMicrogridEnv.from_scenario(n)
            nested_env._flat_spaces = False
            flat_env = DiscreteMicrogridEnv.from_scenario(n)
            np.random.seed(n)
            o_n, o_f = nested_env.reset(), flat_env.reset()
            names = {k for k in ("genset", "battery", "grid") if k in o_n}
            L = BatchLayout(n_grids=1, n_steps=8760, horizon=23, has_genset="genset" in names, has_battery="battery" in names,
                            has_grid="grid" in names, flat_order="gym")
            for k in range(5):
                expect = np.concatenate([np.asarray(o_n[name][0], dtype=np.float64) for name in L.obs_slices()])
                assert o_f.shape == (L.obs_dim,) and np.array_equal(np.asarray(o_f, dtype=np.float64), expect), (n, k)
                a = k % flat_env.action_space.n
                o_n, o_f = nested_env.step(a)[0], flat_env.step(a)[0]
    finally:
        gym.spaces.Dict, base_mod.Dict = old, old_base


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
def test_gym_flat_order_rows_are_the_permuted_module_order_rows(arch, device):
    """flat_order="gym" moves the column bases of the module blocks, nothing else: every observation kernel (H = 0 rows, per-step
    rows, prefetched rings, patched rings of rolling windows, float32 rows) == the module-order rows, permuted."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T = 2000, 120
    g = torch.Generator(device=device); g.manual_seed(6)
    for H, prefetch, dt in ((0, 0, torch.float64), (24, 0, torch.float64), (24, 8, torch.float64), (24, 8, torch.float32)):
        bm = generate(N, n_steps=T, seed=4, arch=arch, device=device, horizon=H, mixed_timers=True)
        bg = generate(N, n_steps=T, seed=4, arch=arch, device=device, horizon=H, mixed_timers=True, flat_order="gym")
        sm, sg = bm.layout.obs_slices(), bg.layout.obs_slices()
        perm = torch.cat([torch.arange(sm[name].start, sm[name].stop) for name in sg]).to(device)
        em = BatchedMicrogridEnv(bm, obs_dtype=dt, obs_prefetch=prefetch)
        eg = BatchedMicrogridEnv(bg, obs_dtype=dt, obs_prefetch=prefetch)
        om, og = em.reset(T - 30), eg.reset(T - 30)
        assert torch.equal(om[:, perm], og)
        for k in range(30):
            a = em.sample_action(generator=g)
            om, og = em.step(a)[0], eg.step(a)[0]
            assert torch.equal(om[:, perm], og), (H, prefetch, k)
        if H and prefetch:                                 # rolling windows: restarted grids are patched into the rings
            starts = torch.randint(0, T - 12, (N,), dtype=torch.int32, device=device, generator=g)
            lengths = torch.randint(1, 12, (N,), dtype=torch.int32, device=device, generator=g)
            om, og = em.reset_windows(starts, lengths, max_length=12, rolling=True), eg.reset_windows(starts, lengths, max_length=12, rolling=True)
            assert torch.equal(om[:, perm], og)
            for k in range(20):
                a = em.sample_action(generator=g)
                om, _, d, _ = em.step(a)
                og = eg.step(a)[0]
                assert torch.equal(om[:, perm], og), k
                if bool(d.any()):
                    ns = torch.randint(0, T - 12, (N,), dtype=torch.int32, device=device, generator=g)
                    nl = torch.randint(1, 12, (N,), dtype=torch.int32, device=device, generator=g)
                    om, og = em.reset_grids(d, ns, nl), eg.reset_grids(d, ns, nl)
                    assert torch.equal(om[:, perm], og), k
        em.close(); eg.close()


@pytest.mark.gpu
def test_factorised_with_grid_before_battery_and_reward_shapers(device):
    """The remaining specialisations of the factorised fused kernels: the grid-before-battery sweep order (F = 14 / 15:
    module_container.py:355-413) and shaped rewards (reward_shaping/*.py: the non-HOT form of the loop)."""
    from dataclasses import replace
    from pymgrid_amd import MicrogridBatch, StepEngine
    from pymgrid_amd._lib import FACTOR_COLUMNS
    N, T, K = 3000, 400, 260
    g = torch.Generator(device=device); g.manual_seed(10)
    for arch in ("battery+grid", "genset+battery+grid"):
        bm, bf = _pair(N, T, arch, device)
        for gfb in (False, True):
            for shaper in (0, 1, 2):
                pair = []
                for b in (bm, bf):
                    cols = {k: (v.clone() if k in MicrogridBatch.STATE_COLUMNS else v) for k, v in b.cols.items()}
                    e = StepEngine(MicrogridBatch(replace(b.layout, grid_before_battery=gfb), cols))
                    e.set_reward_shaper(shaper)
                    e.reset(13, want_obs=False)
                    pair.append(e)
                acts = torch.rand(K, N, bm.layout.action_dim, dtype=torch.float64, device=device, generator=g)
                om = pair[0].step_k(acts, reward=True, soc_trace=True, status_trace=gfb, log=gfb)
                of = pair[1].step_k(acts, reward=True, soc_trace=True, status_trace=gfb, log=gfb)
                for k in om:
                    assert torch.equal(om[k], of[k]), (arch, gfb, shaper, k)
                for k in MicrogridBatch.STATE_COLUMNS:
                    if k in pair[0].batch.cols:
                        assert torch.equal(pair[0].batch.cols[k], pair[1].batch.cols[k]), (arch, gfb, shaper, k)
                for e in pair:
                    e.close()
    # the sweep order matters (else this test would prove nothing): rewards of the two orders differ somewhere
    bm, _ = _pair(N, T, "genset+battery+grid", device)
    outs = []
    for gfb in (False, True):
        cols = {k: (v.clone() if k in MicrogridBatch.STATE_COLUMNS else v) for k, v in bm.cols.items()}
        e = StepEngine(MicrogridBatch(replace(bm.layout, grid_before_battery=gfb), cols))
        outs.append(e.step_k(acts, reward=True)["reward"])
        e.close()
    assert not torch.equal(outs[0], outs[1])


# ---------------------------------------------------------------------------------------------------------------------
# batch-uniform parameter columns (mgx_columns.uniform_mask)
# ---------------------------------------------------------------------------------------------------------------------
def test_uniform_columns_host():
    """generate(uniform_columns=True): the parameters MicrogridGenerator gives every microgrid are stride-0 columns (one value),
    equal to the full columns; the C struct carries their bits; the oracle's copy is contiguous again."""
    from pymgrid_amd import _lib
    from pymgrid_amd.generator import generate
    bu = generate(40, n_steps=50, seed=1, arch="genset+battery+grid", device="cpu", uniform_columns=True)
    bn = generate(40, n_steps=50, seed=1, arch="genset+battery+grid", device="cpu")
    assert bn.uniform_columns() == [] and bn.c_columns().uniform_mask == 0
    names = bu.uniform_columns()
    assert set(names) == {"bat_efficiency", "bat_cost_cycle", "gen_cost", "gen_co2_per_unit", "gen_cost_per_unit_co2", "gen_times",
                          "grid_cost_per_unit_co2", "loss_load_cost", "overgeneration_cost"}
    assert bu.uniform_param_bytes() == 8 * 8 + 4
    assert bu.c_columns().uniform_mask == sum(1 << _lib.UNIFORM_BITS.index(n) for n in names)
    for k in bn.cols:
        assert torch.equal(bu.cols[k], bn.cols[k]), k
    assert all(a.flags.c_contiguous for a in bu.numpy_columns().values() if isinstance(a, np.ndarray))
    mixed = generate(40, n_steps=50, seed=1, arch="genset+battery", device="cpu", uniform_columns=True, mixed_timers=True)
    assert "gen_times" not in mixed.uniform_columns()                  # drawn per grid: a real column


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
@pytest.mark.parametrize("series", ["materialised", "factorised"])
def test_uniform_columns_equal_full_columns(arch, series, device):
    """Every kernel family on a batch with uniform parameter columns == the batch with full [N] columns: single steps (+ log,
    obs), fused steps (HOT and general form), rollouts, the discrete step, observation rows / rings, views, a fleet step."""
    from pymgrid_amd import BatchedMicrogridEnv, StepEngine
    from pymgrid_amd.generator import generate
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    N, T, H = 3000, 700, 24
    kw = dict(n_steps=T, seed=21, arch=arch, device=device, series=series)
    bn, bu = generate(N, **kw), generate(N, uniform_columns=True, **kw)
    assert bu.uniform_columns() and not bn.uniform_columns()
    en, eu = StepEngine(bn), StepEngine(bu)
    g = torch.Generator(device=device); g.manual_seed(4)
    A = bn.layout.action_dim
    acts = torch.rand(140, N, A, dtype=torch.float64, device=device, generator=g)
    for e in (en, eu):
        e.reset(5, want_obs=False)
    for k in range(3):
        rn, ru = en.step(acts[k], want_obs=True, want_log=True), eu.step(acts[k], want_obs=True, want_log=True)
        for x, y in zip(rn, ru):
            assert torch.equal(x, y)
    for kwargs in (dict(reward=True, soc_trace=True), dict(reward=True, done=True, soc_trace=True, status_trace=True, log=True)):
        on, ou = en.step_k(acts, **kwargs), eu.step_k(acts, **kwargs)
        for k in on:
            assert torch.equal(on[k], ou[k]), k
    L = bn.layout
    lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, False)
    table = table_array(lists)
    ids = torch.randint(0, len(lists), (100, N), dtype=torch.uint8, device=device, generator=g)
    for these in (ids, ids[0].contiguous()):
        on, ou = en.rollout_discrete(these, table, 100, reward=True, soc_trace=True), eu.rollout_discrete(these, table, 100, reward=True, soc_trace=True)
        for k in on:
            assert torch.equal(on[k], ou[k]), k
    i32 = ids[1].to(torch.int32)
    for x, y in zip(en.step_discrete(i32, table, want_log=True, want_control=True), eu.step_discrete(i32, table, want_log=True, want_control=True)):
        assert torch.equal(x, y)
    for k in ("charge", "soc", "gen_status"):
        if k in bn.cols:
            assert torch.equal(bn.cols[k], bu.cols[k]), k
    en.close(); eu.close()
    # observations: rows (per step and rings) and views
    kw = dict(kw, horizon=H, n_steps=120)
    for mode in (dict(obs_prefetch=0), dict(obs_prefetch=8), dict(obs_views=True)):
        vn, vu = BatchedMicrogridEnv(generate(N, **kw), **mode), BatchedMicrogridEnv(generate(N, uniform_columns=True, **kw), **mode)
        flat = (lambda o: o.flat()) if "obs_views" in mode else (lambda o: o)
        assert torch.equal(flat(vn.reset(90)), flat(vu.reset(90)))
        for k in range(30):
            a = vn.sample_action(generator=g)
            (o1, r1, d1, _), (o2, r2, d2, _) = vn.step(a), vu.step(a)
            assert torch.equal(flat(o1), flat(o2)) and torch.equal(r1, r2) and torch.equal(d1, d2), (mode, k)
        vn.close(); vu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ARCHS)
def test_factorised_batch_vs_the_oracle_directly(arch, device, oracle):
    """The factorised kernels against the CPU oracle itself (not only against the materialised kernels): fused steps from an
    odd row across two LDS chunks, then a rule-based rollout with the marginal-cost priority lists (RuleBasedControl), on the
    series the factors stand for."""
    from pymgrid_amd import StepEngine
    from pymgrid_amd.generator import generate
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    from pymgrid_amd.rbc import default_priority_ids
    N, T, K = 2048, 400, 300
    bf = generate(N, n_steps=T, seed=33, arch=arch, device=device, mixed_timers=True, series="factorised")
    bm = generate(N, n_steps=T, seed=33, arch=arch, device=device, mixed_timers=True)
    cols = bf.numpy_columns()
    st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
    g = torch.Generator(device=device); g.manual_seed(12)
    acts = torch.rand(K, N, bf.layout.action_dim, dtype=torch.float64, device=device, generator=g)
    e = StepEngine(bf)
    e.reset(41, want_obs=False)
    out = e.step_k(acts, reward=True, soc_trace=True)
    ref = oracle.run_batch(cols, st, 41, K, acts.cpu().numpy(), normalized=True, nthreads=8)
    assert np.array_equal(out["reward"].cpu().numpy(), ref)
    assert np.array_equal(bf.cols["charge"].cpu().numpy(), st["charge"]) and np.array_equal(out["soc_trace"][-1].cpu().numpy(), st["soc"])
    if "gen_status" in st:
        assert np.array_equal(bf.cols["gen_status"].cpu().numpy().view(np.uint32), st["gen_status"])
    # rule-based control: the priority ids of a factorised batch (tariff prices from the pattern) == the materialised batch's
    L = bf.layout
    lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, False)
    ids_f = default_priority_ids(bf, lists, remove_redundant_gensets=False)
    assert np.array_equal(ids_f, default_priority_ids(bm, lists, remove_redundant_gensets=False))
    e.reset(0, want_obs=False)
    st = {k: bf.cols[k].cpu().numpy().view(np.uint32 if k == "gen_status" else np.float64).copy() for k in st}
    r = e.rollout_discrete(torch.from_numpy(ids_f).to(device), table_array(lists), K, reward=True, soc_trace=True)
    ref = oracle.rollout_batch(cols, st, 0, K, ids_f, table_array(lists), nthreads=8)
    assert np.array_equal(r["reward"].cpu().numpy(), ref)
    assert np.array_equal(bf.cols["charge"].cpu().numpy(), st["charge"])
    e.close()


@pytest.mark.gpu
def test_views_and_lockstep_done_through_resets_and_trajectories(device):
    """The host-side bookkeeping of round 3 under stress: observation views, the lock-step `done` constants and the host mirror
    of the step counter through mid-episode resets, a FixedLengthStochasticTrajectory (a new window at every reset) and a fused
    fleet that is reset in the middle of its cached step plans -- always == a rows env driven the same way."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate, generate_fleet
    from pymgrid_amd.hetero import BucketedFleet
    from pymgrid_amd.trajectory import FixedLengthStochasticTrajectory
    N, T, H = 1500, 150, 24
    kw = dict(n_steps=T, seed=77, arch="genset+battery+grid", device=device, horizon=H, mixed_timers=True)
    np.random.seed(5)
    rows = BatchedMicrogridEnv(generate(N, **kw), trajectory_func=FixedLengthStochasticTrajectory(17))
    np.random.seed(5)
    views = BatchedMicrogridEnv(generate(N, series="factorised", **kw), trajectory_func=FixedLengthStochasticTrajectory(17), obs_views=True)
    g = torch.Generator(device=device); g.manual_seed(9)
    for episode in range(4):
        np.random.seed(100 + episode); o = rows.reset()
        np.random.seed(100 + episode); v = views.reset()
        assert rows.current_step == views.current_step == rows.initial_step and (rows.initial_step, rows.final_step) == (views.initial_step, views.final_step)
        assert torch.equal(o, v.flat())
        n_steps = 17 if episode % 2 == 0 else 6                  # every other episode is abandoned mid-way
        for k in range(n_steps):
            a = rows.sample_action(generator=g)
            o, r, d, _ = rows.step(a)
            v, r2, d2, _ = views.step(a)
            assert torch.equal(o, v.flat()) and torch.equal(r, r2) and torch.equal(d, d2), (episode, k)
            assert bool(d.all()) == (k == 16) and rows.current_step == views.current_step == rows.initial_step + k + 1
    rows.close(); views.close()
    # a fused fleet: reset in the middle of its rings / cached plans, several times
    pr = generate_fleet(6000, n_steps=90, seed=3, horizon=H, device=device, mixed_timers=True)
    pv = generate_fleet(6000, n_steps=90, seed=3, horizon=H, device=device, mixed_timers=True, series="factorised")
    names = list(pr)
    fr = BucketedFleet.from_batches([pr[k][0] for k in names], obs_prefetch=8, reuse_outputs=16)
    fv = BucketedFleet.from_batches([pv[k][0] for k in names], obs_views=True, reuse_outputs=16)
    for n_steps in (13, 5, 30, 89):
        o, v = fr.reset(), fv.reset()
        for x, y in zip(o, v):
            assert torch.equal(x, y.flat())
        for k in range(n_steps):
            acts = fr.sample_action(generator=g)
            o, r, d, _ = fr.step(acts)
            v, r2, d2, _ = fv.step(acts)
            for b in range(len(names)):
                assert torch.equal(o[b], v[b].flat()) and torch.equal(r[b], r2[b]) and torch.equal(d[b], d2[b]), (n_steps, k, names[b])
            assert all(e.current_step == k + 1 for e in fr.envs + fv.envs)
    fr.close(); fv.close()


@pytest.mark.gpu
def test_factorised_edge_shapes_vs_the_oracle(device, oracle):
    """Series lengths around the 64-row outage words and the 128-row LDS chunks of the fused kernels (T = 2 ... 257), batches of
    1 ... 257 grids, fused launches that start in the last rows of a chunk and end on the last row, single steps with
    observations whose windows (H = 0, 3, T + 2) run past the end of the series, the rule-based rollout: == the CPU oracle on the
    series the factors stand for; observations == the materialised twin's."""
    from pymgrid_amd import BatchedMicrogridEnv, StepEngine
    from pymgrid_amd.generator import generate
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    from pymgrid_amd.rbc import default_priority_ids
    rs = np.random.RandomState(0)
    for T, N in ((2, 1), (3, 65), (63, 257), (64, 63), (65, 64), (127, 130), (128, 1), (129, 257), (257, 66)):
        arch = ("genset+battery+grid", "battery+grid", "genset+battery")[T % 3]
        bf = generate(N, n_steps=T, seed=100 + T, arch=arch, device=device, mixed_timers=True, series="factorised")
        cols = bf.numpy_columns()
        A = bf.layout.action_dim
        e = StepEngine(bf)
        for t0 in sorted({0, max(0, T - 2), max(0, min(T - 1, 126)), rs.randint(0, T)}):
            K = T - t0
            st = {k: bf.cols[k].cpu().numpy().view(np.uint32 if k == "gen_status" else np.float64).copy()
                  for k in ("charge", "soc", "gen_status") if k in bf.cols}
            acts = torch.rand(K, N, A, dtype=torch.float64, device=device)
            e.reset(t0, want_obs=False)
            out = e.step_k(acts, reward=True, soc_trace=True, done=True)
            ref = oracle.run_batch(cols, st, t0, K, acts.cpu().numpy(), normalized=True)
            assert np.array_equal(out["reward"].cpu().numpy(), ref), (T, N, t0)
            assert np.array_equal(bf.cols["charge"].cpu().numpy(), st["charge"]), (T, N, t0)
            assert bool(out["done"][-1].all()) and (K == 1 or not bool(out["done"][:-1].any()))
        # rule-based rollout over the whole series
        lists = get_priority_lists(bf.layout.has_genset, bf.layout.has_battery, bf.layout.has_grid, False)
        ids = default_priority_ids(bf, lists, remove_redundant_gensets=False)
        st = {k: bf.cols[k].cpu().numpy().view(np.uint32 if k == "gen_status" else np.float64).copy()
              for k in ("charge", "soc", "gen_status") if k in bf.cols}
        e.reset(0, want_obs=False)
        r = e.rollout_discrete(torch.from_numpy(ids).to(device), table_array(lists), T, reward=True)
        ref = oracle.rollout_batch(cols, st, 0, T, ids, table_array(lists))
        assert np.array_equal(r["reward"].cpu().numpy(), ref), (T, N, "rbc")
        e.close()
        # observations: single steps to the end of the series, windows past it; factorised == materialised twin
        for H in (0, 3, T + 2):
            kw = dict(n_steps=T, seed=100 + T, arch=arch, device=device, mixed_timers=True, horizon=H)
            em, ef = BatchedMicrogridEnv(generate(N, **kw)), BatchedMicrogridEnv(generate(N, series="factorised", **kw))
            assert torch.equal(em.reset(), ef.reset()), (T, N, H)
            for k in range(T - 1):
                a = torch.rand(N, A, dtype=torch.float64, device=device)
                (o1, r1, d1, _), (o2, r2, d2, _) = em.step(a), ef.step(a)
                assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), (T, N, H, k)
            em.close(); ef.close()
