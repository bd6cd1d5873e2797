"""Window prefetch on the general path: microgrids with SEVERAL modules of a kind (module_container.py:355-413) get their
forecast windows from the same K-step refill as the single-instance layouts (obs_windows_k_multi_kernel) -- the rows an env
returns out of its rings must equal, bit for bit, the rows the per-step kernel writes (observe_row_multi, itself pinned to the
oracle and to the reference-made multi.npz in test_multi_module.py / test_multi_instances.py)."""
import pytest

torch = pytest.importorskip("torch")


def _envs(device, dt, K, layout="rows", **wide):
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate, widen
    out = []
    for k in (K, 0):
        base = generate(150, n_steps=90, seed=11, arch="genset+battery+grid", horizon=24, device=device)
        out.append(BatchedMicrogridEnv(widen(base, **wide), obs_prefetch=k, obs_dtype=dt, obs_layout=layout if k else "rows"))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["rows", "columns", None])
@pytest.mark.parametrize("dt", ["float64", "float32"])
@pytest.mark.parametrize("wide", [dict(n_genset=2, n_battery=2, n_grid=1), dict(n_genset=1, n_battery=3, n_grid=2, n_load=2),
                                  dict(n_genset=2, n_battery=1, n_grid=1, n_load=2, n_pv=3)])
def test_ring_rows_of_multi_instance_grids_are_the_per_step_rows(device, dt, wide, layout):
    """layout: row-major ring blocks, column-major ones ([D, pitch]; the env returns the transposed view), and the default (column-
    major while the batch walks in lock-step)."""
    dt = getattr(torch, dt)
    ring, plain = _envs(device, dt, 4, layout, **wide)
    assert ring.obs_prefetch == 4 and plain.obs_prefetch == 0 and ring.layout.multi
    L = ring.layout
    gen = torch.Generator(device=device); gen.manual_seed(2)
    o_r, o_p = ring.reset(), plain.reset()
    assert o_r.shape == (150, L.obs_dim) and torch.equal(o_r, o_p)
    assert o_r.stride() == ((L.obs_dim, 1) if layout == "rows" else (1, 160))
    for k in range(70):                                   # 17 ring changes; rows past the end of the series from step 66 on
        a = torch.rand(150, L.action_dim, dtype=torch.float64, device=device, generator=gen)
        o_r, r_r, d_r, _ = ring.step(a)
        o_p, r_p, d_p, _ = plain.step(a)
        assert torch.equal(o_r, o_p), (k, (o_r != o_p).nonzero()[:4])
        assert torch.equal(r_r, r_p) and torch.equal(d_r, d_p)
        if k == 37:                                       # a reset in the middle of a ring
            o_r, o_p = ring.reset(), plain.reset()
            assert torch.equal(o_r, o_p)
    ring.close(); plain.close()


@pytest.mark.gpu
def test_fleet_with_a_multi_instance_bucket_on_rings(device):
    """A fleet of a single-instance bucket and a multi-instance one, both on rings: since round 6 the multi bucket (two of a kind:
    the register form) steps INSIDE the fleet's one launch (fleet_step_kernel_vm) and refills through the general window kernel;
    rows == the envs stepped alone without rings."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate, widen
    from pymgrid_amd.hetero import BucketedFleet

    def batches():
        b0 = generate(100, n_steps=60, seed=3, arch="genset+battery", horizon=24, device=device)
        b1 = widen(generate(70, n_steps=60, seed=4, arch="genset+battery+grid", horizon=24, device=device), n_genset=2, n_battery=2)
        return [b0, b1]
    fleet = BucketedFleet.from_batches(batches(), obs_prefetch=4)
    alone = [BatchedMicrogridEnv(b, obs_prefetch=0) for b in batches()]
    gen = torch.Generator(device=device); gen.manual_seed(5)
    obs = fleet.reset()
    for o, e in zip(obs, alone):
        assert torch.equal(o, e.reset())
    for k in range(30):
        acts = [torch.rand(e.layout.n_grids, e.layout.action_dim, dtype=torch.float64, device=device, generator=gen) for e in alone]
        obs, rew, done, _ = fleet.step(acts)
        for j, e in enumerate(alone):
            o, r, d, _ = e.step(acts[j])
            assert torch.equal(obs[j], o) and torch.equal(rew[j], r), (k, j)
    fleet.close()
    for e in alone:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [0, 4])
@pytest.mark.parametrize("wide", [dict(n_genset=2, n_battery=2, n_grid=1), dict(n_genset=1, n_battery=2, n_grid=2, n_load=2, n_pv=3)])
def test_per_grid_windows_of_multi_instance_grids(device, wide, K):
    """mgx_reset_windows on the general path (per-microgrid trajectories, microgrid.py:205-225): grid i walks its own rows
    start[i] + k and is done after length[i] steps -- rows, rewards and log of every grid == those of the lock-step batch reset at
    that grid's start row (itself pinned to the reference-made multi.npz), forecasts past the end of the series included; with and
    without rings."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate, widen
    N, T, H, steps = 150, 90, 24, 30

    def make(k):
        base = generate(N, n_steps=T, seed=11, arch="genset+battery+grid", horizon=H, device=device)
        return BatchedMicrogridEnv(widen(base, **wide), obs_prefetch=k)
    starts_of = (0, 17, 50)                                    # 50 + 30 + 24 > 90: windows that run past the series
    gen = torch.Generator(device=device); gen.manual_seed(7)
    pick = torch.randint(0, 3, (N,), device=device, generator=gen)
    start = torch.tensor(starts_of, device=device, dtype=torch.int32)[pick]
    length = torch.randint(5, steps + 1, (N,), device=device, generator=gen).to(torch.int32)
    env = make(K)
    assert env.layout.multi and env.obs_prefetch == K
    acts = [torch.rand(N, env.layout.action_dim, dtype=torch.float64, device=device, generator=gen) for _ in range(steps)]
    obs = [env.reset_windows(start, length, max_length=steps).clone()]
    rew, done = [], []
    for a in acts:
        o, r, d, _ = env.step(a)
        obs.append(o.clone()); rew.append(r.clone()); done.append(d.clone())
    assert torch.equal(env.current_steps.to(torch.int32), start + steps)
    for v, s0 in enumerate(starts_of):
        ref = make(0)                                          # the same grids, lock-step from row s0
        mine = pick == v
        o = ref.reset(s0)
        assert torch.equal(obs[0][mine], o[mine]), ("reset", s0)
        for k, a in enumerate(acts):
            o, r, d, _ = ref.step(a)
            assert torch.equal(obs[k + 1][mine], o[mine]), (s0, k)
            assert torch.equal(rew[k][mine], r[mine]), (s0, k)
        ref.close()
    for k in range(steps):                                     # done_i from step length_i - 1 on (base_timeseries_module.py:124-125)
        assert torch.equal(done[k].to(torch.bool), k >= length - 1), k
    env.reset()                                                # a plain reset returns to the shared window
    assert int(env.current_steps.max()) == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("windows, shards", [(True, 1), (False, 2)])
def test_discrete_steps_of_multi_instance_grids_in_per_grid_windows(device, windows, shards):
    """mgx_step_lists (the one-launch discrete step of the general path) on a handle in per-grid windows (mgx_reset_windows: gathered
    window buffers, `done` per grid) -- and, in lock-step, over two shards (shards are not offered during per-grid windows) -- ==
    expansion + continuous step of a twin env, step by step."""
    from pymgrid_amd import DiscreteBatchedMicrogridEnv
    from pymgrid_amd.generator import generate, widen
    N, T, steps = 300, 80, 20

    def make():
        base = generate(N, n_steps=T, seed=21, arch="genset+battery+grid", horizon=4, device=device)
        return DiscreteBatchedMicrogridEnv(widen(base, n_genset=2, n_battery=2, n_grid=1), obs_prefetch=0, remove_redundant_gensets=False)
    gen = torch.Generator(device=device); gen.manual_seed(9)
    start = torch.randint(0, 40, (N,), device=device, generator=gen).to(torch.int32)
    length = torch.randint(5, steps + 1, (N,), device=device, generator=gen).to(torch.int32)
    a, b = make(), make()
    if windows:
        oa, ob = a.reset_windows(start, length, max_length=steps), b.reset_windows(start, length, max_length=steps)
    else:
        oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    if shards > 1:
        a.engine.set_shards(shards); a.engine.fork()
    for k in range(steps):
        ids = a.sample_action(generator=gen)
        oa, ra, da, _ = a.step(ids)
        ob, rb, db, _ = super(DiscreteBatchedMicrogridEnv, b).step(b.get_action(ids), normalized=False)
        if shards > 1:
            a.engine.join(); a.engine.fork()
        assert torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(oa, ob), k
    if shards > 1:
        a.engine.join(); a.engine.set_shards(1)
    for name in ("charge", "soc", "gen_status"):
        assert torch.equal(a.batch.cols[name], b.batch.cols[name])
    a.close(); b.close()
