"""Step + observation row in ONE launch (fleet_rows_kernel, mgx_set_rows_direct / obs_direct=True: mgx_step / mgx_step_discrete /
mgx_fleet_step on a factorised batch with a forecast horizon and whole rows wanted): every row, reward and done flag equals the materialised twin's, which goes through
step_kernel + obs_rows_wave_kernel (itself pinned against the reference's observations in test_gpu_parity.py) -- from an odd start
row to the end of the series (padding), ragged N, float32 rows, discrete steps, logs, fleets of three layouts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(n, T, arch, H, device, seed=5, **kw):
    from pymgrid_amd.generator import generate
    f = generate(n, n_steps=T, seed=seed, arch=arch, horizon=H, device=device, mixed_timers=True, series="factorised", **kw)
    m = generate(n, n_steps=T, seed=seed, arch=arch, horizon=H, device=device, mixed_timers=True, series="materialised", **kw)
    return f, m


@pytest.mark.parametrize("arch,H,dtype", [("genset+battery+grid", 24, torch.float64), ("genset+battery", 24, torch.float64),
                                          ("battery+grid", 24, torch.float32), ("genset+battery+grid", 5, torch.float64),
                                          ("genset+battery", 1, torch.float32), ("genset+battery+grid", 70, torch.float64)])
def test_direct_rows_equal_the_materialised_twin(arch, H, dtype, device):
    from pymgrid_amd import BatchedMicrogridEnv
    N, T = 1003, 140
    f, m = _pair(N, T, arch, H, device)
    ef = BatchedMicrogridEnv(f, obs_dtype=dtype, obs_direct=True, log=True)
    em = BatchedMicrogridEnv(m, obs_dtype=dtype, obs_prefetch=0, log=True)
    g = torch.Generator(device=device); g.manual_seed(7)
    for start, n_steps in ((0, 9), (37, 6), (T - H - 4, H + 3)):
        assert torch.equal(ef.reset(start), em.reset(start))
        for k in range(min(n_steps, T - start - 1)):
            a = torch.rand(N, f.layout.action_dim, dtype=torch.float64, device=device, generator=g) * 1.2 - 0.1
            (o1, r1, d1, i1), (o2, r2, d2, i2) = ef.step(a), em.step(a)
            assert torch.equal(o1, o2), (start, k)
            assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(i1["log"], i2["log"])
    assert torch.equal(f.cols["charge"], m.cols["charge"])
    ef.close(); em.close()


@pytest.mark.parametrize("arch", ["genset+battery", "battery+grid", "genset+battery+grid"])
def test_direct_rows_discrete_steps(arch, device):
    from pymgrid_amd import DiscreteBatchedMicrogridEnv
    N, T, H = 517, 90, 24
    f, m = _pair(N, T, arch, H, device, seed=11)
    ef = DiscreteBatchedMicrogridEnv(f, obs_direct=True, remove_redundant_gensets=False)
    em = DiscreteBatchedMicrogridEnv(m, obs_prefetch=0, remove_redundant_gensets=False)
    g = torch.Generator(device=device); g.manual_seed(3)
    assert torch.equal(ef.reset(), em.reset())
    for k in range(T):                                   # ... to the last row of the series: done, and a row of pure padding
        ids = torch.randint(0, ef.action_space.n, (N,), dtype=torch.int32, device=device, generator=g)
        (o1, r1, d1, _), (o2, r2, d2, _) = ef.step(ids), em.step(ids)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), k
    assert bool(d1.all())
    ef.close(); em.close()


def test_direct_rows_fleet(device):
    """Three layouts through ONE mgx_fleet_step call (one fleet_rows_kernel launch, rotating row buffers) == the three
    materialised envs stepped one by one."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.hetero import BucketedFleet
    archs = ("genset+battery", "battery+grid", "genset+battery+grid")
    T, H = 80, 24
    pairs = [_pair(700 + 19 * k, T, a, H, device, seed=43 + k) for k, a in enumerate(archs)]
    fleet = BucketedFleet.from_batches([f for f, _ in pairs], obs_direct=True, reuse_outputs=4)
    assert fleet.fused
    twins = [BatchedMicrogridEnv(m, obs_prefetch=0) for _, m in pairs]
    g = torch.Generator(device=device); g.manual_seed(5)
    o1 = fleet.reset()
    for j, e in enumerate(twins):
        assert torch.equal(o1[j], e.reset())
    for k in range(T - 1):
        acts = [torch.rand(e.n_grids, e.layout.action_dim, dtype=torch.float64, device=device, generator=g) for e in twins]
        obs, rew, done, _ = fleet.step(acts)
        for j, e in enumerate(twins):
            o2, r2, d2, _ = e.step(acts[j])
            assert torch.equal(obs[j], o2), (k, j)
            assert torch.equal(rew[j], r2) and torch.equal(done[j], d2)
    fleet.close()
    for e in twins:
        e.close()
