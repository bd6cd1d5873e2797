"""Column-major ring blocks (mgx_set_ring_layout / obs_layout="columns"): the observation a step returns is the same [N, D] matrix,
stored [D, pitch] -- a view with strides (1, pitch).  Every value == the row-major rings', through ring changes, resets in the middle
of a ring, the end-of-series padding, float32 rows, discrete steps and a three-layout fleet."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arch,H,K,dtype,discrete", [("genset+battery+grid", 24, 8, torch.float64, False), ("genset+battery", 24, 5, torch.float32, False),
                                                     ("battery+grid", 7, 3, torch.float64, True), ("genset+battery+grid", 23, 16, torch.float32, True)])
def test_column_major_rings_equal_row_major_rings(arch, H, K, dtype, discrete, device):
    from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T = 1003, 130
    cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
    kw = dict(remove_redundant_gensets=False) if discrete else {}

    def make(layout):
        b = generate(N, n_steps=T, seed=5, arch=arch, horizon=H, device=device, mixed_timers=True, series="factorised")
        return cls(b, obs_dtype=dtype, obs_prefetch=K, obs_layout=layout, **kw)
    rows, cols = make("rows"), make("columns")
    g = torch.Generator(device=device); g.manual_seed(2)
    for start, n_steps in ((0, 3 * K + 2), (11, K + 1), (T - H - 5, H + 4)):
        o1, o2 = rows.reset(start), cols.reset(start)
        assert o2.shape == o1.shape and o2.stride() == (1, (N + 31) // 32 * 32) and torch.equal(o1, o2), (start, "reset")
        for k in range(min(n_steps, T - start - 1)):
            a = rows.sample_action(generator=g)
            (o1, r1, d1, _), (o2, r2, d2, _) = rows.step(a), cols.step(a)
            assert torch.equal(o1, o2), (start, k)
            assert torch.equal(r1, r2) and torch.equal(d1, d2)
        assert torch.equal(o2.contiguous(), o1)
    rows.close(); cols.close()


def test_column_major_fleet(device):
    from pymgrid_amd.generator import generate
    from pymgrid_amd.hetero import BucketedFleet
    archs = ("genset+battery", "battery+grid", "genset+battery+grid")

    def fleet(layout):
        batches = [generate(700 + 13 * k, n_steps=100, seed=43 + k, arch=a, horizon=24, device=device, series="factorised")
                   for k, a in enumerate(archs)]
        return BucketedFleet.from_batches(batches, obs_prefetch=8, reuse_outputs=24, obs_layout=layout)
    rows, cols = fleet("rows"), fleet("columns")
    assert rows.fused and cols.fused
    g = torch.Generator(device=device); g.manual_seed(3)
    o1, o2 = rows.reset(), cols.reset()
    assert all(torch.equal(a, b) for a, b in zip(o1, o2))
    for k in range(40):
        acts = [torch.rand(e.n_grids, e.layout.action_dim, dtype=torch.float64, device=device, generator=g) for e in rows.envs]
        (o1, r1, d1, _), (o2, r2, d2, _) = rows.step(acts), cols.step(acts)
        for j in range(3):
            assert torch.equal(o1[j], o2[j]), (k, j)
            assert torch.equal(r1[j], r2[j]) and torch.equal(d1[j], d2[j])
    rows.close(); cols.close()


def test_automatic_layout_follows_the_episode_mode(device):
    """obs_layout=None (the default): column-major blocks while the batch walks in lock-step, row-major rings as soon as per-grid
    episodes begin (restarted grids are patched into row-major rings), column-major again at the next lock-step reset -- with the
    same observations as an env pinned to row-major rings throughout."""
    from pymgrid_amd import BatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T, K, H = 777, 140, 6, 12

    def make(layout):
        b = generate(N, n_steps=T, seed=8, arch="genset+battery+grid", horizon=H, device=device, mixed_timers=True, series="factorised")
        return BatchedMicrogridEnv(b, obs_prefetch=K, obs_layout=layout)
    auto, rows = make(None), make("rows")
    pitch = (N + 31) // 32 * 32
    g = torch.Generator(device=device); g.manual_seed(4)

    def walk(n):
        for k in range(n):
            a = rows.sample_action(generator=g)
            (o1, r1, d1, _), (o2, r2, d2, _) = rows.step(a), auto.step(a)
            assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), k
        return o2
    o1, o2 = rows.reset(), auto.reset()
    assert o2.stride() == (1, pitch) and o1.is_contiguous() and torch.equal(o1, o2)
    assert walk(2 * K + 1).stride() == (1, pitch)
    starts = torch.randint(0, T - 30, (N,), dtype=torch.int32, device=device, generator=g)
    o1, o2 = rows.reset_windows(starts, max_length=20, rolling="inplace"), auto.reset_windows(starts, max_length=20, rolling="inplace")
    assert o2.is_contiguous() and torch.equal(o1, o2)
    assert walk(K + 2).is_contiguous()
    mask = torch.rand(N, device=device, generator=g) < 0.3
    new = torch.randint(0, T - 30, (N,), dtype=torch.int32, device=device, generator=g)
    assert torch.equal(rows.reset_grids(mask, new), auto.reset_grids(mask, new))
    walk(3)
    o1, o2 = rows.reset(3), auto.reset(3)
    assert o2.stride() == (1, pitch) and torch.equal(o1, o2)
    assert walk(K + 3).stride() == (1, pitch)
    with pytest.raises(Exception):              # a pinned column-major env refuses per-grid episodes instead of switching
        make("columns").reset_windows(starts, max_length=20, rolling="inplace")
    rows.close(); auto.close()
