"""Generator rules pinned against the REAL MicrogridGenerator: tests/golden/generator_rules.npz holds, for 96
microgrids the reference generated (seeds 42 and 7), the random draws its code made and everything it derived from them
(tests/golden/make_generator_goldens.py).  Fed the same draws, pymgrid_amd.generator's rule functions must give the same
numbers, bit for bit: sizes, converted module parameters, scaled series, tariffs, co2 series, weak-grid outage series."""
import json

import numpy as np
import pytest
import torch

from conftest import golden


@pytest.fixture(scope="module")
def cases():
    z = golden("generator_rules.npz")
    return z, json.loads(str(z["meta"]))


def _draws(metas):
    keys = ("size_load", "load_file", "pv_pen", "bat_hours", "pv_file", "soc0_randn", "weak", "tariff", "outage_randn",
            "outage_dur", "co2_file")
    return {k: np.array([d[k] for _, d in metas]) for k in keys}


def test_sizing_rules_equal_the_reference(cases):
    from pymgrid_amd import generator as gen
    z, metas = cases
    r = gen.derive(_draws(metas))
    n_gen = n_grid = n_weak = 0
    for j, (key, d) in enumerate(metas):
        p = d["params"]
        assert np.around(r["pv_size"][j], 2) == d["pv_rated"], key                       # df_parameters['PV_rated_power']
        assert r["bat_max_capacity"][j] == d["battery_capacity"] == p["battery"]["max_capacity"], key
        assert r["bat_power"][j] == d["battery_power"] == p["battery"]["max_charge"] == p["battery"]["max_discharge"], key
        assert r["bat_min_capacity"][j] == p["battery"]["min_capacity"], key
        assert r["soc0"][j] == d["battery_soc_0"] == p["battery"]["soc"], key
        assert r["soc0"][j] * r["bat_max_capacity"][j] == p["battery"]["charge"], key
        assert p["battery"]["efficiency"] == 0.9 and p["battery"]["battery_cost_cycle"] == 0.02
        assert p["unbalanced"] == dict(loss_load_cost=10.0, overgeneration_cost=1.0)
        assert (p["horizon"], p["initial_step"], p["final_step"]) == (23, 0, 8760)      # a fresh microgrid runs to the end of its series
        if d["genset"]:
            n_gen += 1
            g = p["genset"]
            assert r["gen_rated"][j] == d["genset_rated"], key
            assert r["gen_running_min"][j] == g["running_min_production"] and r["gen_running_max"][j] == g["running_max_production"], key
            assert (g["genset_cost"], g["co2_per_unit"], g["cost_per_unit_co2"], g["start_up_time"], g["wind_down_time"]) == \
                (0.4, 2.0, 0.1, 0, 0) and g["status"] == [1, 1, 0, 0]
        if d["grid"]:
            n_grid += 1
            assert r["grid_power"][j] == d["grid_power"] == p["grid"]["max_import"] == p["grid"]["max_export"], key
            assert p["grid"]["cost_per_unit_co2"] == 0.1
            n_weak += d["weak"]
    assert n_gen > 30 and n_grid > 30 and n_weak > 10


def test_architecture_rule_equals_the_reference(cases):
    from pymgrid_amd import generator as gen
    _, metas = cases
    arch = gen.architecture_of(dict(bin_rand=np.array([d["bin_rand"] for _, d in metas]),
                                    weak=np.array([d["weak"] for _, d in metas])))
    for a, (key, d) in zip(arch, metas):
        want = "genset+battery+grid" if d["genset"] and d["grid"] else ("battery+grid" if d["grid"] else "genset+battery")
        assert a == want, key


def test_series_rules_equal_the_reference(cases):
    from pymgrid_amd import generator as gen
    z, metas = cases
    P = gen.base_profiles()
    r = gen.derive(_draws(metas), P)
    rows = z["sample_rows"]
    checked = 0
    for j, (key, d) in enumerate(metas):
        load = P["load"][:, d["load_file"]] * r["load_ratio"][j]
        pv = P["pv"][:, d["pv_file"]] * r["pv_ratio"][j]
        assert np.array_equal(-np.abs(load)[rows], z[f"{key}_load_sample"]) and -np.abs(load).sum() == z[f"{key}_load_sum"], key
        assert np.array_equal(np.abs(pv)[rows], z[f"{key}_pv_sample"]) and np.abs(pv).sum() == z[f"{key}_pv_sum"], key
        if d["grid"]:
            assert np.array_equal(gen.electricity_tariff(d["tariff"])[:48], z[f"{key}_price_day0"]), key
            assert np.array_equal(P["co2"][rows, d["co2_file"]], z[f"{key}_co2_sample"]), key
            status = np.unpackbits(z[f"{key}_status"])[:8760].astype(np.float64)
            if not d["weak"]:
                assert status.all(), key
            elif f"{key}_outage_uniforms" in z.files:                    # the reference's own uniform draws as input
                mine = gen.weak_grid_profile(z[f"{key}_outage_uniforms"], r["outage_per_day"][j], d["outage_dur"])
                assert np.array_equal(mine, status), key
                checked += 1
    assert checked == 8


def test_weak_grid_rule_edge_cases():
    from pymgrid_amd.generator import weak_grid_profile
    u = np.full(21, 0.9)
    u[[0, 5, 20]] = 0.0                                   # outages drawn at rows 0, 5 and at the extra row 20
    s = weak_grid_profile(u, 24 * 0.5, 3)                 # threshold 0.5, duration 3: back-fill covers 2 rows before each
    assert s.tolist() == [0, 1, 1, 0, 0, 0] + [1] * 12 + [0, 0]          # row 0 only by its own draw; rows 18, 19 from row 20
    assert weak_grid_profile(np.full(11, 0.9), -1.0, 7).all()           # a negative outage rate (randn * 3/4 + 0.25 < 0): none
    u = np.full(11, 0.9); u[2] = 0.0
    assert weak_grid_profile(u, 24 * 0.5, 7).tolist() == [1, 0, 0] + [1] * 7      # "if i - j > 0": row 0 is never back-filled
    assert weak_grid_profile(u, 24 * 0.5, 1).tolist() == [1, 1, 0] + [1] * 7      # duration 1: no back-fill at all


def test_host_philox_stream_is_a_uniform_stream():
    from pymgrid_amd.generator import synth_uniform_host
    u = synth_uniform_host(42, np.arange(2000)[:, None], np.arange(500)[None, :])
    assert u.shape == (2000, 500) and 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    assert len(np.unique(u)) == u.size                                    # counter-based: no repeats across (grid, row)
    assert not np.array_equal(u, synth_uniform_host(43, np.arange(2000)[:, None], np.arange(500)[None, :]))


def test_generate_on_cpu_follows_the_rules_and_is_shard_invariant():
    from pymgrid_amd import generator as gen
    full = gen.generate(96, n_steps=300, seed=11, arch="genset+battery+grid", device="cpu")
    halves = [gen.generate(96, n_steps=300, seed=11, arch="genset+battery+grid", device="cpu", rank=r, world=2) for r in (0, 1)]
    for k, v in full.cols.items():
        assert torch.equal(v, torch.cat([h.cols[k] for h in halves], dim=-1)), k
    D = gen.draw_scalars(np.arange(96), 11)
    r = gen.derive(D)
    P = gen.base_profiles()
    c = full.cols
    assert np.array_equal(c["bat_max_capacity"].numpy(), r["bat_max_capacity"]) and np.array_equal(c["gen_running_max"].numpy(), r["gen_running_max"])
    assert np.array_equal(c["load_ts"].numpy(), -np.abs(P["load"][:300, D["load_file"]] * r["load_ratio"][None, :]))
    assert np.array_equal(c["grid_ts"][:, 0].numpy(), np.where(D["tariff"][None, :] == 1, gen.electricity_tariff(1, 300)[:, None],
                                                              gen.electricity_tariff(2, 300)[:, None]))
    status = c["grid_ts"][:, 3].numpy()
    weak = D["weak"].astype(bool)
    assert status[:, ~weak].all() and ((status == 0) | (status == 1)).all()
    # fleet split: MicrogridGenerator's architecture mix, every grid exactly once
    fleet = gen.generate_fleet(300, n_steps=48, seed=5, device="cpu")
    idx = np.sort(np.concatenate([i for _, i in fleet.values()]))
    assert np.array_equal(idx, np.arange(300)) and set(fleet) == {"genset+battery", "battery+grid", "genset+battery+grid"}
    arch = gen.architecture_of(gen.draw_scalars(np.arange(300), 5))
    for name, (b, i) in fleet.items():
        assert (arch[i] == name).all() and b.layout.n_grids == len(i)


@pytest.mark.gpu
def test_device_synthesis_equals_the_host_rules(device):
    """mgx_synthesize_series (HIP: base profile x ratio, tariffs, co2, Philox outage uniforms + back-fill) == the numpy rule
    functions on the same draws, bit for bit; contiguous shards and scattered selections."""
    from pymgrid_amd import generator as gen
    for arch in ("genset+battery", "genset+battery+grid"):
        dev_b = gen.generate(1030, n_steps=400, seed=3, arch=arch, device=device, rank=1, world=2, mixed_timers=True)
        cpu_b = gen.generate(1030, n_steps=400, seed=3, arch=arch, device="cpu", rank=1, world=2, mixed_timers=True)
        assert set(dev_b.cols) == set(cpu_b.cols)
        for k, v in cpu_b.cols.items():
            assert torch.equal(dev_b.cols[k].cpu(), v), (arch, k)
    sel = np.sort(np.random.RandomState(0).choice(5000, 700, replace=False))
    a = gen.generate(5000, n_steps=8760, seed=9, arch="genset+battery+grid", device=device, select=sel)
    b = gen.generate(5000, n_steps=8760, seed=9, arch="genset+battery+grid", device="cpu", select=sel)
    for k, v in b.cols.items():
        assert torch.equal(a.cols[k].cpu(), v), k
    st = a.cols["grid_ts"][:, 3]
    weak = gen.draw_scalars(sel, 9)["weak"].astype(bool)
    assert bool(st[:, torch.as_tensor(~weak, device=device)].all()) and float(st[:, torch.as_tensor(weak, device=device)].mean()) < 1.0


@pytest.mark.gpu
def test_device_generator_equals_the_host_rules(device):
    """mgx_generate_columns (HIP: counter-based Philox draws, randint / Irwin-Hall normal, the sizing rules incl. the mean of the
    scaled load in numpy's pairwise order, bounds, packed genset words, architecture codes) == draw_scalars + derive in numpy,
    bit for bit, for contiguous shards at large global offsets and scattered selections."""
    from pymgrid_amd import generator as gen
    for N, T, seed, mixed, g0 in ((1029, 300, 11, True, 0), (4097, 8760, 42, False, 875_000), (300, 64, 7, True, (1 << 33) + 5)):
        G = gen.generate_columns_device(device, N, T, seed, mixed, g0, draws=True)
        H, d, r = gen.generate_columns_host(N, T, seed, mixed, np.arange(g0, g0 + N))
        for k, v in H.items():
            assert torch.equal(G[k].cpu(), v), (N, k)
        assert np.array_equal(G["d_bin_rand"].cpu().numpy(), d["bin_rand"]) and np.array_equal(G["d_size_load"].cpu().numpy(), d["size_load"])
        assert np.array_equal(G["d_soc0_normal"].cpu().numpy(), d["soc0_randn"]) and np.array_equal(G["d_outage_normal"].cpu().numpy(), d["outage_randn"])
        assert np.array_equal(G["d_su"].cpu().numpy(), d["su"]) and np.array_equal(G["d_bat_hours"].cpu().numpy(), d["bat_hours"])
    sel = np.sort(np.random.RandomState(1).choice(1_000_000, 1500, replace=False))
    G = gen.generate_columns_device(device, len(sel), 200, 3, True, 0, torch.as_tensor(sel, device=device))
    H, _, _ = gen.generate_columns_host(len(sel), 200, 3, True, sel)
    for k, v in H.items():
        assert torch.equal(G[k].cpu(), v), k
    # the draws look like what they stand for
    G = gen.generate_columns_device(device, 200_000, 64, 5, True, 0, draws=True)
    z = G["d_soc0_normal"].cpu().numpy()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs(np.mean(z < -1.0) - 0.1587) < 0.005
    assert abs(G["d_bin_rand"].mean().item() - 0.5) < 0.005 and set(np.unique(G["d_bat_hours"].cpu().numpy())) == {3, 4, 5}
    code = G["arch"].cpu().numpy()
    assert abs(np.mean(code == 0) - 0.33) < 0.01 and abs(np.mean(code == 1) - 0.165) < 0.01      # half of the grid-only draws are weak


@pytest.mark.gpu
def test_generating_a_million_grids_builds_no_host_array_of_that_size(device):
    """The CUDA path of generate(): draws, sizing, series factors and outage words are all made on the device -- the host holds
    the base profile tables and a few per-profile numbers, nothing that grows with N."""
    import tracemalloc
    from pymgrid_amd import generator as gen
    gen.base_profiles()                                       # (the package data: loaded once, not part of a generate call)
    gen.generate(1024, n_steps=64, seed=42, arch="genset+battery+grid", device=device, series="factorised")   # warm-up: lazy imports
    N = 1_000_000
    tracemalloc.start()
    b = gen.generate(N, n_steps=64, seed=42, arch="genset+battery+grid", device=device, series="factorised", mixed_timers=True)
    f = gen.generate_fleet(N, n_steps=64, seed=42, device=device, rank=3, world=8, series="factorised")
    _, peak = tracemalloc.get_traced_memory()
    tracemalloc.stop()
    assert b.layout.n_grids == N and sum(len(i) for _, i in f.values()) == N // 8
    assert peak < N, f"host allocations peaked at {peak} bytes: an array of {N} grids was built on the host"


def test_irwin_hall_normals_stand_in_for_randn_statistically():
    """The two normals of MicrogridGenerator -- soc_0 = min(max(randn(), soc_min), soc_max) (_get_battery, MicrogridGenerator.py:
    230-243) and outage_per_day = randn() * 3/4 + 0.25 (_get_grid, :288-292, which sets the rate of _generate_weak_grid_profile,
    :321-340) -- are drawn here as the sum of 12 uniforms - 6 (Irwin-Hall; host and device then agree bit for bit without
    sharing a libm).  That is STATISTICAL parity with np.random.randn, stated and bounded here: over 400 000 grids the draws'
    CDF stays within 0.004 of the Gaussian's (Irwin-Hall(12)'s own worst deviation is ~0.002), their first four moments
    match, and what the generator's rules make of them -- the clip masses of soc_0 at 0.2 and 1.0, the share of grids without
    any outage, the mean outage rate -- sit within four binomial standard errors (+ the Irwin-Hall bias) of the Gaussian
    figures.  What is NOT reproduced: tails beyond 6 sigma (probability 2e-9 per draw)."""
    from scipy import stats
    from pymgrid_amd.generator import draw_scalars
    n = 400_000
    d = draw_scalars(np.arange(n, dtype=np.int64), seed=42)
    for name in ("soc0_randn", "outage_randn"):
        z = d[name]
        assert np.abs(z).max() <= 6.0
        D = stats.kstest(z, "norm").statistic
        assert D < 0.004, (name, D)
        assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1.0) < 0.01
        assert abs(stats.skew(z)) < 0.02 and abs(stats.kurtosis(z) + 0.1) < 0.03        # Irwin-Hall(12): excess kurtosis -0.1
    assert abs(np.corrcoef(d["soc0_randn"], d["outage_randn"])[0, 1]) < 4 / np.sqrt(n)   # separate Philox counters
    # the rules applied to the draws
    soc0 = np.minimum(np.maximum(d["soc0_randn"], 0.2), 1.0)
    for mass, p in (((soc0 == 0.2).mean(), stats.norm.cdf(0.2)), ((soc0 == 1.0).mean(), stats.norm.sf(1.0))):
        assert abs(mass - p) < 4 * np.sqrt(p * (1 - p) / n) + 0.003, (mass, p)
    inner = soc0[(soc0 > 0.2) & (soc0 < 1.0)]
    want_mean = (stats.norm.pdf(0.2) - stats.norm.pdf(1.0)) / (stats.norm.cdf(1.0) - stats.norm.cdf(0.2))   # truncated-normal mean
    assert abs(inner.mean() - want_mean) < 0.003
    rate = np.clip(d["outage_randn"] * 3 / 4 + 0.25, 0.0, None) / 24                    # outages per row (negative = none)
    p_none = stats.norm.cdf(-1 / 3)                                                      # 0.75 z + 0.25 <= 0
    assert abs((rate == 0).mean() - p_none) < 4 * np.sqrt(p_none * (1 - p_none) / n) + 0.003
    want_rate = (0.75 * stats.norm.pdf(-1 / 3) + 0.25 * stats.norm.sf(-1 / 3)) / 24      # E[max(0, 0.75 Z + 0.25)] / 24
    assert abs(rate.mean() - want_rate) < 1e-4
