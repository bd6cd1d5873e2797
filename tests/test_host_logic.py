"""CPU-only tests: host-side packing / layout logic, the exported C ABI, the priority-list enumeration, the
generator's shard invariance.  No compute call is made without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


def test_c_abi_exports_every_declared_symbol():
    """libmgx.so loads on a GPU-less host and exports every function include/mgx.h declares."""
    from pymgrid_amd import _lib
    _lib.build()
    L = C.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "mgx.h")).read()
    declared = set(re.findall(r"\b(mgx_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None
    lib = _lib.lib()
    assert lib.mgx_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define MGX_ABI_VERSION (\d+)", header).group(1))
    assert C.sizeof(_lib.Layout) == 16 * 4
    n_ptr = len(_lib.COLUMN_NAMES)
    assert C.sizeof(_lib.Columns) == 8 + 8 * n_ptr
    # every column of the C struct, in order
    fields = re.search(r"typedef struct mgx_columns \{(.*?)\} mgx_columns;", header, re.S).group(1)
    fields = re.sub(r"/\*.*?\*/", "", fields, flags=re.S)
    names = re.findall(r"\*\s*([a-z_0-9]+)\s*[,;]", fields)
    assert tuple(names) == _lib.COLUMN_NAMES


def test_engine_fails_loudly_without_a_gpu(pymgrid25):
    """No silent CPU fallback: on a host without a HIP device creating an engine is an error."""
    from pymgrid_amd import MgxError, MicrogridBatch, StepEngine, _lib
    b = MicrogridBatch.from_grids([pymgrid25[2]], device="cpu")
    with pytest.raises(MgxError):
        StepEngine(b)
    if not torch.cuda.is_available():
        h = C.c_void_p()
        L, cols = b.c_layout(), b.c_columns()
        rc = _lib.lib().mgx_create(C.byref(L), C.byref(cols), C.byref(h))
        assert rc == _lib.MGX_ERR_DEVICE and not h.value
        assert b"no HIP device" in _lib.lib().mgx_last_error()
    bad = b.c_layout(); bad.struct_size = 12
    h = C.c_void_p()
    assert _lib.lib().mgx_create(C.byref(bad), C.byref(b.c_columns()), C.byref(h)) == _lib.MGX_ERR_INVALID


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pymgrid_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                assert "mgx_oracle" not in text, f


def test_pack_grids_and_layout(pymgrid25):
    from pymgrid_amd import pack_grids, unpack_status
    from pymgrid_amd.scenario import bucket_by_layout
    buckets = bucket_by_layout(pymgrid25)
    assert sorted(len(v) for v in buckets.values()) == [7, 8, 10]        # SURVEY App. B architectures
    tmpl4 = [pymgrid25[n] for n in (2, 3, 5, 7, 15, 17, 19, 20, 21, 23)]
    A, L = pack_grids(tmpl4)
    assert (L.n_grids, L.n_steps, L.horizon, L.final_step) == (10, 8760, 23, 8759)
    assert L.action_dim == 3 and L.obs_dim == 24 + 24 + 4 + 2 and len(L.log_names) == 23
    assert A["load_ts"].shape == (8760, 10) and (A["load_ts"] <= 0).all() and (A["pv_ts"] >= 0).all()
    assert np.array_equal(A["load_lo"], A["load_ts"].min(0)) and (A["load_hi"] == 0).all()
    assert np.array_equal(unpack_status(A["gen_status"]), np.tile([1, 1, 0, 0], (10, 1)))
    assert (A["gen_times"] == 0).all()
    # 189 algorithmic bytes per env-step for the Template-4 core mode
    from pymgrid_amd import BatchLayout
    t4 = BatchLayout(n_grids=1, n_steps=10, has_genset=True, has_battery=True, has_grid=False)
    assert t4.bytes_per_step() == 189
    assert t4.bytes_per_step(log=True, obs=True) == 189 + 8 * 23 + 8 + 8 * 8 + 16
    assert t4.bytes_fused(64) == 140 + 64 * 57                        # params 108 + state 12 r / 20 w; 57 B streamed per step
    full = BatchLayout(n_grids=1, n_steps=10, horizon=24, has_genset=True, has_battery=True, has_grid=True)
    assert full.action_dim == 4 and full.obs_dim == 156                  # D = 156 with grid, H = 24
    with pytest.raises(ValueError):
        pack_grids([pymgrid25[0], pymgrid25[1]])                         # different module sets
    g = dict(pymgrid25[2]); g["battery"] = dict(g["battery"], efficiency=1.5)
    with pytest.raises(ValueError):
        pack_grids([g])
    g = dict(pymgrid25[2]); g["genset"] = dict(g["genset"], start_up_time=300)
    with pytest.raises(ValueError):
        pack_grids([g])


def test_battery_and_genset_initial_state_rules():
    """BatteryModule._init_battery (battery_module.py:96-106) and GensetModule initial status (:91-92,216-227)."""
    from pymgrid_amd import pack_grids, unpack_status
    base = dict(load_ts=np.ones(10), pv_ts=np.ones(10), final_step=10, horizon=0,
                unbalanced=dict(loss_load_cost=10.0, overgeneration_cost=2.0))
    bat = dict(min_capacity=10.0, max_capacity=100.0, max_charge=50.0, max_discharge=50.0, efficiency=0.9,
               battery_cost_cycle=0.02)
    A, _ = pack_grids([dict(base, battery=dict(bat, init_soc=0.3)), dict(base, battery=dict(bat, init_charge=45.0))])
    assert np.array_equal(A["charge"], [0.3 * 100.0, 45.0]) and np.array_equal(A["soc"], [0.3, 45.0 / 100.0])
    with pytest.raises(ValueError):
        pack_grids([dict(base, battery=bat)])
    gen = dict(running_min_production=10.0, running_max_production=50.0, genset_cost=0.5, start_up_time=2,
               wind_down_time=3)
    A, _ = pack_grids([dict(base, battery=dict(bat, init_soc=0.5), genset=dict(gen, init_start_up=True)),
                       dict(base, battery=dict(bat, init_soc=0.5), genset=dict(gen, init_start_up=False))])
    assert np.array_equal(unpack_status(A["gen_status"]), [[1, 1, 0, 3], [0, 0, 2, 0]])
    assert np.array_equal(A["gen_times"], [2 | (3 << 16)] * 2)


def test_priority_lists_match_reference_enumeration(pymgrid25):
    """get_priority_lists reproduces DiscreteMicrogridEnv.actions_list of every scenario (goldens G3)."""
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    from pymgrid_amd.scenario import architecture
    z = golden("discrete.npz")
    for n, p in enumerate(pymgrid25):
        a = architecture(p)
        redundant = "genset" in a and p["genset"]["running_min_production"] == 0
        pls = get_priority_lists("genset" in a, "battery" in a, "grid" in a, redundant)
        assert np.array_equal(table_array(pls), z[f"s{n}_table"][:, :3]), n
    assert len(get_priority_lists(True, True, True)) == 12               # n_modules! * 2^n_gensets
    assert len(get_priority_lists(True, True, False)) == 4
    assert len(get_priority_lists(False, True, True)) == 2
    assert len(get_priority_lists(True, True, True, remove_redundant_gensets=True)) == 6


def test_spaces():
    from pymgrid_amd.spaces import Box, Discrete
    b = Box(0.0, 1.0, shape=(3,))
    assert b.contains(np.array([0.0, 0.5, 1.0])) and not b.contains(np.array([0.0, 0.5, 1.1]))
    assert b.sample().shape == (3,)
    d = Discrete(4)
    assert 3 in d and 4 not in d and -1 not in d and 0 <= d.sample() < 4


def test_generator_is_shard_invariant():
    """Rank r of W draws exactly columns [r*N/W, (r+1)*N/W) of the global batch."""
    from pymgrid_amd.generator import generate
    for arch in ("genset+battery", "genset+battery+grid", "battery+grid"):    # incl. the weak-grid outage draws
        full = generate(96, n_steps=300, seed=5, arch=arch, device="cpu", mixed_timers=True)
        parts = [generate(96, n_steps=300, seed=5, arch=arch, device="cpu", mixed_timers=True, rank=r, world=3)
                 for r in range(3)]
        for name, t in full.cols.items():
            cat = torch.cat([p.cols[name] for p in parts], dim=-1)
            assert torch.equal(cat, t), (arch, name)
        assert parts[0].layout.n_grids == 32
    full = generate(96, n_steps=30, seed=5, arch="genset+battery", device="cpu", mixed_timers=True)
    # sizing rules (MicrogridGenerator.py:214-386)
    c = full.cols
    assert torch.equal(c["bat_min_capacity"], 0.2 * c["bat_max_capacity"])
    assert torch.equal(c["bat_max_charge"], torch.ceil(c["bat_max_capacity"] / 4))
    assert (c["gen_running_max"] / c["gen_running_min"] - 18.0).abs().max() < 1e-9
    assert (c["soc"] >= 0.2).all() and (c["soc"] <= 1.0).all()
    g = generate(30, n_steps=48, seed=1, arch="genset+battery+grid", device="cpu")
    assert g.cols["grid_ts"].shape == (48, 4, 30)
    st = g.cols["grid_ts"][:, 3]
    assert ((st == 0) | (st == 1)).all()


def test_header_is_plain_c(tmp_path):
    """include/mgx.h is a C header (the boundary is a C ABI): gcc -std=c99 -pedantic accepts it on its own."""
    import subprocess
    src = tmp_path / "chk.c"
    src.write_text('#include "mgx.h"\nint main(void) { mgx_layout l; l.struct_size = (int)sizeof l; return l.struct_size ? 0 : MGX_ABI_VERSION; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                    "-fsyntax-only", str(src)], check=True)


def test_create_argument_checks_of_round_3_without_a_gpu(pymgrid25):
    """mgx_create validates layouts / columns before it looks for a device: the round-3 additions (factorised series, flat_order,
    uniform columns) refuse inconsistent input with a message -- checked here on a GPU-less host."""
    from pymgrid_amd import MicrogridBatch, _lib
    from pymgrid_amd.generator import generate
    L = _lib.lib()

    def create(batch, edit_layout=None, edit_cols=None):
        lay, cols = batch.c_layout(), batch.c_columns()
        if edit_layout:
            edit_layout(lay)
        if edit_cols:
            edit_cols(cols)
        h = C.c_void_p()
        rc = L.mgx_create(C.byref(lay), C.byref(cols), C.byref(h))
        assert not h.value or rc == 0
        if h.value:
            L.mgx_destroy(h)
        return rc, L.mgx_last_error().decode()

    bf = generate(8, n_steps=30, seed=1, arch="genset+battery+grid", device="cpu", series="factorised")
    rc, msg = create(bf, edit_cols=lambda c: setattr(c, "load_ratio", None))
    assert rc == _lib.MGX_ERR_INVALID and "load_ratio" in msg
    rc, msg = create(bf, edit_cols=lambda c: setattr(c, "tariff", None))
    assert rc == _lib.MGX_ERR_INVALID and "tariff" in msg
    rc, msg = create(bf, edit_layout=lambda l: setattr(l, "flat_order", 7))
    assert rc == _lib.MGX_ERR_INVALID and "flat_order" in msg
    rc, msg = create(bf, edit_cols=lambda c: setattr(c, "uniform_mask", 1 << 20))
    assert rc == _lib.MGX_ERR_INVALID and "uniform_mask" in msg
    # several modules of a kind: the general kernels read materialised series, native row order, full columns only
    two = dict(pymgrid25[2]); two["genset"] = [pymgrid25[2]["genset"]] * 2
    bm = MicrogridBatch.from_grids([two], device="cpu")
    for edit_l, edit_c, what in ((lambda l: setattr(l, "flat_order", 1), None, "flat_order"),
                                 (None, lambda c: setattr(c, "uniform_mask", 1), "uniform_mask"),
                                 (None, lambda c: setattr(c, "base_load", c.load_ts), "factorised")):
        rc, msg = create(bm, edit_l, edit_c)
        assert rc == _lib.MGX_ERR_UNSUPPORTED and what in msg, (rc, msg)
    if not torch.cuda.is_available():          # everything consistent: only the device is missing
        rc, msg = create(bf)
        assert rc == _lib.MGX_ERR_DEVICE and "no HIP device" in msg
    # the host side refuses the same things earlier
    with pytest.raises(ValueError):
        MicrogridBatch(bm.layout, dict(bm.cols, base_load=torch.zeros(8760, 8, dtype=torch.float64)))
    from dataclasses import replace
    with pytest.raises(ValueError):
        replace(bf.layout, flat_order="alphabetical")


def test_episode_entry_points_refuse_null_handles_without_a_gpu():
    """The ABI v6 entry points (in-place episodes) check their arguments before they touch a device."""
    from pymgrid_amd import _lib
    L = _lib.lib()
    assert L.mgx_reset_episodes(None, None, None, 5, None, None, None, None) == _lib.MGX_ERR_INVALID
    assert b"NULL" in L.mgx_last_error()
    assert L.mgx_set_auto_reset(None, 1, 0, 0, None, None, None) == _lib.MGX_ERR_INVALID
    assert L.mgx_set_final_obs(None, None) == _lib.MGX_ERR_INVALID
    assert b"mgx_set_final_obs" in L.mgx_last_error()


def test_hot_kernels_use_no_scratch_memory():
    """The backend's per-kernel figures of the build (hipcc -Rpass-analysis=kernel-resource-usage, kept by _lib.build): only the
    general multi-instance kernels (LDS lists + a per-lane stack) may use scratch memory.  A private segment on a stepping
    kernel costs launch time: 24 B of it once slowed every single step by 0.4-1 us before anyone noticed."""
    from pymgrid_amd import _lib
    _lib.build()
    usage = _lib.resource_usage()
    if usage is None:
        pytest.skip("libmgx.so was not built on this machine (no resource_usage.json beside the objects)")
    hot = ("step_kernel", "step_discrete_kernel", "step_k_kernel", "rollout_kernel", "fleet_step_kernel", "fleet_step_kernel_v", "fleet_step_kernel_vm", "observe_kernel",
           "step_multi_kernel", "step_k_multi_small_kernel", "rollout_multi_small_kernel", "step_lists_small_kernel",
           "obs_rows_wave_kernel", "obs_windows_k_kernel", "patch_windows_kernel", "expand_kernel", "check_kernel",
           "normalise_series_kernel", "gather_windows_kernel", "synthesize_series_kernel")
    seen = 0
    for name, u in usage.items():
        base = name.split("<")[0].split("::")[-1]
        if base == "step_multi_kernel" and name.rstrip().endswith(", true>"):
            continue                    # (the in-place-episode form of the general step: 36 B of private segment on some layouts; off the lock-step path)
        if base in hot:
            seen += 1
            assert u.get("scratch", 0) == 0 and u.get("vgpr_spill", 0) == 0, (name, u)
    assert seen > 50, seen                                  # every specialisation was looked at
