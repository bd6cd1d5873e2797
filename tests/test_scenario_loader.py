"""Scenario loader: the reference's !Microgrid YAML + csv.gz files -> parameter dicts.
Needs the reference's data directory, so it only runs in the build container."""
import os

import numpy as np
import pytest

REF_SCENARIOS = "/root/reference/src/pymgrid/data/scenario"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SCENARIOS), reason="reference data files not present")


def test_yaml_loader_equals_reference_loader(pymgrid25):
    """All 25 pymgrid25 scenarios: the parameter dict read from YAML + csv.gz by this repo's loader equals the one
    extracted from Microgrid.from_scenario(n) by the reference (fixture pymgrid25_inputs.npz), bit for bit."""
    from pymgrid_amd.scenario import from_scenario
    for n, ref in enumerate(pymgrid25):
        p = from_scenario(n, REF_SCENARIOS)
        order = p.pop("controllable_order")               # module-list order of the YAML: never grid before battery here
        assert p.pop("current_step") == p["initial_step"] == 0       # saved counter: kept apart from the constructor's initial_step
        assert not ("grid" in order and "battery" in order and order.index("grid") < order.index("battery")), n
        assert set(p) == set(ref), (n, set(p) ^ set(ref))
        for k, v in ref.items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(np.asarray(p[k]).reshape(v.shape), v), (n, k)
            elif isinstance(v, dict):
                assert p[k] == v, (n, k, p[k], v)
            else:
                assert p[k] == v, (n, k)


def test_state_checkpoint_roundtrip(tmp_path, pymgrid25):
    import torch
    from pymgrid_amd import MicrogridBatch
    from pymgrid_amd.scenario import load_state, save_state
    b = MicrogridBatch.from_grids([pymgrid25[2], pymgrid25[3]], device="cpu")
    b.cols["charge"] += 1.5
    save_state(b, 17, str(tmp_path / "ck.npz"))
    b2 = MicrogridBatch.from_grids([pymgrid25[2], pymgrid25[3]], device="cpu")
    assert load_state(b2, str(tmp_path / "ck.npz")) == 17
    for k in ("charge", "soc", "gen_status"):
        assert torch.equal(b.cols[k], b2.cols[k])



def test_dump_and_load_round_trip(pymgrid25, tmp_path):
    """dump_scenario_yaml -> load_scenario_yaml gives the parameter dict back, bit for bit (all 25 scenarios; one with the
    grid listed before the battery and a stepped state)."""
    from pymgrid_amd.scenario import dump_scenario_yaml, load_scenario_yaml
    cases = list(enumerate(pymgrid25))
    with_grid = next(q for q in pymgrid25 if q.get("grid") is not None and q.get("genset") is not None)
    odd = dict(with_grid); odd["controllable_order"] = ["grid", "battery"]; odd["initial_step"] = 5
    odd["current_step"] = 9                                # saved mid-episode: the counter is NOT the constructor's initial_step
    odd["battery"] = dict(odd["battery"], charge=odd["battery"]["max_capacity"] * 0.61803, soc=0.61803)
    cases.append(("odd", odd))
    for n, ref in cases:
        os.makedirs(tmp_path / f"mg_{n}", exist_ok=True)
        path = dump_scenario_yaml(ref, str(tmp_path / f"mg_{n}" / "microgrid.yaml"))
        p = load_scenario_yaml(path)
        order = p.pop("controllable_order")
        if n == "odd":
            assert order.index("grid") < order.index("battery")
            assert p["initial_step"] == 5 and p["current_step"] == 9
        else:
            assert p.pop("current_step") == p["initial_step"]
        for k, v in ref.items():
            if k == "controllable_order":
                continue
            if isinstance(v, np.ndarray):
                assert np.array_equal(np.asarray(p[k]).reshape(v.shape), v), (n, k)
            elif k == "battery":
                assert p[k]["charge"] == v["charge"] and abs(p[k]["soc"] - v["soc"]) < 1e-15, (n, k)
                assert {kk: vv for kk, vv in p[k].items() if kk not in ("charge", "soc")} == \
                    {kk: vv for kk, vv in v.items() if kk not in ("charge", "soc", "init_soc")}, (n, k)
            else:
                assert p[k] == v, (n, k, p[k], v)


def test_the_reference_loads_what_we_dump(pymgrid25, tmp_path):
    """Microgrid.load(<file written by dump_scenario_yaml>) in the real reference == the scenario it came from."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import _refenv
    _refenv.import_reference()
    import make_goldens as mg
    from pymgrid import Microgrid
    from pymgrid_amd.scenario import dump_scenario_yaml
    mid = dict(pymgrid25[3], initial_step=4, current_step=11)      # saved mid-episode
    os.makedirs(tmp_path / "mg_mid", exist_ok=True)
    with open(dump_scenario_yaml(mid, str(tmp_path / "mg_mid" / "microgrid.yaml"))) as fh:
        m = Microgrid.load(fh)
    load = mg.find(m, mg.LoadModule)[0]
    assert load.initial_step == 4 and load.current_step == 11       # the reference keeps the two apart the same way
    m.reset()
    assert load.current_step == 4
    for n in (0, 3, 7, 24):
        ref = pymgrid25[n]
        os.makedirs(tmp_path / f"mg_{n}", exist_ok=True)
        path = dump_scenario_yaml(ref, str(tmp_path / f"mg_{n}" / "microgrid.yaml"))
        with open(path) as fh:
            m = Microgrid.load(fh)
        back = mg.extract_params(m)
        assert set(back) == set(ref), (n, set(back) ^ set(ref))
        for k, v in ref.items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(np.asarray(back[k]).reshape(v.shape), v), (n, k)
            else:
                assert back[k] == v, (n, k, back[k], v)


def test_multi_instance_scenarios_round_trip(tmp_path):
    """Several gensets / batteries / grids / loads / pvs per microgrid: dump_scenario_yaml -> load_scenario_yaml gives the
    parameter dict back, and the real reference loads the same file into the same modules (count, order, parameters, state)."""
    from conftest import multi_cases
    from pymgrid_amd.batch import module_list
    from pymgrid_amd.scenario import bucket_by_layout, dump_scenario_yaml, load_scenario_yaml
    ref_ok = os.path.isdir("/root/reference/src/pymgrid")
    if ref_ok:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
        import _refenv
        _refenv.import_reference()
        from pymgrid import Microgrid
    all_p = []
    for ci, p, mt, z in multi_cases():
        all_p.append(p)
        os.makedirs(tmp_path / f"m{ci}", exist_ok=True)
        path = dump_scenario_yaml(p, str(tmp_path / f"m{ci}" / "microgrid.yaml"))
        q = load_scenario_yaml(path)
        # (series go through csv text and pandas' default float parser, as in the reference: the last bit may differ for
        #  arbitrary doubles -- the pymgrid25 series are short decimals and survive bit for bit, see the tests above)
        close = lambda a, b: np.allclose(a, b, rtol=1e-14, atol=0.0)
        assert close(np.asarray(q["load_ts"]).reshape(p["load_ts"].shape), -np.abs(p["load_ts"]))
        assert close(np.asarray(q["pv_ts"]).reshape(p["pv_ts"].shape), np.abs(p["pv_ts"]))
        for kind in ("genset", "battery", "grid"):
            a, b = module_list(p.get(kind)), module_list(q.get(kind))
            assert len(a) == len(b), (ci, kind)
            for x, y in zip(a, b):
                for key in ("running_max_production", "genset_cost", "start_up_time", "max_capacity", "efficiency", "max_import"):
                    if key in x:
                        assert y[key] == x[key], (ci, kind, key)
        for j, ts in enumerate(p["grid_ts"]):
            got = q["grid_ts"] if len(p["grid_ts"]) == 1 else q["grid_ts"][j]
            assert close(got, ts), (ci, j)
        if ref_ok:
            with open(path) as fh:
                m = Microgrid.load(fh)
            for kind in ("genset", "battery", "grid"):
                mods = m.modules[kind] if module_list(p.get(kind)) else []
                assert len(mods) == len(module_list(p.get(kind))), (ci, kind)
            for j, b in enumerate(module_list(p.get("battery"))):
                assert m.modules["battery"][j].max_capacity == b["max_capacity"]
                assert abs(m.modules["battery"][j].soc - b["init_soc"]) < 1e-15
            for j, g in enumerate(module_list(p.get("grid"))):
                assert m.modules["grid"][j].max_import == g["max_import"]
                assert close(m.modules["grid"][j].time_series, p["grid_ts"][j])
            assert len(m.modules["load"]) == p["load_ts"].shape[1]
            assert close(m.modules["load"][-1].time_series[:, 0], -np.abs(p["load_ts"][:, -1]))
    assert len(bucket_by_layout(all_p)) == len(all_p)          # every module mix is a layout of its own


def test_scenario_features_the_reference_serialises(tmp_path):
    """What round 5's loader refused and Microgrid.load (microgrid.py:848-908) accepts, both directions against the REAL reference:
    a trajectory function, modules built with raise_errors=True, time-series modules with forecast horizons of their own, a
    microgrid without a LoadModule -- the reference dumps, this loader reads; this dump writes, the reference loads."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import _refenv
    _refenv.import_reference()
    from pymgrid import Microgrid
    from pymgrid.microgrid.trajectory import DeterministicTrajectory as RefDet, FixedLengthStochasticTrajectory as RefFixed, \
        StochasticTrajectory as RefStoch
    from pymgrid.modules import BatteryModule, GridModule, LoadModule, RenewableModule
    from pymgrid_amd import trajectory as tj
    from pymgrid_amd.scenario import dump_scenario_yaml, load_scenario_yaml
    rs = np.random.RandomState(3)
    T = 60
    load, pv = 40 + 30 * rs.rand(T), 50 * rs.rand(T)
    grid = np.stack([0.2 + rs.rand(T), 0.1 * rs.rand(T), 0.3 * rs.rand(T), (rs.rand(T) > 0.2).astype(float)], axis=1)

    def bat():
        return BatteryModule(min_capacity=20.0, max_capacity=100.0, max_charge=25.0, max_discharge=30.0, efficiency=0.9, init_soc=0.5,
                             raise_errors=True)
    cases = {
        "fixed": (RefFixed(24), tj.FixedLengthStochasticTrajectory), "det": (RefDet(5, 40), tj.DeterministicTrajectory),
        "stoch": (RefStoch(), tj.StochasticTrajectory)}
    for name, (ref_tf, mirror) in cases.items():
        m = Microgrid([("load", LoadModule(time_series=load, forecaster="oracle", forecast_horizon=5, raise_errors=True)),
                       ("pv", RenewableModule(time_series=pv, raise_errors=True)), ("battery", bat()),
                       ("grid", GridModule(max_import=90.0, max_export=50.0, time_series=grid, forecaster="oracle", forecast_horizon=3,
                                           raise_errors=True))], trajectory_func=ref_tf)
        d = tmp_path / f"ref_{name}"
        os.makedirs(d)
        with open(d / "microgrid.yaml", "w") as fh:
            m.dump(fh)
        p = load_scenario_yaml(str(d / "microgrid.yaml"))
        assert isinstance(p["trajectory_func"], mirror), name
        if name == "fixed":
            assert p["trajectory_func"].trajectory_length == 24
        if name == "det":
            assert (p["trajectory_func"].initial_step, p["trajectory_func"].final_step) == (5, 40)
        assert p["raise_errors"] is True
        assert p["horizon"] == 5 and p["horizons"] == {"load": [5], "pv": [0], "grid": [3]}
        # ... and back: the reference loads this repo's dump of that dict into the same microgrid
        d2 = tmp_path / f"ours_{name}"
        os.makedirs(d2)
        with open(dump_scenario_yaml(p, str(d2 / "microgrid.yaml"))) as fh:
            m2 = Microgrid.load(fh)
        assert type(m2.trajectory_func) is type(ref_tf) and vars(m2.trajectory_func) == vars(ref_tf)
        assert m2.modules["load"][0].forecast_horizon == 5 and m2.modules["grid"][0].forecast_horizon == 3
        assert m2.modules["pv"][0].forecast_horizon == 0 and type(m2.modules["pv"][0].forecaster).__name__ == "NoForecaster"
        assert all(mod.raise_errors for mod in m2.modules.to_list() if hasattr(mod, "raise_errors"))
        assert m2.modules["battery"][0].soc == 0.5
    # no LoadModule
    m = Microgrid([("pv", RenewableModule(time_series=pv)), ("battery", bat())])
    d = tmp_path / "ref_noload"
    os.makedirs(d)
    with open(d / "microgrid.yaml", "w") as fh:
        m.dump(fh)
    p = load_scenario_yaml(str(d / "microgrid.yaml"))
    # (series go through csv text and pandas' default float parser, as in the reference: the last bit of an arbitrary double may differ)
    assert np.asarray(p["load_ts"]).shape == (T, 0) and np.allclose(p["pv_ts"], pv, rtol=1e-14, atol=0.0)
    d2 = tmp_path / "ours_noload"
    os.makedirs(d2)
    with open(dump_scenario_yaml(p, str(d2 / "microgrid.yaml"))) as fh:
        m2 = Microgrid.load(fh)
    assert "load" not in m2.modules.to_dict() and np.allclose(m2.modules["pv"][0].time_series[:, 0], pv, rtol=1e-14, atol=0.0)
    # modules that disagree about the window: a ValueError in the reference's constructor (get_attrs(unique=True)) and here
    with pytest.raises(ValueError):
        Microgrid([("load", LoadModule(time_series=load, final_step=50)), ("pv", RenewableModule(time_series=pv))])
    good = load_scenario_yaml(str(tmp_path / "ours_det" / "microgrid.yaml"))
    text = open(tmp_path / "ours_det" / "microgrid.yaml").read()
    assert good["final_step"] == T
    bad = text.replace("      final_step: 60", "      final_step: 50", 1)  # the first module's only
    with open(tmp_path / "ours_det" / "bad.yaml", "w") as fh:
        fh.write(bad)
    with pytest.raises(ValueError):
        load_scenario_yaml(str(tmp_path / "ours_det" / "bad.yaml"))
    # no time-series module at all: the reference cannot build such a microgrid (get_attrs: "No values found for key(s) ['final_step']")
    with pytest.raises(AttributeError):
        Microgrid([("battery", bat())])
