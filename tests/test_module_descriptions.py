"""`Microgrid([modules...])` -- the reference's own way to build a microgrid (microgrid.py:100-173) -- through this package's module
descriptions (pymgrid_amd/modules.py): the parameter dict they turn into == what tests/golden/make_surface_goldens.py extracted from
the REAL modules built with the same constructor arguments (tests/golden/surface.npz), the constructors refuse what the reference's
refuse, and (gpu) the microgrid built that way steps `==` the reference."""
import json

import numpy as np
import pytest

from conftest import golden


def _modules(z, case):
    from pymgrid_amd.modules import BatteryModule, GensetModule, GridModule, LoadModule, RenewableModule
    pre = f"c{case}_"
    p = json.loads(str(z[pre + "params"]))
    H = p["horizon"]
    kw = dict(forecaster="oracle", forecast_horizon=H) if H else {}
    mods = [("load", LoadModule(time_series=np.abs(z[pre + "load_ts"][:, 0]), **kw)), ("pv", RenewableModule(time_series=z[pre + "pv_ts"][:, 0], **kw))]
    if "genset" in p:
        g = dict(p["genset"])
        st = g.pop("status")
        mods.append(("genset", GensetModule(init_start_up=bool(st[0]), **g)))
    b = dict(p["battery"])
    soc = b.pop("soc"); b.pop("charge")
    mods.append(BatteryModule(init_soc=soc, **b))                      # (a bare module: the list may mix tuples and modules)
    if "grid" in p:
        mods.append(("grid", GridModule(time_series=z[pre + "grid_ts"], **kw, **p["grid"])))
    return p, mods


@pytest.mark.parametrize("case", [0, 1, 2])
def test_module_list_gives_the_parameters_the_reference_modules_hold(case):
    from pymgrid_amd.modules import params_from_modules
    z = golden("surface.npz")
    p, mods = _modules(z, case)
    q = params_from_modules(mods, loss_load_cost=10.0, overgeneration_cost=1.0)
    for k in ("horizon", "final_step", "initial_step", "unbalanced", "battery", "genset", "grid"):
        assert q.get(k) == p.get(k), k
    pre = f"c{case}_"
    for k in ("load_ts", "pv_ts", "grid_ts"):
        if pre + k in z.files:
            assert np.array_equal(np.abs(np.asarray(q[k])).reshape(-1), np.abs(z[pre + k]).reshape(-1)), k
    assert q["controllable_order"] == [n for n in ("genset", "battery", "grid") if n in p]


def test_module_constructors_refuse_what_the_reference_refuses():
    from pymgrid_amd.modules import BatteryModule, GensetModule, GridModule, LoadModule, params_from_modules
    with pytest.raises(AssertionError):
        BatteryModule(10, 100, 20, 20, efficiency=1.2, init_soc=0.5)                   # battery_module.py:78
    with pytest.raises(ValueError):
        BatteryModule(10, 100, 20, 20, efficiency=0.9)                                 # :96-106 neither init_charge nor init_soc
    with pytest.raises(ValueError):
        GensetModule(running_min_production=50, running_max_production=10, genset_cost=0.4)      # genset_module.py:75-76
    with pytest.raises(NotImplementedError):
        GensetModule(0, 10, genset_cost=lambda x: x)
    ts = np.ones((10, 4))
    with pytest.raises(ValueError):
        GridModule(-1, 10, ts)                                                           # grid_module.py:103-107
    with pytest.raises(ValueError):
        GridModule(10, 10, np.ones((10, 2)))
    with pytest.raises(ValueError):
        GridModule(10, 10, np.full((10, 4), 0.5))                                        # status column not binary
    assert GridModule(10, 10, np.ones((10, 3))).cls_params["time_series"].shape == (10, 4)      # three columns: an always-up grid
    with pytest.raises(NotImplementedError):
        LoadModule(np.ones(10), forecaster=lambda a, b, n: b)
    with pytest.raises(TypeError):
        params_from_modules([("load", object())])
    with pytest.raises(ValueError):                                                      # no time-series module: no final_step (microgrid.py:113-128)
        params_from_modules([BatteryModule(10, 100, 20, 20, 0.9, init_soc=0.5)])
    # both signs in one series (base_timeseries_module.py:68-79) is refused when the batch is packed
    from pymgrid_amd.batch import pack_grids
    q = params_from_modules([LoadModule(np.array([1.0, 2.0, 3.0])), BatteryModule(10, 100, 20, 20, 0.9, init_charge=50.0)])
    assert q["battery"]["soc"] == 0.5 and q["unbalanced"] == dict(loss_load_cost=10.0, overgeneration_cost=2.0)
    assert pack_grids([q])[1].n_pv == 0


@pytest.mark.gpu
def test_microgrid_built_from_modules_steps_like_the_reference(device):
    from pymgrid_amd import Microgrid
    z = golden("surface.npz")
    for case in range(3):
        pre = f"c{case}_"
        _, mods = _modules(z, case)
        m = Microgrid(mods, device=str(device), loss_load_cost=10.0, overgeneration_cost=1.0)
        m.reset()
        np.random.seed(int(z[pre + "seed"]))
        for k in range(z[pre + "reward"].shape[0]):
            _, reward, done, _ = m.run(m.sample_action())
            assert reward == z[pre + "reward"][k] and done == bool(z[pre + "done"][k]), (case, k)
        m.close()
