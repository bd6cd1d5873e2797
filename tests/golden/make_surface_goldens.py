#!/usr/bin/env python3
"""Golden vectors for the `Microgrid` methods beside `run` that the N = 1 adaptors mirror, produced by the REAL reference (run in
the build container only: needs /root/reference):
    python tests/golden/make_surface_goldens.py          -> tests/golden/surface.npz

Three small microgrids (genset + battery + weak grid with H = 5; battery + grid with H = 0; a genset with start-up / wind-down
times + battery with H = 3), each stepped with seeded `sample_action()` controls to the END of its series (so that the padded
forecasts of the last H steps are in the fixture).  With H = 0 the reference's own state_dict(normalized=True) raises TypeError (a
one-value state normalises to a float, base_module.py:488): stored as the exception's name.  Before every step:
  sd_raw / sd_norm   Microgrid.state_dict(normalized=False / True)   (microgrid.py:699-717), values in iteration order
  cost               Microgrid.get_cost_info()                        (microgrid.py:334-335): [production, absorption] per module
  act / act_raw / act_back   the control drawn, Microgrid.from_normalized(act, act=True) and to_normalized(that, act=True)  (:388-431)
  obs_back / obs_fwd from_normalized(state arrays, obs=True) of the NORMALISED state and to_normalized(..., obs=True) of the raw one
and of the step itself: the nested observation `run` returns (microgrid.py:227-325 -> MicrogridStep.output), its reward and done.
Everything stored is data: parameters, series, seeds, and what the reference computed from them."""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (imports the reference through tests/golden/_refenv.py)
from make_round6_goldens import battery, put_params, series  # noqa: E402

warnings.simplefilter("ignore")
from pymgrid import Microgrid  # noqa: E402
from pymgrid.modules import GensetModule, GridModule, LoadModule, RenewableModule  # noqa: E402


def flat(nested):
    """{name: [array-or-scalar per module]} -> one float64 vector in iteration order"""
    out = []
    for name, lst in nested.items():
        for v in lst:
            out.append(np.asarray(list(v.values()) if isinstance(v, dict) else v, dtype=np.float64).reshape(-1))
    return np.concatenate(out) if out else np.zeros(0)


def build(case, rs, T):
    load, pv, grid = series(rs, T)
    if case == 0:
        H = 5
        mods = [("genset", GensetModule(running_min_production=10.0, running_max_production=90.0, genset_cost=0.4, co2_per_unit=2.0,
                                        cost_per_unit_co2=0.1)),
                ("battery", battery(rs)),
                ("grid", GridModule(max_import=80.0, max_export=50.0, time_series=grid, cost_per_unit_co2=0.1, forecaster="oracle",
                                    forecast_horizon=H))]
    elif case == 1:
        H = 0
        grid[:, 3] = 1.0
        mods = [("battery", battery(rs)),
                ("grid", GridModule(max_import=100.0, max_export=60.0, time_series=grid, cost_per_unit_co2=0.05))]
    else:
        H = 3
        mods = [("genset", GensetModule(running_min_production=5.0, running_max_production=120.0, genset_cost=0.35, co2_per_unit=1.5,
                                        cost_per_unit_co2=0.2, start_up_time=2, wind_down_time=1, init_start_up=False)),
                ("battery", battery(rs))]
    kw = dict(forecaster="oracle", forecast_horizon=H) if H else {}
    mods = [("load", LoadModule(time_series=load, **kw)), ("pv", RenewableModule(time_series=pv, **kw))] + mods
    return Microgrid(mods, loss_load_cost=10.0, overgeneration_cost=1.0)


def main():
    out = {}
    T = 40
    for case in range(3):
        rs = np.random.RandomState(6300 + case)
        m = build(case, rs, T)
        pre = f"c{case}_"
        put_params(out, pre, mg.extract_params(m))
        seed = 880 + case
        np.random.seed(seed)
        out[pre + "seed"] = np.array(seed)
        rows = {k: [] for k in ("sd_raw", "sd_norm", "cost", "act", "act_raw", "act_back", "obs_back", "obs_fwd", "obs", "reward", "done")}
        keys = None
        done = False
        while not done:
            sd_raw = m.state_dict(normalized=False)
            try:
                sd_norm = m.state_dict(normalized=True)
            except TypeError as e:               # H = 0: ModuleSpace.normalize hands a float back for a one-value state (space.py:207-218)
                sd_norm = None
                out[pre + "sd_norm_error"] = np.array(type(e).__name__)
            if keys is None:
                keys = {"state": {n: [list(d) for d in lst] for n, lst in sd_raw.items()}, "cost": list(m.get_cost_info())}
            rows["sd_raw"].append(flat(sd_raw))
            if sd_norm is not None:
                rows["sd_norm"].append(flat(sd_norm))
            rows["cost"].append(np.array([[d["production_marginal_cost"], d["absorption_marginal_cost"]]
                                          for lst in m.get_cost_info().values() for d in lst], dtype=np.float64).reshape(-1))
            as_arrays = lambda sd: {n: [np.array(list(d.values()), dtype=np.float64) for d in lst] for n, lst in sd.items()}  # noqa: E731
            if sd_norm is not None:
                rows["obs_back"].append(flat(m.from_normalized(as_arrays(sd_norm), obs=True)))
            rows["obs_fwd"].append(flat(m.to_normalized(as_arrays(sd_raw), obs=True)))
            a = m.sample_action()
            raw = m.from_normalized(a, act=True)
            rows["act"].append(flat(a)); rows["act_raw"].append(flat(raw)); rows["act_back"].append(flat(m.to_normalized(raw, act=True)))
            obs, reward, done, _ = m.run(a)
            if "obs" not in keys:
                keys["obs"] = list(obs)
                keys["act"] = list(a)
            rows["obs"].append(flat(obs)); rows["reward"].append(reward); rows["done"].append(done)
        out[pre + "keys"] = np.array(json.dumps(keys))
        for k, v in rows.items():
            out[pre + k] = np.array(v)
        print(case, "steps", len(rows["reward"]), "state width", rows["sd_raw"][0].shape, "keys", keys["obs"], keys["cost"])
    out["cases"] = np.array(3)
    np.savez_compressed(os.path.join(HERE, "surface.npz"), **out)


if __name__ == "__main__":
    main()
