#!/usr/bin/env python3
"""Round-6 golden vectors, produced by the REAL reference (run in the build container only: needs /root/reference):
    python tests/golden/make_round6_goldens.py          -> tests/golden/round6.npz

  strict_*    Microgrid.sample_action(strict_bound=True) (microgrid.py:337-362, base_module.py:326-356) on microgrids without a
              genset (battery + weak grid; battery only; grid listed before the battery): numpy's global generator seeded, then K
              rounds of sample_action -> run.  Stored: the controls drawn, the per-step bounds the reference normalised
              ([normalize(-max_consumption), normalize(max_production)] per module), the rewards.  (With a genset the reference
              itself raises: SURVEY App. C Q4 -- one such case is stored as the exception's type name.)
  mixed_*     time-series modules with DIFFERENT forecast horizons in one microgrid (load H = 5 oracle, pv no forecaster, grid
              H = 3): K steps of seeded random controls -> rewards and the flat observation of every step (module order load, pv,
              genset, battery, grid).
  nofix_*     microgrids WITHOUT a LoadModule (pv + battery + grid) and WITHOUT a RenewableModule (load + genset + battery):
              rewards + log columns.
Everything stored is data: parameters, series, seeds, and what the reference computed from them."""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (imports the reference through tests/golden/_refenv.py)

warnings.simplefilter("ignore")
from pymgrid import Microgrid  # noqa: E402
from pymgrid.modules import BatteryModule, GensetModule, GridModule, LoadModule, RenewableModule  # noqa: E402


def series(rs, T):
    t = np.arange(T)
    load = 40 + 30 * rs.rand(T)
    pv = 60 * rs.rand(T) * (rs.rand(T) > 0.35)
    price = np.where((t % 24 >= 17) & (t % 24 < 21), 0.59, np.where(t % 24 >= 8, 0.29, 0.22))
    grid = np.stack([price, 0.4 * price, 0.2 + 0.3 * rs.rand(T), (rs.rand(T) > 0.25).astype(float)], axis=1)
    return load, pv, grid


def battery(rs):
    cap = float(rs.randint(100, 400))
    return BatteryModule(min_capacity=0.2 * cap, max_capacity=cap, max_charge=float(np.ceil(cap / 4)), max_discharge=float(np.ceil(cap / 3)),
                         efficiency=float(rs.choice([0.9, 0.95, 1.0])), battery_cost_cycle=0.02, init_soc=float(rs.uniform(0.25, 0.95)))


def put_params(out, pre, p):
    scalars, arrays = mg.split_params(p)
    out[pre + "params"] = np.array(json.dumps(scalars))
    for k, v in arrays.items():
        out[pre + k] = v


def make_strict(out):
    T, K = 120, 100
    cases = []
    for c, kind in enumerate(("battery+grid", "battery", "grid+battery", "battery+grid")):
        rs = np.random.RandomState(9100 + c)
        load, pv, grid = series(rs, T)
        mods = [("load", LoadModule(time_series=load)), ("pv", RenewableModule(time_series=pv))]
        gridm = GridModule(max_import=float(rs.randint(60, 120)), max_export=float(rs.randint(30, 90)), time_series=grid, cost_per_unit_co2=0.1)
        if kind == "battery":
            mods += [("battery", battery(rs))]
        elif kind == "grid+battery":
            mods += [("grid", gridm), ("battery", battery(rs))]
        else:
            mods += [("battery", battery(rs)), ("grid", gridm)]
        m = Microgrid(mods, loss_load_cost=10.0, overgeneration_cost=1.0)
        p = mg.extract_params(m)
        if kind == "grid+battery":
            p["controllable_order"] = ["grid", "battery"]
        put_params(out, f"strict{c}_", p)
        seed = 4200 + c
        np.random.seed(seed)
        names = [n for n in ("battery", "grid") if n in p]
        ctrl_rows, lo_rows, hi_rows, rew = [], [], [], []
        for k in range(K):
            lo, hi = [], []
            for n in names:                          # what module.sample_action(strict_bound=True) computes, read BEFORE the draw
                mod = m.modules[n][0]
                a, b = mod._action_space.normalize(-1 * mod.max_consumption), mod._action_space.normalize(mod.max_production)
                lo.append(0.0 if np.isnan(a) else float(a)); hi.append(0.0 if np.isnan(b) else float(b))
            ctrl = m.sample_action(strict_bound=True)
            assert list(ctrl) == ([n for n in p.get("controllable_order", names)]), (list(ctrl), names)
            ctrl_rows.append([float(ctrl[n][0]) for n in names])
            lo_rows.append(lo); hi_rows.append(hi)
            _, r, _, _ = m.run(ctrl, normalized=True)
            rew.append(r)
        out[f"strict{c}_seed"] = np.array(seed)
        out[f"strict{c}_control"] = np.array(ctrl_rows)          # columns: battery, grid (the action layout), whatever the draw order
        out[f"strict{c}_lo"] = np.array(lo_rows); out[f"strict{c}_hi"] = np.array(hi_rows)
        out[f"strict{c}_reward"] = np.array(rew)
        cases.append(kind)
    out["strict_cases"] = np.array(cases)
    # with a genset the reference raises
    rs = np.random.RandomState(9200)
    load, pv, _ = series(rs, 50)
    m = Microgrid([("load", LoadModule(time_series=load)), ("pv", RenewableModule(time_series=pv)),
                   ("genset", GensetModule(running_min_production=5.0, running_max_production=80.0, genset_cost=0.4)),
                   ("battery", battery(rs))])
    try:
        m.sample_action(strict_bound=True)
        err = ""
    except Exception as e:      # noqa: BLE001
        err = type(e).__name__
    out["strict_genset_error"] = np.array(err)
    print("strict: genset raises", err)


def make_mixed(out):
    T, K = 90, 80
    rs = np.random.RandomState(9300)
    load, pv, grid = series(rs, T)
    m = Microgrid([("load", LoadModule(time_series=load, forecaster="oracle", forecast_horizon=5)),
                   ("pv", RenewableModule(time_series=pv)),
                   ("battery", battery(rs)),
                   ("grid", GridModule(max_import=100.0, max_export=60.0, time_series=grid, forecaster="oracle", forecast_horizon=3,
                                       cost_per_unit_co2=0.1))], loss_load_cost=10.0, overgeneration_cost=1.0)
    p = mg.extract_params(m)
    p["horizon"] = 5
    p["horizons"] = {"load": [5], "pv": [0], "grid": [3]}
    put_params(out, "mixed_", p)
    acts = np.random.RandomState(9301).rand(K, 2)
    obs0 = mg.flat_obs(m, m.state_dict(normalized=True)) if False else None
    res = mg.run_episode(m, p, acts, want_obs=True)
    out["mixed_actions"], out["mixed_reward"], out["mixed_obs"] = acts, res["reward"], res["obs"]
    print("mixed horizons: obs row", res["obs"].shape)


def make_nofix(out):
    T, K = 80, 79
    for c, kind in enumerate(("no_load", "no_pv")):
        rs = np.random.RandomState(9400 + c)
        load, pv, grid = series(rs, T)
        if kind == "no_load":
            mods = [("pv", RenewableModule(time_series=pv)), ("battery", battery(rs)),
                    ("grid", GridModule(max_import=100.0, max_export=60.0, time_series=grid, cost_per_unit_co2=0.1))]
        else:
            mods = [("load", LoadModule(time_series=load)),
                    ("genset", GensetModule(running_min_production=5.0, running_max_production=90.0, genset_cost=0.4, co2_per_unit=2.0,
                                            cost_per_unit_co2=0.1, start_up_time=1, wind_down_time=1)),
                    ("battery", battery(rs))]
        m = Microgrid(mods, loss_load_cost=10.0, overgeneration_cost=1.0)
        p = mg.extract_params(m)
        put_params(out, f"nofix{c}_", p)
        acts = np.random.RandomState(9410 + c).rand(K, mg.action_dims(p))
        res = mg.run_episode(m, p, acts, want_obs=True)
        out[f"nofix{c}_actions"], out[f"nofix{c}_reward"], out[f"nofix{c}_log"] = acts, res["reward"], res["log"]
        out[f"nofix{c}_obs"] = res["obs"]
        out[f"nofix{c}_charge"], out[f"nofix{c}_status"] = res["charge"], res["status"]
    out["nofix_cases"] = np.array(["no_load", "no_pv"])
    out["log_names"] = np.array(mg.LOG_NAMES)


if __name__ == "__main__":
    out = {}
    make_strict(out)
    make_mixed(out)
    make_nofix(out)
    mg.save("round6.npz", **out)
