#!/usr/bin/env python3
"""Golden fixture for the states in which the REFERENCE gives up with an AssertionError (tests/golden/asserts.npz).

Run in the build container only (needs /root/reference):   python tests/golden/make_assert_goldens.py

``DiscreteMicrogridEnv.step`` asserts its way through ``PriorityListAlgo._populate_action`` (priority_list.py:73,121,124,135,154)
and ``Microgrid.run`` through ``BaseMicrogridModule.as_sink`` (base_module.py:272) and ``BatteryModule.update``
(battery_module.py:114,118).  The states that trip them are reachable:
a lossy battery charged at its limit lands one ulp ABOVE ``max_capacity`` ((x / eta) * eta rounds up), after which its
``max_consumption`` is negative.  Every case below is a one-step probe of a microgrid in such a state -- some placed there by
hand (``nextafter``), some driven there by the reference's own arithmetic -- under EVERY priority list of its discrete env
and under a few continuous controls.  Stored: the parameters and pre-step state (inputs) and, per probe, whether the reference
raised, at which file:line, else the control / reward / post-step charge it produced (outputs).  No reference source text.
"""
import json
import os
import sys
import warnings
from copy import deepcopy

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

warnings.simplefilter("ignore")
_refenv.import_reference()

import make_goldens as mg  # noqa: E402
from pymgrid import Microgrid  # noqa: E402
from pymgrid.envs import DiscreteMicrogridEnv  # noqa: E402
from pymgrid.modules import BatteryModule, GensetModule, GridModule, LoadModule, RenewableModule  # noqa: E402

MOD_ID = {GensetModule: 0, BatteryModule: 1, GridModule: 2}
SITES = {"priority_list.py": 1, "base_module.py": 2, "genset_module.py": 3, "battery_module.py": 4}


def site_of(exc):
    tb = exc.__traceback__
    while tb.tb_next is not None:
        tb = tb.tb_next
    return os.path.basename(tb.tb_frame.f_code.co_filename), tb.tb_lineno


def build(kind, load, pv, bat, rs, grid_first=False, weak=False):
    """kind: bit 0 genset, bit 2 grid (a battery always)."""
    T = len(load)
    mods = [("load", LoadModule(time_series=np.asarray(load, dtype=float))), ("pv", RenewableModule(time_series=np.asarray(pv, dtype=float)))]
    ctrl = []
    if kind & 1:
        ctrl.append(("genset", GensetModule(running_min_production=float(rs.choice([0.0, 5.0])), running_max_production=60.0,
                                            genset_cost=0.4, co2_per_unit=2.0, cost_per_unit_co2=0.1,
                                            start_up_time=int(rs.randint(0, 2)), wind_down_time=0, init_start_up=True)))
    battery = ("battery", BatteryModule(**bat))
    if kind & 4:
        status = np.ones(T)
        if weak:
            status[0] = 0.0
        gts = np.stack([0.2 + 0.1 * rs.rand(T), 0.05 * rs.rand(T), 0.3 * rs.rand(T), status], axis=1)
        grid = ("grid", GridModule(max_import=80.0, max_export=float(rs.choice([0.0, 40.0])), time_series=gts, cost_per_unit_co2=0.1))
        ctrl += [grid, battery] if grid_first else [battery, grid]
    else:
        ctrl.append(battery)
    return Microgrid(mods + ctrl, loss_load_cost=10.0, overgeneration_cost=1.0)


def set_charge(m, charge):
    b = mg.find(m, BatteryModule)[0]
    b._current_charge = float(charge)
    b._soc = float(charge) / b.max_capacity


def drive_overfull(rs):
    """A lossy battery charged at its limit by the reference itself until (and if) its charge exceeds max_capacity."""
    for _ in range(4000):
        cap = float(10 ** rs.uniform(0.5, 4))
        bat = dict(min_capacity=cap * float(rs.choice([0.0, 0.2])), max_capacity=cap, max_charge=cap * float(rs.uniform(0.05, 1.2)),
                   max_discharge=cap * float(rs.uniform(0.05, 1.2)), efficiency=float(rs.choice([0.9, 0.5, rs.uniform(0.3, 0.99)])),
                   battery_cost_cycle=0.02, init_soc=float(rs.uniform(0.2, 1.0)))
        m = build(0, np.zeros(40), np.full(40, 3 * cap), bat, rs)
        b = mg.find(m, BatteryModule)[0]
        for _k in range(30):
            try:
                m.run({"battery": [-1e9]}, normalized=False)
            except AssertionError:
                break
            if b.current_charge > b.max_capacity:
                return bat, float(b.current_charge)
    raise RuntimeError("no overfull battery found")


def probe(m):
    """One-step probes of a microgrid standing at its current step: every priority list, and a few continuous controls."""
    p = mg.extract_params(m)
    A = mg.action_dims(p)
    env0 = DiscreteMicrogridEnv.from_microgrid(deepcopy(m))
    n = env0.action_space.n
    table = -np.ones((n, 4, 2), np.int32)
    for i, pl in enumerate(env0.actions_list):
        for j, el in enumerate(pl):
            table[i, j] = (MOD_ID[type(env0.modules[el.module[0]][el.module[1]])], el.action)
    d_site, d_line = np.zeros(n, np.int32), np.zeros(n, np.int32)
    d_control, d_reward, d_charge = np.zeros((n, A)), np.zeros(n), np.zeros(n)
    for a in range(n):
        env = deepcopy(env0)
        try:
            ctrl = env._get_action(a)
            c = 0
            if "genset" in p:
                d_control[a, c:c + 2] = np.asarray(ctrl[mg.mod_name(env, GensetModule)][0], dtype=np.float64); c += 2
            if "battery" in p:
                d_control[a, c] = ctrl[mg.mod_name(env, BatteryModule)][0]; c += 1
            if "grid" in p:
                d_control[a, c] = ctrl[mg.mod_name(env, GridModule)][0]; c += 1
            _, d_reward[a], _, _ = env.step(a)
            d_charge[a] = mg.post_state(env)[0]
        except AssertionError as exc:
            f, d_line[a] = site_of(exc)
            d_site[a] = SITES[f]
    # continuous controls (raw units): charge hard, charge a little, idle, discharge -- the battery's request in the last-but-grid column
    rows = []
    for x in (-1e9, -1e-3, 0.0, 1.0):
        row = np.zeros(A)
        c = 0
        if "genset" in p:
            row[0], row[1] = 1.0, 10.0; c = 2
        row[c] = x
        rows.append(row)
    rows = np.array(rows)
    c_site, c_line, c_reward, c_charge = (np.zeros(len(rows), np.int32), np.zeros(len(rows), np.int32), np.zeros(len(rows)),
                                          np.zeros(len(rows)))
    for k, row in enumerate(rows):
        mm = deepcopy(m)
        try:
            _, c_reward[k], _, _ = mm.run(mg.control_from_row(mm, p, row), normalized=False)
            c_charge[k] = mg.post_state(mm)[0]
        except AssertionError as exc:
            f, c_line[k] = site_of(exc)
            c_site[k] = SITES[f]
    return p, dict(table=table, d_site=d_site, d_line=d_line, d_control=d_control, d_reward=d_reward, d_charge=d_charge,
                   c_rows=rows, c_site=c_site, c_line=c_line, c_reward=c_reward, c_charge=c_charge)


def main():
    rs = np.random.RandomState(7272)
    out, meta = {}, []
    cases = []
    T = 6
    # (a) placed by hand: one ulp above max_capacity / one ulp below min_capacity / exactly full / exactly empty, under excess and
    #     deficit, for every layout that has a battery (grid before / after the battery; an outage row)
    for kind in (0, 1, 4, 5):
        for gf in ((False, True) if kind & 4 else (False,)):
            for weak in ((False, True) if kind & 4 else (False,)):
                for eta in (0.9, 0.5):
                    cap = 100.0
                    bat = dict(min_capacity=20.0, max_capacity=cap, max_charge=30.0, max_discharge=25.0, efficiency=eta,
                               battery_cost_cycle=0.02, init_soc=0.5)
                    for charge in (np.nextafter(cap, np.inf), cap, np.nextafter(20.0, -np.inf), 20.0, np.nextafter(cap, -np.inf)):
                        for load, pv in ((10.0, 50.0), (50.0, 10.0), (30.0, 30.00005)):       # excess, deficit, |difference| <= 1e-4
                            m = build(kind, np.full(T, load), np.full(T, pv), bat, rs, grid_first=gf, weak=weak)
                            set_charge(m, charge)
                            cases.append((m, dict(kind=kind, grid_first=gf, weak=weak, how="placed")))
    # (b) driven there by the reference's own arithmetic
    for j in range(24):
        bat, charge = drive_overfull(rs)
        kind = (0, 1, 4, 5)[j % 4]
        cap = bat["max_capacity"]
        m = build(kind, np.full(T, 0.1 * cap), np.full(T, 0.1 * cap * float(rs.choice([0.2, 3.0]))), bat, rs, grid_first=bool(j % 8 >= 4))
        set_charge(m, charge)
        cases.append((m, dict(kind=kind, grid_first=bool(j % 8 >= 4), weak=False, how="driven")))
    n_raise = {}
    for i, (m, info) in enumerate(cases):
        p, res = probe(m)
        p["controllable_order"] = [{GensetModule: "genset", BatteryModule: "battery", GridModule: "grid"}[type(lst[0])]
                                   for _, lst in m.controllable.iterdict()]
        scalars, arrays = mg.split_params(p)
        scalars.update(info)
        meta.append(scalars)
        for k, v in arrays.items():
            out[f"c{i}_{k}"] = v
        for k, v in res.items():
            out[f"c{i}_{k}"] = v
        for s, ln in list(zip(res["d_site"], res["d_line"])) + list(zip(res["c_site"], res["c_line"])):
            if s:
                n_raise[(int(s), int(ln))] = n_raise.get((int(s), int(ln)), 0) + 1
    out["meta"] = np.array(json.dumps(meta))
    out["sites"] = np.array(json.dumps(SITES))
    mg.save("asserts.npz", **out)
    print(f"{len(cases)} cases; raises by (site, line): {n_raise}")


if __name__ == "__main__":
    main()
