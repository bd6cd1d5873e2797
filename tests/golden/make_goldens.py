#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference (pymgrid 1.2.2).

Run in the build container only (needs /root/reference):   python tests/golden/make_goldens.py
Everything stored is DATA: inputs (parameters, time series, seeds) and the outputs the reference's own
code produced for them.  No reference source text is stored.

Fixture families:
  pymgrid25_inputs.npz   parameters + time series of the 25 benchmark scenarios (as loaded by
                         Microgrid.from_scenario: microgrid.py:958-980)
  pymgrid25_run.npz      G1/G2: full-year runs with seeded random normalised actions -> per-step reward,
                         SoC, charge, genset status, done + full log rows on a step subsample
  discrete.npz           G3: DiscreteMicrogridEnv priority-list tables + expanded controls + rewards
  genset_fsm.npz         G4: GensetModule.update_status transition table (start_up, wind_down in 0..4)
  obs.npz                G5/G7/G8: observations (incl. end-of-series padding, H=23/24, weak grid), reset semantics
  generated.npz          G6/G7: generator-style grids built as real pymgrid modules (genset FSM timers,
                         grid module with outages, normalised and raw actions)
  loadpv.npz             multi-module load/pv-only grids (the reference's TestMicrogridLoadPV family)
  order.npz              grids whose module list names the GridModule before the BatteryModule (sweep order)
  helper_microgrid.npz   the fixture microgrid of the reference's own tests: random actions, RBC, discrete env
"""
import itertools
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refenv  # noqa: E402

warnings.simplefilter("ignore")
_refenv.import_reference()

from pymgrid import Microgrid  # noqa: E402
from pymgrid.envs import DiscreteMicrogridEnv  # noqa: E402
from pymgrid.modules import (BatteryModule, GensetModule, GridModule, LoadModule,  # noqa: E402
                             RenewableModule, UnbalancedEnergyModule)


# --------------------------------------------------------------------------------------------- #
def find(m, cls):
    return [mod for _, lst in m.modules.iterdict() for mod in lst if isinstance(mod, cls)]


def extract_params(m):
    """Reference Microgrid -> plain parameter dict (arrays + scalars) in this repo's vocabulary."""
    loads, pvs = find(m, LoadModule), find(m, RenewableModule)
    p = {
        "load_ts": np.stack([x.time_series[:, 0] for x in loads], axis=1) if loads else np.zeros((0, 0)),
        "pv_ts": np.stack([x.time_series[:, 0] for x in pvs], axis=1) if pvs else np.zeros((0, 0)),
        "horizon": int((loads + pvs)[0].forecast_horizon),
        "final_step": int((loads + pvs)[0].final_step),
        "initial_step": int((loads + pvs)[0].initial_step),
    }
    if not loads:
        p["load_ts"] = np.zeros((p["pv_ts"].shape[0], 0))
    if not pvs:
        p["pv_ts"] = np.zeros((p["load_ts"].shape[0], 0))
    ub = find(m, UnbalancedEnergyModule)[0]
    p["unbalanced"] = dict(loss_load_cost=float(ub.loss_load_cost), overgeneration_cost=float(ub.overgeneration_cost))
    bats, gens, grids = find(m, BatteryModule), find(m, GensetModule), find(m, GridModule)
    assert len(bats) <= 1 and len(gens) <= 1 and len(grids) <= 1
    if bats:
        b = bats[0]
        p["battery"] = dict(min_capacity=float(b.min_capacity), max_capacity=float(b.max_capacity),
                            max_charge=float(b.max_charge), max_discharge=float(b.max_discharge),
                            efficiency=float(b.efficiency), battery_cost_cycle=float(b.battery_cost_cycle),
                            charge=float(b.current_charge), soc=float(b.soc))
    if gens:
        g = gens[0]
        assert not callable(g.genset_cost)
        p["genset"] = dict(running_min_production=float(g.running_min_production),
                           running_max_production=float(g.running_max_production),
                           genset_cost=float(g.genset_cost), co2_per_unit=float(g.co2_per_unit),
                           cost_per_unit_co2=float(g.cost_per_unit_co2),
                           start_up_time=int(g.start_up_time), wind_down_time=int(g.wind_down_time),
                           status=[int(g._current_status), int(g._goal_status),
                                   int(g._steps_until_up), int(g._steps_until_down)])
        if not g.allow_abortion:
            p["genset"]["allow_abortion"] = False
    if grids:
        g = grids[0]
        p["grid"] = dict(max_import=float(g.max_import), max_export=float(g.max_export),
                         cost_per_unit_co2=float(g.cost_per_unit_co2))
        p["grid_ts"] = np.array(g.time_series, dtype=np.float64)
    return p


def split_params(p):
    """-> (json-able scalars, dict of arrays)"""
    arrays = {k: np.asarray(v, dtype=np.float64) for k, v in p.items() if isinstance(v, np.ndarray)}
    scalars = {k: v for k, v in p.items() if not isinstance(v, np.ndarray)}
    return scalars, arrays


def mod_name(m, cls):
    for name, lst in m.modules.iterdict():
        if isinstance(lst[0], cls):
            return name
    return None


LOG_COLUMNS = [  # (our name, module class or 'balance', reference log field)
    ("reward", "balance", "reward"),
    ("fixed_provided", "balance", "fixed_provided_to_microgrid"),
    ("fixed_absorbed", "balance", "fixed_absorbed_from_microgrid"),
    ("controllable_provided", "balance", "controllable_provided_to_microgrid"),
    ("controllable_absorbed", "balance", "controllable_absorbed_from_microgrid"),
    ("overall_provided", "balance", "overall_provided_to_microgrid"),
    ("overall_absorbed", "balance", "overall_absorbed_from_microgrid"),
    ("load_met", LoadModule, "load_met"),
    ("renewable_used", RenewableModule, "renewable_used"),
    ("curtailment", RenewableModule, "curtailment"),
    ("loss_load", UnbalancedEnergyModule, "loss_load"),
    ("overgeneration", UnbalancedEnergyModule, "overgeneration"),
    ("unbalanced_reward", UnbalancedEnergyModule, "reward"),
    ("genset_production", GensetModule, "genset_production"),
    ("genset_co2_production", GensetModule, "co2_production"),
    ("genset_reward", GensetModule, "reward"),
    ("gen_cur", GensetModule, "current_status"),
    ("gen_goal", GensetModule, "goal_status"),
    ("gen_up", GensetModule, "steps_until_up"),
    ("gen_down", GensetModule, "steps_until_down"),
    ("discharge_amount", BatteryModule, "discharge_amount"),
    ("charge_amount", BatteryModule, "charge_amount"),
    ("battery_reward", BatteryModule, "reward"),
    ("soc_pre", BatteryModule, "soc"),
    ("charge_pre", BatteryModule, "current_charge"),
    ("grid_import", GridModule, "grid_import"),
    ("grid_export", GridModule, "grid_export"),
    ("grid_co2_production", GridModule, "co2_production"),
    ("grid_reward", GridModule, "reward"),
]
LOG_NAMES = [c[0] for c in LOG_COLUMNS]


def log_matrix(m):
    """Reference log -> [steps, len(LOG_COLUMNS)] float64 (NaN where the module is absent).
    Multi-module grids: per-module columns summed in module order (left-to-right running sum)."""
    log = m.get_log()
    n = len(log)
    out = np.full((n, len(LOG_COLUMNS)), np.nan)
    for j, (_, cls, field) in enumerate(LOG_COLUMNS):
        name = "balance" if cls == "balance" else mod_name(m, cls)
        if name is None:
            continue
        cols = [c for c in log.columns if c[0] == name and c[2] == field]
        if not cols:
            out[:, j] = np.zeros(n)
            continue
        acc = log[cols[0]].values.astype(np.float64).copy()     # (no "0.0 +": a module's -0.0 keeps its sign)
        for c in cols[1:]:
            acc = acc + log[c].values.astype(np.float64)
        out[:, j] = acc
    return out


def action_dims(p):
    return 2 * ("genset" in p) + ("battery" in p) + ("grid" in p)


def control_from_row(m, p, row):
    """Row of the flat action matrix -> Microgrid.run control dict.  Column order: genset(goal, energy),
    battery, grid (= the reference's controllable sweep order, module_container.py:355-413)."""
    ctrl, c = {}, 0
    if "genset" in p:
        ctrl[mod_name(m, GensetModule)] = [np.array([row[c], row[c + 1]])]; c += 2
    if "battery" in p:
        ctrl[mod_name(m, BatteryModule)] = [float(row[c])]; c += 1
    if "grid" in p:
        ctrl[mod_name(m, GridModule)] = [float(row[c])]; c += 1
    return ctrl


def flat_obs(m, obs):
    """Nested obs dict -> flat vector in this repo's documented order: load*, pv*, genset, battery, grid."""
    parts = []
    for cls in (LoadModule, RenewableModule, GensetModule, BatteryModule, GridModule):
        name = mod_name(m, cls)
        if name is not None and name in obs:
            for o in obs[name]:
                parts.append(np.asarray(o, dtype=np.float64).reshape(-1))
    return np.concatenate(parts) if parts else np.zeros(0)


def post_state(m):
    bats, gens = find(m, BatteryModule), find(m, GensetModule)
    ch = float(bats[0].current_charge) if bats else np.nan
    soc = float(bats[0].soc) if bats else np.nan
    st = ([int(gens[0]._current_status), int(gens[0]._goal_status), int(gens[0]._steps_until_up),
           int(gens[0]._steps_until_down)] if gens else [0, 0, 0, 0])
    return ch, soc, st


def run_episode(m, p, actions, normalized=True, want_obs=False):
    K = actions.shape[0]
    reward = np.zeros(K); done = np.zeros(K, np.uint8)
    charge = np.zeros(K); soc = np.zeros(K); status = np.zeros((K, 4), np.int32)
    obs_rows = []
    for k in range(K):
        obs, r, d, _ = m.run(control_from_row(m, p, actions[k]), normalized=normalized)
        reward[k], done[k] = r, d
        charge[k], soc[k], status[k] = post_state(m)
        if want_obs:
            obs_rows.append(flat_obs(m, obs))
    res = dict(reward=reward, done=done, charge=charge, soc=soc, status=status, log=log_matrix(m))
    if want_obs:
        res["obs"] = np.stack(obs_rows)
    return res


def subsample_steps(K):
    idx = set(range(min(128, K))) | set(range(max(0, K - 64), K)) | set(range(0, K, 97))
    return np.array(sorted(idx), dtype=np.int64)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrays)} arrays")


# --------------------------------------------------------------------------------------------- #
def make_pymgrid25():
    inputs, runs = {}, {}
    for n in range(25):
        m = Microgrid.from_scenario(n)
        p = extract_params(m)
        scalars, arrays = split_params(p)
        inputs[f"s{n}_params"] = np.array(json.dumps(scalars))
        for k, v in arrays.items():
            inputs[f"s{n}_{k}"] = v
        K = p["final_step"] - p["initial_step"]            # 8759 steps to done
        A = action_dims(p)
        seed = 1000 + n
        actions = np.random.RandomState(seed).rand(K, A)   # regenerated from the seed by the tests
        res = run_episode(m, p, actions, normalized=True)
        assert res["done"][-1] == 1 and res["done"][:-1].sum() == 0
        idx = subsample_steps(K)
        runs[f"s{n}_seed"] = np.array(seed)
        runs[f"s{n}_reward"] = res["reward"]
        runs[f"s{n}_soc"] = res["soc"]
        runs[f"s{n}_charge_sub"] = res["charge"][idx]
        runs[f"s{n}_status"] = res["status"].astype(np.int8)
        runs[f"s{n}_log_idx"] = idx
        runs[f"s{n}_log_sub"] = res["log"][idx]
        runs[f"s{n}_log_colsum"] = np.nansum(res["log"], axis=0)
        print(f"scenario {n}: A={A} sum(reward)={res['reward'].sum():.6f}")
    inputs["log_names"] = np.array(LOG_NAMES)
    runs["log_names"] = np.array(LOG_NAMES)
    save("pymgrid25_inputs.npz", **inputs)
    save("pymgrid25_run.npz", **runs)


# --------------------------------------------------------------------------------------------- #
def make_discrete():
    out = {}
    mod_id = {GensetModule: 0, BatteryModule: 1, GridModule: 2}
    for n in range(25):
        env = DiscreteMicrogridEnv.from_scenario(n)
        p = extract_params(env)
        # the priority-list table: [n_actions, max_len, 2] (module id, action id), -1 padded
        table = -np.ones((len(env.actions_list), 4, 2), np.int32)
        for i, pl in enumerate(env.actions_list):
            for j, el in enumerate(pl):
                cls = type(env.modules[el.module[0]][el.module[1]])
                table[i, j] = (mod_id[cls], el.action)
        K = 400
        rs = np.random.RandomState(2000 + n)
        ids = rs.randint(0, env.action_space.n, size=K)
        A = action_dims(p)
        control = np.zeros((K, A)); reward = np.zeros(K); soc = np.zeros(K)
        for k in range(K):
            ctrl = env._get_action(int(ids[k]))
            c = 0
            if "genset" in p:
                control[k, c:c + 2] = np.asarray(ctrl[mod_name(env, GensetModule)][0], dtype=np.float64); c += 2
            if "battery" in p:
                control[k, c] = ctrl[mod_name(env, BatteryModule)][0]; c += 1
            if "grid" in p:
                control[k, c] = ctrl[mod_name(env, GridModule)][0]; c += 1
            _, reward[k], _, _ = env.step(int(ids[k]))
            soc[k] = post_state(env)[1]
        out[f"s{n}_table"] = table
        out[f"s{n}_ids"] = ids.astype(np.int32)
        out[f"s{n}_control"] = control
        out[f"s{n}_reward"] = reward
        out[f"s{n}_soc"] = soc
        print(f"discrete {n}: n_actions={env.action_space.n}")
    save("discrete.npz", **out)


# --------------------------------------------------------------------------------------------- #
def make_genset_fsm():
    rows = []
    for su, wd, init in itertools.product(range(5), range(5), (0, 1)):
        for seq in itertools.product((0, 1), repeat=7):
            g = GensetModule(running_min_production=10, running_max_production=100, genset_cost=1.0,
                             start_up_time=su, wind_down_time=wd, init_start_up=bool(init))
            for goal in seq:
                pre = (g._current_status, g._goal_status, g._steps_until_up, g._steps_until_down)
                nxt = g.next_status(goal)
                g.update_status(goal)
                post = (g._current_status, g._goal_status, g._steps_until_up, g._steps_until_down)
                rows.append((su, wd, goal, *map(int, pre), *map(int, post), int(nxt)))
    tab = np.unique(np.array(rows, dtype=np.int16), axis=0)
    # the same table for GensetModule(allow_abortion=False) (genset_module.py:78-88,291,328)
    rows = []
    for su, wd, init in itertools.product(range(5), range(5), (0, 1)):
        for seq in itertools.product((0, 1), repeat=7):
            g = GensetModule(running_min_production=10, running_max_production=100, genset_cost=1.0, start_up_time=su,
                             wind_down_time=wd, init_start_up=bool(init), allow_abortion=False)
            for goal in seq:
                pre = (g._current_status, g._goal_status, g._steps_until_up, g._steps_until_down)
                nxt = g.next_status(goal)
                g.update_status(goal)
                post = (g._current_status, g._goal_status, g._steps_until_up, g._steps_until_down)
                rows.append((su, wd, goal, *map(int, pre), *map(int, post), int(nxt)))
    tab_noabort = np.unique(np.array(rows, dtype=np.int16), axis=0)
    # fractional goal values: round-half-even (genset_module.py:281)
    fr = []
    for v in (0.0, 0.25, 0.5, 0.5000000001, 0.4999999999, 0.75, 1.0):
        g = GensetModule(10, 100, 1.0, start_up_time=0, wind_down_time=0, init_start_up=False)
        g.update_status(v)
        fr.append((v, int(g._current_status)))
    save("genset_fsm.npz", transitions=tab, transitions_no_abortion=tab_noabort, fractional=np.array(fr))
    print("genset transitions:", tab.shape, "without abortion:", tab_noabort.shape)


# --------------------------------------------------------------------------------------------- #
def make_obs():
    out = {}
    # G5: end-of-series padding. scenario 1 (genset+battery+weak grid, H=23), start late, run to done.
    for n, start in ((1, 8700), (0, 8730), (2, 8740)):
        m = Microgrid.from_scenario(n)
        m.initial_step = start
        obs0 = flat_obs(m, m.reset())
        p = extract_params(m)
        K = p["final_step"] - start
        actions = np.random.RandomState(3000 + n).rand(K, action_dims(p))
        res = run_episode(m, p, actions, want_obs=True)
        # two more steps past `done` (the series has 8760 rows; obs then contains only padding rows)
        extra = []
        for _ in range(1):
            o, r, d, _ = m.run(control_from_row(m, p, np.full(action_dims(p), 0.5)))
            extra.append(flat_obs(m, o))
        out[f"tail{n}_start"] = np.array(start)
        out[f"tail{n}_obs0"] = obs0
        out[f"tail{n}_obs"] = res["obs"]
        out[f"tail{n}_obs_extra"] = np.stack(extra)
        out[f"tail{n}_reward"] = res["reward"]
        out[f"tail{n}_done"] = res["done"]
        out[f"tail{n}_charge0"] = np.array(p["battery"]["charge"])
    # head-of-episode obs for every scenario (H=23): reset obs + 40 steps
    for n in range(25):
        m = Microgrid.from_scenario(n)
        p = extract_params(m)
        obs0 = flat_obs(m, m.reset())
        actions = np.random.RandomState(3100 + n).rand(40, action_dims(p))
        res = run_episode(m, p, actions, want_obs=True)
        out[f"head{n}_obs0"] = obs0
        out[f"head{n}_obs"] = res["obs"]
    # G8: reset semantics -- state persists, counter returns
    m = Microgrid.from_scenario(1)
    p = extract_params(m)
    actions = np.random.RandomState(3200).rand(30, action_dims(p))
    res1 = run_episode(m, p, actions[:15])
    obs_r = flat_obs(m, m.reset())
    ch, soc, st = post_state(m)
    res2 = run_episode(m, p, actions[15:])
    out["reset_obs"] = obs_r
    out["reset_state"] = np.array([ch, soc, *st])
    out["reset_reward1"] = res1["reward"]
    out["reset_reward2"] = res2["reward"]
    out["reset_soc2"] = res2["soc"]
    save("obs.npz", **out)


# --------------------------------------------------------------------------------------------- #
def draw_generated(rs, n_grids, T):
    """Parameter draw for generator-style grids (sizing rules of MicrogridGenerator.py:214-603, SURVEY App. B;
    the draw itself is this repo's, only the module arithmetic is the reference's)."""
    t = np.arange(T)
    grids = []
    for i in range(n_grids):
        peak = float(rs.randint(100, 100001))
        shape = 0.55 + 0.25 * np.sin(2 * np.pi * (t % 24) / 24 + rs.rand() * 6.28) + 0.2 * rs.rand(T)
        load = peak * shape / shape.max()
        pv_peak = peak * rs.randint(30, 151) / 100.0
        sun = np.clip(np.sin(np.pi * ((t % 24) - 6) / 12), 0, None) * (0.6 + 0.4 * rs.rand(T))
        pv = pv_peak * sun / max(sun.max(), 1e-9)
        cap = float(np.ceil(rs.choice([3, 4, 5]) * load.mean()))
        kind = i % 3          # 0: genset only, 1: grid only, 2: both
        g = dict(load=load, pv=pv,
                 battery=dict(min_capacity=0.2 * cap, max_capacity=cap, max_charge=float(np.ceil(cap / 4)),
                              max_discharge=float(np.ceil(cap / 4)), efficiency=float(rs.choice([0.9, 0.95, 1.0, 0.8])),
                              battery_cost_cycle=float(rs.choice([0.02, 0.0, 0.05])),
                              init_soc=float(np.clip(rs.randn(), 0.2, 1.0))),
                 loss_load_cost=float(rs.choice([10.0, 8.5])), overgeneration_cost=float(rs.choice([1.0, 2.0])))
        if kind in (0, 2):
            rated = float(np.ceil(peak / 0.9))
            g["genset"] = dict(running_min_production=0.05 * rated, running_max_production=0.9 * rated,
                               genset_cost=0.4, co2_per_unit=2.0, cost_per_unit_co2=0.1,
                               start_up_time=int(rs.randint(0, 4)), wind_down_time=int(rs.randint(0, 4)),
                               init_start_up=bool(rs.randint(0, 2)))
        if kind in (1, 2):
            price = np.where((t % 24 >= 17) & (t % 24 < 21), 0.59, np.where((t % 24 >= 8), 0.29, 0.22))
            status = np.ones(T)
            if rs.rand() < 0.5:   # weak grid: random outages
                k = 0
                while k < T:
                    if rs.rand() < 0.03:
                        d = rs.randint(1, 6); status[k:k + d] = 0; k += d
                    k += 1
            ts = np.stack([price, 0.1 * price * (i % 2), 0.2 + 0.3 * rs.rand(T), status], axis=1)
            g["grid"] = dict(max_import=2 * peak, max_export=2 * peak, cost_per_unit_co2=0.1, ts=ts)
        grids.append(g)
    return grids


def build_reference_grid(g, horizon):
    fc = dict(forecaster="oracle", forecast_horizon=horizon) if horizon else dict()
    mods = [("load", LoadModule(time_series=g["load"], **fc)),
            ("pv", RenewableModule(time_series=g["pv"], **fc))]
    if "genset" in g:
        mods.append(("genset", GensetModule(**g["genset"])))
    b = g["battery"]
    mods.append(("battery", BatteryModule(**b)))
    if "grid" in g:
        q = g["grid"]
        mods.append(("grid", GridModule(max_import=q["max_import"], max_export=q["max_export"],
                                        time_series=q["ts"], cost_per_unit_co2=q["cost_per_unit_co2"], **fc)))
    return Microgrid(mods, loss_load_cost=g["loss_load_cost"], overgeneration_cost=g["overgeneration_cost"])


def make_generated():
    out = {}
    T, K, n_grids = 400, 300, 48
    rs = np.random.RandomState(4000)
    grids = draw_generated(rs, n_grids, T)
    meta = []
    for i, g in enumerate(grids):
        horizon = 24 if i % 4 == 3 else 0
        m = build_reference_grid(g, horizon)
        p = extract_params(m)
        A = action_dims(p)
        normalized = (i % 2 == 0)
        a = np.random.RandomState(4100 + i).rand(K, A)
        if not normalized:     # raw (unnormalised) controls, including out-of-range requests
            c = 0
            if "genset" in p:
                # (a negative genset request is an error in the reference: as_sink on a pure source)
                a[:, c + 1] = np.maximum(a[:, c + 1] * 1.3 - 0.1, 0.0) * p["genset"]["running_max_production"]; c += 2
            if "battery" in p:
                a[:, c] = (a[:, c] * 2 - 1) * 1.5 * p["battery"]["max_charge"]; c += 1
            if "grid" in p:
                a[:, c] = (a[:, c] * 2 - 1) * 1.2 * p["grid"]["max_import"]; c += 1
            a[::17, :] = 0.0   # exact zeros: the x == 0 routing (base_module.py:166-171)
        obs0 = flat_obs(m, m.reset())
        res = run_episode(m, p, a, normalized=normalized, want_obs=(horizon > 0 or i < 6))
        scalars, arrays = split_params(p)
        scalars["normalized"] = normalized
        meta.append(scalars)
        for k, v in arrays.items():
            out[f"g{i}_{k}"] = v
        out[f"g{i}_actions"] = a
        out[f"g{i}_reward"] = res["reward"]
        out[f"g{i}_soc"] = res["soc"]
        out[f"g{i}_charge"] = res["charge"]
        out[f"g{i}_status"] = res["status"].astype(np.int8)
        out[f"g{i}_log"] = res["log"][:, :]
        out[f"g{i}_obs0"] = obs0
        if "obs" in res:
            out[f"g{i}_obs"] = res["obs"][::5]
    out["meta"] = np.array(json.dumps(meta))
    out["log_names"] = np.array(LOG_NAMES)
    save("generated.npz", **out)


# --------------------------------------------------------------------------------------------- #
def make_loadpv():
    """Load/PV-only grids with 1..9 modules of each (reference tests/microgrid/test_microgrid.py:188-427)."""
    out = {}
    T = 100
    case = 0
    for n_load, n_pv in ((1, 1), (2, 1), (1, 2), (3, 3), (5, 2), (9, 9), (7, 1), (1, 8)):
        rs = np.random.RandomState(5000 + case)
        loads = [60 * rs.rand(T) for _ in range(n_load)]
        pvs = [50 * rs.rand(T) * (rs.rand(T) > 0.3) for _ in range(n_pv)]
        mods = [("load", LoadModule(time_series=x)) for x in loads] + \
               [("pv", RenewableModule(time_series=x)) for x in pvs]
        m = Microgrid(mods, loss_load_cost=10.0, overgeneration_cost=2.0)
        p = extract_params(m)
        K = T - 1
        res = run_episode(m, p, np.zeros((K, 0)))
        out[f"c{case}_load_ts"] = p["load_ts"]
        out[f"c{case}_pv_ts"] = p["pv_ts"]
        out[f"c{case}_reward"] = res["reward"]
        out[f"c{case}_done"] = res["done"]
        out[f"c{case}_log"] = res["log"]
        case += 1
    out["n_cases"] = np.array(case)
    out["log_names"] = np.array(LOG_NAMES)
    save("loadpv.npz", **out)


# --------------------------------------------------------------------------------------------- #
def make_rbc():
    """RuleBasedControl.run (algos/rbc/rbc.py:64-93) on every pymgrid25 scenario for a full year, plus on the
    generator-style grids (shorter): the sorted priority list, per-step reward, final state."""
    from pymgrid.algos import RuleBasedControl
    out = {}
    mod_id = {GensetModule: 0, BatteryModule: 1, GridModule: 2}

    def plist(rbc):
        arr = -np.ones((3, 2), np.int32)
        for j, el in enumerate(rbc.priority_list):
            arr[j] = (mod_id[type(rbc.microgrid.modules[el.module[0]][el.module[1]])], el.action)
        return arr
    for n in range(25):
        m = Microgrid.from_scenario(n)
        rbc = RuleBasedControl(m)
        log = rbc.run()
        mg = rbc.microgrid
        out[f"s{n}_plist"] = plist(rbc)
        out[f"s{n}_reward"] = log[("balance", 0, "reward")].values.astype(np.float64)
        ch, soc, st = post_state(mg)
        out[f"s{n}_final"] = np.array([ch, soc, *st])
        out[f"s{n}_logsum"] = np.nansum(log_matrix(mg), axis=0)
        print(f"rbc {n}: steps={len(log)} cost={-out[f's{n}_reward'].sum():.2f} plist={out[f's{n}_plist'].tolist()}")
    # generator-style grids (timers, weak grids): same draw as make_generated
    T, n_grids = 400, 48
    grids = draw_generated(np.random.RandomState(4000), n_grids, T)
    for i, g in enumerate(grids):
        m = build_reference_grid(g, 0)
        rbc = RuleBasedControl(m)
        log = rbc.run()
        out[f"g{i}_plist"] = plist(rbc)
        out[f"g{i}_reward"] = log[("balance", 0, "reward")].values.astype(np.float64)
        ch, soc, st = post_state(rbc.microgrid)
        out[f"g{i}_final"] = np.array([ch, soc, *st])
    out["log_names"] = np.array(LOG_NAMES)
    save("rbc.npz", **out)


# --------------------------------------------------------------------------------------------- #
def make_shaping():
    """Reward shapers (microgrid/reward_shaping/*.py) and trajectory functions (microgrid/trajectory/*.py)."""
    from pymgrid.microgrid.reward_shaping import BatteryDischargeShaper, PVCurtailmentShaper
    from pymgrid.microgrid.trajectory import (DeterministicTrajectory, FixedLengthStochasticTrajectory,
                                              StochasticTrajectory)
    out = {}
    K = 300
    for n in (1, 2, 0):
        # PV curtailment shaper under random normalised controls
        m = Microgrid.from_scenario(n)
        m.reward_shaping_func = PVCurtailmentShaper()
        p = extract_params(m)
        a = np.random.RandomState(6000 + n).rand(K, action_dims(p))
        a[::11] = np.round(a[::11])
        shaped = np.zeros(K)
        for k in range(K):
            _, shaped[k], _, _ = m.run(control_from_row(m, p, a[k]))
        log = m.get_log()
        out[f"shape_pv_{n}_shaped"] = shaped
        out[f"shape_pv_{n}_raw"] = log[("balance", 0, "reward")].values.astype(np.float64)
        assert np.array_equal(shaped, log[("balance", 0, "shaped_reward")].values)
        # Battery discharge shaper: it asserts a value in [-1, 1] (battery_discharge_shaper.py:33), which random
        # controls violate, so it is driven through the discrete env (priority-list controls never over-produce)
        env = DiscreteMicrogridEnv.from_scenario(n)
        env.reward_shaping_func = BatteryDischargeShaper()
        ids = np.random.RandomState(6050 + n).randint(0, env.action_space.n, size=K)
        shaped = np.zeros(K)
        for k in range(K):
            _, shaped[k], _, _ = env.step(int(ids[k]))
        log = env.get_log()
        out[f"shape_bat_{n}_ids"] = ids.astype(np.int32)
        out[f"shape_bat_{n}_shaped"] = shaped
        out[f"shape_bat_{n}_raw"] = log[("balance", 0, "reward")].values.astype(np.float64)
    # trajectories: windows drawn from numpy's global RNG at every reset (microgrid.py:221-225)
    for tag, func in (("det", DeterministicTrajectory(100, 160)), ("stoch", StochasticTrajectory()),
                      ("fixed", FixedLengthStochasticTrajectory(48))):
        m = Microgrid.from_scenario(2)
        m.trajectory_func = func
        p = extract_params(m)
        np.random.seed(777)
        windows, rewards, obs0 = [], [], []
        for ep in range(3):
            o = m.reset()
            lo = find(m, LoadModule)[0]
            windows.append((lo.initial_step, lo.final_step))
            obs0.append(flat_obs(m, o))
            rs = np.random.RandomState(6100 + ep)
            ep_r = []
            for k in range(400):
                _, r, d, _ = m.run(control_from_row(m, p, rs.rand(action_dims(p))))
                ep_r.append(r)
                if d:
                    break
            rewards.append(np.array(ep_r))
        out[f"traj_{tag}_windows"] = np.array(windows)
        for ep in range(3):
            out[f"traj_{tag}_reward{ep}"] = rewards[ep]
            out[f"traj_{tag}_obs0_{ep}"] = obs0[ep]
        print(tag, windows, [len(r) for r in rewards])
    save("shaping.npz", **out)


# --------------------------------------------------------------------------------------------- #
def make_obskeys():
    """BaseMicrogridEnv(observation_keys=...) (envs/base/base.py:109-163,211-223; tests/envs/test_discrete.py:82-95)."""
    out = {}
    keys = ["soc", "load_current", "import_price_current", "current_status", "renewable_forecast_3",
            "grid_status_forecast_0", "steps_until_down"]
    all_keys = keys
    for n in (1, 0, 2):
        m0 = Microgrid.from_scenario(n)
        have = set(m0.state_series().index.get_level_values(-1))
        keys = [k for k in all_keys if k in have]                  # a key absent from the state is a NameError
        out[f"s{n}_keys"] = np.array(keys)
        env = DiscreteMicrogridEnv.from_scenario(n, observation_keys=keys)
        obs0 = np.asarray(env.reset(), dtype=np.float64)
        ids = np.random.RandomState(7000 + n).randint(0, env.action_space.n, size=30)
        rows = []
        for a in ids:
            o, r, d, _ = env.step(int(a))
            rows.append(np.asarray(o, dtype=np.float64))
        out[f"s{n}_obs0"] = obs0
        out[f"s{n}_obs"] = np.stack(rows)
        out[f"s{n}_ids"] = ids.astype(np.int32)
        ss = env.state_series(normalized=True)
        import pandas as pd
        sel = ss.loc[pd.IndexSlice[:, :, keys]]
        out[f"s{n}_key_order"] = np.array([k[2] for k in sel.index])
        print(n, obs0.shape, list(out[f"s{n}_key_order"]))
    save("obskeys.npz", **out)


def make_order():
    """Module-list ORDER: the controllable sweep is pure sources first, then sources-and-sinks in the order of the
    module list (module_container.py:355-413).  Generator-style grids whose GridModule is listed BEFORE the
    BatteryModule: random normalised + raw actions (rewards, state, log rows) and DiscreteMicrogridEnv (priority-list
    table in the reference's enumeration order, expanded controls, rewards)."""
    T, n_grids = 300, 8
    grids = [g for g in draw_generated(np.random.RandomState(9100), 40, T) if "grid" in g][:n_grids]
    assert len(grids) == n_grids and any("genset" in g for g in grids) and any("genset" not in g for g in grids)
    mod_id = {GensetModule: 0, BatteryModule: 1, GridModule: 2}
    kind = {GensetModule: "genset", BatteryModule: "battery", GridModule: "grid"}
    out, meta = {}, []

    def build(g):
        q, b = g["grid"], g["battery"]
        mods = [("load", LoadModule(time_series=g["load"])), ("pv", RenewableModule(time_series=g["pv"])),
                ("grid", GridModule(max_import=q["max_import"], max_export=q["max_export"], time_series=q["ts"],
                                    cost_per_unit_co2=q["cost_per_unit_co2"])),
                ("battery", BatteryModule(**b))]
        if "genset" in g:
            mods.append(("genset", GensetModule(**g["genset"])))
        return Microgrid(mods, loss_load_cost=g["loss_load_cost"], overgeneration_cost=g["overgeneration_cost"])
    for i, g in enumerate(grids):
        m = build(g)
        p = extract_params(m)
        p["controllable_order"] = [kind[type(lst[0])] for _, lst in m.controllable.iterdict()]
        assert p["controllable_order"].index("grid") < p["controllable_order"].index("battery")
        scalars, arrays = split_params(p)
        meta.append(scalars)
        out.update({f"g{i}_{k}": v for k, v in arrays.items()})
        rs = np.random.RandomState(9200 + i)
        K = T - 1
        acts = rs.rand(K, action_dims(p)) * 1.3 - 0.15          # some requests outside [0, 1]: clipped by the modules
        if "genset" in p:
            acts[:, 0] = rs.rand(K)                              # goal status must stay inside [0, 1]
            acts[:, 1] = np.clip(acts[:, 1], 0, None)
        m.reset()
        res = run_episode(m, p, acts, normalized=True)
        out[f"g{i}_actions"] = acts
        for k in ("reward", "charge", "soc", "status", "log"):
            out[f"g{i}_{k}"] = res[k]
        env = DiscreteMicrogridEnv.from_microgrid(build(g))
        table = -np.ones((len(env.actions_list), 4, 2), np.int32)
        for a, pl in enumerate(env.actions_list):
            for j, el in enumerate(pl):
                table[a, j] = (mod_id[type(env.modules[el.module[0]][el.module[1]])], el.action)
        ids = rs.randint(0, env.action_space.n, size=200)
        env.reset()
        out[f"g{i}_table"], out[f"g{i}_ids"] = table, ids.astype(np.int32)
        out[f"g{i}_disc_reward"] = np.array([env.step(int(a))[1] for a in ids], np.float64)
        print(f"order {i}: {p['controllable_order']} n_actions={env.action_space.n} reward sum {res['reward'].sum():.3f}")
    out["meta"] = np.array(json.dumps(meta))
    out["log_names"] = np.array(LOG_NAMES)
    save("order.npz", **out)


def make_helper():
    """The fixture microgrid of the reference's OWN test suite (tests/helpers/modular_microgrid.py:14-37: genset 10..50
    at cost 0.5, lossless 100-unit battery at SoC 0.5, renewable 50, load 60, import-only grid with unit prices) --
    rebuilt here through the public module constructors with the same arguments -- run (a) with 60 seeded random
    normalised actions, observations included, (b) by RuleBasedControl for 10 steps (tests/control/test_rbc.py:28-38),
    (c) through DiscreteMicrogridEnv with 40 random priority-list ids."""
    from pymgrid.algos import RuleBasedControl

    def build(T=100):
        return Microgrid([GensetModule(running_min_production=10, running_max_production=50, genset_cost=0.5),
                          BatteryModule(min_capacity=0, max_capacity=100, max_charge=50, max_discharge=50, efficiency=1.0,
                                        init_soc=0.5),
                          RenewableModule(time_series=50 * np.ones(T)),
                          LoadModule(time_series=60 * np.ones(T)),
                          GridModule(max_import=100, max_export=0, time_series=np.ones((T, 3)), raise_errors=True)])
    m = build()
    p = extract_params(m)
    scalars, arrays = split_params(p)
    out = {"meta": np.array(json.dumps(scalars))}
    out.update({f"in_{k}": v for k, v in arrays.items()})
    rs = np.random.RandomState(77)
    acts = rs.rand(60, action_dims(p))
    m.reset()
    res = run_episode(m, p, acts, normalized=True, want_obs=True)
    out["rand_actions"] = acts
    for k in ("reward", "done", "charge", "soc", "status", "obs", "log"):
        out[f"rand_{k}"] = res[k]
    rbc = RuleBasedControl(build())
    mod_id = {GensetModule: 0, BatteryModule: 1, GridModule: 2}
    out["rbc_plist"] = np.array([(mod_id[type(rbc.microgrid.modules[el.module[0]][el.module[1]])], el.action)
                                 for el in rbc.priority_list], np.int32)
    out["rbc_marginal_cost"] = np.array([el.marginal_cost for el in rbc.priority_list], np.float64)
    log = rbc.run(10)
    out["rbc_reward"] = log[("balance", 0, "reward")].values.astype(np.float64)
    out["rbc_final"] = np.array(post_state(rbc.microgrid)[:2])
    env = DiscreteMicrogridEnv.from_microgrid(build())
    ids = rs.randint(0, env.action_space.n, size=40)
    env.reset()
    r = []
    for a in ids:
        r.append(env.step(int(a))[1])
    out["disc_ids"], out["disc_reward"], out["disc_n"] = ids.astype(np.int32), np.array(r, np.float64), np.array(env.action_space.n)
    out["log_names"] = np.array(LOG_NAMES)
    save("helper_microgrid.npz", **out)
    print("helper: rand reward sum", res["reward"].sum(), "rbc", out["rbc_reward"], "plist", out["rbc_plist"].tolist())


if __name__ == "__main__":
    which = sys.argv[1:] or ["pymgrid25", "discrete", "genset_fsm", "obs", "generated", "loadpv", "rbc", "shaping", "obskeys", "helper", "order"]
    for w in which:
        globals()["make_" + w]()
