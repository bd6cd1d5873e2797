#!/usr/bin/env python3
"""Golden vectors for microgrids with SEVERAL modules of a kind (two gensets, three batteries, two grids, ...), produced by
the REAL reference (the container takes any number of modules per name: module_container.py:355-413; the sweep visits them in
list order, microgrid.py:255-314; priority lists range over module instances, priority_list.py:15-67).

Run in the build container only (needs /root/reference):   python tests/golden/make_multi_goldens.py
Writes tests/golden/multi.npz: inputs (parameters, series, seeded actions, priority-list ids) and what the reference computed
from them (rewards, done, every log column of every module instance, observations, states, expanded controls).
"""
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
from pymgrid.algos import RuleBasedControl  # noqa: E402
from pymgrid.envs import DiscreteMicrogridEnv  # noqa: E402
from pymgrid.modules import (BatteryModule, GensetModule, GridModule, LoadModule,  # noqa: E402
                             RenewableModule, UnbalancedEnergyModule)

KINDS = (("genset", GensetModule), ("battery", BatteryModule), ("grid", GridModule))
COMMON = [("reward", "balance", "reward"),
          ("fixed_provided", "balance", "fixed_provided_to_microgrid"),
          ("fixed_absorbed", "balance", "fixed_absorbed_from_microgrid"),
          ("controllable_provided", "balance", "controllable_provided_to_microgrid"),
          ("controllable_absorbed", "balance", "controllable_absorbed_from_microgrid"),
          ("overall_provided", "balance", "overall_provided_to_microgrid"),
          ("overall_absorbed", "balance", "overall_absorbed_from_microgrid"),
          ("load_met", "load", "load_met"), ("renewable_used", "pv", "renewable_used"), ("curtailment", "pv", "curtailment"),
          ("loss_load", "balancing", "loss_load"), ("overgeneration", "balancing", "overgeneration"),
          ("unbalanced_reward", "balancing", "reward")]
PER_KIND = {"genset": [("genset_production", "genset_production"), ("genset_co2_production", "co2_production"),
                       ("genset_reward", "reward"), ("gen_cur", "current_status"), ("gen_goal", "goal_status"),
                       ("gen_up", "steps_until_up"), ("gen_down", "steps_until_down")],
            "battery": [("discharge_amount", "discharge_amount"), ("charge_amount", "charge_amount"),
                        ("battery_reward", "reward"), ("soc_pre", "soc"), ("charge_pre", "current_charge")],
            "grid": [("grid_import", "grid_import"), ("grid_export", "grid_export"),
                     ("grid_co2_production", "co2_production"), ("grid_reward", "reward")]}


def inst_name(name, j):
    return name if j == 0 else f"{name}[{j}]"


def draw_case(rs, T, n_gen, n_bat, n_grid, n_load, n_pv, horizon, order):
    t = np.arange(T)
    g = dict(T=T, horizon=horizon, order=order,
             load=[float(rs.randint(50, 400)) * (0.5 + 0.5 * rs.rand(T)) for _ in range(n_load)],
             pv=[float(rs.randint(30, 300)) * rs.rand(T) * (rs.rand(T) > 0.3) for _ in range(n_pv)],
             loss_load_cost=10.0, overgeneration_cost=float(rs.choice([1.0, 2.0])), genset=[], battery=[], grid=[])
    for _ in range(n_gen):
        rated = float(rs.randint(60, 300))
        g["genset"].append(dict(running_min_production=float(rs.choice([0.0, 0.05, 0.2])) * rated, running_max_production=0.9 * rated,
                                genset_cost=float(rs.choice([0.3, 0.4, 0.55])), co2_per_unit=2.0, cost_per_unit_co2=0.1,
                                start_up_time=int(rs.randint(0, 3)), wind_down_time=int(rs.randint(0, 3)),
                                init_start_up=bool(rs.randint(0, 2))))
    for _ in range(n_bat):
        cap = float(rs.randint(80, 500))
        g["battery"].append(dict(min_capacity=0.2 * cap, max_capacity=cap, max_charge=float(np.ceil(cap / 4)),
                                 max_discharge=float(np.ceil(cap / 3)), efficiency=float(rs.choice([0.9, 0.95, 1.0, 0.8])),
                                 battery_cost_cycle=float(rs.choice([0.02, 0.0, 0.05])),
                                 init_soc=float(np.clip(0.6 + 0.3 * rs.randn(), 0.2, 1.0))))
    for q in range(n_grid):
        price = np.where((t % 24 >= 17) & (t % 24 < 21), 0.59, np.where((t % 24 >= 8), 0.29, 0.22)) * (1 + 0.3 * q)
        status = (rs.rand(T) > 0.1 * (1 + q)).astype(float)
        ts = np.stack([price, 0.4 * price * ((q + 1) % 2), 0.2 + 0.3 * rs.rand(T), status], axis=1)
        g["grid"].append(dict(max_import=float(rs.randint(100, 400)), max_export=float(rs.randint(50, 300)),
                              cost_per_unit_co2=0.1, ts=ts))
    return g


def build(g):
    fc = dict(forecaster="oracle", forecast_horizon=g["horizon"]) if g["horizon"] else dict()
    by_kind = {
        "load": [("load", LoadModule(time_series=x, **fc)) for x in g["load"]],
        "pv": [("pv", RenewableModule(time_series=x, **fc)) for x in g["pv"]],
        "genset": [("genset", GensetModule(**q)) for q in g["genset"]],
        "battery": [("battery", BatteryModule(**q)) for q in g["battery"]],
        "grid": [("grid", GridModule(max_import=q["max_import"], max_export=q["max_export"], time_series=q["ts"],
                                     cost_per_unit_co2=q["cost_per_unit_co2"], **fc)) for q in g["grid"]],
    }
    mods = [m for kind in g["order"] for m in by_kind[kind]]
    return mods


def control_of(m, counts, row):
    ctrl, c = {}, 0
    if counts["genset"]:
        ctrl["genset"] = [np.array([row[c + 2 * j], row[c + 2 * j + 1]]) for j in range(counts["genset"])]
        c += 2 * counts["genset"]
    for kind in ("battery", "grid"):
        if counts[kind]:
            ctrl[kind] = [float(row[c + j]) for j in range(counts[kind])]
            c += counts[kind]
    return ctrl


def flat_obs(obs):
    parts = []
    for name in ("load", "pv", "genset", "battery", "grid"):
        for o in obs.get(name, []):
            parts.append(np.asarray(o, dtype=np.float64).reshape(-1))
    return np.concatenate(parts)


def log_names(counts):
    names = [c[0] for c in COMMON]
    for kind, _ in KINDS:
        for j in range(counts[kind]):
            names += [inst_name(n, j) for n, _ in PER_KIND[kind]]
    return names


def log_matrix(m, counts):
    log = m.get_log()
    n = len(log)
    cols = []
    for _, mod, field in COMMON:          # load / pv columns: summed over the instances in module order
        acc = np.zeros(n)
        for c in [c for c in log.columns if c[0] == mod and c[2] == field]:
            acc = acc + log[c].values.astype(np.float64)
        cols.append(acc)
    for kind, _ in KINDS:
        for j in range(counts[kind]):
            for _, field in PER_KIND[kind]:
                cols.append(log[(kind, j, field)].values.astype(np.float64))
    return np.stack(cols, axis=1)


def states(m, counts):
    ch = [float(b.current_charge) for b in m.modules["battery"]] if counts["battery"] else []
    soc = [float(b.soc) for b in m.modules["battery"]] if counts["battery"] else []
    st = [[int(g._current_status), int(g._goal_status), int(g._steps_until_up), int(g._steps_until_down)]
          for g in m.modules["genset"]] if counts["genset"] else []
    return ch, soc, st


def main():
    cases = [  # n_gen, n_bat, n_grid, n_load, n_pv, horizon, module-list order
        (2, 2, 1, 1, 1, 0, ("load", "pv", "genset", "battery", "grid")),
        (1, 3, 2, 2, 2, 3, ("load", "pv", "genset", "battery", "grid")),
        (3, 1, 0, 1, 1, 0, ("load", "pv", "genset", "battery", "grid")),
        (0, 2, 2, 1, 2, 2, ("load", "pv", "genset", "grid", "battery")),          # grids before the batteries
        (4, 3, 2, 3, 2, 0, ("load", "pv", "genset", "battery", "grid")),          # lists of >= 8 addends: pairwise np.sum
        (2, 0, 1, 1, 1, 1, ("load", "pv", "genset", "battery", "grid")),
        (1, 2, 0, 1, 1, 0, ("battery", "load", "genset", "pv", "grid")),
    ]
    out, meta = {}, []
    T, K = 96, 90
    for ci, (n_gen, n_bat, n_grid, n_load, n_pv, H, order) in enumerate(cases):
        rs = np.random.RandomState(7000 + ci)
        g = draw_case(rs, T, n_gen, n_bat, n_grid, n_load, n_pv, H, order)
        counts = dict(genset=n_gen, battery=n_bat, grid=n_grid)
        A = 2 * n_gen + n_bat + n_grid
        for normalized in (True, False):
            m = Microgrid(build(g), loss_load_cost=g["loss_load_cost"], overgeneration_cost=g["overgeneration_cost"])
            a = np.random.RandomState(7100 + 2 * ci + int(normalized)).rand(K, A)
            a[::7, :] = np.round(a[::7, :])
            if not normalized:          # raw requests, some beyond the modules' limits
                c = 0
                for j in range(n_gen):
                    a[:, c + 1] = np.maximum(a[:, c + 1] * 1.3 - 0.1, 0.0) * g["genset"][j]["running_max_production"]; c += 2
                for j in range(n_bat):
                    a[:, c] = (a[:, c] * 2 - 1) * 1.5 * g["battery"][j]["max_charge"]; c += 1
                for j in range(n_grid):
                    a[:, c] = (a[:, c] * 2 - 1) * 1.2 * g["grid"][j]["max_import"]; c += 1
                a[::11, :] = 0.0
            tag = f"c{ci}_{'n' if normalized else 'r'}"
            out[f"{tag}_obs0"] = flat_obs(m.reset())
            reward, done, obs_rows, charge, soc, status = [], [], [], [], [], []
            for k in range(K):
                obs, r, d, _ = m.run(control_of(m, counts, a[k]), normalized=normalized)
                reward.append(r); done.append(d); obs_rows.append(flat_obs(obs))
                ch, so, st = states(m, counts)
                charge.append(ch); soc.append(so); status.append(st)
            out[f"{tag}_actions"] = a
            out[f"{tag}_reward"] = np.array(reward)
            out[f"{tag}_done"] = np.array(done, dtype=np.uint8)
            out[f"{tag}_obs"] = np.stack(obs_rows)
            out[f"{tag}_charge"] = np.array(charge, dtype=np.float64).reshape(K, n_bat)
            out[f"{tag}_soc"] = np.array(soc, dtype=np.float64).reshape(K, n_bat)
            out[f"{tag}_status"] = np.array(status, dtype=np.int32).reshape(K, n_gen, 4)
            out[f"{tag}_log"] = log_matrix(m, counts)
        out[f"c{ci}_log_names"] = np.array(log_names(counts))
        # discrete env: priority lists over module instances, expanded controls, rewards.  The reference enumerates all
        # permutations of (module instance, action) elements: 7 elements = 5 040 permutations is what it can do in seconds
        n_lists = 0
        if 2 * n_gen + n_bat + n_grid <= 7:
            n_lists = discrete_case(out, ci, g, n_gen, n_bat, n_grid, A)
        # inputs
        for j, x in enumerate(g["load"]):
            out[f"c{ci}_load_{j}"] = x
        for j, x in enumerate(g["pv"]):
            out[f"c{ci}_pv_{j}"] = x
        for j, q in enumerate(g["grid"]):
            out[f"c{ci}_grid_ts_{j}"] = q["ts"]
        meta.append(dict(T=T, horizon=H, order=list(order), loss_load_cost=g["loss_load_cost"],
                         overgeneration_cost=g["overgeneration_cost"], genset=g["genset"], battery=g["battery"],
                         grid=[{k: v for k, v in q.items() if k != "ts"} for q in g["grid"]], n_load=n_load, n_pv=n_pv,
                         n_lists=n_lists))
        print(f"case {ci}: gensets {n_gen} batteries {n_bat} grids {n_grid} loads {n_load} pvs {n_pv} H {H}: "
              f"{n_lists} priority lists, sum(reward) {np.sum(out[f'c{ci}_n_reward']):.4f}", flush=True)
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "multi.npz")
    np.savez_compressed(path, **out)
    print(f"multi.npz: {os.path.getsize(path) / 1e3:.0f} KB")


def discrete_case(out, ci, g, n_gen, n_bat, n_grid, A):
    if True:
        env = DiscreteMicrogridEnv(build(g), loss_load_cost=g["loss_load_cost"], overgeneration_cost=g["overgeneration_cost"])
        L = max(len(pl) for pl in env.actions_list)
        kind_id = {"genset": 0, "battery": 1, "grid": 2}
        table = -np.ones((len(env.actions_list), L, 3), np.int32)
        for i, pl in enumerate(env.actions_list):
            for j, el in enumerate(pl):
                table[i, j] = (kind_id[el.module[0]], el.module[1], el.action)
        Kd = 60
        ids = np.random.RandomState(7300 + ci).randint(0, env.action_space.n, size=Kd)
        control, dreward = np.zeros((Kd, A)), np.zeros(Kd)
        env.reset()
        for k in range(Kd):
            ctrl = env._get_action(int(ids[k]))
            c = 0
            for j in range(n_gen):
                control[k, c:c + 2] = np.asarray(ctrl["genset"][j], dtype=np.float64); c += 2
            for kind, n in (("battery", n_bat), ("grid", n_grid)):
                for j in range(n):
                    control[k, c] = ctrl[kind][j]; c += 1
            _, dreward[k], _, _ = env.step(int(ids[k]))
        out[f"c{ci}_pl_table"] = table
        out[f"c{ci}_ids"] = ids.astype(np.int32)
        out[f"c{ci}_control"] = control
        out[f"c{ci}_dreward"] = dreward
        # RuleBasedControl: the modules sorted by marginal cost (rbc.py:31-50), deployed for the whole series
        m = Microgrid(build(g), loss_load_cost=g["loss_load_cost"], overgeneration_cost=g["overgeneration_cost"])
        rbc = RuleBasedControl(m)
        out[f"c{ci}_rbc_list"] = np.array([(kind_id[el.module[0]], el.module[1], el.action) for el in rbc._priority_list], np.int32)
        log = rbc.run()
        out[f"c{ci}_rbc_reward"] = log[("balance", 0, "reward")].values.astype(np.float64)
        # reward shapers on several batteries / renewables (reward_shaping/*.py sum over the module instances); the balancing
        # module under the name the scenario files give it ("unbalanced_energy": the name BatteryDischargeShaper looks up)
        from pymgrid.microgrid.reward_shaping import BatteryDischargeShaper, PVCurtailmentShaper
        for tag, shaper in (("bat", BatteryDischargeShaper()), ("pv", PVCurtailmentShaper())):
            mods = build(g) + [("unbalanced_energy", UnbalancedEnergyModule(raise_errors=False, loss_load_cost=g["loss_load_cost"],
                                                                             overgeneration_cost=g["overgeneration_cost"]))]
            senv = DiscreteMicrogridEnv(mods, add_unbalanced_module=False, reward_shaping_func=shaper)
            senv.reset()
            sids = np.random.RandomState(7400 + ci).randint(0, senv.action_space.n, size=40)
            shaped = np.zeros(40)
            for k in range(40):
                _, shaped[k], _, _ = senv.step(int(sids[k]))
            out[f"c{ci}_shape_{tag}_ids"] = sids.astype(np.int32)
            out[f"c{ci}_shape_{tag}"] = shaped
        return int(env.action_space.n)


if __name__ == "__main__":
    main()
