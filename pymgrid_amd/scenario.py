"""Parameter-dict <-> npz helpers.

A *parameter dict* describes one microgrid: ``load_ts`` / ``pv_ts`` (and ``grid_ts``), optional ``battery`` /
``genset`` / ``grid`` sub-dicts, ``unbalanced`` costs, ``horizon``, ``initial_step``, ``final_step`` -- the content
of one ``!Microgrid`` scenario YAML of the reference (data/scenario/pymgrid25/microgrid_*/microgrid_*.yaml,
microgrid.py:874-908) after loading.
"""
import json

import numpy as np


def load_npz_grids(path, prefix_fmt="s{}_", count=None):
    """Read parameter dicts stored as ``<prefix>params`` (JSON scalars) + ``<prefix>load_ts`` ... arrays."""
    z = np.load(path, allow_pickle=False)
    grids, n = [], 0
    while (count is None or n < count) and (prefix_fmt.format(n) + "params") in z.files:
        pre = prefix_fmt.format(n)
        p = json.loads(str(z[pre + "params"]))
        for k in ("load_ts", "pv_ts", "grid_ts"):
            if pre + k in z.files:
                p[k] = z[pre + k]
        grids.append(p)
        n += 1
    return grids


def architecture(p):
    """Module set of a parameter dict, e.g. ('genset', 'battery')."""
    return tuple(k for k in ("genset", "battery", "grid") if p.get(k) is not None)


def bucket_by_layout(grids):
    """Group microgrids that can share one SoA batch (same module set and multiplicities, sweep order, series length,
    horizon, window).
    Returns {key: [indices]} in first-seen order."""
    buckets = {}
    for i, p in enumerate(grids):
        from .batch import grid_first
        load, pv = np.asarray(p["load_ts"]), np.asarray(p["pv_ts"])
        from .batch import module_list
        hz = p.get("horizons") or {}
        key = (architecture(p), tuple(len(module_list(p.get(k))) for k in ("genset", "battery", "grid")), load.shape[0],
               int(p.get("horizon", 0)), tuple((k, tuple(v)) for k, v in sorted(hz.items())),
               int(p.get("initial_step", 0)), int(p.get("final_step", 0)), grid_first(p),
               1 if load.ndim == 1 else load.shape[1], 1 if pv.ndim == 1 else pv.shape[1])
        buckets.setdefault(key, []).append(i)
    return buckets


# ---------------------------------------------------------------------------------------------------------
# Scenario files of the reference: `!Microgrid` YAML + csv.gz arrays
#   Microgrid.from_scenario / load      microgrid/microgrid.py:847-893,958-980
#   BaseMicrogridModule.from_yaml       modules/base/base_module.py:771-799  (cls_params -> __init__, then state)
#   !NDArray constructor                utils/serialize.py:91-112            (pd.read_csv(path, index_col=0).values)
# ---------------------------------------------------------------------------------------------------------
_MODULE_TAGS = ("!LoadModule", "!RenewableModule", "!UnbalancedEnergyModule", "!Genset", "!BatteryModule",
                "!GridModule")


def _make_loader(base_dir):
    import os

    import pandas as pd
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    def ndarray(loader, node):
        if isinstance(node, yaml.SequenceNode):
            return np.array(loader.construct_sequence(node, deep=True))
        path = loader.construct_scalar(node)
        if not os.path.isabs(path):
            path = os.path.join(base_dir, path)
        return pd.read_csv(path, index_col=0).values

    def tagged(tag):
        def construct(loader, node):
            d = loader.construct_mapping(node, deep=True)
            d["__tag__"] = tag
            return d
        return construct

    Loader.add_constructor("!NDArray", ndarray)
    for tag in _MODULE_TAGS + ("!Microgrid", "!DiscreteMicrogridEnv"):
        Loader.add_constructor(tag, tagged(tag))

    # trajectory functions / reward shapers (yaml.YAMLObject subclasses in the reference: the mapping holds the instance's
    # attributes, microgrid/trajectory/*.py, microgrid/reward_shaping/*.py) -> this package's mirrors of those classes
    from . import trajectory as tj

    def obj(cls, *fields):
        def construct(loader, node):
            d = loader.construct_mapping(node, deep=True) if isinstance(node, yaml.MappingNode) else {}
            return cls(*[d[f] for f in fields])
        return construct
    Loader.add_constructor("!DeterministicTrajectory", obj(tj.DeterministicTrajectory, "initial_step", "final_step"))
    Loader.add_constructor("!StochasticTrajectory", obj(tj.StochasticTrajectory))
    Loader.add_constructor("!FixedLengthStochasticTrajectory", obj(tj.FixedLengthStochasticTrajectory, "trajectory_length"))
    Loader.add_constructor("!PVCurtailmentShaper", obj(tj.PVCurtailmentShaper))
    Loader.add_constructor("!BatteryDischargeShaper", obj(tj.BatteryDischargeShaper))
    return Loader


def load_scenario_yaml(path):
    """Read one serialised microgrid (``Microgrid.dump`` format, e.g. data/scenario/pymgrid25/microgrid_3/
    microgrid_3.yaml) into a parameter dict, applying the reference's deserialisation rules: constructor arguments
    from ``cls_params``, then the state attributes (battery: ``soc`` then ``current_charge`` setters,
    battery_module.py:356-362 -> charge = current_charge, soc = charge / max_capacity; genset: the four private
    status fields, genset_module.py:426-427).

    Beyond the plain vocabulary the dict may carry: ``trajectory_func`` / ``reward_shaping_func`` (instances of this package's
    mirrors of the reference's classes; the N = 1 envs take them as defaults), ``raise_errors`` (True when any module was built
    with it: the envs' default), ``horizons`` ({"load": [h per module], "pv": [...], "grid": [...]} when the time-series modules do
    not share one forecast horizon: ``horizon`` is then their maximum and the envs drop the columns a module does not have)."""
    import os

    import yaml
    with open(path) as fh:
        doc = yaml.load(fh, Loader=_make_loader(os.path.dirname(os.path.abspath(path))))
    if doc.get("__tag__") not in ("!Microgrid", "!DiscreteMicrogridEnv"):
        raise ValueError(f"{path}: not a !Microgrid document")
    return params_from_module_docs(doc, path)


def params_from_module_docs(doc, path="<modules>"):
    """The parameter dict of a microgrid described as ``{"modules": [(name, {"__tag__", "cls_params", "state"}), ...],
    "trajectory_func": .., "reward_shaping_func": ..}`` -- what a serialised ``!Microgrid`` document holds (load_scenario_yaml) and
    what a list of this package's module descriptions (modules.py: ``Microgrid([BatteryModule(...), ...])``) turns into."""
    p = {"load_ts": [], "pv_ts": [], "grid_ts": [], "grid": [], "genset": [], "battery": []}
    for key in ("trajectory_func", "reward_shaping_func"):           # (Microgrid._serialization_data writes the former only)
        if doc.get(key) is not None:
            if isinstance(doc[key], dict):
                raise ValueError(f"{path}: {key} {doc[key].get('__tag__')!r} is not one of the reference's classes")
            p[key] = doc[key]
    ts_meta, horizons = [], {"load": [], "pv": [], "grid": []}
    raise_errors = False
    order = []                                          # controllable modules in list order (module_container.py:355-413)
    for name, mod in doc["modules"]:
        tag, cp, state = mod["__tag__"], mod["cls_params"], mod.get("state", {})
        if tag in ("!Genset", "!BatteryModule", "!GridModule"):
            kind = {"!Genset": "genset", "!BatteryModule": "battery", "!GridModule": "grid"}[tag]
            if kind not in order:
                order.append(kind)
        raise_errors = raise_errors or bool(cp.get("raise_errors"))     # base_module.py:79-93: the envs' dry-run check (mgx_check_step)
        if tag in ("!LoadModule", "!RenewableModule", "!GridModule"):
            fc = cp.get("forecaster")
            if isinstance(fc, (int, float)) and not isinstance(fc, bool):         # GaussianNoiseForecaster
                noise = dict(std=float(fc), relative_noise=bool(cp.get("forecaster_relative_noise", False)),
                             increase_uncertainty=bool(cp.get("forecaster_increase_uncertainty", False)))
                if p.setdefault("forecast_noise", noise) != noise:
                    raise NotImplementedError("different noisy forecasters per module are not supported")
            elif fc not in (None, "oracle"):
                raise NotImplementedError(f"forecaster {fc!r}: only None, 'oracle' and a noise std are supported")
            horizon = int(cp.get("forecast_horizon", 0)) if fc is not None else 0   # base_timeseries_module.py:40
            ts = np.asarray(cp["time_series"], dtype=np.float64)
            final = int(cp.get("final_step", -1))
            final = ts.shape[0] if final <= 0 else final
            # the constructor's initial_step and the saved step counter are two things (Microgrid.from_yaml builds the module
            # from cls_params and only then restores _current_step, base_module.py:771-957): reset() goes back to the former
            init = int(cp.get("initial_step", 0))
            ts_meta.append((final, init, int(state.get("_current_step", init)), ts.shape[0]))
            horizons[{"!LoadModule": "load", "!RenewableModule": "pv", "!GridModule": "grid"}[tag]].append(horizon)
            if tag == "!LoadModule":
                p["load_ts"].append(-np.abs(ts.reshape(ts.shape[0], -1)[:, 0]))
            elif tag == "!RenewableModule":
                p["pv_ts"].append(np.abs(ts.reshape(ts.shape[0], -1)[:, 0]))
            else:
                p["grid_ts"].append(ts)
                p["grid"].append(dict(max_import=float(cp["max_import"]), max_export=float(cp["max_export"]),
                                      cost_per_unit_co2=float(cp.get("cost_per_unit_co2", 0.0))))
        elif tag == "!UnbalancedEnergyModule":
            p["unbalanced"] = dict(loss_load_cost=float(cp["loss_load_cost"]),
                                   overgeneration_cost=float(cp["overgeneration_cost"]))
        elif tag == "!Genset":
            su, wd = int(cp.get("start_up_time", 0)), int(cp.get("wind_down_time", 0))
            on = int(bool(cp.get("init_start_up", True)))
            status = [on, on, 0, wd] if on else [0, 0, su, 0]                   # genset_module.py:91-92,216-227
            if "_current_status" in state:
                status = [int(state["_current_status"]), int(state["_goal_status"]),
                          int(state["_steps_until_up"]), int(state["_steps_until_down"])]
            p["genset"].append(dict(running_min_production=float(cp["running_min_production"]),
                                    running_max_production=float(cp["running_max_production"]),
                                    genset_cost=float(cp["genset_cost"]), co2_per_unit=float(cp.get("co2_per_unit", 0.0)),
                                    cost_per_unit_co2=float(cp.get("cost_per_unit_co2", 0.0)),
                                    start_up_time=su, wind_down_time=wd, status=status))
            if not cp.get("allow_abortion", True):
                p["genset"][-1]["allow_abortion"] = False
        elif tag == "!BatteryModule":
            if cp.get("battery_transition_model") is not None:
                raise NotImplementedError("custom battery_transition_model is not supported")
            cap = float(cp["max_capacity"])
            if "current_charge" in state:                # the setters of a restored state: soc = charge / max_capacity
                charge = float(state["current_charge"])
                soc = charge / cap
            elif cp.get("init_charge") is not None:      # the constructor's rules (battery_module.py:96-106)
                charge = float(cp["init_charge"])
                soc = charge / cap
            else:
                soc = float(cp["init_soc"])
                charge = soc * cap
            p["battery"].append(dict(min_capacity=float(cp["min_capacity"]), max_capacity=cap,
                                     max_charge=float(cp["max_charge"]), max_discharge=float(cp["max_discharge"]),
                                     efficiency=float(cp["efficiency"]),
                                     battery_cost_cycle=float(cp.get("battery_cost_cycle", 0.0)),
                                     charge=charge, soc=soc))
    # one module of a kind: the plain vocabulary (a dict, a [T] series); several: lists / [T, n] (module_container.py:355-413
    # keeps a list per name)
    if not ts_meta:         # (the reference cannot build such a microgrid either: Microgrid.__init__ -> get_attrs('final_step') finds no value)
        raise ValueError(f"{path}: no time-series module (No values found for key(s) ['final_step'])")
    # the reference refuses modules that disagree about the window (Microgrid.__init__ -> get_attrs(..., unique=True):
    # module_container.py:97-180 raises ValueError); a saved microgrid's modules all stand at the same counter (Microgrid.run steps them together)
    if len(set(ts_meta)) != 1:
        raise ValueError(f"{path}: the time-series modules disagree about (final_step, initial_step, current step, length): {sorted(set(ts_meta))}")
    T = ts_meta[0][3]
    for key in ("load_ts", "pv_ts"):            # a microgrid without a LoadModule / RenewableModule: a [T, 0] series (the general kernels)
        cols = p[key]
        p[key] = cols[0] if len(cols) == 1 else (np.stack(cols, axis=1) if cols else np.zeros((T, 0)))
    for key in ("genset", "battery", "grid", "grid_ts"):
        if not p[key]:
            del p[key]
        elif len(p[key]) == 1:
            p[key] = p[key][0]
    if "unbalanced" not in p:
        raise ValueError("scenario has no UnbalancedEnergyModule")
    p["final_step"], p["initial_step"], p["current_step"] = ts_meta[0][:3]
    hs = [h for v in horizons.values() for h in v]
    p["horizon"] = max(hs) if hs else 0
    if len(set(hs)) > 1:                        # forecast_horizon is per module (base_timeseries_module.py:40): keep who has what
        p["horizons"] = {k: list(v) for k, v in horizons.items() if v}
    if raise_errors:
        p["raise_errors"] = True
    p["controllable_order"] = order
    return p


def dump_scenario_yaml(p, path):
    """Write a parameter dict as a serialised microgrid in the reference's format (``Microgrid.dump``: microgrid.py:820-846,
    ``!Microgrid`` YAML + ``data/cls_params/<Module>/time_series.csv.gz`` next to it, utils/serialize.py:24-83) so that
    ``pymgrid.Microgrid.load(open(path))`` -- and ``load_scenario_yaml(path)`` -- give the same microgrid back: constructor
    arguments under ``cls_params``, the dynamic state (battery charge / SoC, the four genset status fields, the step
    counter) under ``state``.  The controllable modules are listed in ``controllable_order``; several modules of a kind are
    written one after the other (the series of instance j > 0 under ``data/cls_params/<Module>_<j>/``: the reference's own
    dump writes every instance of a class to the same file, utils/serialize.py:33-41)."""
    import os

    import pandas as pd
    import yaml
    from .batch import grid_series_list, module_list
    load, pv = np.asarray(p["load_ts"], dtype=np.float64), np.asarray(p["pv_ts"], dtype=np.float64)
    load, pv = load.reshape(load.shape[0], -1), pv.reshape(pv.shape[0], -1)
    base = os.path.dirname(os.path.abspath(path))
    H, t0, final = int(p.get("horizon", 0)), int(p.get("initial_step", 0)), int(p.get("final_step", 0)) or load.shape[0]
    cur = int(p.get("current_step", t0))               # the saved step counter (state), not the constructor's initial_step
    noise = p.get("forecast_noise")

    def series(tag, arr, j=0):
        rel = os.path.join("data", "cls_params", tag if j == 0 else f"{tag}_{j}", "time_series.csv.gz")
        os.makedirs(os.path.dirname(os.path.join(base, rel)), exist_ok=True)
        pd.DataFrame(np.asarray(arr, dtype=np.float64).reshape(arr.shape[0], -1)).to_csv(os.path.join(base, rel))
        return _Tagged("!NDArray", rel)

    hz, rerr = p.get("horizons") or {}, bool(p.get("raise_errors", False))

    def ts_params(tag, arr, extra=None, j=0):
        kind = {"LoadModule": "load", "RenewableModule": "pv", "GridModule": "grid"}[tag]
        h = int(hz[kind][j]) if kind in hz else H           # forecast_horizon is per module (base_timeseries_module.py:40)
        fc = (float(noise["std"]) if noise else "oracle") if h > 0 else None
        d = dict(final_step=final, forecast_horizon=h, forecaster=fc,
                 forecaster_increase_uncertainty=bool(noise and noise.get("increase_uncertainty", False)),
                 forecaster_relative_noise=bool(noise and noise.get("relative_noise", False)),
                 initial_step=t0, raise_errors=rerr, time_series=series(tag, arr, j))
        d.update(extra or {})
        return d

    def module(name, tag, cls_params, state, j=0):
        return [name, _Tagged(tag, dict(cls_params=cls_params, name=[name, j], state=dict(state, _current_step=cur)))]
    mods = [module("load", "!LoadModule", ts_params("LoadModule", np.abs(load[:, j]), j=j), {}, j) for j in range(load.shape[1])]
    mods += [module("pv", "!RenewableModule", ts_params("RenewableModule", np.abs(pv[:, j]),
                                                        dict(provided_energy_name="renewable_used"), j=j), {}, j)
             for j in range(pv.shape[1])]
    mods.append(module("unbalanced_energy", "!UnbalancedEnergyModule",
                       dict(initial_step=t0, loss_load_cost=float(p["unbalanced"]["loss_load_cost"]),
                            overgeneration_cost=float(p["unbalanced"]["overgeneration_cost"]), raise_errors=rerr), {}))
    order = [k for k in (p.get("controllable_order") or []) if p.get(k) is not None]
    order += [k for k in ("genset", "battery", "grid") if p.get(k) is not None and k not in order]
    grid_series = grid_series_list(p)
    for kind, j, q in [(kind, j, q) for kind in order for j, q in enumerate(module_list(p[kind]))]:
        if kind == "genset":
            su, wd = int(q.get("start_up_time", 0)), int(q.get("wind_down_time", 0))
            if q.get("status") is not None:
                st = [int(v) for v in q["status"]]
            else:
                on = int(bool(q.get("init_start_up", True)))
                st = [on, on, 0, wd] if on else [0, 0, su, 0]
            mods.append(module("genset", "!Genset", dict(
                allow_abortion=bool(q.get("allow_abortion", True)), co2_per_unit=float(q.get("co2_per_unit", 0.0)),
                cost_per_unit_co2=float(q.get("cost_per_unit_co2", 0.0)), genset_cost=float(q["genset_cost"]),
                init_start_up=bool(st[0]), initial_step=t0, provided_energy_name="genset_production", raise_errors=rerr,
                running_max_production=float(q["running_max_production"]),
                running_min_production=float(q["running_min_production"]), start_up_time=su, wind_down_time=wd),
                dict(_current_status=st[0], _goal_status=st[1], _steps_until_up=st[2], _steps_until_down=st[3]), j))
        elif kind == "battery":
            cap = float(q["max_capacity"])
            if q.get("charge") is not None:
                charge = float(q["charge"])
            elif q.get("init_charge") is not None:
                charge = float(q["init_charge"])
            else:
                charge = float(q["init_soc"]) * cap
            mods.append(module("battery", "!BatteryModule", dict(
                battery_cost_cycle=float(q.get("battery_cost_cycle", 0.0)), battery_transition_model=None,
                efficiency=float(q["efficiency"]), init_charge=None, init_soc=charge / cap, initial_step=t0,
                max_capacity=cap, max_charge=float(q["max_charge"]), max_discharge=float(q["max_discharge"]),
                min_capacity=float(q["min_capacity"]), raise_errors=rerr), dict(current_charge=charge, soc=charge / cap), j))
        else:
            mods.append(module("grid", "!GridModule", ts_params("GridModule", np.asarray(grid_series[j], dtype=np.float64), dict(
                cost_per_unit_co2=float(q.get("cost_per_unit_co2", 0.0)), max_export=float(q["max_export"]),
                max_import=float(q["max_import"])), j=j), {}, j))
    from . import trajectory as tj
    tf = p.get("trajectory_func")
    if isinstance(tf, tj.DeterministicTrajectory):
        tf = _Tagged("!DeterministicTrajectory", dict(initial_step=int(tf.initial_step), final_step=int(tf.final_step)))
    elif isinstance(tf, tj.FixedLengthStochasticTrajectory):
        tf = _Tagged("!FixedLengthStochasticTrajectory", dict(trajectory_length=int(tf.trajectory_length)))
    elif isinstance(tf, tj.StochasticTrajectory):
        tf = _Tagged("!StochasticTrajectory", {})
    elif tf is not None:
        raise TypeError(f"trajectory_func {tf!r}: only the reference's three trajectory classes have a YAML form")
    doc = _Tagged("!Microgrid", dict(final_step=final, initial_step=t0, modules=mods, trajectory_func=tf))

    class Dumper(yaml.SafeDumper):
        pass

    def represent(dumper, obj):
        if isinstance(obj.value, dict):
            return dumper.represent_mapping(obj.tag, obj.value)
        return dumper.represent_scalar(obj.tag, obj.value)
    Dumper.add_representer(_Tagged, represent)
    with open(path, "w") as fh:
        yaml.dump(doc, fh, Dumper=Dumper, default_flow_style=False, sort_keys=True)
    return path


class _Tagged:
    """A YAML node with an application tag (``!Microgrid``, ``!LoadModule``, ``!NDArray`` ...)."""

    def __init__(self, tag, value):
        self.tag, self.value = tag, value


def from_scenario(microgrid_number, root):
    """``Microgrid.from_scenario(n)`` (microgrid.py:958-980) given the directory that holds ``pymgrid25/``."""
    import os
    n = int(microgrid_number)
    return load_scenario_yaml(os.path.join(root, "pymgrid25", f"microgrid_{n}", f"microgrid_{n}.yaml"))


def save_state(batch, step, path):
    """Checkpoint of the dynamic state (the YAML ``state`` blocks of the reference, base_module.py:826-850)."""
    arrays = {k: v.cpu().numpy() for k, v in batch.state().items()}
    np.savez_compressed(path, current_step=np.int64(step), **arrays)


def load_state(batch, path):
    """Restore a ``save_state`` checkpoint into the batch; returns the saved step counter."""
    import torch
    z = np.load(path)
    batch.load_state({k: torch.from_numpy(z[k]).to(batch.device) for k in z.files if k != "current_step"})
    return int(z["current_step"])
