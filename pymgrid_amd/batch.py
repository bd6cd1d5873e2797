"""Struct-of-arrays batch of N microgrids in device memory.

A *batch* is what ``Microgrid.__init__`` receives in the reference (a module list, microgrid.py:100-128) for N
microgrids at once: every module parameter becomes an fp64 column ``[N]``, every time series a time-major
``[T, N]`` array (one step reads one contiguous row), the battery charge / SoC and the packed genset status are the
dynamic state columns.  Column names are exactly the fields of ``mgx_columns`` in ``include/mgx.h``.
"""
from dataclasses import dataclass, asdict

import numpy as np
import torch

from . import _lib

F64 = torch.float64


@dataclass(frozen=True)
class BatchLayout:
    n_grids: int
    n_steps: int                 # T
    horizon: int = 0             # forecast horizon H (oracle forecaster), 0 = none
    initial_step: int = 0
    final_step: int = 0          # <= 0: n_steps (base_timeseries_module.py:321-326)
    has_genset: bool = True
    has_battery: bool = True
    has_grid: bool = False
    n_load: int = 1
    n_pv: int = 1
    # the GridModule precedes the BatteryModule in the microgrid's module list: it is stepped and summed first
    # (module_container.py:355-413).  Column orders (actions, log, observation) do not depend on it.
    grid_before_battery: bool = False

    def __post_init__(self):
        if self.final_step <= 0:
            object.__setattr__(self, "final_step", self.n_steps)
        if self.grid_before_battery and not (self.has_grid and self.has_battery):
            object.__setattr__(self, "grid_before_battery", False)

    @property
    def action_dim(self):
        return 2 * int(self.has_genset) + int(self.has_battery) + int(self.has_grid)

    @property
    def obs_dim(self):
        w = 1 + self.horizon
        return (self.n_load + self.n_pv) * w + 4 * int(self.has_genset) + 2 * int(self.has_battery) \
            + 4 * w * int(self.has_grid)

    @property
    def log_names(self):
        names = ["reward", "fixed_provided", "fixed_absorbed", "controllable_provided", "controllable_absorbed",
                 "overall_provided", "overall_absorbed", "load_met", "renewable_used", "curtailment", "loss_load",
                 "overgeneration", "unbalanced_reward"]
        if self.has_genset:
            names += ["genset_production", "genset_co2_production", "genset_reward", "genset_status"]
        if self.has_battery:
            names += ["discharge_amount", "charge_amount", "battery_reward", "soc_pre", "charge_pre"]
        if self.has_grid:
            names += ["grid_import", "grid_export", "grid_co2_production", "grid_reward"]
        names += ["violations"]      # bit mask of requests the reference refuses with raise_errors=True
        return names

    @property
    def action_names(self):
        names = []
        if self.has_genset:
            names += ["genset_goal_status", "genset_energy"]
        if self.has_battery:
            names += ["battery"]
        if self.has_grid:
            names += ["grid"]
        return names

    @property
    def obs_names(self):
        """State-key name of every flat observation column, as the reference's ``state_dict`` spells them
        (base_timeseries_module.py:90-97 ``<component>_current`` / ``<component>_forecast_<j>``; genset_module.py:426-431;
        battery_module.py:280-281; grid_module.py:70)."""
        H = self.horizon

        def window(components):
            names = [f"{c}_current" for c in components]
            for j in range(H):
                names += [f"{c}_forecast_{j}" for c in components]
            return names
        names = []
        for _ in range(self.n_load):
            names += window(["load"])
        for _ in range(self.n_pv):
            names += window(["renewable"])
        if self.has_genset:
            names += ["current_status", "goal_status", "steps_until_up", "steps_until_down"]
        if self.has_battery:
            names += ["soc", "current_charge"]
        if self.has_grid:
            names += window(["import_price", "export_price", "co2_per_kwh", "grid_status"])
        return names

    def obs_slices(self):
        """name -> slice of the flat observation (order load, pv, genset, battery, grid)."""
        w, k, out = 1 + self.horizon, 0, {}
        for name, n in (("load", self.n_load * w), ("pv", self.n_pv * w), ("genset", 4 * int(self.has_genset)),
                        ("battery", 2 * int(self.has_battery)), ("grid", 4 * w * int(self.has_grid))):
            if n:
                out[name] = slice(k, k + n)
                k += n
        return out

    def bytes_per_step(self, log=False, obs=False):
        """Algorithmic HBM bytes of one env-step of one grid for the single-step kernel:
        B = 8*(A + P_f + C_ts + 2*S_f + 2) + 2*S_i + P_i + 1 [+ 8*L] [+ 8*D + 8*C_ts]."""
        A = self.action_dim
        P_f = 2 + 6 * int(self.has_battery) + 5 * int(self.has_genset) + 3 * int(self.has_grid)
        C_ts = self.n_load + self.n_pv + 4 * int(self.has_grid)
        S_f = int(self.has_battery)
        S_i = 4 * int(self.has_genset)
        P_i = 4 * int(self.has_genset)
        B = 8 * (A + P_f + C_ts + 2 * S_f + 2) + 2 * S_i + P_i + 1
        if log:
            B += 8 * len(self.log_names) + 8 * S_f     # the log also reads the pre-step SoC
        if obs:
            B += 8 * self.obs_dim + 8 * C_ts
        return B

    def bytes_fused(self, K, reward=True, done=True, soc_trace=True, status_trace=False, log=False, action_bytes=8):
        """Compulsory HBM bytes of ONE grid for a K-step fused launch: parameters and state move once, the
        per-step streams (actions, series rows, requested outputs) K times."""
        A = self.action_dim
        P_f = 2 + 6 * int(self.has_battery) + 5 * int(self.has_genset) + 3 * int(self.has_grid)
        C_ts = self.n_load + self.n_pv + 4 * int(self.has_grid)
        # once: float params, packed genset times, charge read + charge/soc write, genset status read + write
        once = 8 * P_f + 4 * int(self.has_genset) + (8 + 16) * int(self.has_battery) + 8 * int(self.has_genset) \
            + 8 * int(log and self.has_battery)      # the log also reads the pre-launch SoC
        per = action_bytes * A + 8 * C_ts + 8 * int(reward) + int(done) + 8 * int(soc_trace and self.has_battery) \
            + 4 * int(status_trace and self.has_genset) + 8 * len(self.log_names) * int(log)
        return once + K * per


def pack_status(cur, goal, up, down):
    """current | goal<<8 | steps_until_up<<16 | steps_until_down<<24 (genset_module.py:426-431 state keys)."""
    cur, goal, up, down = (np.asarray(x, dtype=np.int64) for x in (cur, goal, up, down))
    if np.any((up < 0) | (up > 255) | (down < 0) | (down > 255)):
        raise ValueError("steps_until_up / steps_until_down must fit in 8 bits")
    return (cur | (goal << 8) | (up << 16) | (down << 24)).astype(np.uint32)


def unpack_status(word):
    w = np.asarray(word).astype(np.int64)
    return np.stack([w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff, (w >> 24) & 0xff], axis=-1).astype(np.int32)


def pack_times(start_up, wind_down, allow_abortion=True):
    """start_up_time | (not allow_abortion) << 8 | wind_down_time << 16 (GensetModule arguments, genset_module.py:61-98)."""
    su, wd = np.asarray(start_up, dtype=np.int64), np.asarray(wind_down, dtype=np.int64)
    if np.any((su < 0) | (su > 255) | (wd < 0) | (wd > 255)):
        raise ValueError("start_up_time / wind_down_time must be in [0, 255] on the device path")
    no_abort = (~np.asarray(allow_abortion, dtype=bool)).astype(np.int64)
    return (su | (no_abort << 8) | (wd << 16)).astype(np.uint32)


class MicrogridBatch:
    """Device-resident SoA columns of N microgrids that share one layout."""

    STATE_COLUMNS = ("charge", "soc", "gen_status")

    def __init__(self, layout, cols, forecast_noise=None):
        self.layout = layout
        self.cols = cols
        # GaussianNoiseForecaster switches: dict(seed=int, increase_uncertainty=bool) or None (oracle forecaster)
        self.forecast_noise = forecast_noise
        self._validate()

    # ------------------------------------------------------------------------------------------------
    def _validate(self):
        L, N, T = self.layout, self.layout.n_grids, self.layout.n_steps
        need = ["loss_load_cost", "overgeneration_cost"] + (["load_ts"] if L.n_load else []) + \
            (["pv_ts"] if L.n_pv else [])
        if L.has_battery:
            need += ["bat_min_capacity", "bat_max_capacity", "bat_max_charge", "bat_max_discharge", "bat_efficiency",
                     "bat_cost_cycle", "charge", "soc"]
        if L.has_genset:
            need += ["gen_running_min", "gen_running_max", "gen_cost", "gen_co2_per_unit", "gen_cost_per_unit_co2",
                     "gen_times", "gen_status"]
        if L.has_grid:
            need += ["grid_max_import", "grid_max_export", "grid_cost_per_unit_co2", "grid_ts"]
        shapes = {"load_ts": (T, N), "pv_ts": (T, N), "grid_ts": (T, 4, N), "grid_lo": (4, N), "grid_hi": (4, N)}
        if L.n_load != 1:            # several (or no) load modules per grid: [T, n_load, N], bounds [n_load, N]
            shapes.update(load_ts=(T, L.n_load, N), load_lo=(L.n_load, N), load_hi=(L.n_load, N))
        if L.n_pv != 1:
            shapes.update(pv_ts=(T, L.n_pv, N), pv_lo=(L.n_pv, N), pv_hi=(L.n_pv, N))
        for name in need:
            if name not in self.cols:
                raise ValueError(f"column {name} is required by the layout")
        dev = None
        for name, t in self.cols.items():
            if name not in _lib.COLUMN_NAMES:
                raise ValueError(f"unknown column {name}")
            want = torch.int32 if name in ("gen_times", "gen_status") else F64   # uint32 bit patterns
            if t.dtype != want:
                raise TypeError(f"column {name}: dtype {t.dtype}, expected {want}")
            if tuple(t.shape) != shapes.get(name, (N,)):
                raise ValueError(f"column {name}: shape {tuple(t.shape)}, expected {shapes.get(name, (N,))}")
            if not t.is_contiguous():
                raise ValueError(f"column {name} must be contiguous")
            dev = dev or t.device
            if t.device != dev:
                raise ValueError("all columns must live on one device")
        self.device = dev

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_numpy(cls, layout, arrays, device):
        cols = {}
        arrays = dict(arrays)
        noise = arrays.pop("__forecast_noise__", None)
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if name in ("gen_times", "gen_status"):
                t = torch.from_numpy(a.astype(np.uint32).view(np.int32).copy())
            else:
                t = torch.from_numpy(a.astype(np.float64, copy=False).copy())
            cols[name] = t.to(device)
        return cls(layout, cols, forecast_noise=noise)

    @classmethod
    def from_grids(cls, grids, device="cuda"):
        """Pack a list of per-microgrid parameter dicts (the vocabulary of ``scenario.load_fixture`` /
        ``tests/golden/make_goldens.py:extract_params``) that share a layout."""
        arrays, layout = pack_grids(grids)
        return cls.from_numpy(layout, arrays, device)

    # ------------------------------------------------------------------------------------------------
    def state(self):
        """Copy of the dynamic state (BatteryModule charge/soc, GensetModule status)."""
        return {k: self.cols[k].clone() for k in self.STATE_COLUMNS if k in self.cols}

    def load_state(self, state):
        for k, v in state.items():
            self.cols[k].copy_(v)

    def numpy_columns(self):
        out = {}
        for k, v in self.cols.items():
            out[k] = v.cpu().numpy().view(np.uint32) if v.dtype == torch.int32 else v.cpu().numpy()
        out["layout"] = dict(N=self.layout.n_grids, T=self.layout.n_steps, horizon=self.layout.horizon,
                             final_step=self.layout.final_step, has_genset=int(self.layout.has_genset),
                             has_battery=int(self.layout.has_battery), has_grid=int(self.layout.has_grid),
                             grid_before_battery=int(self.layout.grid_before_battery))
        return out

    def c_layout(self):
        L = _lib.Layout()
        d = asdict(self.layout)
        for k, v in d.items():
            setattr(L, k, int(v))
        L.struct_size = _lib.C.sizeof(_lib.Layout)
        return L

    def c_columns(self):
        c = _lib.Columns()
        c.struct_size = _lib.C.sizeof(_lib.Columns)
        for name in _lib.COLUMN_NAMES:
            t = self.cols.get(name)
            setattr(c, name, t.data_ptr() if t is not None else None)
        return c


def series_bounds(ts):
    """Observation bounds of a load / renewable series: base_timeseries_module.py:81-88."""
    lo, hi = ts.min(axis=0), ts.max(axis=0)
    return np.minimum(lo, 0.0), np.maximum(hi, 0.0)


def grid_first(p):
    """True when the parameter dict says the GridModule comes before the BatteryModule in the module list
    (``controllable_order``: names of the controllable modules in list order, e.g. ("grid", "battery", "genset"))."""
    order = p.get("controllable_order")
    if not order or p.get("grid") is None or p.get("battery") is None:
        return False
    order = [str(x) for x in order]
    return "grid" in order and "battery" in order and order.index("grid") < order.index("battery")


def pack_grids(grids):
    """list of parameter dicts -> (numpy column dict, BatchLayout)."""
    if not grids:
        raise ValueError("need at least one microgrid")
    g0 = grids[0]
    has = {k: g0.get(k) is not None for k in ("genset", "battery", "grid")}
    T = np.asarray(g0["load_ts"]).shape[0]
    N = len(grids)

    def n_modules(a):
        a = np.asarray(a)
        return 1 if a.ndim == 1 else a.shape[1]
    n_load, n_pv = n_modules(g0["load_ts"]), n_modules(g0["pv_ts"])
    for g in grids:
        for k in has:
            if (g.get(k) is not None) != has[k]:
                raise ValueError("all microgrids of a batch must have the same module set (bucket them by layout)")
        for k, n in (("load_ts", n_load), ("pv_ts", n_pv)):
            a = np.asarray(g[k])
            if a.shape[0] != T or n_modules(a) != n:
                raise ValueError(f"{k}: every microgrid of a batch needs {n} series of length {T}")
        for k in ("horizon", "final_step", "initial_step"):
            if g.get(k, g0.get(k)) != g0.get(k):
                raise ValueError(f"all microgrids of a batch must share {k}")
    layout = BatchLayout(n_grids=N, n_steps=T, horizon=int(g0.get("horizon", 0)),
                         initial_step=int(g0.get("initial_step", 0)), final_step=int(g0.get("final_step", 0)),
                         has_genset=has["genset"], has_battery=has["battery"], has_grid=has["grid"],
                         n_load=n_load, n_pv=n_pv, grid_before_battery=grid_first(g0))
    if any(grid_first(g) != layout.grid_before_battery for g in grids if has["grid"] and has["battery"]):
        raise ValueError("all microgrids of a batch must step battery and grid in the same order (bucket them by layout)")

    def col(fn):
        return np.array([fn(g) for g in grids], dtype=np.float64)

    A = {}
    def stack(key, n):           # -> [T, N] for one module per grid, else [T, n, N]
        a = np.stack([np.asarray(g[key], dtype=np.float64).reshape(T, n) for g in grids], axis=2)
        return a[:, 0, :] if n == 1 else a
    # sign convention of the stored series (base_timeseries_module.py:68-79)
    raw_load, raw_pv = stack("load_ts", n_load), stack("pv_ts", n_pv)
    for ts in (raw_load, raw_pv):                       # _sign_check: a pure sink / source has one sign
        if not ((np.sign(ts) <= 0).all(axis=0) | (np.sign(ts) >= 0).all(axis=0)).all():
            raise ValueError('time_series cannot contain both positive and negative values unless it is both '
                             'a source and a sink.')
    A["load_ts"], A["pv_ts"] = -np.abs(raw_load), np.abs(raw_pv)
    if n_load:
        A["load_lo"], A["load_hi"] = series_bounds(A["load_ts"])
    else:
        del A["load_ts"]
    if n_pv:
        A["pv_lo"], A["pv_hi"] = series_bounds(A["pv_ts"])
    else:
        del A["pv_ts"]
    A["loss_load_cost"] = col(lambda g: g["unbalanced"]["loss_load_cost"])
    A["overgeneration_cost"] = col(lambda g: g["unbalanced"]["overgeneration_cost"])
    if has["battery"]:
        for k in ("min_capacity", "max_capacity", "max_charge", "max_discharge", "efficiency"):
            A["bat_" + k] = col(lambda g, k=k: g["battery"][k])
        A["bat_cost_cycle"] = col(lambda g: g["battery"]["battery_cost_cycle"])
        if np.any((A["bat_efficiency"] <= 0) | (A["bat_efficiency"] > 1)):
            raise ValueError("battery efficiency must be in (0, 1]")          # battery_module.py:78
        charge, soc = [], []
        for g in grids:            # BatteryModule._init_battery, battery_module.py:96-106
            b = g["battery"]
            if b.get("charge") is not None:
                c = float(b["charge"])
                s = float(b["soc"]) if b.get("soc") is not None else c / b["max_capacity"]
            elif b.get("init_charge") is not None:
                c = float(b["init_charge"]); s = c / b["max_capacity"]
            elif b.get("init_soc") is not None:
                s = float(b["init_soc"]); c = s * b["max_capacity"]
            else:
                raise ValueError("Must set one of init_charge and init_soc.")
            charge.append(c); soc.append(s)
        A["charge"], A["soc"] = np.array(charge), np.array(soc)
    if has["genset"]:
        A["gen_running_min"] = col(lambda g: g["genset"]["running_min_production"])
        A["gen_running_max"] = col(lambda g: g["genset"]["running_max_production"])
        if np.any(A["gen_running_min"] > A["gen_running_max"]):
            raise ValueError("parameter min_production must not be greater than parameter max_production.")
        A["gen_cost"] = col(lambda g: g["genset"]["genset_cost"])
        A["gen_co2_per_unit"] = col(lambda g: g["genset"].get("co2_per_unit", 0.0))
        A["gen_cost_per_unit_co2"] = col(lambda g: g["genset"].get("cost_per_unit_co2", 0.0))
        su = [int(g["genset"].get("start_up_time", 0)) for g in grids]
        wd = [int(g["genset"].get("wind_down_time", 0)) for g in grids]
        A["gen_times"] = pack_times(su, wd, [bool(g["genset"].get("allow_abortion", True)) for g in grids])
        st = []
        for g, s_, w_ in zip(grids, su, wd):      # genset_module.py:91-92,216-227
            q = g["genset"]
            if q.get("status") is not None:
                st.append([int(v) for v in q["status"]])
            else:
                on = int(bool(q.get("init_start_up", True)))
                st.append([on, on, 0, w_] if on else [0, 0, s_, 0])
        st = np.array(st)
        A["gen_status"] = pack_status(st[:, 0], st[:, 1], st[:, 2], st[:, 3])
    if has["grid"]:
        A["grid_max_import"] = col(lambda g: g["grid"]["max_import"])
        A["grid_max_export"] = col(lambda g: g["grid"]["max_export"])
        A["grid_cost_per_unit_co2"] = col(lambda g: g["grid"].get("cost_per_unit_co2", 0.0))
        gts = []
        for g in grids:                           # GridModule._check_params, grid_module.py:103-123
            ts = np.asarray(g["grid_ts"], dtype=np.float64)
            if ts.ndim != 2 or ts.shape[1] not in (3, 4) or ts.shape[0] != T:
                raise ValueError("Time series must be two dimensional with three or four columns.")
            if ts.shape[1] == 3:
                ts = np.concatenate([ts, np.ones((T, 1))], axis=1)
            elif not np.all((ts[:, 3] == 0) | (ts[:, 3] == 1)):
                raise ValueError("Last column (grid status) must contain binary values.")
            if (ts < 0).any():
                raise ValueError("Time series must be non-negative.")
            gts.append(ts)
        gts = np.stack(gts, axis=2)               # [T, 4, N]
        A["grid_ts"] = gts
        A["grid_lo"], A["grid_hi"] = gts.min(axis=0), gts.max(axis=0)
    # GaussianNoiseForecaster (forecast/forecaster.py:220-275): p["forecast_noise"] = dict(std=, relative_noise=,
    # increase_uncertainty=, seed=) -> per-grid, per-module noise standard deviation columns
    fn0 = g0.get("forecast_noise")
    if any((g.get("forecast_noise") is None) != (fn0 is None) for g in grids):
        raise ValueError("all microgrids of a batch must agree on having a noisy forecaster")
    if fn0 is not None:
        if layout.horizon == 0:
            raise ValueError("forecast_noise needs a forecast horizon > 0")
        lo_, hi_ = layout.initial_step, layout.final_step
        for g in grids:
            f = g["forecast_noise"]
            if bool(f.get("increase_uncertainty", False)) != bool(fn0.get("increase_uncertainty", False)):
                raise ValueError("all microgrids of a batch must share increase_uncertainty")

        def stds(series_of):          # _get_noise_std :236-249: std * |mean(time_series[initial:final])| if relative
            out = []
            for j, g in enumerate(grids):
                f = g["forecast_noise"]
                sd = float(f["std"])
                if f.get("relative_noise", False):
                    sd *= abs(float(series_of(j)[lo_:hi_].mean()))
                out.append(sd)
            return np.array(out)
        A["load_noise_std"] = stds(lambda j: A["load_ts"][:, j])
        A["pv_noise_std"] = stds(lambda j: A["pv_ts"][:, j])
        if has["grid"]:
            A["grid_noise_std"] = stds(lambda j: A["grid_ts"][:, :, j])
        A["__forecast_noise__"] = dict(seed=int(fn0.get("seed", 0)),
                                       increase_uncertainty=bool(fn0.get("increase_uncertainty", False)))
    return A, layout
