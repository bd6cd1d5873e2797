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
    # module multiplicities (the reference's container holds a list of modules per name, module_container.py:355-413):
    # -1 = as has_* says (0 or 1).  With n > 1 the columns of that kind are [n, N], grid_ts [T, n_grid, 4, N], and the
    # batch runs on the general kernels.
    n_genset: int = -1
    n_battery: int = -1
    n_grid: int = -1
    # order of the module blocks in a flat observation row: "module" (load, pv, genset, battery, grid) or "gym" (battery,
    # genset, grid, load, pv: alphabetical by module name -- what the reference's flat observation is under gym's key-sorting
    # ``Dict``, envs/base/base.py:128-163,211-223).  Column bases only: no kernel cost.
    flat_order: str = "module"

    def __post_init__(self):
        if self.flat_order not in ("module", "gym"):
            raise ValueError("flat_order must be 'module' or 'gym'")
        if self.final_step <= 0:
            object.__setattr__(self, "final_step", self.n_steps)
        for kind in ("genset", "battery", "grid"):
            n = getattr(self, "n_" + kind)
            if n < 0:
                object.__setattr__(self, "n_" + kind, int(getattr(self, "has_" + kind)))
            else:
                object.__setattr__(self, "has_" + kind, n > 0)
        if self.grid_before_battery and not (self.has_grid and self.has_battery):
            object.__setattr__(self, "grid_before_battery", False)

    @property
    def multi(self):
        """True when the batch runs on the general kernels: not exactly one module of every present kind."""
        return self.n_load != 1 or self.n_pv != 1 or self.n_genset > 1 or self.n_battery > 1 or self.n_grid > 1

    @property
    def action_dim(self):
        return 2 * self.n_genset + self.n_battery + self.n_grid

    @property
    def obs_dim(self):
        w = 1 + self.horizon
        return (self.n_load + self.n_pv) * w + 4 * self.n_genset + 2 * self.n_battery + 4 * w * self.n_grid

    @property
    def log_names(self):
        names = ["reward", "fixed_provided", "fixed_absorbed", "controllable_provided", "controllable_absorbed",
                 "overall_provided", "overall_absorbed", "load_met", "renewable_used", "curtailment", "loss_load",
                 "overgeneration", "unbalanced_reward"]
        def block(cols, n):          # instance 0 keeps the plain names, instance j > 0 is "name[j]"
            return [c if j == 0 else f"{c}[{j}]" for j in range(n) for c in cols]
        names += block(["genset_production", "genset_co2_production", "genset_reward", "genset_status"], self.n_genset)
        names += block(["discharge_amount", "charge_amount", "battery_reward", "soc_pre", "charge_pre"], self.n_battery)
        names += block(["grid_import", "grid_export", "grid_co2_production", "grid_reward"], self.n_grid)
        names += ["violations"]      # bit mask of requests the reference refuses with raise_errors=True
        return names

    @property
    def action_names(self):
        def block(cols, n):
            return [c if j == 0 else f"{c}[{j}]" for j in range(n) for c in cols]
        return block(["genset_goal_status", "genset_energy"], self.n_genset) + block(["battery"], self.n_battery) \
            + block(["grid"], self.n_grid)

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
        per_module = {"load": window(["load"]), "pv": window(["renewable"]),
                      "genset": ["current_status", "goal_status", "steps_until_up", "steps_until_down"],
                      "battery": ["soc", "current_charge"],
                      "grid": window(["import_price", "export_price", "co2_per_kwh", "grid_status"])}
        names = []
        for name, n, _ in self._blocks():
            names += per_module[name] * n
        return names

    def _blocks(self):
        """(module name, instances, columns per instance) of the flat observation, in ``flat_order``."""
        w = 1 + self.horizon
        blocks = [("load", self.n_load, w), ("pv", self.n_pv, w), ("genset", self.n_genset, 4),
                  ("battery", self.n_battery, 2), ("grid", self.n_grid, 4 * w)]
        if self.flat_order == "gym":
            blocks.sort(key=lambda b: b[0])              # gym.spaces.Dict sorts the keys of a plain dict
        return blocks

    def obs_slices(self):
        """name -> slice of the flat observation (``flat_order``: load, pv, genset, battery, grid -- or alphabetical)."""
        k, out = 0, {}
        for name, n, width in self._blocks():
            if n:
                out[name] = slice(k, k + n * width)
                k += n * width
        return out

    def obs_instances(self):
        """module name -> list of slices of the flat observation, one per module instance (the reference's nested
        observation: {name: [array per module]}, microgrid.py:316-319)."""
        k, out = 0, {}
        for name, n, width in self._blocks():
            if n:
                out[name] = [slice(k + j * width, k + (j + 1) * width) for j in range(n)]
                k += n * width
        return out

    def bytes_per_step(self, log=False, obs=False):
        """Algorithmic HBM bytes of one env-step of one grid for the single-step kernel:
        B = 8*(A + P_f + C_ts + 2*S_f + 2) + 2*S_i + P_i + 1 [+ 8*L] [+ 8*D + 8*C_ts]."""
        A = self.action_dim
        # (per module INSTANCE: a microgrid with two gensets reads two parameter sets, module_container.py:355-413)
        P_f = 2 + 6 * self.n_battery + 5 * self.n_genset + 3 * self.n_grid
        C_ts = self.n_load + self.n_pv + 4 * self.n_grid
        S_f = self.n_battery
        S_i = 4 * self.n_genset
        P_i = 4 * self.n_genset
        B = 8 * (A + P_f + C_ts + 2 * S_f + 2) + 2 * S_i + P_i + 1
        if log:
            B += 8 * len(self.log_names) + 8 * S_f     # the log also reads the pre-step SoC
        if obs:
            B += 8 * self.obs_dim + 8 * C_ts
        return B

    def bytes_fused(self, K, reward=True, done=True, soc_trace=True, status_trace=False, log=False, action_bytes=8,
                    factorised=False, done_bits=False):
        """Compulsory HBM bytes of ONE grid for a K-step fused launch: parameters and state move once, the
        per-step streams (actions, series rows, requested outputs) K times.  ``factorised``: the series are formed in the
        kernel from base profile x ratio -- per launch the grid's factors (two ratios + two profile ids; with a GridModule a
        co2 profile id, a tariff byte and one 8-byte outage word per 64 steps) instead of 8 * C_ts bytes of rows per step
        (the base-profile rows are shared by all grids: K * 64 B per table and WORKGROUP, out of the caches, not counted).
        ``done``: one byte per step, 1/8 with ``done_bits``, none when the caller derives it from the step counter."""
        A = self.action_dim
        P_f = 2 + 6 * int(self.has_battery) + 5 * int(self.has_genset) + 3 * int(self.has_grid)
        C_ts = self.n_load + self.n_pv + 4 * int(self.has_grid)
        # once: float params, packed genset times, charge read + charge/soc write, genset status read + write
        once = 8 * P_f + 4 * int(self.has_genset) + (8 + 16) * int(self.has_battery) + 8 * int(self.has_genset) \
            + 8 * int(log and self.has_battery)      # the log also reads the pre-launch SoC
        series = 8 * C_ts
        if factorised:
            once += 2 * 8 + 2 + 2 * int(self.has_grid)
            series = 0.125 * int(self.has_grid)      # the outage word: 8 B per 64 steps
        per = action_bytes * A + series + 8 * int(reward) + (0.125 if done_bits else 1) * int(done) \
            + 8 * int(soc_trace and self.has_battery) + 4 * int(status_trace and self.has_genset) \
            + 8 * len(self.log_names) * int(log)
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
        # time-series modules with forecast horizons of their own (``horizons`` of the parameter dicts): the columns of the
        # H = max row the microgrids really have, or None (all share ``layout.horizon``); the envs return these columns only
        self.obs_keep = None
        self._validate()

    # ------------------------------------------------------------------------------------------------
    def _validate(self):
        L, N, T = self.layout, self.layout.n_grids, self.layout.n_steps
        fact = self.factorised
        need = ["loss_load_cost", "overgeneration_cost"] + (["load_ts"] if L.n_load and not fact else []) + \
            (["pv_ts"] if L.n_pv and not fact else [])
        if fact:
            if L.multi:
                raise ValueError("factorised series need exactly one module of every kind per grid")
            need += ["base_load", "base_pv", "load_profile", "pv_profile", "load_ratio", "pv_ratio"]
            if L.has_grid:
                need += ["base_co2", "co2_profile", "tariff"]
        if L.has_battery:
            need += ["bat_min_capacity", "bat_max_capacity", "bat_max_charge", "bat_max_discharge", "bat_efficiency",
                     "bat_cost_cycle", "charge", "soc"]
        if L.has_genset:
            need += ["gen_running_min", "gen_running_max", "gen_cost", "gen_co2_per_unit", "gen_cost_per_unit_co2",
                     "gen_times", "gen_status"]
        if L.has_grid:
            need += ["grid_max_import", "grid_max_export", "grid_cost_per_unit_co2"] + ([] if fact else ["grid_ts"])
        PPITCH = _lib.PROFILE_PITCH
        shapes = {"load_ts": (T, N), "pv_ts": (T, N), "grid_ts": (T, 4, N), "grid_lo": (4, N), "grid_hi": (4, N),
                  "base_load": (T, PPITCH), "base_pv": (T, PPITCH), "base_co2": (T, PPITCH), "outage_bits": ((T + 63) // 64, N)}
        dtypes = {"gen_times": torch.int32, "gen_status": torch.int32, "load_profile": torch.uint8, "pv_profile": torch.uint8,
                  "co2_profile": torch.uint8, "tariff": torch.uint8, "outage_bits": torch.int64}   # (u)int bit patterns
        if L.n_load != 1:            # several (or no) load modules per grid: [T, n_load, N], bounds [n_load, N]
            shapes.update(load_ts=(T, L.n_load, N), load_lo=(L.n_load, N), load_hi=(L.n_load, N), load_noise_std=(L.n_load, N))
        if L.n_pv != 1:
            shapes.update(pv_ts=(T, L.n_pv, N), pv_lo=(L.n_pv, N), pv_hi=(L.n_pv, N), pv_noise_std=(L.n_pv, N))
        # several modules of a controllable kind: instance-major columns
        per_kind = {"bat_": L.n_battery, "gen_": L.n_genset, "grid_m": L.n_grid, "grid_c": L.n_grid}
        for name in _lib.COLUMN_NAMES:
            for prefix, n in per_kind.items():
                if name.startswith(prefix) and n > 1:
                    shapes[name] = (n, N)
        if L.n_battery > 1:
            shapes.update(charge=(L.n_battery, N), soc=(L.n_battery, N))
        if L.n_grid > 1:
            shapes.update(grid_ts=(T, L.n_grid, 4, N), grid_lo=(L.n_grid, 4, N), grid_hi=(L.n_grid, 4, N),
                          grid_noise_std=(L.n_grid, N))
        for name in need:
            if name not in self.cols:
                raise ValueError(f"column {name} is required by the layout")
        dev = None
        for name, t in self.cols.items():
            if name not in _lib.COLUMN_NAMES:
                raise ValueError(f"unknown column {name}")
            want = dtypes.get(name, F64)
            if t.dtype != want:
                raise TypeError(f"column {name}: dtype {t.dtype}, expected {want}")
            if tuple(t.shape) != shapes.get(name, (N,)):
                raise ValueError(f"column {name}: shape {tuple(t.shape)}, expected {shapes.get(name, (N,))}")
            if name in _lib.UNIFORM_BITS and t.dim() == 1 and N > 1 and t.stride(0) == 0:
                if L.multi:              # a batch-uniform parameter: ONE value, expanded (mgx_columns.uniform_mask)
                    raise ValueError(f"column {name}: uniform (stride-0) columns need one module of every kind per grid")
            elif not t.is_contiguous():
                raise ValueError(f"column {name} must be contiguous")
            dev = dev or t.device
            if t.device != dev:
                raise ValueError("all columns must live on one device")
        self.device = dev
        if fact:
            # Profile ids index the [T, MGX_PROFILE_PITCH] base tables and the tariff selects a price pattern: the C ABI cannot
            # check device memory (include/mgx.h states the precondition), so it is checked here, once, at construction
            # (one device -> host read): an id beyond the pitch would read a neighbouring row / table.
            ids = [k for k in ("load_profile", "pv_profile") + (("co2_profile",) if L.has_grid else ()) if k in self.cols]
            top = max(int(self.cols[k].max()) for k in ids) if ids and N else 0
            if top >= PPITCH:
                raise ValueError(f"profile id {top} >= MGX_PROFILE_PITCH ({PPITCH}): base tables hold {PPITCH} profile columns")
            if L.has_grid and N and int(self.cols["tariff"].max()) > 2:
                raise ValueError("tariff pattern must be 0 (no import price), 1 or 2 (MicrogridGenerator.py:253-285)")

    def with_flat_order(self, flat_order):
        """The same columns (shared, not copied) under a layout whose flat observation rows are in ``flat_order``
        ("module" / "gym": BatchLayout.flat_order)."""
        from dataclasses import replace
        return MicrogridBatch(replace(self.layout, flat_order=flat_order), self.cols, forecast_noise=self.forecast_noise)

    @property
    def factorised(self):
        """True when the batch holds its series as base profile x per-grid ratio (``mgx_columns.base_load``, include/mgx.h)
        instead of [T, N] arrays: what ``generator.generate(series="factorised")`` builds."""
        return self.cols.get("base_load") is not None

    def materialise(self):
        """The materialised twin of a factorised batch: the same columns (the state columns are SHARED, not copied) with the
        [T, N] series written by ``mgx_synthesize_series``' arithmetic -- one multiply per value, the same the kernels of the
        factorised form perform, so both step bit-identically."""
        if not self.factorised:
            return self
        from .generator import materialise_series
        cols = {k: v for k, v in self.cols.items() if k not in _lib.FACTOR_COLUMNS}
        cols.update(materialise_series(self))
        return MicrogridBatch(self.layout, cols, forecast_noise=self.forecast_noise)

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
            elif name in ("load_profile", "pv_profile", "co2_profile", "tariff"):
                t = torch.from_numpy(a.astype(np.uint8).copy())
            elif name == "outage_bits":
                t = torch.from_numpy(a.astype(np.uint64).view(np.int64).copy())
            else:
                t = torch.from_numpy(a.astype(np.float64, copy=False).copy())
            cols[name] = t.to(device)
        return cls(layout, cols, forecast_noise=noise)

    @classmethod
    def from_grids(cls, grids, device="cuda", flat_order="module"):
        """Pack a list of per-microgrid parameter dicts (the vocabulary of ``scenario.load_fixture`` /
        ``tests/golden/make_goldens.py:extract_params``) that share a layout."""
        arrays, layout = pack_grids(grids, flat_order=flat_order)
        b = cls.from_numpy(layout, arrays, device)
        hz = grids[0].get("horizons")
        if any(g.get("horizons") != hz for g in grids):
            raise ValueError("all microgrids of a batch must share the per-module forecast horizons (bucket them by layout)")
        b.obs_keep = obs_keep_columns(layout, hz)
        return b

    # ------------------------------------------------------------------------------------------------
    def state(self):
        """Copy of the dynamic state (BatteryModule charge/soc, GensetModule status)."""
        return {k: self.cols[k].clone() for k in self.STATE_COLUMNS if k in self.cols}

    def load_state(self, state):
        for k, v in state.items():
            self.cols[k].copy_(v)

    def numpy_columns(self):
        """Host copies of the columns (the CPU oracle's input in the tests); a factorised batch is materialised first."""
        if self.factorised:
            return self.materialise().numpy_columns()
        out = {}
        for k, v in self.cols.items():
            a = np.ascontiguousarray(v.cpu().numpy())          # (a uniform column is a stride-0 view: the oracle wants [N])
            out[k] = a.view(np.uint32) if v.dtype == torch.int32 else a
        out["layout"] = dict(N=self.layout.n_grids, T=self.layout.n_steps, horizon=self.layout.horizon,
                             final_step=self.layout.final_step, has_genset=int(self.layout.has_genset),
                             has_battery=int(self.layout.has_battery), has_grid=int(self.layout.has_grid),
                             grid_before_battery=int(self.layout.grid_before_battery))
        return out

    def c_layout(self):
        L = _lib.Layout()
        d = asdict(self.layout)
        d["flat_order"] = {"module": 0, "gym": 1}[self.layout.flat_order]
        for k, v in d.items():
            setattr(L, k, int(v))
        L.struct_size = _lib.C.sizeof(_lib.Layout)
        return L

    def uniform_columns(self):
        """Names of the parameter columns that hold ONE value for the whole batch (a stride-0 ``tensor.expand(N)``): the kernels
        read element 0 for every grid (``mgx_columns.uniform_mask``), nothing per grid moves."""
        N = self.layout.n_grids
        return [n for n in _lib.UNIFORM_BITS if n in self.cols and self.cols[n].dim() == 1 and N > 1 and self.cols[n].stride(0) == 0]

    def uniform_param_bytes(self):
        """Bytes per grid a step does NOT read because these parameters are batch-uniform (8 per fp64 column, 4 for gen_times)."""
        return sum(4 if n == "gen_times" else 8 for n in self.uniform_columns())

    def c_columns(self):
        c = _lib.Columns()
        c.struct_size = _lib.C.sizeof(_lib.Columns)
        for name in _lib.COLUMN_NAMES:
            t = self.cols.get(name)
            setattr(c, name, t.data_ptr() if t is not None else None)
        mask = 0
        for name in self.uniform_columns():
            mask |= 1 << _lib.UNIFORM_BITS.index(name)
        c.uniform_mask = mask
        return c


def obs_keep_columns(layout, horizons):
    """Columns of the flat observation row (built for H = ``layout.horizon``, the LONGEST forecast horizon) that a microgrid
    whose time-series modules have horizons of their own really has: ``horizons`` = {"load": [h per module], "pv": [...],
    "grid": [...]} (forecast_horizon is a per-module argument, base_timeseries_module.py:30-45; a module without a forecaster
    observes its current value only).  A window block is [current x C, forecast_0 x C, ...] (C = 1, or the 4 grid components):
    module j keeps its first C * (1 + h_j) columns -- the normalisation of a column does not depend on the horizon.  None when
    every module has the layout's horizon."""
    if not horizons:
        return None
    H, keep, k = layout.horizon, [], 0
    for name, n, width in layout._blocks():
        hs = horizons.get(name)
        for j in range(n):
            if name in ("load", "pv", "grid"):
                C = 4 if name == "grid" else 1
                h = int(hs[j]) if hs is not None else H
                if not 0 <= h <= H:
                    raise ValueError(f"horizons[{name!r}][{j}] = {h} outside [0, {H}]")
                keep += list(range(k, k + C * (1 + h)))
            else:
                keep += list(range(k, k + width))
            k += width
    return None if len(keep) == layout.obs_dim else keep


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


def module_list(v):
    """A parameter dict's ``genset`` / ``battery`` / ``grid`` entry -> list of per-instance dicts (a single dict = one
    instance, None = none)."""
    if v is None:
        return []
    return list(v) if isinstance(v, (list, tuple)) else [v]


def grid_series_list(g):
    ts = g.get("grid_ts")
    if ts is None:
        return []
    return list(ts) if isinstance(ts, (list, tuple)) else [ts]


def pack_grids(grids, flat_order="module"):
    """list of parameter dicts -> (numpy column dict, BatchLayout).  ``genset`` / ``battery`` / ``grid`` may be lists of
    dicts (several modules of a kind, ``grid_ts`` then a list of [T, 4] arrays): columns [n, N], instance-major."""
    if not grids:
        raise ValueError("need at least one microgrid")
    g0 = grids[0]
    count = {k: len(module_list(g0.get(k))) for k in ("genset", "battery", "grid")}
    has = {k: n > 0 for k, n in count.items()}
    T = np.asarray(g0["load_ts"]).shape[0]
    N = len(grids)

    def n_modules(a):
        a = np.asarray(a)
        return 1 if a.ndim == 1 else a.shape[1]
    n_load, n_pv = n_modules(g0["load_ts"]), n_modules(g0["pv_ts"])
    for g in grids:
        for k in count:
            if len(module_list(g.get(k))) != count[k]:
                raise ValueError("all microgrids of a batch must have the same module set (bucket them by layout)")
        if len(grid_series_list(g)) != count["grid"]:
            raise ValueError("grid_ts: one [T, 4] series per GridModule")
        for k, n in (("load_ts", n_load), ("pv_ts", n_pv)):
            a = np.asarray(g[k])
            if a.shape[0] != T or n_modules(a) != n:
                raise ValueError(f"{k}: every microgrid of a batch needs {n} series of length {T}")
        for k in ("horizon", "final_step", "initial_step"):
            if g.get(k, g0.get(k)) != g0.get(k):
                raise ValueError(f"all microgrids of a batch must share {k}")
    if max(count.values()) > _lib.MAX_INSTANCES:
        raise ValueError(f"at most {_lib.MAX_INSTANCES} gensets, batteries and grids per microgrid on the device path")
    layout = BatchLayout(n_grids=N, n_steps=T, horizon=int(g0.get("horizon", 0)),
                         initial_step=int(g0.get("initial_step", 0)), final_step=int(g0.get("final_step", 0)),
                         has_genset=has["genset"], has_battery=has["battery"], has_grid=has["grid"],
                         n_load=n_load, n_pv=n_pv, grid_before_battery=grid_first(g0),
                         n_genset=count["genset"], n_battery=count["battery"], n_grid=count["grid"], flat_order=flat_order)
    if any(grid_first(g) != layout.grid_before_battery for g in grids if has["grid"] and has["battery"]):
        raise ValueError("all microgrids of a batch must step battery and grid in the same order (bucket them by layout)")

    def col(fn):
        return np.array([fn(g) for g in grids], dtype=np.float64)

    def inst_col(kind, fn, dtype=np.float64):          # -> [N] for one instance per grid, else [n, N]
        a = np.array([[fn(q) for q in module_list(g.get(kind))] for g in grids], dtype=dtype).T
        return a[0] if count[kind] == 1 else np.ascontiguousarray(a)

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
            A["bat_" + k] = inst_col("battery", lambda b, k=k: b[k])
        A["bat_cost_cycle"] = inst_col("battery", lambda b: b["battery_cost_cycle"])
        if np.any((A["bat_efficiency"] <= 0) | (A["bat_efficiency"] > 1)):
            raise ValueError("battery efficiency must be in (0, 1]")          # battery_module.py:78

        def init_battery(b):       # BatteryModule._init_battery, battery_module.py:96-106 -> (charge, soc)
            if b.get("charge") is not None:
                c = float(b["charge"])
                return c, (float(b["soc"]) if b.get("soc") is not None else c / b["max_capacity"])
            if b.get("init_charge") is not None:
                c = float(b["init_charge"])
                return c, c / b["max_capacity"]
            if b.get("init_soc") is not None:
                s = float(b["init_soc"])
                return s * b["max_capacity"], s
            raise ValueError("Must set one of init_charge and init_soc.")
        A["charge"] = inst_col("battery", lambda b: init_battery(b)[0])
        A["soc"] = inst_col("battery", lambda b: init_battery(b)[1])
    if has["genset"]:
        A["gen_running_min"] = inst_col("genset", lambda q: q["running_min_production"])
        A["gen_running_max"] = inst_col("genset", lambda q: q["running_max_production"])
        if np.any(A["gen_running_min"] > A["gen_running_max"]):
            raise ValueError("parameter min_production must not be greater than parameter max_production.")
        A["gen_cost"] = inst_col("genset", lambda q: q["genset_cost"])
        A["gen_co2_per_unit"] = inst_col("genset", lambda q: q.get("co2_per_unit", 0.0))
        A["gen_cost_per_unit_co2"] = inst_col("genset", lambda q: q.get("cost_per_unit_co2", 0.0))
        su = inst_col("genset", lambda q: int(q.get("start_up_time", 0)), np.int64)
        wd = inst_col("genset", lambda q: int(q.get("wind_down_time", 0)), np.int64)
        A["gen_times"] = pack_times(su, wd, inst_col("genset", lambda q: bool(q.get("allow_abortion", True)), bool))

        def init_status(q):        # genset_module.py:91-92,216-227
            if q.get("status") is not None:
                return [int(v) for v in q["status"]]
            on = int(bool(q.get("init_start_up", True)))
            return [on, on, 0, int(q.get("wind_down_time", 0))] if on else [0, 0, int(q.get("start_up_time", 0)), 0]
        st = [inst_col("genset", lambda q, c=c: init_status(q)[c], np.int64) for c in range(4)]
        A["gen_status"] = pack_status(*st)
    if has["grid"]:
        A["grid_max_import"] = inst_col("grid", lambda q: q["max_import"])
        A["grid_max_export"] = inst_col("grid", lambda q: q["max_export"])
        A["grid_cost_per_unit_co2"] = inst_col("grid", lambda q: q.get("cost_per_unit_co2", 0.0))
        gts = []
        for g in grids:                           # GridModule._check_params, grid_module.py:103-123
            per = []
            for ts in grid_series_list(g):
                ts = np.asarray(ts, dtype=np.float64)
                if ts.ndim != 2 or ts.shape[1] not in (3, 4) or ts.shape[0] != T:
                    raise ValueError("Time series must be two dimensional with three or four columns.")
                if ts.shape[1] == 3:
                    ts = np.concatenate([ts, np.ones((T, 1))], axis=1)
                elif not np.all((ts[:, 3] == 0) | (ts[:, 3] == 1)):
                    raise ValueError("Last column (grid status) must contain binary values.")
                if (ts < 0).any():
                    raise ValueError("Time series must be non-negative.")
                per.append(ts)
            gts.append(np.stack(per, axis=1))     # [T, n_grid, 4]
        gts = np.stack(gts, axis=3)               # [T, n_grid, 4, N]
        lo, hi = gts.min(axis=0), gts.max(axis=0)
        if count["grid"] == 1:
            gts, lo, hi = gts[:, 0], lo[0], hi[0]  # [T, 4, N]
        A["grid_ts"] = np.ascontiguousarray(gts)
        A["grid_lo"], A["grid_hi"] = lo, hi
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
        def inst_stds(key, n):    # one std per module INSTANCE: a column [N], or [n, N] with several modules of the kind
            if n == 0:
                return None
            per = [stds(lambda j, q=q: (A[key] if n == 1 else A[key][:, q])[..., j]) for q in range(n)]
            return per[0] if n == 1 else np.stack(per)
        for key, name, n in (("load_ts", "load_noise_std", layout.n_load), ("pv_ts", "pv_noise_std", layout.n_pv),
                             ("grid_ts", "grid_noise_std", layout.n_grid)):
            col_ = inst_stds(key, n)
            if col_ is not None:
                A[name] = col_
        A["__forecast_noise__"] = dict(seed=int(fn0.get("seed", 0)),
                                       increase_uncertainty=bool(fn0.get("increase_uncertainty", False)))
    return A, layout
