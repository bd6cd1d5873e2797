"""The reference's modules as this package sees them, in two halves.

Descriptions: ``BatteryModule`` / ``GensetModule`` / ``GridModule`` / ``LoadModule`` / ``RenewableModule`` / ``UnbalancedEnergyModule``
take the reference constructors' arguments (battery_module.py:66-106, genset_module.py:61-98, grid_module.py:73-123,
load_module.py:58-80, renewable_module.py:61-84, unbalanced_energy_module.py:13-26), check what those check, and keep them as data:
``Microgrid([("load", LoadModule(...)), BatteryModule(...), ...], loss_load_cost=..)`` (microgrid.py:100-173) builds the N = 1 batch
from them.  They do not step -- the microgrid steps on the device.

Views: ``microgrid.modules`` / ``.fixed`` / ``.flex`` / ``.controllable`` / ``.module_list`` (microgrid.py:761-818,
modules/module_container.py:355-413) are read-only views over the batch's columns: every attribute is read from the batch when it
is asked for, so it shows the state the kernels left (``microgrid.modules.battery[0].soc``).  Writing is not offered (the device
path has no per-module Python objects to write to -- ``set_module_attr`` covers the step window)."""
import numpy as np

from .batch import unpack_status

DEFAULT_HORIZON = 23          # microgrid/__init__.py:1


class _ModuleSpec:
    """The constructor arguments of one of the reference's modules, kept as data: ``Microgrid([...])`` turns a list of these into
    the columns of an N = 1 batch (scenario.params_from_module_docs: the rules of the modules' constructors).  Not a stepping
    object -- the microgrid steps on the device; read a module's live state through ``microgrid.modules``."""
    tag, default_name = None, None

    def __init__(self, **cls_params):
        self.cls_params = cls_params

    def doc(self):
        return {"__tag__": self.tag, "cls_params": dict(self.cls_params), "state": {}}

    def __repr__(self):
        args = ", ".join(f"{k}={type(v).__name__ if isinstance(v, np.ndarray) else v!r}" for k, v in self.cls_params.items())
        return f"{type(self).__name__}({args})"


class BatteryModule(_ModuleSpec):
    """``pymgrid.modules.BatteryModule`` (battery_module.py:66-106)."""
    tag, default_name = "!BatteryModule", "battery"

    def __init__(self, min_capacity, max_capacity, max_charge, max_discharge, efficiency, battery_cost_cycle=0.0,
                 battery_transition_model=None, init_charge=None, init_soc=None, initial_step=0, raise_errors=False):
        assert 0 < efficiency <= 1
        if init_charge is None and init_soc is None:
            raise ValueError("Must set one of init_charge and init_soc.")                      # battery_module.py:96-106
        if init_charge is not None and init_soc is not None:
            import warnings
            warnings.warn("Passed both init_capa and init_soc. Using init_charge and ignoring init_soc")
            init_soc = None
        super().__init__(min_capacity=min_capacity, max_capacity=max_capacity, max_charge=max_charge, max_discharge=max_discharge,
                         efficiency=efficiency, battery_cost_cycle=battery_cost_cycle, battery_transition_model=battery_transition_model,
                         init_charge=init_charge, init_soc=init_soc, initial_step=initial_step, raise_errors=raise_errors)


class GensetModule(_ModuleSpec):
    """``pymgrid.modules.GensetModule`` (genset_module.py:61-98)."""
    tag, default_name = "!Genset", "genset"

    def __init__(self, running_min_production, running_max_production, genset_cost, co2_per_unit=0.0, cost_per_unit_co2=0.0,
                 start_up_time=0, wind_down_time=0, allow_abortion=True, init_start_up=True, initial_step=0, raise_errors=False,
                 provided_energy_name="genset_production"):
        if running_min_production > running_max_production:
            raise ValueError("parameter min_production must not be greater than parameter max_production.")
        if callable(genset_cost):
            raise NotImplementedError("a callable genset_cost is not offered on the device path")
        super().__init__(running_min_production=running_min_production, running_max_production=running_max_production,
                         genset_cost=genset_cost, co2_per_unit=co2_per_unit, cost_per_unit_co2=cost_per_unit_co2,
                         start_up_time=start_up_time, wind_down_time=wind_down_time, allow_abortion=allow_abortion,
                         init_start_up=init_start_up, initial_step=initial_step, raise_errors=raise_errors)


class _SeriesSpec(_ModuleSpec):
    def __init__(self, time_series, forecaster, forecast_horizon, forecaster_increase_uncertainty, forecaster_relative_noise,
                 initial_step, final_step, raise_errors, **more):
        if callable(forecaster):
            raise NotImplementedError("user-defined forecasters (callables) are not offered on the device path")
        super().__init__(time_series=np.asarray(time_series, dtype=np.float64), forecaster=forecaster, forecast_horizon=forecast_horizon,
                         forecaster_increase_uncertainty=forecaster_increase_uncertainty,
                         forecaster_relative_noise=forecaster_relative_noise, initial_step=initial_step, final_step=final_step,
                         raise_errors=raise_errors, **more)


class LoadModule(_SeriesSpec):
    """``pymgrid.modules.LoadModule`` (load_module.py:58-80)."""
    tag, default_name = "!LoadModule", "load"

    def __init__(self, time_series, forecaster=None, forecast_horizon=DEFAULT_HORIZON, forecaster_increase_uncertainty=False,
                 forecaster_relative_noise=False, initial_step=0, final_step=-1, raise_errors=False):
        super().__init__(time_series, forecaster, forecast_horizon, forecaster_increase_uncertainty, forecaster_relative_noise,
                         initial_step, final_step, raise_errors)


class RenewableModule(_SeriesSpec):
    """``pymgrid.modules.RenewableModule`` (renewable_module.py:61-84)."""
    tag, default_name = "!RenewableModule", "pv"

    def __init__(self, time_series, raise_errors=False, forecaster=None, forecast_horizon=DEFAULT_HORIZON,
                 forecaster_increase_uncertainty=False, forecaster_relative_noise=False, initial_step=0, final_step=-1,
                 provided_energy_name="renewable_used"):
        super().__init__(time_series, forecaster, forecast_horizon, forecaster_increase_uncertainty, forecaster_relative_noise,
                         initial_step, final_step, raise_errors)


class GridModule(_SeriesSpec):
    """``pymgrid.modules.GridModule`` (grid_module.py:73-123): ``time_series`` [T, 3] (import price, export price, co2 per kWh:
    the grid is always up) or [T, 4] (+ grid status)."""
    tag, default_name = "!GridModule", "grid"

    def __init__(self, max_import, max_export, time_series, forecaster=None, forecast_horizon=DEFAULT_HORIZON,
                 forecaster_increase_uncertainty=False, forecaster_relative_noise=False, initial_step=0, final_step=-1,
                 cost_per_unit_co2=0.0, raise_errors=False):
        ts = np.asarray(time_series, dtype=np.float64)
        if max_import < 0 or max_export < 0:                                              # grid_module.py:98-104
            raise ValueError("parameter max_import / max_export must be non-negative.")
        if ts.ndim != 2 or ts.shape[1] not in (3, 4):
            raise ValueError("Time series must be two dimensional with three or four columns."
                             "See docstring for details.")
        if ts.shape[1] == 3:                                                              # :106-109: the status column is all ones
            ts = np.concatenate([ts, np.ones((ts.shape[0], 1))], axis=1)
        if (ts < 0).any():
            raise ValueError("Time series must be non-negative.")
        if not ((ts[:, 3] == 0) | (ts[:, 3] == 1)).all():
            raise ValueError("Last column (grid status) must contain binary values.")
        super().__init__(ts, forecaster, forecast_horizon, forecaster_increase_uncertainty, forecaster_relative_noise,
                         initial_step, final_step, raise_errors, max_import=max_import, max_export=max_export,
                         cost_per_unit_co2=cost_per_unit_co2)


class UnbalancedEnergyModule(_ModuleSpec):
    """``pymgrid.modules.UnbalancedEnergyModule`` (unbalanced_energy_module.py:13-26)."""
    tag, default_name = "!UnbalancedEnergyModule", "unbalanced_energy"

    def __init__(self, raise_errors, initial_step=0, loss_load_cost=10, overgeneration_cost=2.0):
        super().__init__(raise_errors=raise_errors, initial_step=initial_step, loss_load_cost=loss_load_cost,
                         overgeneration_cost=overgeneration_cost)


def params_from_modules(modules, add_unbalanced_module=True, loss_load_cost=10.0, overgeneration_cost=2.0,
                        reward_shaping_func=None, trajectory_func=None):
    """``Microgrid(modules, add_unbalanced_module, loss_load_cost, overgeneration_cost, ...)`` (microgrid.py:100-173): a list of
    module descriptions -- bare or ``(name, module)`` tuples; the names are this package's fixed ones whatever the tuple says --
    to the parameter dict the N = 1 adaptors are built from."""
    from .scenario import params_from_module_docs
    docs = []
    for item in modules:
        mod = item[1] if isinstance(item, tuple) else item
        if not isinstance(mod, _ModuleSpec):
            raise TypeError(f"modules must be list-like of modules, not {type(mod).__name__}")
        docs.append((mod.default_name, mod.doc()))
    if add_unbalanced_module:
        docs.append(("unbalanced_energy", UnbalancedEnergyModule(False, loss_load_cost=loss_load_cost,
                                                                 overgeneration_cost=overgeneration_cost).doc()))
    return params_from_module_docs({"modules": docs, "trajectory_func": trajectory_func, "reward_shaping_func": reward_shaping_func})

# reference attribute -> batch column, per kind
_PARAMS = {
    "battery": dict(min_capacity="bat_min_capacity", max_capacity="bat_max_capacity", max_charge="bat_max_charge",
                    max_discharge="bat_max_discharge", efficiency="bat_efficiency", battery_cost_cycle="bat_cost_cycle",
                    soc="soc", current_charge="charge"),
    "genset": dict(running_min_production="gen_running_min", running_max_production="gen_running_max", genset_cost="gen_cost",
                   co2_per_unit="gen_co2_per_unit", cost_per_unit_co2="gen_cost_per_unit_co2"),
    "grid": dict(max_import="grid_max_import", max_export="grid_max_export", cost_per_unit_co2="grid_cost_per_unit_co2"),
    "unbalanced_energy": dict(loss_load_cost="loss_load_cost", overgeneration_cost="overgeneration_cost"),
    "load": {}, "pv": {},
}
_KIND = {"load": "fixed", "pv": "flex", "unbalanced_energy": "flex", "genset": "controllable", "battery": "controllable",
         "grid": "controllable"}
_CLASS = {"load": "LoadModule", "pv": "RenewableModule", "unbalanced_energy": "UnbalancedEnergyModule", "genset": "GensetModule",
          "battery": "BatteryModule", "grid": "GridModule"}


class ModuleView:
    """One module of the microgrid: ``name`` = (module name, number) as in the reference (base_module.py:600-612)."""

    def __init__(self, env, kind, j):
        self._env, self._kind, self._j = env, kind, j

    @property
    def name(self):
        return (self._kind, self._j)

    @property
    def module_type(self):
        return _KIND[self._kind]

    def _n(self):
        L = self._env.layout
        return dict(load=L.n_load, pv=L.n_pv, unbalanced_energy=1, genset=L.n_genset, battery=L.n_battery, grid=L.n_grid)[self._kind]

    def __getattr__(self, attr):
        if attr.startswith("_"):
            raise AttributeError(attr)
        env, kind, j, n = self._env, self._kind, self._j, self._n()
        cols = _PARAMS[kind]
        if attr in cols:
            v = env._col(cols[attr]) if kind == "unbalanced_energy" else env._col(cols[attr], n)[j]
            return float(v)
        if kind == "genset":
            if attr in ("current_status", "goal_status", "steps_until_up", "steps_until_down"):
                st = unpack_status(env._col("gen_status", n)[j])
                return int(st[("current_status", "goal_status", "steps_until_up", "steps_until_down").index(attr)])
            tm = int(env._col("gen_times", n)[j])
            if attr == "start_up_time":
                return tm & 0xff
            if attr == "wind_down_time":
                return (tm >> 16) & 0xff
            if attr == "allow_abortion":
                return not ((tm >> 8) & 1)
        if kind == "battery":
            if attr == "min_soc":
                return self.min_capacity / self.max_capacity
            if attr == "max_soc":
                return 1.0
        if kind in ("load", "pv", "grid") and attr == "time_series":
            T = env.layout.n_steps
            if kind == "grid":
                return env._series_rows("grid_ts", (T, n, 4), np.arange(T))[:, j, :]
            return env._series_rows(kind + "_ts", (T, n), np.arange(T))[:, j:j + 1]
        if kind in ("load", "pv", "grid") and attr == "forecast_horizon":
            keys = self.state_dict()
            comps = 4 if kind == "grid" else 1
            return len(keys) // comps - 1
        if attr == "current_load" and kind == "load":
            return -1 * next(iter(self.state_dict().values()))            # load_module.py:104-111 (the series is stored negated)
        if attr == "current_renewable" and kind == "pv":
            return next(iter(self.state_dict().values()))
        if attr in ("current_step", "initial_step", "final_step"):
            return int(getattr(env, attr))
        if attr in ("production_marginal_cost", "absorption_marginal_cost"):
            return env.get_cost_info()[kind][j][attr]
        if attr == "marginal_cost":
            return env.get_cost_info()[kind][j]["production_marginal_cost"]
        raise AttributeError(f"'{_CLASS[kind]}' view has no attribute '{attr}'")

    def __setattr__(self, attr, value):
        if attr.startswith("_"):
            return object.__setattr__(self, attr, value)
        raise AttributeError("module views are read-only: the device batch holds the parameters (see MicrogridBatch.from_grids)")

    def state_dict(self, normalized=False):
        """``BaseMicrogridModule.state_dict`` (base_module.py:473-490)."""
        return self._env.state_dict(normalized=normalized)[self._kind][self._j]

    @property
    def state(self):
        return np.array(list(self.state_dict().values()), dtype=np.float64)

    def __repr__(self):
        return f"{_CLASS[self._kind]}View{self.name}"


class ModuleContainerView:
    """``ModuleContainer`` (modules/module_container.py): module name -> list of modules, in the container's order (fixed, flex,
    controllable); ``.fixed`` / ``.flex`` / ``.controllable`` are sub-containers; attribute and item access by name."""

    def __init__(self, env, only=None):
        self._env, self._only = env, only

    def _names(self):
        return [(n, k) for n, k in self._env._container_order() if self._only is None or _KIND[n] == self._only]

    def names(self):
        return [n for n, _ in self._names()]

    def to_dict(self):
        return {n: [ModuleView(self._env, n, j) for j in range(k)] for n, k in self._names()}

    def iterdict(self):
        return iter(self.to_dict().items())

    def to_list(self):
        return [m for lst in self.to_dict().values() for m in lst]

    def iterlist(self):
        return iter(self.to_list())

    def to_tuples(self):
        return [(n, m) for n, lst in self.to_dict().items() for m in lst]

    @property
    def fixed(self):
        return ModuleContainerView(self._env, "fixed")

    @property
    def flex(self):
        return ModuleContainerView(self._env, "flex")

    @property
    def controllable(self):
        return ModuleContainerView(self._env, "controllable")

    def __getitem__(self, name):
        d = self.to_dict()
        if name not in d:
            raise KeyError(name)
        return d[name]

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __contains__(self, name):
        return name in self.names()

    def __iter__(self):
        return iter(self.names())

    def __len__(self):
        return sum(k for _, k in self._names())

    def __repr__(self):
        return repr({n: [repr(m) for m in lst] for n, lst in self.to_dict().items()})
