"""Read-only views of ONE microgrid's modules in the reference's vocabulary: ``microgrid.modules`` / ``.fixed`` / ``.flex`` /
``.controllable`` / ``.module_list`` (microgrid.py:761-818, modules/module_container.py:355-413) over the columns of an N = 1
batch.  A view holds no data: every attribute is read from the batch when it is asked for, so it shows the state the kernels
left.  Parameters are those of the modules' constructors (battery_module.py:60-106, genset_module.py:61-98, grid_module.py:58-96,
load_module.py:45-70, renewable_module.py:45-70, unbalanced_energy_module.py:9-26); writing is not offered (the device path has
no per-module Python objects to write to -- ``set_module_attr`` covers the step window)."""
import numpy as np

from .batch import unpack_status

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
