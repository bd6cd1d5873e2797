"""Rule-based control on device -- the batched counterpart of ``pymgrid.algos.RuleBasedControl``
(algos/rbc/rbc.py:8-140): one fixed priority list per microgrid (by default the controllable modules sorted by
marginal cost, rbc.py:31-50 with PriorityListElement.__lt__, priority_list_element.py:70-80), deployed for a whole
episode.  The control is expanded inside the fused rollout kernel, so a year of N microgrids runs with no host in
the loop and no action stream in HBM (``mgx_rollout_discrete`` with a constant id per grid).
"""
import numpy as np
import torch

from .priority_list import BATTERY, GENSET, GRID, get_instance_priority_lists, get_priority_lists, lists_array, table_array


def marginal_costs(cols, layout, t):
    """module -> [N] marginal cost as the reference's modules report it at step t:
    genset get_cost(1.0) (genset_module.py:188-205,519-521), battery battery_cost_cycle (battery_module.py:340-346),
    grid the current import price (grid_module.py:322-324)."""
    out = {}
    if layout.has_genset:
        production = 1.0
        co2 = cols["gen_co2_per_unit"] * production
        out[GENSET] = cols["gen_cost"] * production + cols["gen_cost_per_unit_co2"] * co2
    if layout.has_battery:
        out[BATTERY] = cols["bat_cost_cycle"]
    if layout.has_grid:
        if cols.get("grid_ts") is not None:
            out[GRID] = cols["grid_ts"][t, 0]
        else:                              # factorised series: the import price is the tariff pattern's value at hour t % 24
            from .generator import electricity_tariff
            pat = np.asarray(cols["tariff"].cpu())
            p1, p2 = electricity_tariff(1, t + 1)[t], electricity_tariff(2, t + 1)[t]
            out[GRID] = np.where(pat == 1, p1, np.where(pat == 2, p2, 0.0))
    return {k: np.asarray(v.cpu() if torch.is_tensor(v) else v, dtype=np.float64) for k, v in out.items()}


def default_priority_ids(batch, actions_list, remove_redundant_gensets=True, t=None):
    """Per-grid index (uint8 [N]) into ``actions_list`` of ``sorted(priority_lists[0])`` (rbc.py:46-47)."""
    L = batch.layout
    t = L.initial_step if t is None else t
    N = L.n_grids
    if remove_redundant_gensets and L.has_genset:
        rmin = batch.cols["gen_running_min"].cpu().numpy()
        zero = rmin == 0
        if zero.any() and not zero.all():
            raise ValueError("remove_redundant_gensets: mixed running_min_production == 0 / > 0 in one batch")
        redundant = bool(zero.all())
    else:
        redundant = False
    first = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, redundant, L.grid_before_battery)[0]
    costs = marginal_costs(batch.cols, L, t)
    # stable sort of the elements of `first` by (marginal cost ascending, action descending)
    cost = np.stack([costs[m] for m, _ in first], axis=1)                  # [N, n_el]
    action = np.array([a for _, a in first])
    order = np.argsort(-action, kind="stable")                             # ties: higher action first
    order = np.broadcast_to(order, (N, len(first))).copy()
    key = np.take_along_axis(cost, order, axis=1)
    order = np.take_along_axis(order, np.argsort(key, axis=1, kind="stable"), axis=1)
    index = {pl: j for j, pl in enumerate(actions_list)}
    ids = np.empty(N, dtype=np.uint8)
    cache = {}
    for code in np.unique(order, axis=0):
        pl = tuple(first[j] for j in code)
        cache[tuple(code)] = index[pl]
    for code, j in cache.items():
        ids[(order == np.array(code)).all(axis=1)] = j
    return ids


def instance_marginal_costs(cols, layout, t):
    """(kind, instance) -> [N] marginal cost for layouts with several gensets / batteries / grids (columns [n, N])."""
    N, out = layout.n_grids, {}

    def rows(name, n):
        return np.asarray(cols[name].cpu(), dtype=np.float64).reshape(n, N)
    if layout.has_genset:
        co2 = rows("gen_co2_per_unit", layout.n_genset) * 1.0
        cost = rows("gen_cost", layout.n_genset) * 1.0 + rows("gen_cost_per_unit_co2", layout.n_genset) * co2
        for j in range(layout.n_genset):
            out[(GENSET, j)] = cost[j]
    if layout.has_battery:
        c = rows("bat_cost_cycle", layout.n_battery)
        for j in range(layout.n_battery):
            out[(BATTERY, j)] = c[j]
    if layout.has_grid:
        price = np.asarray(cols["grid_ts"][t].cpu(), dtype=np.float64).reshape(layout.n_grid, 4, N)
        for j in range(layout.n_grid):
            out[(GRID, j)] = price[j, 0]
    return out


def default_instance_priority_ids(batch, actions_list, t=None):
    """Per-grid index (int32 [N]) into ``actions_list`` (lists of (kind, instance, action)) of ``sorted(priority_lists[0])``."""
    L = batch.layout
    t = L.initial_step if t is None else t
    first = actions_list[0]
    costs = instance_marginal_costs(batch.cols, L, t)
    cost = np.stack([costs[(k, j)] for k, j, _ in first], axis=1)          # [N, n_el]
    action = np.array([a for _, _, a in first])
    order = np.argsort(-action, kind="stable")
    order = np.broadcast_to(order, (L.n_grids, len(first))).copy()
    key = np.take_along_axis(cost, order, axis=1)
    order = np.take_along_axis(order, np.argsort(key, axis=1, kind="stable"), axis=1)
    index = {pl: j for j, pl in enumerate(actions_list)}
    ids = np.empty(L.n_grids, dtype=np.int32)
    for code in np.unique(order, axis=0):
        ids[(order == code).all(axis=1)] = index[tuple(first[j] for j in code)]
    return ids


class RuleBasedControl:
    """``RuleBasedControl(microgrid).run()`` for a batch.

    Parameters mirror the reference: ``priority_list`` None -> marginal-cost order per grid; or one list of
    ``(module, action)`` pairs applied to every grid.  ``run`` returns ``{"reward": [K, N], ...}`` (the balance
    log's reward column; pass ``log=True`` for every log column) and leaves the engine at the end of the episode.  On an N = 1
    adaptor (``MicrogridEnv`` / ``DiscreteMicrogridEnv`` with ``log=True``) ``run()`` returns what the reference's does (rbc.py:64-93):
    the microgrid's log as a DataFrame (``as_frame=False`` for the tensors).
    """

    def __init__(self, env, priority_list=None, remove_redundant_gensets=True):
        from .envs import BatchedMicrogridEnv
        if not isinstance(env, BatchedMicrogridEnv):
            raise TypeError("env must be a (Discrete)BatchedMicrogridEnv")
        self.env, self.engine, self.batch, self.layout = env, env.engine, env.batch, env.layout
        L = self.layout
        self._instances = L.n_genset > 1 or L.n_battery > 1 or L.n_grid > 1
        if self._instances:                  # priority lists over module instances: (kind, instance, action) elements
            redundant = []
            if remove_redundant_gensets and L.has_genset:
                rmin = self.batch.cols["gen_running_min"].reshape(L.n_genset, L.n_grids)
                for j in range(L.n_genset):
                    zero = rmin[j] == 0
                    if bool(zero.any()) and not bool(zero.all()):
                        raise ValueError("remove_redundant_gensets: mixed running_min_production == 0 / > 0 in one batch")
                    if bool(zero.all()):
                        redundant.append(j)
            self.actions_list = get_instance_priority_lists(L.n_genset, L.n_battery, L.n_grid, redundant, L.grid_before_battery)
            self._table = None
            self._lists = torch.as_tensor(lists_array(self.actions_list), device=self.batch.device).contiguous()
            if priority_list is None:
                ids = default_instance_priority_ids(self.batch, self.actions_list)
            else:
                pl = tuple((int(k), int(j), int(a)) for k, j, a in priority_list)
                if pl not in self.actions_list:
                    raise ValueError("Invalid priority list. Use RuleBasedControl.get_priority_lists to view all "
                                     "valid priority lists.")
                ids = np.full(L.n_grids, self.actions_list.index(pl), dtype=np.int32)
            self.priority_ids = ids
            self._ids_dev = torch.from_numpy(ids).to(self.batch.device)
            return
        redundant = False
        if remove_redundant_gensets and L.has_genset:
            redundant = bool((self.batch.cols["gen_running_min"] == 0).all().item())
        self.actions_list = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, redundant, L.grid_before_battery)
        self._table = table_array(self.actions_list)
        if priority_list is None:
            ids = default_priority_ids(self.batch, self.actions_list, remove_redundant_gensets)
        else:
            pl = tuple((int(m), int(a)) for m, a in priority_list)
            if pl not in self.actions_list:
                raise ValueError("Invalid priority list. Use RuleBasedControl.get_priority_lists to view all "
                                 "valid priority lists.")
            ids = np.full(L.n_grids, self.actions_list.index(pl), dtype=np.uint8)
        self.priority_ids = ids
        self._ids_dev = torch.from_numpy(ids).to(self.batch.device)

    def get_priority_lists(self):
        return self.actions_list

    @property
    def priority_list(self):
        """Per-grid priority lists as tuples of (module, action)."""
        return [self.actions_list[j] for j in self.priority_ids]

    def reset(self):
        return self.env.reset()

    @property
    def microgrid(self):
        """``PriorityListAlgo.microgrid`` (priority_list.py:169-180): the microgrid(s) under control -- the env."""
        return self.env

    @property
    def modules(self):
        return self.env.modules

    @property
    def fixed(self):
        return self.env.fixed

    @property
    def flex(self):
        return self.env.flex

    def get_empty_action(self):
        return self.env.get_empty_action()

    def run(self, max_steps=None, chunk=512, log=False, soc_trace=False, reward=True, restore_state=False, as_frame=None, verbose=False):
        """``RuleBasedControl.run`` (rbc.py:64-93): reset the microgrids -- through the env, so a ``trajectory_func``
        redraws the episode window as ``Microgrid.reset`` does (microgrid.py:205-225) -- then deploy the priority lists
        until ``done`` (the end of the CURRENT episode window) or for ``max_steps`` steps.  The reference works on a deep
        copy of the microgrid; here the batch itself is stepped unless ``restore_state=True`` puts battery charge / SoC and
        genset status back afterwards."""
        from .envs import _SingleMixin
        if as_frame is None:                              # the reference returns the microgrid's log (rbc.py:93): so does an N = 1 adaptor
            as_frame = isinstance(self.env, _SingleMixin) and getattr(self.env, "_keep_log", False) and not (log or soc_trace)
        if as_frame:
            log = True
        self.env.reset()
        lo, hi = self.engine.window                       # the window the reset has just installed
        total = hi - lo                                   # done fires at counter hi - 1: hi - lo steps in all
        if max_steps is not None:
            total = min(total, int(max_steps))
        saved = self.batch.state() if restore_state else None
        try:
            res = self._run(total, chunk, log, soc_trace, reward)
            if as_frame:                                  # the steps' log rows become the env's log, as if it had been stepped: get_log() works after
                self.env._log_rows = list(res["log"])
                self.env._shaped_rows = list(res["reward"])
                return self.env.get_log()
            return res
        finally:
            if saved is not None:
                self.batch.load_state(saved)
            self.env._after_external_steps()              # the env's observation rings follow the counter again

    def _run(self, total, chunk, log, soc_trace, reward):
        L = self.layout
        ret = torch.zeros(L.n_grids, dtype=torch.float64, device=self.batch.device)
        parts = {}
        done = 0
        if L.multi:
            # several modules of a kind per grid: the general path's K-step kernel with the lists over module instances
            # (a single-instance table is the same thing with instance 0 everywhere)
            lists = self._lists if self._instances else torch.as_tensor(
                lists_array([tuple((m, 0, a) for m, a in pl) for pl in self.actions_list]), device=self.batch.device)
            ids = self._ids_dev.to(torch.int32).contiguous()
            while done < total:
                k = min(chunk, total - done)
                out = self.engine.rollout_lists(ids, lists, k, reward=reward, soc_trace=soc_trace, log=log, ret_acc=ret)
                for name, v in out.items():
                    if name != "ret_acc":
                        parts.setdefault(name, []).append(v)
                done += k
            res = {name: torch.cat(v) for name, v in parts.items()}
            res["episode_return"] = ret
            return res
        while done < total:
            k = min(chunk, total - done)
            out = self.engine.rollout_discrete(self._ids_dev, self._table, k, reward=reward, soc_trace=soc_trace,
                                               log=log, ret_acc=ret)
            for name, v in out.items():
                if name != "ret_acc":
                    parts.setdefault(name, []).append(v)
            done += k
        res = {name: torch.cat(v) for name, v in parts.items()}
        res["episode_return"] = ret
        return res
