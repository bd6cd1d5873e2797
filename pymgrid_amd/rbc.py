"""Rule-based control on device -- the batched counterpart of ``pymgrid.algos.RuleBasedControl``
(algos/rbc/rbc.py:8-140): one fixed priority list per microgrid (by default the controllable modules sorted by
marginal cost, rbc.py:31-50 with PriorityListElement.__lt__, priority_list_element.py:70-80), deployed for a whole
episode.  The control is expanded inside the fused rollout kernel, so a year of N microgrids runs with no host in
the loop and no action stream in HBM (``mgx_rollout_discrete`` with a constant id per grid).
"""
import numpy as np
import torch

from .priority_list import BATTERY, GENSET, GRID, get_priority_lists, table_array


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
        out[GRID] = cols["grid_ts"][t, 0]
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


class RuleBasedControl:
    """``RuleBasedControl(microgrid).run()`` for a batch.

    Parameters mirror the reference: ``priority_list`` None -> marginal-cost order per grid; or one list of
    ``(module, action)`` pairs applied to every grid.  ``run`` returns ``{"reward": [K, N], ...}`` (the balance
    log's reward column; pass ``log=True`` for every log column) and leaves the engine at the end of the episode.
    """

    def __init__(self, env, priority_list=None, remove_redundant_gensets=True):
        from .envs import BatchedMicrogridEnv
        if not isinstance(env, BatchedMicrogridEnv):
            raise TypeError("env must be a (Discrete)BatchedMicrogridEnv")
        self.env, self.engine, self.batch, self.layout = env, env.engine, env.batch, env.layout
        L = self.layout
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

    def run(self, max_steps=None, chunk=512, log=False, soc_trace=False, reward=True, restore_state=False):
        """``RuleBasedControl.run`` (rbc.py:64-93): reset the microgrids -- through the env, so a ``trajectory_func``
        redraws the episode window as ``Microgrid.reset`` does (microgrid.py:205-225) -- then deploy the priority lists
        until ``done`` (the end of the CURRENT episode window) or for ``max_steps`` steps.  The reference works on a deep
        copy of the microgrid; here the batch itself is stepped unless ``restore_state=True`` puts battery charge / SoC and
        genset status back afterwards."""
        L = self.layout
        self.env.reset()
        lo, hi = self.engine.window                       # the window the reset has just installed
        total = hi - lo                                   # done fires at counter hi - 1: hi - lo steps in all
        if max_steps is not None:
            total = min(total, int(max_steps))
        saved = self.batch.state() if restore_state else None
        try:
            return self._run(total, chunk, log, soc_trace, reward)
        finally:
            if saved is not None:
                self.batch.load_state(saved)
            self.env._after_external_steps()              # the env's observation rings follow the counter again

    def _run(self, total, chunk, log, soc_trace, reward):
        L = self.layout
        ret = torch.zeros(L.n_grids, dtype=torch.float64, device=self.batch.device)
        parts = {}
        done = 0
        if L.n_load != 1 or L.n_pv != 1:
            # several load / renewable modules per grid: no fused kernel for that layout -- one expand + step per env-step
            ids = self._ids_dev.to(torch.int32)
            rows = {"reward": [], "soc_trace": [], "log": []}
            for _ in range(total):
                control = self.engine.expand_discrete(ids, self._table)
                _, r, _, lg = self.engine.step(control, normalized=False, want_obs=False, want_log=log)
                ret += r
                if reward:
                    rows["reward"].append(r)
                if soc_trace and L.has_battery:
                    rows["soc_trace"].append(self.batch.cols["soc"].clone())
                if log:
                    rows["log"].append(lg)
            res = {name: torch.stack(v) for name, v in rows.items() if v}
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
