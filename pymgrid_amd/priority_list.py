"""Priority lists of the discrete environment: host-side mirror of ``PriorityListAlgo.get_priority_lists``
(algos/priority_list/priority_list.py:15-67) and ``PriorityListElement`` (priority_list_element.py:8-80).

An element is ``(module, action)`` with module 0 genset / 1 battery / 2 grid.  A genset contributes two elements
(its action space has two entries: goal status 0 or 1), battery and grid one each.
"""
from itertools import permutations

import numpy as np

GENSET, BATTERY, GRID = 0, 1, 2
MODULE_NAMES = {GENSET: "genset", BATTERY: "battery", GRID: "grid"}


def get_priority_lists(has_genset, has_battery, has_grid, remove_redundant_gensets=False, grid_before_battery=False):
    """All priority lists in the reference's order (``grid_before_battery``: the source-and-sink modules in the order of
    the microgrid's module list, BatchLayout.grid_before_battery).

    controllable sources first (genset), then source_and_sinks (battery, grid) -- priority_list.py:26-33;
    every permutation, later repeats of a module dropped (:40-47), duplicates removed keeping first
    occurrence (:48); with ``remove_redundant_gensets`` (gensets whose running_min_production == 0) lists that
    hold the genset "off" element are dropped (:53-67)."""
    elements = []
    if has_genset:
        elements += [(GENSET, 0), (GENSET, 1)]
    sas = ([(BATTERY, 0)] if has_battery else []) + ([(GRID, 0)] if has_grid else [])
    elements += sas[::-1] if (grid_before_battery and has_battery and has_grid) else sas
    pls = []
    for perm in permutations(elements):
        seen, pl = set(), []
        for mod, act in perm:
            if mod not in seen:
                seen.add(mod)
                pl.append((mod, act))
        pls.append(tuple(pl))
    unique = list(dict.fromkeys(pls))
    if remove_redundant_gensets:
        unique = [pl for pl in unique if (GENSET, 0) not in pl]
    return unique


def table_array(priority_lists):
    """-> int32 [n_actions, 3, 2] (module, action), -1 padded: the layout ``mgx_expand_discrete`` takes."""
    tab = -np.ones((len(priority_lists), 3, 2), dtype=np.int32)
    for i, pl in enumerate(priority_lists):
        for j, (mod, act) in enumerate(pl):
            tab[i, j] = (mod, act)
    return tab


def get_instance_priority_lists(n_genset, n_battery, n_grid, redundant_gensets=(), grid_before_battery=False):
    """Priority lists over module INSTANCES for microgrids with several gensets / batteries / grids: elements are
    ``(kind, instance, action)``.  The reference's enumeration, literally (priority_list.py:15-67): elements = every
    controllable source (gensets in list order, two actions each), then every source-and-sink (names in module-list order,
    instances in order); all permutations; a module met again in a permutation is dropped; duplicates removed keeping the
    first occurrence; lists holding the "off" element of a genset in ``redundant_gensets`` (running_min_production == 0)
    are removed.  The count grows factorially with the number of elements, as in the reference."""
    elements = [(GENSET, j, a) for j in range(n_genset) for a in (0, 1)]
    bats, grids = [(BATTERY, j, 0) for j in range(n_battery)], [(GRID, j, 0) for j in range(n_grid)]
    elements += (grids + bats) if (grid_before_battery and n_battery and n_grid) else (bats + grids)
    if len(elements) > 9:
        raise ValueError(f"{len(elements)} priority-list elements: the reference's enumeration is factorial "
                         "(priority_list.py:35: every permutation) -- more than 9 elements is not offered")
    pls = []
    for perm in permutations(elements):
        seen, pl = set(), []
        for kind, inst, act in perm:
            if (kind, inst) not in seen:
                seen.add((kind, inst))
                pl.append((kind, inst, act))
        pls.append(tuple(pl))
    unique = list(dict.fromkeys(pls))
    off = {(GENSET, int(j), 0) for j in redundant_gensets}
    if off:
        unique = [pl for pl in unique if not any(el in off for el in pl)]
    return unique


def lists_array(priority_lists):
    """-> int32 [n_lists, list_len, 3] (kind, instance, action), -1 padded: the layout ``mgx_expand_lists`` takes."""
    width = max(len(pl) for pl in priority_lists)
    tab = -np.ones((len(priority_lists), width, 3), dtype=np.int32)
    for i, pl in enumerate(priority_lists):
        for j, el in enumerate(pl):
            tab[i, j] = el
    return tab
