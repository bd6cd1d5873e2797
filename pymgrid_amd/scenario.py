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
    """Group microgrids that can share one SoA batch (same module set, series length, horizon, window).
    Returns {key: [indices]} in first-seen order."""
    buckets = {}
    for i, p in enumerate(grids):
        key = (architecture(p), np.asarray(p["load_ts"]).shape[0], int(p.get("horizon", 0)),
               int(p.get("initial_step", 0)), int(p.get("final_step", 0)))
        buckets.setdefault(key, []).append(i)
    return buckets
