import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refcheck: imports the real reference from /root/reference (build container only)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def pymgrid25():
    """The 25 benchmark scenarios as parameter dicts (fixture derived from the reference's data files)."""
    from pymgrid_amd.scenario import load_npz_grids
    return load_npz_grids(os.path.join(GOLDEN, "pymgrid25_inputs.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pymgrid_amd import _lib
    _lib.build()
    return torch.device("cuda:0")


def actions_for(p, row):
    """flat action row -> oracle action dict (column order genset(goal, energy), battery, grid)."""
    a, c = {}, 0
    if p.get("genset") is not None:
        a["genset"] = row[c:c + 2]; c += 2
    if p.get("battery") is not None:
        a["battery"] = row[c]; c += 1
    if p.get("grid") is not None:
        a["grid"] = row[c]; c += 1
    return a


def action_dim(p):
    return 2 * (p.get("genset") is not None) + (p.get("battery") is not None) + (p.get("grid") is not None)


def multi_cases():
    """The reference-made fixtures for microgrids with several gensets / batteries / grids (tests/golden/multi.npz,
    make_multi_goldens.py): yields (case index, parameter dict in this repo's vocabulary, meta, npz)."""
    import json
    z = golden("multi.npz")
    for ci, mt in enumerate(json.loads(str(z["meta"]))):
        p = dict(load_ts=np.stack([z[f"c{ci}_load_{j}"] for j in range(mt["n_load"])], axis=1),
                 pv_ts=np.stack([z[f"c{ci}_pv_{j}"] for j in range(mt["n_pv"])], axis=1),
                 horizon=mt["horizon"], final_step=mt["T"], initial_step=0,
                 unbalanced=dict(loss_load_cost=mt["loss_load_cost"], overgeneration_cost=mt["overgeneration_cost"]),
                 genset=mt["genset"], battery=mt["battery"], grid=mt["grid"],
                 grid_ts=[z[f"c{ci}_grid_ts_{j}"] for j in range(len(mt["grid"]))],
                 controllable_order=[k for k in mt["order"] if k in ("genset", "battery", "grid")])
        yield ci, p, mt, z
