"""Scenario loader: the reference's !Microgrid YAML + csv.gz files -> parameter dicts.
Needs the reference's data directory, so it only runs in the build container."""
import os

import numpy as np
import pytest

REF_SCENARIOS = "/root/reference/src/pymgrid/data/scenario"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SCENARIOS), reason="reference data files not present")


def test_yaml_loader_equals_reference_loader(pymgrid25):
    """All 25 pymgrid25 scenarios: the parameter dict read from YAML + csv.gz by this repo's loader equals the one
    extracted from Microgrid.from_scenario(n) by the reference (fixture pymgrid25_inputs.npz), bit for bit."""
    from pymgrid_amd.scenario import from_scenario
    for n, ref in enumerate(pymgrid25):
        p = from_scenario(n, REF_SCENARIOS)
        order = p.pop("controllable_order")               # module-list order of the YAML: never grid before battery here
        assert not ("grid" in order and "battery" in order and order.index("grid") < order.index("battery")), n
        assert set(p) == set(ref), (n, set(p) ^ set(ref))
        for k, v in ref.items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(np.asarray(p[k]).reshape(v.shape), v), (n, k)
            elif isinstance(v, dict):
                assert p[k] == v, (n, k, p[k], v)
            else:
                assert p[k] == v, (n, k)


def test_state_checkpoint_roundtrip(tmp_path, pymgrid25):
    import torch
    from pymgrid_amd import MicrogridBatch
    from pymgrid_amd.scenario import load_state, save_state
    b = MicrogridBatch.from_grids([pymgrid25[2], pymgrid25[3]], device="cpu")
    b.cols["charge"] += 1.5
    save_state(b, 17, str(tmp_path / "ck.npz"))
    b2 = MicrogridBatch.from_grids([pymgrid25[2], pymgrid25[3]], device="cpu")
    assert load_state(b2, str(tmp_path / "ck.npz")) == 17
    for k in ("charge", "soc", "gen_status"):
        assert torch.equal(b.cols[k], b2.cols[k])

