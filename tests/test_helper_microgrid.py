"""The fixture microgrid of the reference's OWN test suite (tests/helpers/modular_microgrid.py:14-37), used there by
tests/control/test_rbc.py and tests/microgrid/test_microgrid.py: genset + lossless battery + renewable + load +
import-only grid.  Golden data from the real reference (tests/golden/make_goldens.py make_helper): 60 seeded random
steps with observations, RuleBasedControl.run(10), 40 DiscreteMicrogridEnv steps.  Oracle on the CPU, HIP engine on
the GPU; every comparison is exact."""
import json

import numpy as np
import pytest
import torch

from conftest import action_dim, actions_for, golden


def _params(z):
    p = json.loads(str(z["meta"]))
    for k in z.files:
        if k.startswith("in_"):
            p[k[3:]] = z[k]
    return p


def test_oracle_random_actions_rbc_and_discrete(oracle):
    from pymgrid_amd import MicrogridBatch
    from pymgrid_amd.priority_list import MODULE_NAMES, get_priority_lists, table_array
    from pymgrid_amd.rbc import default_priority_ids, marginal_costs
    z = golden("helper_microgrid.npz")
    p = _params(z)
    names = [str(s) for s in z["log_names"]]
    om = oracle.OracleMicrogrid(p)
    acts = z["rand_actions"]
    obs = om.reset()
    for k in range(len(acts)):
        out = om.run(actions_for(p, acts[k]), True)
        assert out.reward == z["rand_reward"][k] and out.done == z["rand_done"][k], k
        assert om.s.charge == z["rand_charge"][k] and om.s.soc == z["rand_soc"][k], k
        assert list(om.status) == z["rand_status"][k].tolist(), k
        assert np.array_equal(om.observe(), z["rand_obs"][k]), k
        d = out.as_dict()
        for j, name in enumerate(names):
            ref = z["rand_log"][k, j]
            if not np.isnan(ref):
                assert d[name] == ref, (k, name)
    # RuleBasedControl: sorted by marginal cost (tests/control/test_rbc.py:20-26), 10 steps (:28-38)
    b = MicrogridBatch.from_grids([p], device="cpu")
    L = b.layout
    lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, False)
    ids = default_priority_ids(b, lists)
    assert lists[ids[0]] == tuple((int(m), int(a)) for m, a in z["rbc_plist"])
    mc = z["rbc_marginal_cost"]
    assert (mc[:-1] <= mc[1:]).all()
    costs = marginal_costs(b.cols, L, 0)
    assert [float(costs[int(m)][0]) for m, _ in z["rbc_plist"]] == mc.tolist()
    cols = b.numpy_columns()
    st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status")}
    r = oracle.rollout_batch(cols, st, 0, 10, ids, table_array(lists))
    assert np.array_equal(r[:, 0], z["rbc_reward"])
    assert st["charge"][0] == z["rbc_final"][0] and st["soc"][0] == z["rbc_final"][1]
    # DiscreteMicrogridEnv (tests/envs/test_discrete.py:73-80: n = 3! * 2)
    assert int(z["disc_n"]) == len(lists) == 12
    om = oracle.OracleMicrogrid(p)
    for k, a in enumerate(z["disc_ids"]):
        act = om.populate_action([(MODULE_NAMES[m], act_) for m, act_ in lists[int(a)]])
        assert om.run(act, normalized=False).reward == z["disc_reward"][k], k


@pytest.mark.gpu
def test_device_random_actions_rbc_and_discrete(device):
    from pymgrid_amd import (DiscreteBatchedMicrogridEnv, DiscreteMicrogridEnv, MicrogridBatch, MicrogridEnv,
                             RuleBasedControl, unpack_status)
    z = golden("helper_microgrid.npz")
    p = _params(z)
    names = [str(s) for s in z["log_names"]]
    env = MicrogridEnv(p, device=device)
    assert env.current_step == 0                                   # test_microgrid.py:104-112 (test_current_step)
    env.reset()
    acts = z["rand_actions"]
    for k in range(len(acts)):
        ctrl = {n: ([v] if n != "genset" else [np.asarray(v)]) for n, v in actions_for(p, acts[k]).items()}
        obs, reward, done, info = env.step(ctrl, normalized=True)
        assert env.current_step == k + 1
        assert reward == z["rand_reward"][k] and int(done) == z["rand_done"][k], k
        assert np.array_equal(np.asarray(obs, dtype=np.float64), z["rand_obs"][k]), k
        st = unpack_status(np.array([env.last_log["genset_status"]], dtype=np.uint32))[0].tolist()
        assert st == z["rand_status"][k].tolist(), k
        for j, name in enumerate(names):
            ref = z["rand_log"][k, j]
            if not np.isnan(ref) and name in env.last_log:
                assert env.last_log[name] == ref, (k, name)
    env.reset()                                                    # :114-122 (test_current_step_after_reset)
    assert env.current_step == 0
    env.close()
    venv = DiscreteBatchedMicrogridEnv(MicrogridBatch.from_grids([p], device=device), remove_redundant_gensets=False)
    rbc = RuleBasedControl(venv, remove_redundant_gensets=False)
    assert rbc.priority_list[0] == tuple((int(m), int(a)) for m, a in z["rbc_plist"])
    res = rbc.run(10)
    assert res["reward"].shape == (10, 1)                          # test_rbc.py:28-38: len(log) == n_steps
    assert np.array_equal(res["reward"][:, 0].cpu().numpy(), z["rbc_reward"])
    assert venv.batch.cols["charge"][0].item() == z["rbc_final"][0]
    venv.close()
    denv = DiscreteMicrogridEnv(p, device=device, remove_redundant_gensets=False)
    assert denv.action_space.n == int(z["disc_n"])
    denv.reset()
    for k, a in enumerate(z["disc_ids"]):
        assert denv.step(int(a))[1] == z["disc_reward"][k], k
    denv.close()
