"""Rule-based control: RuleBasedControl.run of the reference (algos/rbc/rbc.py:64-93) vs the oracle's
rollout (CPU) and the fused on-device rollout (GPU), bit-exact, for a full year of every pymgrid25 scenario."""
import json

import numpy as np
import pytest
import torch

from conftest import golden


def _plist_tuple(arr):
    return tuple((int(m), int(a)) for m, a in arr if m >= 0)


def _generated_grids():
    z = golden("generated.npz")
    meta = json.loads(str(z["meta"]))
    grids = []
    for i, m in enumerate(meta):
        p = dict(m)
        p["horizon"] = 0
        for k in ("load_ts", "pv_ts", "grid_ts"):
            if f"g{i}_{k}" in z.files:
                p[k] = z[f"g{i}_{k}"]
        grids.append(p)
    return grids


def _cases(pymgrid25):
    return [(f"s{n}", p) for n, p in enumerate(pymgrid25)] + [(f"g{i}", p) for i, p in enumerate(_generated_grids())]


def test_rbc_priority_list_and_oracle_rollout(pymgrid25, oracle):
    """Host: sorted(priority_lists[0]) by marginal cost == the reference's RBC list.  Oracle: the whole episode."""
    from pymgrid_amd import MicrogridBatch
    from pymgrid_amd.priority_list import get_priority_lists, table_array
    from pymgrid_amd.rbc import default_priority_ids
    z = golden("rbc.npz")
    for key, p in _cases(pymgrid25):
        b = MicrogridBatch.from_grids([p], device="cpu")
        L = b.layout
        redundant = L.has_genset and p["genset"]["running_min_production"] == 0
        lists = get_priority_lists(L.has_genset, L.has_battery, L.has_grid, redundant)
        ids = default_priority_ids(b, lists)
        assert lists[ids[0]] == _plist_tuple(z[f"{key}_plist"]), key
        cols = b.numpy_columns()
        st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
        ref = z[f"{key}_reward"]
        r = oracle.rollout_batch(cols, st, L.initial_step, len(ref), ids, table_array(lists))
        assert np.array_equal(r[:, 0], ref), key
        fin = z[f"{key}_final"]
        assert st["charge"][0] == fin[0] and st["soc"][0] == fin[1]


@pytest.mark.gpu
def test_rbc_device_full_year_vs_reference(pymgrid25, device):
    """GPU: RuleBasedControl over batched scenarios (one fused rollout, control expanded in-kernel, no action
    stream) reproduces the reference's per-step reward for the whole year and the final battery / genset state."""
    from pymgrid_amd import DiscreteBatchedMicrogridEnv, MicrogridBatch, RuleBasedControl, unpack_status
    from pymgrid_amd.scenario import bucket_by_layout
    z = golden("rbc.npz")
    cases = _cases(pymgrid25)
    grids = [p for _, p in cases]
    for idx in bucket_by_layout(grids).values():
        sub = [grids[i] for i in idx]
        rmin0 = [g.get("genset") is not None and g["genset"]["running_min_production"] == 0 for g in sub]
        assert all(rmin0) or not any(rmin0)
        env = DiscreteBatchedMicrogridEnv(MicrogridBatch.from_grids(sub, device=device), observations=False)
        rbc = RuleBasedControl(env)
        for j, i in enumerate(idx):
            assert rbc.priority_list[j] == _plist_tuple(z[f"{cases[i][0]}_plist"])
        res = rbc.run(chunk=1000)
        reward = res["reward"].cpu().numpy()
        charge, soc = env.batch.cols["charge"].cpu().numpy(), env.batch.cols["soc"].cpu().numpy()
        for j, i in enumerate(idx):
            key = cases[i][0]
            assert np.array_equal(reward[:, j], z[f"{key}_reward"]), key
            fin = z[f"{key}_final"]
            assert charge[j] == fin[0] and soc[j] == fin[1]
            if env.layout.has_genset:
                st = unpack_status(env.batch.cols["gen_status"].cpu().numpy().view(np.uint32))[j]
                assert np.array_equal(st, fin[2:].astype(np.int32))
        assert torch.allclose(res["episode_return"], res["reward"].sum(0), rtol=1e-12)
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["genset+battery", "battery+grid", "genset+battery+grid"])
def test_discrete_rollout_equals_step_loop_and_oracle(arch, device, oracle):
    """Per-step ids [K, N]: the fused rollout == K x (expand + step) through the Gym surface == oracle."""
    from pymgrid_amd import DiscreteBatchedMicrogridEnv
    from pymgrid_amd.generator import generate
    N, T, K = 3000, 60, 41
    e1 = DiscreteBatchedMicrogridEnv(generate(N, n_steps=T, seed=21, arch=arch, device=device, mixed_timers=True),
                                     observations=False, remove_redundant_gensets=False)
    e2 = DiscreteBatchedMicrogridEnv(generate(N, n_steps=T, seed=21, arch=arch, device=device, mixed_timers=True),
                                     observations=False, remove_redundant_gensets=False)
    gen = torch.Generator(device=device); gen.manual_seed(3)
    ids = torch.randint(0, e1.action_space.n, (K, N), dtype=torch.uint8, device=device, generator=gen)
    cols = e1.batch.numpy_columns()
    st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
    out = e1.engine.rollout_discrete(ids, e1._table, K, reward=True, soc_trace=True, status_trace=True, log=True)
    loop = torch.stack([e2.step(ids[k].to(torch.int32))[1] for k in range(K)])
    assert torch.equal(out["reward"], loop)
    for k in ("charge", "soc", "gen_status"):
        if k in e1.batch.cols:
            assert torch.equal(e1.batch.cols[k], e2.batch.cols[k]), k
    ref = oracle.rollout_batch(cols, st, 0, K, ids.cpu().numpy(), e1._table, nthreads=8)
    assert np.array_equal(out["reward"].cpu().numpy(), ref)
    assert np.array_equal(e1.batch.cols["charge"].cpu().numpy(), st["charge"])
    # fixed list per grid (RBC form): ids [N]
    e1.engine.reset(want_obs=False); e2.engine.reset(want_obs=False)
    fixed = ids[0].contiguous()
    r1 = e1.engine.rollout_discrete(fixed, e1._table, K)["reward"]
    r2 = torch.stack([e2.step(fixed.to(torch.int32))[1] for _ in range(K)])
    assert torch.equal(r1, r2)
    e1.close(); e2.close()


@pytest.mark.gpu
def test_rbc_run_respects_the_episode_window_and_leaves_the_env_usable(pymgrid25, device):
    """RuleBasedControl.run resets THROUGH the env: a trajectory_func redraws the window (microgrid.py:205-225) and the run
    stops at that window's `done` (rbc.py:86-91), not at the end of the series; afterwards env.step returns observations
    of the current counter (the prefetch rings were refilled); restore_state=True mirrors the reference's deep copy."""
    from pymgrid_amd import BatchedMicrogridEnv, MicrogridBatch, RuleBasedControl
    from pymgrid_amd.trajectory import DeterministicTrajectory
    tmpl4 = [pymgrid25[n] for n in (2, 3, 5, 7)]
    mk = lambda **kw: BatchedMicrogridEnv(MicrogridBatch.from_grids(tmpl4, device=device), observations=True, **kw)
    env = mk(trajectory_func=DeterministicTrajectory(100, 160), obs_prefetch=8)
    whole = mk()
    state0 = env.batch.state()
    res = RuleBasedControl(env).run(restore_state=True)
    assert res["reward"].shape[0] == 60 and env.current_step == 160           # done fired at counter 159
    for k, v in state0.items():
        assert torch.equal(env.batch.cols[k], v)                              # the batch was handed back as it was
    ref = RuleBasedControl(whole)
    whole.engine.set_window(100, 160)
    r2 = ref.run()                                                            # same window set by hand, state kept
    assert torch.equal(res["reward"], r2["reward"]) and r2["reward"].shape[0] == 60
    res3 = RuleBasedControl(env).run(max_steps=7)
    assert res3["reward"].shape[0] == 7 and env.current_step == 107
    # the env is usable after the run: its observation equals a fresh env brought to the same counter and state
    probe = mk()
    probe.batch.load_state(env.batch.state())
    probe.reset(107)
    a = env.sample_action()
    o1, r1, _, _ = env.step(a)
    o2, r2_, _, _ = probe.step(a)
    assert torch.equal(o1, o2) and torch.equal(r1, r2_)
    env.close(); whole.close(); probe.close()


@pytest.mark.gpu
def test_rbc_on_one_microgrid_returns_the_log_frame(pymgrid25, device):
    """``RuleBasedControl(microgrid).run(max_steps)`` on an N = 1 adaptor returns the microgrid's log as a DataFrame like the
    reference (rbc.py:64-93, microgrid.py:434-475): its ('balance', 0, 'reward') column == the fixture's rewards; `microgrid`,
    `get_empty_action` and `priority_list` read like PriorityListAlgo's (priority_list.py:169-180)."""
    from pymgrid_amd import RuleBasedControl
    from pymgrid_amd.envs import DiscreteMicrogridEnv, MicrogridEnv
    z = golden("rbc.npz")
    for n in (0, 3, 7):
        for cls in (MicrogridEnv, DiscreteMicrogridEnv):
            env = cls(pymgrid25[n], device=str(device), log=True)
            rbc = RuleBasedControl(env)
            assert rbc.microgrid is env and rbc.get_empty_action() == env.get_empty_action()
            assert rbc.priority_list[0] == _plist_tuple(z[f"s{n}_plist"])
            frame = rbc.run(max_steps=60)
            assert len(frame) == 60 and frame.columns.names == ["module_name", "module_number", "field"]
            assert np.array_equal(frame[("balance", 0, "reward")].to_numpy(), z[f"s{n}_reward"][:60])
            assert len(env.log) == 60 and env.current_step == env.initial_step + 60
            res = rbc.run(max_steps=10, as_frame=False)
            assert res["reward"].shape == (10, 1)                              # (tensors on request; the state carries over from the first run)
            env.close()
