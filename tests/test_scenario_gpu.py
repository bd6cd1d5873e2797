"""SURVEY 8(f2) on the GPU box: the reference's on-disk formats feed the device path.

`!Microgrid` YAML + csv.gz files (microgrid.py:820-908, utils/serialize.py:24-112) -> parameter dicts -> SoA device tensors ->
a full year of steps `==` what the REFERENCE produced (tests/golden/pymgrid25_run.npz); and the SoA state checkpoint
(base_module.py:826-850's `state` blocks) saved mid-year, loaded into a fresh batch built from the files, continued `==` the
uninterrupted run.  The files are written by this repo's dump (the CPU suite checks that the reference loads what it dumps and that
the loader equals the reference's on the reference's own 25 scenario directories: tests/test_scenario_loader.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import action_dim, golden

pytestmark = pytest.mark.gpu


def _t(a, device, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=device)


@pytest.fixture(scope="module")
def scenario_root(pymgrid25, tmp_path_factory):
    """The 25 scenarios written as pymgrid25/microgrid_<n>/microgrid_<n>.yaml + data/cls_params/*/time_series.csv.gz."""
    from pymgrid_amd.scenario import dump_scenario_yaml
    root = tmp_path_factory.mktemp("scenario")
    for n, p in enumerate(pymgrid25):
        d = root / "pymgrid25" / f"microgrid_{n}"
        os.makedirs(d)
        dump_scenario_yaml(p, str(d / f"microgrid_{n}.yaml"))
    return str(root)


def _year(eng, acts, device, k0=0, k1=None, chunk=512):
    K = acts.shape[0] if k1 is None else k1
    rew, soc, status = [], [], []
    for k in range(k0, K, chunk):
        out = eng.step_k(_t(acts[k:min(k + chunk, K)], device), normalized=True, reward=True, soc_trace=True, status_trace=True)
        rew.append(out["reward"]); soc.append(out["soc_trace"])
        if "status_trace" in out:
            status.append(out["status_trace"])
    return (torch.cat(rew).cpu().numpy(), torch.cat(soc).cpu().numpy(),
            torch.cat(status).cpu().numpy().view(np.uint32) if status else None)


def test_scenario_files_to_device_full_year(scenario_root, device):
    """from_scenario(n, root) for all 25 -> one SoA batch per module set -> 8 759 steps `==` the reference's rewards, SoC and genset
    status of every step."""
    from pymgrid_amd import MicrogridBatch, StepEngine, unpack_status
    from pymgrid_amd.scenario import bucket_by_layout, from_scenario
    z = golden("pymgrid25_run.npz")
    grids = [from_scenario(n, scenario_root) for n in range(25)]
    seen = 0
    for idx in bucket_by_layout(grids).values():
        sub = [grids[n] for n in idx]
        eng = StepEngine(MicrogridBatch.from_grids(sub, device=device))
        K, A = sub[0]["final_step"] - sub[0]["initial_step"], action_dim(sub[0])
        acts = np.stack([np.random.RandomState(int(z[f"s{n}_seed"])).rand(K, A) for n in idx], axis=1)
        rew, soc, status = _year(eng, acts, device)
        for j, n in enumerate(idx):
            assert np.array_equal(rew[:, j], z[f"s{n}_reward"]), f"scenario {n}: reward"
            assert np.array_equal(soc[:, j], z[f"s{n}_soc"]), f"scenario {n}: soc"
            if status is not None:
                assert np.array_equal(unpack_status(status[:, j]), z[f"s{n}_status"].astype(np.int32)), f"scenario {n}: genset status"
            seen += 1
        eng.close()
    assert seen == 25


def test_state_checkpoint_resumes_the_year(scenario_root, device, tmp_path):
    """Step 3 000 rows, save the SoA state (.npz), build a FRESH batch from the scenario files, load the checkpoint, continue to the
    end of the year: every later reward / SoC / status `==` the reference's uninterrupted year."""
    from pymgrid_amd import MicrogridBatch, StepEngine, unpack_status
    from pymgrid_amd.scenario import bucket_by_layout, from_scenario, load_state, save_state
    z = golden("pymgrid25_run.npz")
    grids = [from_scenario(n, scenario_root) for n in range(25)]
    CUT = 3000
    for b, idx in enumerate(bucket_by_layout(grids).values()):
        sub = [grids[n] for n in idx]
        K, A = sub[0]["final_step"] - sub[0]["initial_step"], action_dim(sub[0])
        acts = np.stack([np.random.RandomState(int(z[f"s{n}_seed"])).rand(K, A) for n in idx], axis=1)
        first = MicrogridBatch.from_grids(sub, device=device)
        eng = StepEngine(first)
        _year(eng, acts, device, 0, CUT)
        torch.cuda.synchronize()
        ck = str(tmp_path / f"bucket{b}.npz")
        save_state(first, eng.current_step, ck)
        eng.close()
        fresh = MicrogridBatch.from_grids(sub, device=device)
        eng2 = StepEngine(fresh)
        t = load_state(fresh, ck)
        assert t == CUT
        eng2.reset(t, want_obs=False)
        rew, soc, status = _year(eng2, acts, device, CUT)
        for j, n in enumerate(idx):
            assert np.array_equal(rew[:, j], z[f"s{n}_reward"][CUT:]), f"scenario {n}: reward after the checkpoint"
            assert np.array_equal(soc[:, j], z[f"s{n}_soc"][CUT:]), f"scenario {n}: soc after the checkpoint"
            if status is not None:
                assert np.array_equal(unpack_status(status[:, j]), z[f"s{n}_status"].astype(np.int32)[CUT:]), f"scenario {n}: status"
        eng2.close()


def test_mid_year_microgrid_dump_resumes_on_device(scenario_root, device, tmp_path):
    """The reference's OWN checkpoint format: a microgrid dumped mid-year (`state` blocks + `_current_step` in the YAML,
    base_module.py:826-850) and loaded again goes on exactly where it stopped -- scenario 3 (genset + battery + grid) through the
    N = 1 adaptor: 400 steps, dump, load, 400 more `==` the reference's rewards."""
    from pymgrid_amd.envs import MicrogridEnv
    from pymgrid_amd.scenario import dump_scenario_yaml, from_scenario
    z = golden("pymgrid25_run.npz")
    n = next(k for k in range(25) if from_scenario(k, scenario_root).get("grid") is not None
             and from_scenario(k, scenario_root).get("genset") is not None)
    p = from_scenario(n, scenario_root)
    env = MicrogridEnv(p, device=str(device), log=False)
    A = env.layout.action_dim
    acts = np.random.RandomState(int(z[f"s{n}_seed"])).rand(p["final_step"] - p["initial_step"], A)
    env.reset()
    rew = [env.step(_t(acts[k][None], device))[1] for k in range(400)]
    assert np.array_equal(np.array(rew), z[f"s{n}_reward"][:400])
    mid = MicrogridEnv.from_microgrid(env)._params                      # parameters WITH the dynamic state
    mid = dict(mid, current_step=env.current_step)
    path = dump_scenario_yaml(mid, str(tmp_path / "mid.yaml"))
    env.close()
    env2 = MicrogridEnv.load(path, device=str(device), log=False)
    assert env2.current_step == 400
    rew2 = [env2.step(_t(acts[k][None], device))[1] for k in range(400, 800)]
    assert np.array_equal(np.array(rew2), z[f"s{n}_reward"][400:800])
    env2.close()
