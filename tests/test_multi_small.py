"""The register form of the general step (step_multi_small: at most two modules of every kind per grid -- parameters, state, controls
and series rows requested up front, the sweep on registers, a K-step loop that keeps parameters and state there) against the run-time
count form it specialises (step_multi_core, mgx_set_tunable(MGX_TUNE_MULTI_GENERIC, 1)): the same batches, controls and calls in two processes, every
reward, log column, observation row and final state `==`.  (Both forms are pinned against the reference-made fixtures of
tests/golden/multi.npz and the oracle by tests/test_multiplicity.py / test_multi_module.py, which run whichever form the layout
gets; this test makes sure those pins hold for BOTH.)  Reference: module_container.py:355-413, microgrid.py:255-314."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from pymgrid_amd import BatchedMicrogridEnv, StepEngine, _lib
from pymgrid_amd.generator import generate, widen
_lib.set_tunable("multi_generic", int(sys.argv[3]))
assert _lib.get_tunable("multi_generic") == (int(sys.argv[3]), 0)
dev = torch.device("cuda:0")
out = {}
g = torch.Generator(device=dev); g.manual_seed(11)
# "d": two of EVERYTHING -- up to 9 addends in the provided list (2 gensets, 2 discharging batteries, 2 importing grids, 2 renewables,
# loss load): the sums with 8 and 9 addends take numpy's pairwise order, which the register form rebuilds from its static slots
for tag, (ng, nb, nr, nl, npv), arch in (("a", (2, 2, 1, 1, 1), "genset+battery+grid"), ("b", (1, 2, 2, 1, 1), "genset+battery+grid"),
                                        ("c", (2, 1, 0, 1, 1), "genset+battery"), ("d", (2, 2, 2, 2, 2), "genset+battery+grid")):
    N, T, K = 1500, 90, 24
    def batch():
        return widen(generate(N, n_steps=T, seed=21, arch=arch, horizon=3, device=dev, mixed_timers=True), n_genset=ng, n_battery=nb, n_grid=nr,
                     n_load=nl, n_pv=npv)
    env = BatchedMicrogridEnv(batch(), log=True, obs_prefetch=0)
    A = env.layout.action_dim
    acts = torch.rand(K + 6, N, A, dtype=torch.float64, device=dev, generator=g)
    obs0 = env.reset()
    rows, rew, logs = [obs0], [], []
    for k in range(6):                                   # single steps with observation rows and log columns
        o, r, d, info = env.step(acts[k])
        rows.append(o); rew.append(r.clone()); logs.append(info["log"].clone())
    e = env.engine
    res = e.step_k(acts[6:], normalized=True, reward=True, soc_trace=True, status_trace=True, log=True)   # K fused steps
    out[tag + "_rows"] = torch.stack(rows).cpu().numpy(); out[tag + "_rew"] = torch.stack(rew).cpu().numpy()
    out[tag + "_log"] = torch.stack(logs).cpu().numpy()
    for name, v in res.items():
        out[tag + "_k_" + name] = v.cpu().numpy()
    for name in ("charge", "soc", "gen_status"):
        if name in env.batch.cols and env.batch.cols[name] is not None:
            out[tag + "_" + name] = env.batch.cols[name].cpu().numpy()
    # float32 controls, unnormalised
    e32 = StepEngine(batch(), action_dtype=torch.float32)
    a32 = (acts[:4] * 50).float()
    rr = torch.empty(4, N, dtype=torch.float64, device=dev)
    for k in range(4):
        e32.step(a32[k], normalized=False, want_obs=False, out=dict(reward=rr[k]))
    out[tag + "_f32_rew"] = rr.cpu().numpy()
    out[tag + "_f32_k"] = e32.step_k(a32, normalized=False, reward=True)["reward"].cpu().numpy()
    env.close(); e32.close()
np.savez(sys.argv[2], **out)
'''


def _run(tmp_path, generic):
    path = str(tmp_path / ("generic.npz" if generic else "small.npz"))
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, path, "1" if generic else "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return np.load(path)


def test_register_form_equals_the_run_time_count_form(device, tmp_path):
    small, generic = _run(tmp_path, False), _run(tmp_path, True)
    assert set(small.files) == set(generic.files) and len(small.files) > 20
    for k in small.files:
        a, b = small[k], generic[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)) or np.array_equal(a, b), k      # == (zeros may differ in sign)
    assert np.isfinite(small["a_k_reward"]).all() and small["a_k_reward"].std() > 0


def test_compile_time_counts_equal_run_time_counts(device):
    """mgx_step_k on small general layouts: the specialisations with the instance counts fixed at compile time
    (step_k_multi_small_kernel<F, CountsCT<...>>, mgx_fused.hip part 5) against the run-time-count form of the same kernel
    (mgx_set_tunable(MGX_TUNE_MULTI_STATIC, 0)): rewards, SoC / status traces, every log column and the final state `==`."""
    import torch
    from pymgrid_amd import StepEngine, _lib
    from pymgrid_amd.generator import generate, widen
    g = torch.Generator(device=device); g.manual_seed(19)
    for (ng, nb, nr, nl, npv), arch in (((2, 2, 1, 1, 1), "genset+battery+grid"), ((2, 2, 2, 2, 2), "genset+battery+grid"),
                                        ((1, 2, 2, 1, 1), "genset+battery+grid"), ((2, 1, 0, 1, 1), "genset+battery"),
                                        ((0, 2, 1, 1, 1), "battery+grid"), ((2, 2, 1, 2, 1), "genset+battery+grid"),   # (no specialisation)
                                        # THREE of a kind (M = 3 slots; up to 9 addends in the mid-sweep sums, 10 / 11 in the final ones:
                                        # numpy's pairwise order rebuilt from the slots) against the run-time-count kernel with its LDS lists
                                        ((3, 3, 1, 1, 1), "genset+battery+grid"), ((3, 2, 1, 1, 1), "genset+battery+grid"),
                                        ((2, 3, 1, 1, 1), "genset+battery+grid"), ((3, 3, 0, 1, 1), "genset+battery"),
                                        ((0, 3, 1, 1, 1), "battery+grid")):
        N, T, K = 2100, 80, 33

        def batch():
            return widen(generate(N, n_steps=T, seed=31, arch=arch, horizon=0, device=device, mixed_timers=True), n_genset=ng, n_battery=nb,
                         n_grid=nr, n_load=nl, n_pv=npv)
        res = []
        acts = None
        for static in (1, 0):
            _lib.set_tunable("multi_static", static)
            try:
                e = StepEngine(batch())
                if acts is None:
                    acts = torch.rand(K, N, e.layout.action_dim, dtype=torch.float64, device=device, generator=g)
                e.reset(3, want_obs=False)
                out = e.step_k(acts, normalized=True, reward=True, done=True, soc_trace=True, status_trace=True, log=True)
                out2 = e.step_k(acts[:7].float().double() * 40, normalized=False, reward=True)        # a second launch from the carried state
                torch.cuda.synchronize()
                res.append(({k: v.clone() for k, v in out.items()}, out2["reward"].clone(),
                            {k: e.batch.cols[k].clone() for k in ("charge", "soc", "gen_status") if k in e.batch.cols}))
                e.close()
            finally:
                _lib.set_tunable("multi_static", 1)
        (o1, r1, s1), (o0, r0, s0) = res
        assert set(o1) == set(o0)
        for k in o1:
            assert torch.equal(o1[k], o0[k]), ((ng, nb, nr, nl, npv), k)
        assert torch.equal(r1, r0) and all(torch.equal(s1[k], s0[k]) for k in s1)
        assert float(o1["reward"].std()) > 0


@pytest.mark.gpu
def test_list_rollouts_on_the_register_form_equal_the_run_time_form(device):
    """mgx_rollout_lists on the compile-time-count specialisations (rollout_multi_small_kernel: the priority-list walk out of registers,
    the list packed into a 64-bit word) against the run-time-count kernel with its walk over the batch's columns
    (mgx_set_tunable(MGX_TUNE_MULTI_STATIC, 0)): one fixed list per grid (RuleBasedControl) and an id per step (a discrete roll-out),
    rewards, done, traces, every log column (the expansion's assert mask in it) and the final state `==` -- two and three of a kind."""
    import torch
    from pymgrid_amd import StepEngine, _lib
    from pymgrid_amd.generator import generate, widen
    g = torch.Generator(device=device); g.manual_seed(23)
    for (ng, nb, nr, nl, npv), arch in (((2, 2, 1, 1, 1), "genset+battery+grid"), ((2, 2, 2, 2, 2), "genset+battery+grid"),
                                        ((1, 2, 1, 1, 1), "genset+battery+grid"), ((2, 1, 0, 1, 1), "genset+battery"),
                                        ((0, 2, 1, 1, 1), "battery+grid"), ((2, 2, 1, 2, 1), "genset+battery+grid"),      # (no compile-time specialisation)
                                        ((3, 3, 1, 1, 1), "genset+battery+grid"),
                                        ((2, 3, 1, 1, 1), "genset+battery+grid"), ((0, 3, 1, 1, 1), "battery+grid")):
        N, T, K = 1500, 70, 21
        # 24 lists drawn by hand (the reference's enumeration is factorial in the elements): every module somewhere, genset goals 0 / 1,
        # modules named twice (the second is skipped, priority_list.py:82-88), padding and elements the layout does not have in between
        rs = np.random.RandomState(100 * ng + 10 * nb + nr)
        mods = [(0, j) for j in range(ng)] + [(1, j) for j in range(nb)] + [(2, j) for j in range(nr)]
        rows = []
        for _ in range(24):
            order = [mods[q] for q in rs.permutation(len(mods))]
            els = [(k, j, int(rs.randint(0, 2)) if k == 0 else 0) for k, j in order]
            els.insert(int(rs.randint(0, len(els) + 1)), (-1, -1, -1))
            els.insert(int(rs.randint(0, len(els) + 1)), (int(rs.randint(0, 3)), 5, 0))
            els.insert(int(rs.randint(1, len(els) + 1)), els[0])
            rows.append(els[: len(mods) + 3] if rs.rand() < 0.8 else els[: max(1, len(mods) - 1)])
        width = max(len(r) for r in rows)
        tab = -np.ones((len(rows), width, 3), dtype=np.int32)
        for q, r in enumerate(rows):
            tab[q, :len(r)] = r
        lists = torch.as_tensor(tab, device=device).contiguous()

        def batch():
            return widen(generate(N, n_steps=T, seed=47, arch=arch, horizon=0, device=device, mixed_timers=True), n_genset=ng, n_battery=nb,
                         n_grid=nr, n_load=nl, n_pv=npv)
        ids_fixed = torch.randint(0, lists.shape[0], (N,), dtype=torch.int32, device=device, generator=g)
        ids_step = torch.randint(-1, lists.shape[0] + 1, (K, N), dtype=torch.int32, device=device, generator=g)   # (ids outside: list 0)
        res = []
        for static, generic in ((1, 0), (0, 0), (0, 1)):     # compile-time counts / run-time counts in registers / the walk over the columns
            _lib.set_tunable("multi_static", static)
            _lib.set_tunable("multi_generic", generic)
            try:
                e = StepEngine(batch())
                e.reset(2, want_obs=False)
                o1 = e.rollout_lists(ids_fixed, lists, K, reward=True, done=True, soc_trace=True, status_trace=True, log=True)
                o2 = e.rollout_lists(ids_step, lists, K, reward=True, log=True)          # a second launch from the carried state
                torch.cuda.synchronize()
                res.append(({k: v.clone() for k, v in o1.items()}, {k: v.clone() for k, v in o2.items()},
                            {k: e.batch.cols[k].clone() for k in ("charge", "soc", "gen_status") if k in e.batch.cols}))
                e.close()
            finally:
                _lib.set_tunable("multi_static", 1)
                _lib.set_tunable("multi_generic", 0)
        (a1, a2, s1) = res[0]
        for (b1, b2, s0) in res[1:]:
            for x, y in ((a1, b1), (a2, b2), (s1, s0)):
                assert set(x) == set(y)
                for k in x:
                    assert torch.equal(x[k], y[k]), ((ng, nb, nr, nl, npv), k)
        assert float(a1["reward"].std()) > 0


@pytest.mark.gpu
def test_discrete_step_over_instance_lists_in_one_call(device):
    """mgx_step_lists (ABI minor 2): DiscreteMicrogridEnv.step over priority lists of module instances -- one launch for layouts of at
    most two of a kind (the walk in registers), expand + step through the control buffer for three of a kind -- against the two calls
    it replaces (mgx_expand_lists -> control -> mgx_step(normalized=0)): control, reward, done, observation rows (H = 0 and H = 6, float64
    and float32), every log column and the state, step after step; the discrete env takes the call."""
    import torch
    from pymgrid_amd import DiscreteBatchedMicrogridEnv, StepEngine
    from pymgrid_amd.generator import generate, widen
    g = torch.Generator(device=device); g.manual_seed(31)
    for (ng, nb, nr, nl, npv), arch, H in (((2, 2, 1, 1, 1), "genset+battery+grid", 0), ((2, 2, 2, 2, 1), "genset+battery+grid", 6),
                                           ((1, 2, 0, 1, 1), "genset+battery", 6), ((0, 2, 2, 1, 2), "battery+grid", 0),
                                           ((3, 2, 1, 1, 1), "genset+battery+grid", 0)):
        N, T, K = 900, 50, 14
        rs = np.random.RandomState(7 + 10 * ng + nb)
        mods = [(0, j) for j in range(ng)] + [(1, j) for j in range(nb)] + [(2, j) for j in range(nr)]
        tab = -np.ones((16, len(mods) + 2, 3), dtype=np.int32)
        for q in range(16):
            els = [(k, j, int(rs.randint(0, 2)) if k == 0 else 0) for k, j in (mods[p] for p in rs.permutation(len(mods)))]
            els.insert(int(rs.randint(0, len(els) + 1)), (-1, -1, -1))
            els.insert(int(rs.randint(1, len(els) + 1)), els[0])
            tab[q, :len(els)] = els
        lists = torch.as_tensor(tab, device=device).contiguous()

        def batch():
            return widen(generate(N, n_steps=T, seed=5, arch=arch, horizon=H, device=device, mixed_timers=True), n_genset=ng, n_battery=nb,
                         n_grid=nr, n_load=nl, n_pv=npv)
        for f32 in (False, True):
            one, two = StepEngine(batch()), StepEngine(batch())
            if f32:
                one.set_obs_dtype(torch.float32); two.set_obs_dtype(torch.float32)
            one.reset(1, want_obs=False); two.reset(1, want_obs=False)
            for k in range(K):
                ids = torch.randint(-1, 17, (N,), dtype=torch.int32, device=device, generator=g)
                ctrl1 = torch.empty(N, one.action_dim, dtype=torch.float64, device=device)
                o1, r1, d1, l1 = one.step_lists(ids, lists, want_obs=True, want_log=True, out=dict(control=ctrl1))
                ctrl2 = two.expand_lists(ids, lists)
                o2, r2, d2, l2 = two.step(ctrl2, normalized=False, want_obs=True, want_log=True)
                assert torch.equal(ctrl1, ctrl2), ((ng, nb, nr), k, "control")
                assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(l1, l2), ((ng, nb, nr), k)
                assert torch.equal(o1, o2), ((ng, nb, nr), k, "obs")
            for name in ("charge", "soc", "gen_status"):
                if name in one.batch.cols:
                    assert torch.equal(one.batch.cols[name], two.batch.cols[name])
            assert one.current_step == two.current_step == 1 + K
            one.close(); two.close()
    # the env: ids -> (obs, reward, done, info) through the one call == expansion + continuous step
    a, b = DiscreteBatchedMicrogridEnv(batch(), log=True), DiscreteBatchedMicrogridEnv(batch(), log=True)
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    for k in range(10):
        ids = a.sample_action(generator=g)
        oa, ra, da, ia = a.step(ids)
        ob, rb, db, ib = super(DiscreteBatchedMicrogridEnv, b).step(b.get_action(ids), normalized=False)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(ia["log"], ib["log"]), k
    a.close(); b.close()
