"""The resident step server (mgx_server_start / _post / _wait / _stop): ONE kernel stays on the device for a burst of env-steps,
parameters and state in registers, controls and outputs through a ring of buffer slots, steps released through a mailbox word
(a stream memory operation, or a host store) and acknowledged through a signal word.  Every reward, observation row, done flag
and the final state == the per-step launches (mgx_step) and == the CPU oracle; a burst that runs dry ends by itself at one step
for all grids and says so."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gen(n, T, arch, device, series, seed=21, **kw):
    from pymgrid_amd.generator import generate
    return generate(n, n_steps=T, seed=seed, arch=arch, device=device, mixed_timers=True, series=series, **kw)


@pytest.mark.parametrize("arch,series,immediate,obs_dtype,act_dtype", [
    ("genset+battery", "materialised", False, torch.float64, torch.float64),
    ("genset+battery", "factorised", True, torch.float32, torch.float32),
    ("genset+battery+grid", "factorised", False, torch.float64, torch.float64),
    ("battery+grid", "materialised", True, torch.float64, torch.float32)])
def test_server_steps_equal_launched_steps_and_the_oracle(arch, series, immediate, obs_dtype, act_dtype, device, oracle):
    from pymgrid_amd import StepEngine
    N, T, K, R = 5003, 90, 41, 4
    plain = StepEngine(_gen(N, T, arch, device, series), obs_dtype=obs_dtype, action_dtype=act_dtype)
    b = _gen(N, T, arch, device, series)
    cols = b.numpy_columns()
    st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status") if k in cols}
    srv = StepEngine(b, obs_dtype=obs_dtype, action_dtype=act_dtype)
    g = torch.Generator(device=device); g.manual_seed(4)
    acts = (torch.rand(K, N, b.layout.action_dim, dtype=torch.float64, device=device, generator=g) * 1.2 - 0.1).to(act_dtype)
    t0 = 7
    plain.reset(t0, want_obs=False); srv.reset(t0, want_obs=False)
    ref = [plain.step(acts[k], want_obs=True, want_log=False) for k in range(K)]          # (obs, reward, done, log)
    slots = srv.server_start(n_slots=R, max_steps=K, want_obs=True, want_done=True, immediate=immediate)
    got = []
    if immediate:                                           # controls already on the device: bursts of R steps, released by host stores
        for k0 in range(0, K, R):
            n = min(R, K - k0)
            for j in range(n):
                slots[(k0 + j) % R]["actions"].copy_(acts[k0 + j])
            torch.cuda.current_stream(device).synchronize()            # (the copies are done: the host may now say so)
            for j in range(n):
                srv.server_post()
            srv.server_wait()
            for j in range(n):
                sl = slots[(k0 + j) % R]
                got.append((sl["obs"].clone(), sl["reward"].clone(), sl["done"].clone()))
    else:                                                   # stream-ordered: write, post, wait, read -- all on torch's current stream
        for k in range(K):
            sl = slots[k % R]
            sl["actions"].copy_(acts[k])
            srv.server_post()
            srv.server_wait()
            got.append((sl["obs"].clone(), sl["reward"].clone(), sl["done"].clone()))
    torch.cuda.current_stream(device).synchronize()
    assert srv.server_stop() == K and srv.current_step == t0 + K == plain.current_step
    for k in range(K):
        assert torch.equal(got[k][0], ref[k][0]), (k, "obs")
        assert torch.equal(got[k][1], ref[k][1]), (k, "reward")
        assert torch.equal(got[k][2].view(torch.bool), ref[k][2].view(torch.bool)), (k, "done")
    for name in ("charge", "soc", "gen_status"):
        if name in b.cols:
            assert torch.equal(b.cols[name], plain.batch.cols[name]), name
    want = oracle.run_batch(cols, st, t0, K, acts.double().cpu().numpy(), normalized=True, nthreads=8)
    assert np.array_equal(torch.stack([r for _, r, _ in got]).cpu().numpy(), want)
    # the engine is an ordinary engine again
    o1 = srv.step(acts[0], want_obs=True)
    o2 = plain.step(acts[0], want_obs=True)
    assert torch.equal(o1[0], o2[0]) and torch.equal(o1[1], o2[1])
    plain.close(); srv.close()


def test_server_refuses_other_calls_and_ends_by_itself(device):
    from pymgrid_amd import StepEngine
    from pymgrid_amd._lib import MGX_ERR_INVALID, MGX_ERR_RANGE, MgxError
    N, T = 2000, 60
    b = _gen(N, T, "genset+battery", device, "factorised")
    twin = StepEngine(_gen(N, T, "genset+battery", device, "factorised"))
    e = StepEngine(b)
    slots = e.server_start(n_slots=2, max_steps=30, idle_timeout_ms=40, immediate=True)
    a = torch.rand(N, 3, dtype=torch.float64, device=device)
    for sl in slots:
        sl["actions"].copy_(a)
    torch.cuda.current_stream(device).synchronize()
    with pytest.raises(MgxError) as ei:                       # the server owns the state
        e.step(a)
    assert ei.value.code == MGX_ERR_INVALID
    e.server_post(); e.server_post()
    e.server_wait()
    torch.cuda.current_stream(device).synchronize()
    r_served = [slots[0]["reward"].clone(), slots[1]["reward"].clone()]
    time.sleep(0.3)                                           # ... and runs dry: the kernel leaves after 40 ms of silence
    e.server_post()                                           # nobody is listening any more
    with pytest.raises(MgxError) as ei:
        e.server_stop()
    assert ei.value.code == MGX_ERR_RANGE and "2 of 3" in str(ei.value)
    assert e.current_step == 2                                # every grid took exactly the two steps that were served
    for k in range(2):
        assert torch.equal(twin.step(a)[1], r_served[k])
    assert torch.equal(b.cols["charge"], twin.batch.cols["charge"]) and torch.equal(b.cols["gen_status"], twin.batch.cols["gen_status"])
    assert torch.equal(e.step(a)[1], twin.step(a)[1])         # and stepping goes on from there
    e.close(); twin.close()
