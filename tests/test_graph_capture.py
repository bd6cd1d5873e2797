"""HIP-graph capture of env steps (device-resident step counter): a graph of 8 single steps, replayed, must equal the
same steps issued eagerly -- rewards, observations, state and the step counter."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("series", ["materialised", "factorised"])
def test_captured_fused_launches_replay_like_eager(series, device):
    """A graph of ONE mgx_step_k launch (K = 48 steps, device-resident counter) replayed five times == the same launches issued
    eagerly; the factorised form stages its base-profile rows from the counter's row."""
    from pymgrid_amd import StepEngine
    from pymgrid_amd.generator import generate
    N, T, K, R = 4000, 300, 48, 5
    gen = torch.Generator(device=device); gen.manual_seed(3)
    acts = torch.rand(R, K, N, 4, dtype=torch.float64, device=device, generator=gen)
    kw = dict(n_steps=T, seed=4, arch="genset+battery+grid", device=device, mixed_timers=True, series=series)
    eager, eng = StepEngine(generate(N, **kw)), StepEngine(generate(N, **kw))
    ref = [eager.step_k(acts[r], reward=True, soc_trace=True) for r in range(R)]
    ref = [{k: v.clone() for k, v in o.items()} for o in ref]
    eng.use_device_counter(True)
    static_a = torch.zeros(K, N, 4, dtype=torch.float64, device=device)
    out = dict(reward=torch.empty(K, N, dtype=torch.float64, device=device), soc_trace=torch.empty(K, N, dtype=torch.float64, device=device))
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            eng.step_k(static_a, reward=True, soc_trace=True, out=out)
    torch.cuda.current_stream(device).wait_stream(side)
    for r in range(R):
        static_a.copy_(acts[r])
        graph.replay()
        torch.cuda.synchronize(device)
        assert torch.equal(out["reward"], ref[r]["reward"]) and torch.equal(out["soc_trace"], ref[r]["soc_trace"]), r
    eng.use_device_counter(False)
    assert eng.current_step == R * K == eager.current_step
    for name in ("charge", "soc", "gen_status"):
        assert torch.equal(eng.batch.cols[name], eager.batch.cols[name])
    eng.close(); eager.close()


@pytest.mark.parametrize("series", ["materialised", "factorised"])
def test_captured_steps_replay_like_eager(series, device):
    from pymgrid_amd import StepEngine
    from pymgrid_amd.generator import generate
    N, T, S, R = 5000, 200, 8, 5                 # graph of S steps, replayed R times
    gen = torch.Generator(device=device); gen.manual_seed(1)
    acts = torch.rand(R * S, N, 3, dtype=torch.float64, device=device, generator=gen)
    eager = StepEngine(generate(N, n_steps=T, seed=4, device=device, mixed_timers=True, series=series))
    ref_r, ref_o = [], []
    for k in range(R * S):
        o, r, d, _ = eager.step(acts[k])
        ref_r.append(r.clone()); ref_o.append(o.clone())

    eng = StepEngine(generate(N, n_steps=T, seed=4, device=device, mixed_timers=True, series=series))
    eng.use_device_counter(True)
    static_a = torch.zeros(S, N, 3, dtype=torch.float64, device=device)
    bufs = [dict(reward=torch.empty(N, dtype=torch.float64, device=device),
                 done=torch.empty(N, dtype=torch.uint8, device=device),
                 obs=torch.empty(N, eng.obs_dim, dtype=torch.float64, device=device)) for _ in range(S)]
    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream(device))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for k in range(S):
                eng.step(static_a[k], want_obs=True, out=bufs[k])
    torch.cuda.current_stream(device).wait_stream(side)
    # capturing does not execute: the counter is still at 0
    assert eng.current_step == 0
    for rep in range(R):
        static_a.copy_(acts[rep * S:(rep + 1) * S])
        graph.replay()
        torch.cuda.synchronize(device)
        for k in range(S):
            assert torch.equal(bufs[k]["reward"], ref_r[rep * S + k]), (rep, k)
            assert torch.equal(bufs[k]["obs"], ref_o[rep * S + k]), (rep, k)
    assert eng.current_step == R * S == eager.current_step
    for name in ("charge", "soc", "gen_status"):
        assert torch.equal(eng.batch.cols[name], eager.batch.cols[name])
    eng.use_device_counter(False)                 # counter copied back to the host, eager stepping continues
    assert eng.current_step == R * S
    o1, r1, _, _ = eng.step(acts[0]); o2, r2, _, _ = eager.step(acts[0])
    assert torch.equal(r1, r2) and torch.equal(o1, o2)
    eng.close(); eager.close()


def test_replay_past_the_series_is_flagged(device):
    from pymgrid_amd import MgxError, StepEngine
    from pymgrid_amd.generator import generate
    N, T = 1000, 12
    eng = StepEngine(generate(N, n_steps=T, seed=4, device=device))
    eng.use_device_counter(True)
    a = torch.rand(N, 3, dtype=torch.float64, device=device)
    for _ in range(T + 3):                        # no launch-time range check in this mode: clamped + flagged in-kernel
        eng.step(a, want_obs=False)
    with pytest.raises(MgxError) as e:
        eng.use_device_counter(False)
    assert e.value.code == 3
    eng.close()


@pytest.mark.parametrize("discrete", [False, True])
def test_graphed_rollout_equals_the_eager_loop(discrete, device):
    """GraphedRollout: n_steps iterations of policy -> env.step captured as ONE graph (a small fp64 network as policy),
    replayed three times in a row, then reset and replayed again == the same loop issued eagerly on a twin env."""
    from pymgrid_amd import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv, GraphedRollout
    from pymgrid_amd.generator import generate
    N, T, S = 3000, 120, 8
    cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv

    def make():
        # (row-major rings for the eager twin: `obs @ W1` on a column-major view takes another GEMM path, whose sums differ in the
        # last bit from those on the graph's contiguous rows -- the env's values are the same either way, tests/test_ring_layout.py)
        return cls(generate(N, n_steps=T, seed=6, arch="genset+battery+grid", horizon=5, device=device, mixed_timers=True), obs_layout="rows")
    env, twin = make(), make()
    g = torch.Generator(device=device); g.manual_seed(2)
    D = env.layout.obs_dim
    W1 = torch.randn(D, 16, dtype=torch.float64, device=device, generator=g) * 0.3
    if discrete:
        n = env.action_space.n
        W2 = torch.randn(16, n, dtype=torch.float64, device=device, generator=g)

        def policy(obs):
            return torch.argmax(torch.tanh(obs @ W1) @ W2, dim=1).to(torch.int32)
    else:
        W2 = torch.randn(16, env.layout.action_dim, dtype=torch.float64, device=device, generator=g)

        def policy(obs):
            return torch.sigmoid(torch.tanh(obs @ W1) @ W2)
    roll = GraphedRollout(env, policy, S)
    obs = twin.reset()
    for rep in range(3):
        r, d, o = roll.run()
        for k in range(S):
            obs, rr, dd, _ = twin.step(policy(obs))
            assert torch.equal(r[k], rr) and torch.equal(d[k], dd), (rep, k)
        assert torch.equal(o, obs), rep
    for name in ("charge", "soc", "gen_status"):
        assert torch.equal(env.batch.cols[name], twin.batch.cols[name]), name
    o0 = roll.reset()                                  # counter back to the start, state untouched (like env.reset)
    obs = twin.reset()
    assert torch.equal(o0, obs)
    r, d, o = roll.run()
    for k in range(S):
        obs, rr, dd, _ = twin.step(policy(obs))
        assert torch.equal(r[k], rr), k
    roll.close()
    assert env.current_step == twin.current_step == S
    env.close(); twin.close()
