"""Policy-in-the-loop rollouts as ONE HIP graph.

A Gym loop ``obs -> policy -> action -> env.step -> obs`` at N <= 10^5 grids is bound by the host's launch rate (a step
kernel takes ~5 us, issuing it from Python ~10 us, a small policy network several launches more).  With the engine's
device-resident step counter (``mgx_use_device_counter``) the whole loop body is stream-capturable, so ``n_steps``
iterations -- policy kernels included -- are recorded once and replayed with a single launch each time.

The reference has no counterpart (its loop is Python per instance, envs/base/base.py:169-209); values are those of the
eager loop, bit for bit (tests/test_graph_capture.py).
"""
import torch

from .envs import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv


class GraphedRollout:
    """``n_steps`` iterations of ``action = policy(obs); obs, reward, done, _ = env.step(action)`` captured in a HIP graph.

    policy: callable(obs [N, D]) -> action tensor for ``env.step`` ([N, A] floats, or int32 ids [N] for a discrete env);
            it must be capturable (device tensors in, device tensors out, static shapes, no host synchronisation).
    ``run()`` replays the graph from the engine's CURRENT step and state and returns (reward [n_steps, N],
    done [n_steps, N] bool, obs [N, D]): views of static buffers that the next ``run()`` overwrites.  ``obs`` is also the
    policy's input of the next replay, so consecutive ``run()`` calls continue the same episode; ``reset()`` restarts it.
    """

    def __init__(self, env, policy, n_steps, warmup=2):
        if not isinstance(env, BatchedMicrogridEnv):
            raise TypeError("env must be a (Discrete)BatchedMicrogridEnv")
        if env.raise_errors or env._keep_log:
            raise ValueError("GraphedRollout needs an env with raise_errors=False, log=False (their bookkeeping lives on the host)")
        env.set_obs_prefetch(0)        # the ring bookkeeping lives on the host too: captured steps write whole rows
        if not env._observations:
            raise ValueError("the policy consumes observations: build the env with observations=True")
        self.env, self.policy, self.n_steps = env, policy, int(n_steps)
        eng = env.engine
        dev = eng.device
        N = eng.N
        self.reward = torch.empty(self.n_steps, N, dtype=torch.float64, device=dev)
        self.done = torch.empty(self.n_steps, N, dtype=torch.bool, device=dev)
        self._start = eng.current_step
        self.obs = env.reset(self._start).clone()          # static buffer: policy input of step 0, final obs of a replay
        # warm-up on a side stream (lazy initialisation of the policy's kernels / workspaces must not be captured), then
        # rewind: reset() restores the step counter only, so the dynamic state is put back by hand
        state = eng.batch.state()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            o = self.obs
            for _ in range(max(0, int(warmup))):
                o, _, _, _ = env.step(policy(o))
        torch.cuda.current_stream(dev).wait_stream(side)
        eng.batch.load_state(state)
        eng.reset(self._start, want_obs=False)
        eng.use_device_counter(True)
        self.graph = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            with torch.cuda.graph(self.graph, stream=side):
                o = self.obs
                for k in range(self.n_steps):
                    o, r, d, _ = env.step(policy(o))
                    self.reward[k].copy_(r)
                    self.done[k].copy_(d)
                self.obs.copy_(o)
        torch.cuda.current_stream(dev).wait_stream(side)

    def run(self):
        self.graph.replay()
        return self.reward, self.done, self.obs

    def reset(self, initial_step=None):
        """Back to ``initial_step`` (default: where the rollout was built); like ``env.reset`` the dynamic state stays."""
        eng = self.env.engine
        eng.use_device_counter(False)
        self.obs.copy_(self.env.reset(self._start if initial_step is None else initial_step))
        eng.use_device_counter(True)
        return self.obs

    def close(self):
        self.env.engine.use_device_counter(False)


__all__ = ["GraphedRollout", "BatchedMicrogridEnv", "DiscreteBatchedMicrogridEnv"]
