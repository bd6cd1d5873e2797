"""Heterogeneous fleets (BASELINE config 5): microgrids with different module sets / horizons cannot share one SoA
batch (a batch has one layout = one kernel specialisation), so a fleet is bucketed by layout
(``scenario.bucket_by_layout``) and every bucket gets its own ``MicrogridBatch`` + engine.  Buckets are independent:
each is stepped on its own HIP stream so small buckets overlap instead of queueing behind each other.
"""
import numpy as np
import torch

from .batch import MicrogridBatch
from .envs import BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv
from .scenario import bucket_by_layout


class BucketedFleet:
    """N microgrids of mixed layouts behind one step()/reset() surface.

    ``step`` takes / returns lists with one entry per bucket (tensors of that bucket's shape); ``scatter`` puts a
    per-grid quantity (reward, done, ...) back into fleet order.
    """

    def __init__(self, grids, device="cuda", discrete=False, **env_kwargs):
        self.n_grids = len(grids)
        self.device = torch.device(device)
        self.buckets = list(bucket_by_layout(grids).items())          # [(key, [indices])]
        cls = DiscreteBatchedMicrogridEnv if discrete else BatchedMicrogridEnv
        self.envs, self.index = [], []
        for _, idx in self.buckets:
            self.envs.append(cls(MicrogridBatch.from_grids([grids[i] for i in idx], device=device), **env_kwargs))
            self.index.append(torch.as_tensor(np.asarray(idx), device=self.device))
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.envs] if self.device.type == "cuda" else []

    def __len__(self):
        return self.n_grids

    def _each(self, fn):
        """Run fn(env, k) for every bucket, each on its own stream; the caller's stream waits for all of them."""
        cur = torch.cuda.current_stream(self.device)
        out = []
        for k, (env, st) in enumerate(zip(self.envs, self.streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out.append(fn(env, k))
        for st in self.streams:
            cur.wait_stream(st)
        return out

    def reset(self):
        return self._each(lambda env, k: env.reset())

    def step(self, actions, **kw):
        """actions: list with one tensor per bucket.  Returns (obs_list, reward_list, done_list, info_list)."""
        res = self._each(lambda env, k: env.step(actions[k], **kw))
        return tuple(list(x) for x in zip(*res))

    def sample_action(self, generator=None):
        return [env.sample_action(generator=generator) for env in self.envs]

    def scatter(self, per_bucket):
        """[tensor [n_b, ...] per bucket] -> one tensor [N, ...] in the order the grids were given."""
        first = per_bucket[0]
        out = torch.empty((self.n_grids,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
        for idx, v in zip(self.index, per_bucket):
            out[idx] = v
        return out

    def close(self):
        for env in self.envs:
            env.close()
