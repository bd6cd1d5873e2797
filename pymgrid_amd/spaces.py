"""Minimal Gym-style spaces (gym itself is not a dependency): just what the reference's env surface exposes --
``Box`` for continuous action / observation spaces, ``Discrete`` for the priority-list action space
(envs/base/base.py:128-163, envs/discrete/discrete.py:76-80)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float64):
        shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
        self.shape, self.dtype = shape, np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape[-len(self.shape):] == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains

    def sample(self, rng=None):
        rng = rng or np.random
        return rng.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.shape}, {self.dtype})"


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape, self.dtype = (), np.dtype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return bool(np.all((x >= 0) & (x < self.n)) and np.issubdtype(x.dtype, np.integer))

    __contains__ = contains

    def sample(self, rng=None):
        rng = rng or np.random
        return int(rng.randint(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"
