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


class Tuple:
    """``gym.spaces.Tuple``: one space per module of a name (envs/base/base.py:141-160)."""

    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, j):
        return self.spaces[j]

    def __iter__(self):
        return iter(self.spaces)

    def contains(self, x):
        return isinstance(x, (tuple, list)) and len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))

    __contains__ = contains

    def sample(self, rng=None):
        return tuple(s.sample(rng) for s in self.spaces)

    def __repr__(self):
        return "Tuple(" + ", ".join(repr(s) for s in self.spaces) + ")"


class Dict:
    """``gym.spaces.Dict`` over module names -> ``Tuple`` of per-module ``Box``es: the nested observation space a
    ``flat_spaces=False`` env exposes (envs/base/base.py:128-163).  Keys keep the order given (the reference builds it in
    ``modules.iterdict()`` order; gym itself sorts the keys of a plain dict -- ``sorted_keys()`` gives that order)."""

    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def sorted_keys(self):
        return sorted(self.spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def contains(self, x):
        return isinstance(x, dict) and set(x) == set(self.spaces) and all(self.spaces[k].contains(v) for k, v in x.items())

    __contains__ = contains

    def sample(self, rng=None):
        return {k: s.sample(rng) for k, s in self.spaces.items()}

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k}: {s!r}" for k, s in self.spaces.items()) + ")"
