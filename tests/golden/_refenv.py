"""Build-container-only helper: make the pure-Python reference importable.

TEST INFRASTRUCTURE.  Used only by ``tests/golden/make_goldens.py`` and by the
``refcheck`` tests (which skip themselves when ``/root/reference`` is absent,
i.e. always on the GPU box).  Nothing in ``pymgrid_amd/`` imports this file.

The reference (pymgrid 1.2.2) needs ``gym``, ``IPython``, ``cvxpy`` and
``statsmodels`` at import time; none of them are installed
and there is no network, so minimal in-memory stand-ins are registered in
``sys.modules`` before ``import pymgrid``.  Only container classes are shimmed
-- every number the goldens hold is computed by the reference's own code.
"""
import os
import sys
import types

import numpy as np

REFERENCE_SRC = "/root/reference/src"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_SRC, "pymgrid"))


# --------------------------------------------------------------------------- #
# gym shim (containers only)
# --------------------------------------------------------------------------- #
class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    @shape.setter
    def shape(self, value):
        self._shape = value

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)

    def sample(self):
        raise NotImplementedError


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.shape(low) if not np.isscalar(low) else np.shape(high)
        shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
        super().__init__(shape, dtype, seed)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.allclose(self.low, other.low) and np.allclose(self.high, other.high))

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class Discrete(Space):
    def __init__(self, n, seed=None):
        self.n = int(n)
        super().__init__((), np.int64, seed)

    def contains(self, x):
        try:
            return int(x) == x and 0 <= int(x) < self.n
        except (TypeError, ValueError):
            return False

    def sample(self):
        return int(self._rng.integers(self.n))

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        if isinstance(spaces, Dict):
            spaces = spaces.spaces
        self.spaces = dict(spaces or {}, **kw)   # insertion order kept
        super().__init__(None, None, seed)

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def values(self):
        return self.spaces.values()

    def items(self):
        return self.spaces.items()

    def contains(self, x):
        return (isinstance(x, dict) and x.keys() == self.spaces.keys()
                and all(x[k] in s for k, s in self.spaces.items()))

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def __eq__(self, other):
        return isinstance(other, Dict) and self.spaces == other.spaces


class Tuple(Space):
    def __init__(self, spaces, seed=None):
        self.spaces = tuple(spaces)
        super().__init__(None, None, seed)

    def __getitem__(self, i):
        return self.spaces[i]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(xi in s for xi, s in zip(x, self.spaces))

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def __eq__(self, other):
        return isinstance(other, Tuple) and self.spaces == other.spaces


def flatten_space(space):
    if isinstance(space, Box):
        return Box(space.low.reshape(-1), space.high.reshape(-1), dtype=space.dtype)
    if isinstance(space, (Dict, Tuple)):
        subs = [flatten_space(s) for s in (space.spaces.values() if isinstance(space, Dict) else space.spaces)]
        if not subs:
            return Box(np.zeros(0), np.zeros(0), dtype=np.float64)
        return Box(np.concatenate([s.low for s in subs]), np.concatenate([s.high for s in subs]),
                   dtype=np.result_type(*[s.dtype for s in subs]))
    raise NotImplementedError(type(space))


def flatten(space, x):
    if isinstance(space, Box):
        return np.asarray(x, dtype=space.dtype).reshape(-1)
    if isinstance(space, Dict):
        parts = [flatten(s, x[k]) for k, s in space.spaces.items()]
    elif isinstance(space, Tuple):
        parts = [flatten(s, xi) for s, xi in zip(space.spaces, x)]
    else:
        raise NotImplementedError(type(space))
    return np.concatenate(parts) if parts else np.zeros(0)


class Env:
    @property
    def unwrapped(self):
        return self


def _install_stubs():
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        gym.__version__ = "0.0-shim"
        spaces = types.ModuleType("gym.spaces")
        for obj in (Space, Box, Discrete, Dict, Tuple, flatten, flatten_space):
            setattr(spaces, obj.__name__, obj)
        utils = types.ModuleType("gym.utils")
        seeding = types.ModuleType("gym.utils.seeding")
        seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
        utils.seeding = seeding
        gym.Env, gym.spaces, gym.utils = Env, spaces, utils
        sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.utils": utils,
                            "gym.utils.seeding": seeding})
    if "IPython" not in sys.modules:
        ipy = types.ModuleType("IPython")
        disp = types.ModuleType("IPython.display")
        disp.display = lambda *a, **k: None
        ipy.display, ipy.get_ipython = disp, (lambda: None)
        sys.modules.update({"IPython": ipy, "IPython.display": disp})
    if "cvxpy" not in sys.modules:
        sys.modules["cvxpy"] = types.ModuleType("cvxpy")
    if "statsmodels" not in sys.modules:
        sm = types.ModuleType("statsmodels")
        reg = types.ModuleType("statsmodels.regression")
        qr = types.ModuleType("statsmodels.regression.quantile_regression")
        qr.QuantReg = object
        sm.regression, reg.quantile_regression = reg, qr
        sys.modules.update({"statsmodels": sm, "statsmodels.regression": reg,
                            "statsmodels.regression.quantile_regression": qr})
    if not hasattr(np, "product"):
        np.product = np.prod   # numpy>=2 dropped it; error path only (base_module.py:145)


def import_reference():
    """Return the imported reference package (``pymgrid``) or raise if it is absent."""
    if not reference_available():
        raise RuntimeError("/root/reference is not present (expected on the GPU box)")
    _install_stubs()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import pymgrid  # noqa: F401  (seeds np.random with 123 at import: SURVEY.md Q10)
    return pymgrid
