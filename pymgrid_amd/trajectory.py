"""Episode-window samplers and reward shapers -- host-side mirrors of ``pymgrid.microgrid.trajectory`` and
``pymgrid.microgrid.reward_shaping``.

A trajectory function is called at every ``reset()`` with the environment's ``(initial_step, final_step)`` and returns
the window of the next episode (microgrid.py:221-225).  The three classes draw from numpy's global RNG with exactly the
calls the reference makes, so a seeded run picks the same windows.  In a batch all grids share the window (they share
the step counter).
"""
import numpy as np


def _randint(low, high):
    """One draw from numpy's GLOBAL stream, as the reference's samplers make it: a seeded run picks the same windows."""
    return int(np.random.randint(low, high))


class DeterministicTrajectory:
    """The same window at every reset (trajectory/deterministic.py:4-12)."""

    def __init__(self, initial_step, final_step):
        self.window = (initial_step, final_step)

    @property
    def initial_step(self):
        return self.window[0]

    @property
    def final_step(self):
        return self.window[1]

    def __call__(self, initial_step, final_step):
        return self.window


class StochasticTrajectory:
    """A random start and a random end behind it: two draws per reset, in the reference's order and with its bounds
    (trajectory/stochastic.py:6-12: start in [lo, hi - 2), end in [start, hi))."""

    def __call__(self, initial_step, final_step):
        start = _randint(initial_step, final_step - 2)
        return start, _randint(start, final_step)


class FixedLengthStochasticTrajectory:
    """A window of ``trajectory_length`` steps at a random start (trajectory/stochastic.py:15-30: one draw per reset, start in
    [lo, hi - length)); an env window shorter than the length is a ValueError, as in the reference."""

    def __init__(self, trajectory_length):
        self.trajectory_length = int(trajectory_length)

    def __call__(self, initial_step, final_step):
        room = final_step - initial_step
        if room < self.trajectory_length:
            raise ValueError(f"the env's window [{initial_step}, {final_step}) holds {room} steps: too short for episodes of "
                             f"{self.trajectory_length}")
        start = _randint(initial_step, final_step - self.trajectory_length)
        return start, start + self.trajectory_length


def check_trajectory_output(output):
    """What Microgrid._check_trajectory_func (microgrid.py:181-203) demands of a trajectory function: a pair of integers."""
    ok = isinstance(output, (tuple, list)) and len(output) == 2 and all(isinstance(v, (int, np.integer)) for v in output)
    if not ok:
        raise TypeError(f"a trajectory function returns (initial_step, final_step) as two integers; got {output!r}")
    return int(output[0]), int(output[1])


class PVCurtailmentShaper:
    """reward = -curtailment (reward_shaping/pv_curtailment_shaper.py)"""
    kind = 1


class BatteryDischargeShaper:
    """reward = (battery discharge - loss load) / load, 0 when there is no load
    (reward_shaping/battery_discharge_shaper.py)"""
    kind = 2


def shaper_kind(func):
    """None / shaper instance / name -> enum mgx_reward_shaper."""
    if func is None:
        return 0
    if isinstance(func, str):
        return {"pv_curtailment": 1, "battery_discharge": 2}[func]
    kind = getattr(func, "kind", None)
    if kind in (1, 2):
        return kind
    raise NotImplementedError("only PVCurtailmentShaper and BatteryDischargeShaper run on device; arbitrary Python "
                              "reward_shaping_func callables are not supported")
