"""Episode-window samplers and reward shapers -- host-side mirrors of ``pymgrid.microgrid.trajectory`` and
``pymgrid.microgrid.reward_shaping``.

A trajectory function is called at every ``reset()`` with the environment's ``(initial_step, final_step)`` and returns
the window of the next episode (microgrid.py:221-225).  The three classes draw from numpy's global RNG with exactly the
calls the reference makes, so a seeded run picks the same windows.  In a batch all grids share the window (they share
the step counter).
"""
import numpy as np


class DeterministicTrajectory:
    """trajectory/deterministic.py:4-12"""

    def __init__(self, initial_step, final_step):
        self.initial_step, self.final_step = initial_step, final_step

    def __call__(self, initial_step, final_step):
        return self.initial_step, self.final_step


class StochasticTrajectory:
    """trajectory/stochastic.py:6-12"""

    def __call__(self, initial_step, final_step):
        initial = np.random.randint(initial_step, final_step - 2)
        final = np.random.randint(initial, final_step)
        return initial, final


class FixedLengthStochasticTrajectory:
    """trajectory/stochastic.py:15-30"""

    def __init__(self, trajectory_length):
        self.trajectory_length = trajectory_length

    def __call__(self, initial_step, final_step):
        if final_step - initial_step < self.trajectory_length:
            raise ValueError(f'Cannot create a trajectory of length {self.trajectory_length}'
                             f'between initial_step ({initial_step}) and final_step ({final_step})')
        initial = np.random.randint(initial_step, final_step - self.trajectory_length)
        return initial, initial + self.trajectory_length


def check_trajectory_output(output):
    """Microgrid._check_trajectory_func (microgrid.py:181-203): two Python ints."""
    try:
        initial_step, final_step = output
        if not (isinstance(initial_step, (int, np.integer)) and isinstance(final_step, (int, np.integer))):
            raise ValueError
    except (TypeError, ValueError):
        raise TypeError(f'trajectory func must return two integer values, not {output}')
    return int(initial_step), int(final_step)


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
