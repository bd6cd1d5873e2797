"""pymgrid_amd -- MI355X-native batched microgrid-step engine.

One data-parallel hot path of Total-RD/pymgrid, rebuilt for gfx950: the per-instance ``Microgrid.run()`` /
``module.step()`` loop becomes hand-written HIP kernels over a struct-of-arrays batch of N microgrids, behind a
C ABI (``include/mgx.h``) and a ``pymgrid.envs``-style Gym surface (``pymgrid_amd.envs``).
"""
from .batch import BatchLayout, MicrogridBatch, pack_grids, pack_status, unpack_status  # noqa: F401
from ._lib import MgxError, build, lib  # noqa: F401
from .engine import StepEngine  # noqa: F401
from .envs import (BatchedMicrogridEnv, DiscreteBatchedMicrogridEnv, DiscreteMicrogridEnv,  # noqa: F401
                   MicrogridEnv)
from .graph import GraphedRollout  # noqa: F401
from .priority_list import get_priority_lists  # noqa: F401
from .rbc import RuleBasedControl  # noqa: F401
from .trajectory import (BatteryDischargeShaper, DeterministicTrajectory,  # noqa: F401
                         FixedLengthStochasticTrajectory, PVCurtailmentShaper, StochasticTrajectory)

Microgrid = MicrogridEnv      # ``pymgrid.Microgrid``'s stepping surface: run / step / reset / sample_action / get_log / from_scenario

__version__ = "0.1.0"
