#!/usr/bin/env python3
"""Build-container only: time the REAL reference (pymgrid 1.2.2 imported from /root/reference) on Template-4 grids
drawn with the benchmark's generator (same sizing rules), next to this repo's C oracle on the same grids.
Usage: python tests/golden/time_reference.py [n_grids] [steps]   -> prints env-steps/s on one core."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
warnings.simplefilter("ignore")
import _refenv  # noqa: E402
_refenv.import_reference()
from pymgrid import Microgrid  # noqa: E402
from pymgrid.modules import BatteryModule, GensetModule, LoadModule, RenewableModule  # noqa: E402
from pymgrid_amd.generator import generate  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_grids = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
b = generate(n_grids, n_steps=steps + 1, seed=42, arch="genset+battery", device="cpu")
c = {k: v.numpy() for k, v in b.cols.items()}
acts = np.random.RandomState(7).rand(steps, n_grids, 3)

t_ref = 0.0
rewards = np.zeros((steps, n_grids))
for i in range(n_grids):
    m = Microgrid([("load", LoadModule(time_series=-c["load_ts"][:, i])),
                   ("pv", RenewableModule(time_series=c["pv_ts"][:, i])),
                   ("genset", GensetModule(running_min_production=c["gen_running_min"][i],
                                           running_max_production=c["gen_running_max"][i], genset_cost=0.4,
                                           co2_per_unit=2.0, cost_per_unit_co2=0.1)),
                   ("battery", BatteryModule(min_capacity=c["bat_min_capacity"][i], max_capacity=c["bat_max_capacity"][i],
                                             max_charge=c["bat_max_charge"][i], max_discharge=c["bat_max_discharge"][i],
                                             efficiency=0.9, battery_cost_cycle=0.02, init_soc=float(c["soc"][i])))],
                  loss_load_cost=10.0, overgeneration_cost=1.0)
    t0 = time.perf_counter()
    for k in range(steps):
        _, r, _, _ = m.run({"genset": [acts[k, i, :2]], "battery": [acts[k, i, 2]]})
        rewards[k, i] = r
    t_ref += time.perf_counter() - t0
cols = b.numpy_columns()
st = {k: cols[k].copy() for k in ("charge", "soc", "gen_status")}
t0 = time.perf_counter()
r_orc = orc.run_batch(cols, st, 0, steps, acts)
t_orc = time.perf_counter() - t0
print(f"reference (pymgrid 1.2.2, Python, 1 core): {n_grids * steps / t_ref:10.0f} env-steps/s  "
      f"({n_grids} Template-4 grids x {steps} steps, Microgrid.run only)")
print(f"C oracle (oracle/mgx_oracle.c, 1 thread):  {n_grids * steps / t_orc:10.0f} env-steps/s   "
      f"rewards identical: {np.array_equal(rewards, r_orc)}")
