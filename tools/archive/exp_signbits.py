#!/usr/bin/env python3
"""Where do the signs of ZERO results differ from the reference's?  (v_min_f64 / v_max_f64 order -0 < +0, Python's min / max
do not: mgx_core.hpp py_min / py_max.)  Steps the 25 benchmark scenarios for 128 logged steps and counts, per log column, the
values that are == the golden but differ in np.signbit."""
import os
import sys
from collections import Counter

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from pymgrid_amd import MicrogridBatch, StepEngine
from pymgrid_amd.scenario import bucket_by_layout, load_npz_grids

z = np.load(os.path.join(ROOT, "tests", "golden", "pymgrid25_run.npz"))
grids = load_npz_grids(os.path.join(ROOT, "tests", "golden", "pymgrid25_inputs.npz"))
names = [str(s) for s in z["log_names"]]
dev = torch.device("cuda:0")
diff, total = Counter(), Counter()
for idx in bucket_by_layout(grids).values():
    sub = [grids[n] for n in idx]
    eng = StepEngine(MicrogridBatch.from_grids(sub, device=dev))
    A = eng.action_dim
    K = sub[0]["final_step"] - sub[0]["initial_step"]
    acts = np.stack([np.random.RandomState(int(z[f"s{n}_seed"])).rand(K, A)[:128] for n in idx], axis=1)
    for k in range(128):
        _, _, _, log = eng.step(torch.as_tensor(acts[k], dtype=torch.float64, device=dev), want_obs=False, want_log=True)
        log = log.cpu().numpy()
        for j, n in enumerate(idx):
            row = z[f"s{n}_log_sub"][k]
            d = dict(zip(eng.log_names, log[:, j]))
            for c, name in enumerate(names):
                if name in d and not np.isnan(row[c]) and d[name] == row[c] == 0.0:
                    total[name] += 1
                    if np.signbit(d[name]) != np.signbit(row[c]):
                        diff[name] += 1
    eng.close()
for name in names:
    if total[name]:
        print(f"{name:28s} zeros {total[name]:6d}  sign differs {diff[name]:6d}")
