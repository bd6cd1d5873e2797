#!/bin/bash
# Run ON THE GPU BOX: config-5 fleet with observation ROWS, the ring refills on a CU-masked prefetch stream
# (MGX_PREFETCH_CU_PERCENT = share of every XCD's CUs the refills may use): wall us per fleet step, float64 / float32 rows.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for pct in 0 75 50 25; do
  for dt in float64 float32; do
    MGX_PREFETCH_CU_PERCENT=$pct python - $dt $pct <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import BucketedFleet
dt = torch.float32 if sys.argv[1] == "float32" else torch.float64
dev = torch.device("cuda:0")
batches = [generate(33333, n_steps=8760, seed=43 + k, arch=a, horizon=24, device=dev, series="factorised")
           for k, a in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
fleet = BucketedFleet.from_batches(batches, obs_dtype=dt, obs_prefetch=16, reuse_outputs=48)
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = [torch.rand(33333, e.layout.action_dim, dtype=torch.float64, device=dev, generator=gen) for e in fleet.envs]
fleet.reset()
for _ in range(2000):
    fleet.step(acts)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(1600):
        fleet.step(acts)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 1600 * 1e6)
print(f"refill CUs {sys.argv[2]:>3s} % (0 = unmasked)  {sys.argv[1]} rows: {best:6.2f} us per fleet step", flush=True)
fleet.close()
PY
  done
done
