"""Round 6: the discrete Gym step of a layout with several modules of a kind (2 gensets + 2 batteries + 1 grid, N = 100 000) --
mgx_step_lists (one launch, the list walk in registers) against the two calls it replaces (mgx_expand_lists -> control -> mgx_step)."""
import time

import torch

from pymgrid_amd import StepEngine
from pymgrid_amd.generator import generate, widen
from pymgrid_amd.priority_list import get_instance_priority_lists, lists_array

dev = torch.device("cuda:0")
N, T = 100_000, 1200
b = widen(generate(N, n_steps=T, seed=5, arch="genset+battery+grid", horizon=0, device=dev), n_genset=2, n_battery=2, n_grid=1)
e = StepEngine(b)
pls = get_instance_priority_lists(2, 2, 1, (), False)
lists = torch.as_tensor(lists_array(pls), device=dev).contiguous()
g = torch.Generator(device=dev); g.manual_seed(1)
ids = torch.randint(0, len(pls), (N,), dtype=torch.int32, device=dev, generator=g)
out = dict(reward=torch.empty(N, dtype=torch.float64, device=dev))
ctrl = torch.empty(N, e.action_dim, dtype=torch.float64, device=dev)
for name, fn in (("expand + step (two launches)", lambda: e.step(e.expand_lists(ids, lists, out=ctrl), normalized=False, want_obs=False, out=out, want_done=False)),
                 ("mgx_step_lists (one launch)", lambda: e.step_lists(ids, lists, want_obs=False, out=out, want_done=False))):
    for rep in range(3):
        e.reset(0, want_obs=False)
        for _ in range(50):
            fn()
        e.reset(0, want_obs=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(1000):
            fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name}: {dt / 1000 * 1e6:.2f} us per env-step of {N} grids")
e.close()
