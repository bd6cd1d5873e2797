"""Round 6: how fast is RuleBasedControl on a layout with several modules of a kind (mgx_rollout_lists: the run-time-count kernel
with its priority-list walk in private memory)?  2 gensets + 2 batteries + 1 grid, N = 100 000, K = 32 per launch."""
import time

import torch

from pymgrid_amd import BatchedMicrogridEnv, RuleBasedControl
from pymgrid_amd.generator import generate, widen

dev = torch.device("cuda:0")
N, T, K = 100_000, 8760, 32
for mix in ((2, 2, 1), (1, 1, 1)):
    ng, nb, nr = mix
    b = widen(generate(N, n_steps=T, seed=5, arch="genset+battery+grid", horizon=0, device=dev), n_genset=ng, n_battery=nb, n_grid=nr,
              n_load=1, n_pv=1) if mix != (1, 1, 1) else generate(N, n_steps=T, seed=5, arch="genset+battery+grid", horizon=0, device=dev)
    env = BatchedMicrogridEnv(b, observations=False)
    rbc = RuleBasedControl(env)
    env.reset()
    for rep in range(3):
        env.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rbc._run(K * 20, K, False, False, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(mix, "multi" if env.layout.multi else "single", f"{dt / (K * 20) * 1e6:.2f} us per env-step of {N} grids")
    env.close()
