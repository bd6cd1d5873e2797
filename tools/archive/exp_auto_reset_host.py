import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import PerGridWindowEnv
dev = torch.device("cuda:0")
for N in (1000, 100_000):
    w = PerGridWindowEnv(generate(N, n_steps=8760, seed=1, arch="genset+battery+grid", horizon=24, device=dev), trajectory_length=168, auto_reset=True)
    a = w.env.sample_action(); w.reset()
    for _ in range(100): w.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): w.step(a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"N={N}: host issue {1e6*(t1-t0)/400:.1f} us/step, wall {1e6*(t2-t0)/400:.1f} us/step")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(400): w.step(a)
    pr.disable(); torch.cuda.synchronize()
    if N == 1000:
        pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    w.close()
