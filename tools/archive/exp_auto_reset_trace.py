import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import PerGridWindowEnv
dev = torch.device("cuda:0")
w = PerGridWindowEnv(generate(100_000, n_steps=8760, seed=1, arch="genset+battery+grid", horizon=24, device=dev), trajectory_length=168, auto_reset=True)
a = w.env.sample_action(); w.reset()
for _ in range(120):
    w.step(a)
torch.cuda.synchronize()
