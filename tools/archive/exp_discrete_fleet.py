import sys, time, torch
sys.path.insert(0, "/root/repo")
from pymgrid_amd.generator import generate
from pymgrid_amd.hetero import BucketedFleet
dev = torch.device("cuda:0")
per = 33333
batches = [generate(per, n_steps=5000, seed=43 + k, arch=arch, horizon=24, device=dev) for k, arch in enumerate(("genset+battery", "battery+grid", "genset+battery+grid"))]
fleet = BucketedFleet.from_batches(batches, discrete=True, obs_prefetch=8, reuse_outputs=24, remove_redundant_gensets=False)
g = torch.Generator(device=dev); g.manual_seed(1)
acts = fleet.sample_action(generator=g)
fleet.reset()
for rep in range(3):
    fleet.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000): fleet.step(acts)
    torch.cuda.synchronize()
    print(f"discrete fleet fused={fleet.fused}: {1e3 * (time.perf_counter() - t0):.1f} us/step")
