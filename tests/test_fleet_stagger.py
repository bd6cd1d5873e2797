"""Staggered ring phases (hetero.BucketedFleet(stagger=True), envs._ring_phase): bucket j walks its observation rings with an
offset so that the buckets' ring refills start at different fleet steps.  Pure scheduling: every observation row, reward and done
flag equals the unstaggered fleet's, across several ring changes and a reset in the middle of a ring."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obs_dtype,K", [(torch.float64, 8), (torch.float32, 5)])
def test_staggered_fleet_equals_the_plain_one(obs_dtype, K, device):
    from pymgrid_amd.generator import generate
    from pymgrid_amd.hetero import BucketedFleet
    archs = ("genset+battery", "battery+grid", "genset+battery+grid")

    def fleet(stagger):
        batches = [generate(700 + 13 * k, n_steps=120, seed=43 + k, arch=a, horizon=24, device=device, series="factorised")
                   for k, a in enumerate(archs)]
        return BucketedFleet.from_batches(batches, obs_dtype=obs_dtype, obs_prefetch=K, reuse_outputs=3 * K, stagger=stagger)
    plain, stag = fleet(False), fleet(True)
    assert [e._ring_phase for e in plain.envs] == [0, 0, 0] and [e._ring_phase for e in stag.envs] == [0, K // 3, 2 * K // 3]
    g = torch.Generator(device=device); g.manual_seed(3)
    for episode in range(2):
        o1, o2 = plain.reset(), stag.reset()
        assert all(torch.equal(a, b) for a, b in zip(o1, o2))
        for k in range(4 * K + 3):
            acts = [torch.rand(e.n_grids, e.layout.action_dim, dtype=torch.float64, device=device, generator=g) for e in plain.envs]
            (o1, r1, d1, _), (o2, r2, d2, _) = plain.step(acts), stag.step(acts)
            for j in range(3):
                assert torch.equal(o1[j], o2[j]), (episode, k, j)
                assert torch.equal(r1[j], r2[j]) and torch.equal(d1[j], d2[j])
    plain.close(); stag.close()
