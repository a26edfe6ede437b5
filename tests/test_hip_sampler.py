"""GPU checks of the device-side surface sampler (stage A).

The reference consumes torch's CPU generator with data-dependent trip counts, so its stream cannot be reproduced
on the device: parity here is (i) the invariants the reference's algorithm guarantees, checked
exactly, and (ii) agreement of the sample distribution with the oracle's (same algorithm, CPU generator).
"""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def small(dev):
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    return field_handle_from_ckpt(util.ckpt("small"), dev)


def rho_of(ck):
    """sampling.py:518-521."""
    g = torch.tensor(ck["kwargs"]["gridSize"], dtype=torch.long)
    size = ck["kwargs"]["aabb"][1] - ck["kwargs"]["aabb"][0]
    return float((torch.max(g) * 0.1) * torch.max(size / g))


def test_invariants_and_determinism(small, dev):
    ck = util.ckpt("small")
    rho = rho_of(ck)
    P = 1500
    s0, a0, _ = small.surface_sample(P, rho, n_epochs=0, seed=123)
    # seeds: inside occupied voxels, alpha consistent with compute_alpha
    assert (small.mask_sample(s0) > 0).all()
    assert torch.equal(small.point_alpha(s0), a0)
    s1, a1, st1 = small.surface_sample(P, rho, n_epochs=1, seed=123)
    st1 = st1.cpu()
    thresh0 = st1[0, 2:3].view(torch.float32).item()
    want = torch.quantile(a0.cpu(), q=0.6).item()
    assert abs(thresh0 - want) <= 1e-6 * max(1.0, abs(want)), (thresh0, want)
    assert int(st1[0, 1]) == 0 and 1 <= int(st1[0, 0]) <= 200
    assert (a1 > thresh0).all(), "every sample accepted in an epoch beats that epoch's threshold"
    assert torch.equal(small.point_alpha(s1), a1)
    # full run: thresholds rise, all samples valid, alpha bookkeeping exact, bitwise reproducible
    s4, a4, st4 = small.surface_sample(P, rho, n_epochs=4, max_iterations=200, seed=123)
    s4b, a4b, st4b = small.surface_sample(P, rho, n_epochs=4, max_iterations=200, seed=123)
    assert torch.equal(s4, s4b) and torch.equal(a4, a4b) and torch.equal(st4, st4b)
    st4 = st4.cpu()
    th = st4[:, 2].contiguous().view(torch.float32)
    assert (st4[:, 3] != -1).all() and (st4[:, 1] == 0).all()
    assert torch.all(th[1:] >= th[:-1])
    assert (a4 > th[-1].item()).all()
    assert torch.equal(small.point_alpha(s4), a4)
    s5, _, _ = small.surface_sample(P, rho, n_epochs=4, seed=124)
    assert not torch.equal(s4, s5)


def test_distribution_matches_oracle(small, dev):
    from oracle import emit as oemit, field as ofield
    ck = util.ckpt("small")
    f = ofield.field_from_ckpt(ck)
    P = 3000
    torch.manual_seed(7)
    so, ao, stats = oemit.surface_samples(f, P, 4, 200, return_stats=True)
    sh, ah, st = small.surface_sample(P, rho_of(ck), n_epochs=4, max_iterations=200, seed=99)
    sh, ah = sh.cpu(), ah.cpu()
    c = (ck["kwargs"]["aabb"][0] + ck["kwargs"]["aabb"][1]) / 2
    ro, rh = torch.linalg.norm(so - c, dim=-1), torch.linalg.norm(sh - c, dim=-1)
    # two independent draws of 3000 samples from the same process: compare quantiles of radius and of alpha
    for qv in (0.1, 0.5, 0.9):
        a, b = torch.quantile(ro, qv).item(), torch.quantile(rh, qv).item()
        assert abs(a - b) < 0.05 * max(a, b) + 0.01, ("radius quantile", qv, a, b)
    sig = lambda al: -torch.log1p(-al.double().clamp(max=1 - 1e-12))     # compare in sigma space: alpha saturates near 1
    for qv in (0.1, 0.5, 0.9):
        a, b = torch.quantile(sig(ao), qv).item(), torch.quantile(sig(ah), qv).item()
        assert abs(a - b) < 0.12 * max(a, b) + 0.05, ("sigma quantile", qv, a, b)
    # per-axis means: the blob is centred, so both should sit near the centre
    assert torch.allclose(so.mean(0), sh.mean(0), atol=0.03)
    # iteration counts are of the same order
    it_o = sum(s[1] for s in stats)
    it_h = int(st.cpu()[:, 0].sum())
    assert 0.4 * it_o <= it_h <= 2.5 * it_o + 4, (it_o, it_h)


def test_small_and_degenerate(small, dev):
    ck = util.ckpt("small")
    s, a, st = small.surface_sample(1, rho_of(ck), n_epochs=2, seed=1)
    assert s.shape == (1, 3) and torch.isfinite(s).all()
    # a threshold nobody can beat (rho = 0: candidates == sample, alpha == thresh is not > thresh) ends at max_iterations
    s, a, st = small.surface_sample(64, 0.0, n_epochs=1, max_iterations=5, seed=3)
    st = st.cpu()
    assert int(st[0, 0]) == 5 and int(st[0, 1]) > 0
    with pytest.raises(RuntimeError):
        small.surface_sample(0, 0.1)


def test_one_lane_candidates_draw_the_same_samples(dev):
    """The sampler with one lane per candidate (a quarter of the workgroups) against the four-lane form
    (iff_field_desc.density_lanes = 1 / 4): same random streams, same alpha bits (density_full), so samples / alpha /
    statistics are equal."""
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    ck = util.ckpt("small")
    rho = rho_of(ck)
    one, four = field_handle_from_ckpt(ck, dev, density_lanes=1), field_handle_from_ckpt(ck, dev, density_lanes=4)
    for P in (75, 593, 5):
        for a, b in zip(one.surface_sample(P, rho, 4, 200, seed=11), four.surface_sample(P, rho, 4, 200, seed=11)):
            assert torch.equal(a, b), P
        for a, b in zip(one.surface_sample_batched(3, P, rho, 4, 200, seed=11), four.surface_sample_batched(3, P, rho, 4, 200, seed=11)):
            assert torch.equal(a, b), P


def test_batched_runs_large_point_counts_and_residency(small, dev):
    """Batched launches: every run equals the single call with its strided seed, at point counts below and above the
    LDS-cache limit (4096) and in both lane forms (batches of 8 or more switch to one lane per candidate); the
    residency query bounds how many launches may be in flight."""
    from iffnerf_amd.hip_field import SAMPLER_SEED_STRIDE
    rho = rho_of(util.ckpt("small"))
    for B, P in ((3, 700), (8, 300), (2, 5000)):
        s, a, st = small.surface_sample_batched(B, P, rho, n_epochs=3, max_iterations=200, seed=77)
        assert s.shape == (B, P, 3) and a.shape == (B, P) and st.shape == (B, 3, 4)
        assert (st[..., 3] != -1).all(), "in-kernel barrier timed out"
        for b in range(B):
            s1, a1, st1 = small.surface_sample(P, rho, n_epochs=3, max_iterations=200, seed=(77 + b * SAMPLER_SEED_STRIDE) % 2 ** 64)
            assert torch.equal(s[b], s1) and torch.equal(a[b], a1) and torch.equal(st[b], st1), (B, P, b)
        assert torch.equal(small.point_alpha(s.reshape(-1, 3)), a.reshape(-1))
        assert not torch.equal(s[0], s[1])
    w1, cap = small.sampler_residency(593, 1)
    w16, cap16 = small.sampler_residency(593, 16)
    assert cap == cap16 and cap >= 256 and w1 == 47 and w16 == 12          # 4 lanes / 1 lane per candidate at P = 593
    w_big, _ = small.sampler_residency(20000, 1)
    assert w_big <= 256                                                     # never more than one workgroup per CU and run
