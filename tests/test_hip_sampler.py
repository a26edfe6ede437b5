"""GPU checks of the device-side surface sampler (stage A).

The reference consumes torch's CPU generator with data-dependent trip counts, so its stream cannot be reproduced
on the device: parity here is (i) the invariants the reference's algorithm guarantees, checked
exactly, and (ii) agreement of the sample distribution with the oracle's (same algorithm, CPU generator).
"""
import os

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


def _ks(a: torch.Tensor, b: torch.Tensor) -> float:
    """Two-sample Kolmogorov-Smirnov statistic sup |F_a - F_b|."""
    a, b = torch.sort(a.double().flatten()).values, torch.sort(b.double().flatten()).values
    allv = torch.cat((a, b))
    fa = torch.searchsorted(a, allv, right=True).double() / a.numel()
    fb = torch.searchsorted(b, allv, right=True).double() / b.numel()
    return float((fa - fb).abs().max())


@pytest.mark.parametrize("model", ["small", "small_sharp"])
def test_distribution_matches_oracle(model, dev):
    """Epoch by epoch, CONDITIONALLY: whole runs are not i.i.d. samples (all P points of a run share the epoch thresholds, which
    are quantiles of that run's own random alphas -- two oracle runs with different seeds already differ by KS 0.05-0.1), so
    the device sampler and the oracle are compared one epoch at a time from the SAME state.  The device run with
    n_epochs = e gives the state after e epochs (its first e epochs do not depend on how many follow); from it, epoch e + 1 is
    taken once by the device (n_epochs = e + 1) and once by the oracle's sampling_epoch (torch CPU generator, the reference's
    op chain, sampling.py:143-213).  Given the state the P samples move independently, so the two-sample Kolmogorov-Smirnov
    statistic applies: every feature of the moved samples -- displacement, distance from the blob's centre, coordinates,
    density -- must stay below the 1e-4 critical value for 3000 vs 3000 points, 2.23 sqrt(2 / 3000) = 0.058, in all four
    epochs on both models, and the epochs' iteration counts must agree to within one."""
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    from oracle import emit as oemit, field as ofield
    ck = util.ckpt("small") if model == "small" else util.ckpt("small", seed=33, peak=26.0, blob_sigma=0.36)
    f = ofield.field_from_ckpt(ck)
    h = field_handle_from_ckpt(ck, dev)
    P, rho, seed = 3000, rho_of(ck), 77
    c = (ck["kwargs"]["aabb"][0] + ck["kwargs"]["aabb"][1]) / 2
    sig = lambda al: -torch.log1p(-al.double().clamp(max=1 - 1e-12))     # compare in sigma space: alpha saturates near 1
    worst = 0.0
    for e in range(4):
        s_e, a_e, _ = h.surface_sample(P, rho, n_epochs=e, max_iterations=200, seed=seed)
        s_h, a_h, st = h.surface_sample(P, rho, n_epochs=e + 1, max_iterations=200, seed=seed)
        st = st.cpu()
        assert (st[:, 3] != -1).all() and int(st[e, 1]) == 0
        s_e, a_e, s_h, a_h = s_e.cpu(), a_e.cpu(), s_h.cpu(), a_h.cpu()
        torch.manual_seed(100 + e)
        s_o, a_o, it_o, left_o = oemit.sampling_epoch(f, s_e.clone(), a_e.clone(), rho, max_iterations=200)
        assert int(left_o) == 0 and abs(int(st[e, 0]) - it_o) <= 1, (e, int(st[e, 0]), it_o)
        thr = st[e, 2:3].view(torch.float32).item()
        assert abs(thr - torch.quantile(a_e, q=0.6).item()) <= 1e-6 and bool((a_o > thr).all()) and bool((a_h > thr).all())
        feats = {"step": lambda s, a: torch.linalg.norm(s - s_e, dim=-1), "radius": lambda s, a: torch.linalg.norm(s - c, dim=-1),
                 "x": lambda s, a: s[:, 0], "y": lambda s, a: s[:, 1], "z": lambda s, a: s[:, 2], "sigma": lambda s, a: sig(a)}
        for name, fn in feats.items():
            d = _ks(fn(s_o, a_o), fn(s_h, a_h))
            worst = max(worst, d)
            assert d < 0.058, (model, e, name, d)
    assert worst > 0.0


def _philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al.), the counter-based generator of csrc/sampler_kernels.hip, restated with Python ints."""
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    x, y, z, w = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * x, M1 * z
        x, y, z, w = ((p1 >> 32) ^ y ^ k0) & mask, p1 & mask, ((p0 >> 32) ^ w ^ k1) & mask, p0 & mask
        k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
    return x, y, z, w


def test_one_candidate_step_is_the_references_arithmetic(small, dev):
    """One iteration of one epoch from known seeds: every sample that moved must sit on one of ITS m = 5 jittered candidates
    as the reference constructs them (pose_estimation/sampling.py:38-66: direction from theta = 2 pi u, phi = arccos(1 - 2 u),
    radius |N(0, rho)|) from the very uniforms the kernel's counter-based generator yields -- recomputed here in float64
    from an independent restatement of Philox and of the formulas -- and must beat the epoch's threshold."""
    import math
    ck = util.ckpt("small")
    rho, P, seed = rho_of(ck), 200, 4242
    s0, a0, _ = small.surface_sample(P, rho, n_epochs=0, seed=seed)
    s1, a1, st = small.surface_sample(P, rho, n_epochs=1, max_iterations=1, seed=seed)
    st = st.cpu()
    assert int(st[0, 0]) == 1 and int(st[0, 3]) == 5               # one iteration, m = 5P // P candidates per sample
    thresh = st[0, 2:3].view(torch.float32).item()
    s0c, s1c, a1c = s0.cpu().double(), s1.cpu().double(), a1.cpu()
    moved = (s1c != s0c).any(-1)
    assert 0.2 * P < int(moved.sum()) <= P and int(st[0, 1]) == P - int(moved.sum())
    u01 = lambda v: (v >> 8) / 16777216.0
    for i in torch.nonzero(moved).flatten().tolist():
        best = float("inf")
        for j in range(5):
            r = _philox4x32_10((i, j, 0 * 4096 + 0, 0xA5), (seed & 0xFFFFFFFF, seed >> 32))
            theta, phi = 2 * math.pi * u01(r[0]), math.acos(1 - 2 * u01(r[1]))
            g = math.sqrt(-2 * math.log(1 - u01(r[2]))) * math.cos(2 * math.pi * u01(r[3]))        # Box-Muller normal
            d = torch.tensor([math.sin(phi) * math.cos(theta), math.sin(phi) * math.sin(theta), math.cos(phi)], dtype=torch.float64)
            cand = s0c[i] + d * abs(g * rho)
            best = min(best, float((cand - s1c[i]).abs().max()))
        assert best < 5e-6, (i, best)
        assert float(a1c[i]) > thresh
    assert torch.equal(s1[~moved.to(dev)], s0[~moved.to(dev)])


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
    LDS-cache limit (4096); the residency query (what bounds the launches in flight of the persistent form)."""
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
    # the stepped form (default) has no co-residency requirement; the persistent form reports the device's capacity
    w1, cap = small.sampler_residency(593, 1)
    w16, cap16 = small.sampler_residency(593, 16)
    assert cap == cap16 == 2 ** 31 - 1 and w1 == 12 and w16 == 12          # one lane per candidate at P = 593 (quad form) whatever the batch
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    pers = field_handle_from_ckpt(util.ckpt("small"), dev, sampler_persistent=True)
    w1, cap = pers.sampler_residency(593, 1)
    w16, cap16 = pers.sampler_residency(593, 16)
    assert cap == cap16 and 256 <= cap < 2 ** 31 - 1 and w1 == 47 and w16 == 12      # batches of 8 or more: one lane per candidate
    w_big, _ = pers.sampler_residency(20000, 1)
    assert w_big <= 256                                                 # never more than one workgroup per CU and run


def test_stepped_launches_equal_the_persistent_launch(small, dev):
    """The sampler as a chain of short launches (the default: seeds | per epoch 8 iteration launches + finisher + apply, the kernel
    boundary as the grid barrier) draws the SAME samples, alphas and statistics as the one persistent launch with in-kernel grid
    barriers (a handle made with iff_field_desc.sampler_persistent = 1), bit for bit: both lane forms, point counts below and above the LDS-cache limit, an
    iteration bound below the number of static launches, and a jitter so small that an epoch needs more than the 8 static
    iteration launches (the finisher's loop)."""
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    pers = field_handle_from_ckpt(util.ckpt("small"), dev, sampler_persistent=True)
    rho = rho_of(util.ckpt("small"))
    cases = [(1, 593, rho, 4, 200), (16, 593, rho, 4, 200), (3, 5000, rho, 3, 200), (8, 300, rho, 2, 3), (2, 700, rho * 0.02, 2, 200),
             (9, 400, rho * 0.02, 2, 12), (2, 3000, rho * 0.02, 2, 200), (2, 20000, rho, 2, 200)]     # > 2560 points: list + candidate launches
    long_epochs = 0
    for B, P, r, E, mi in cases:
        got = small.surface_sample_batched(B, P, r, n_epochs=E, max_iterations=mi, seed=4321)
        want = pers.surface_sample_batched(B, P, r, n_epochs=E, max_iterations=mi, seed=4321)
        assert (want[2][..., 3] != -1).all(), "in-kernel barrier timed out"
        for x, y, what in zip(got, want, ("samples", "alpha", "stats")):
            assert torch.equal(x, y), (B, P, r, E, mi, what, (got[2] - want[2]).abs().max().item() if what == "stats" else None)
        long_epochs += int((got[2][..., 0] > 8).sum())
    assert long_epochs > 0, "no case went past the static iteration launches: the finisher was not exercised"


def _philox_np(i, j, c2, seed):
    """Vectorised Philox4x32-10 for counters (i, j, c2, 0xA5) and the 64-bit key ``seed`` -> four uint32 arrays."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xFFFFFFFF)
    x, y = i.astype(np.uint64), j.astype(np.uint64)
    z, w = np.full_like(x, c2), np.full_like(x, 0xA5)
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    for _ in range(10):
        p0, p1 = M0 * x, M1 * z
        x, y, z, w = ((p1 >> np.uint64(32)) ^ y ^ np.uint64(k0)) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ w ^ np.uint64(k1)) & mask, p0 & mask
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return x, y, z, w


def test_accept_and_pick_decisions_of_one_iteration(small, dev):
    """Every accept / select decision of one sampler iteration, replayed from the kernel's own random stream: for P samples x
    m = 5 candidates x 8 seeds the candidates are rebuilt (independent Philox + the reference's formulas, float64), their alphas
    taken from iff_point_alpha, and then (sampling.py:160-200)
      * a sample moves iff one of its candidates beats the epoch threshold (outside a 1e-5 band around it),
      * the candidate it moved to is one of the passing ones, and
      * the pick is UNIFORM among the passing candidates: the rank of the chosen one among k passing candidates is tested
        against the uniform distribution on {0..k-1} by chi-square for every k (the reference picks index
        floor(u (k - 1 + 0.99)) in alpha order; the kernel gives every passing candidate a random priority and keeps the maximum)."""
    ck = util.ckpt("small")
    rho, P, m = rho_of(ck), 3000, 5
    counts = {k: np.zeros(k) for k in range(2, m + 1)}
    n_checked = 0
    for seed in range(9000, 9008):
        s0, a0, _ = small.surface_sample(P, rho, n_epochs=0, seed=seed)
        s1, a1, st = small.surface_sample(P, rho, n_epochs=1, max_iterations=1, seed=seed)
        st = st.cpu()
        assert int(st[0, 0]) == 1 and int(st[0, 3]) == m
        thresh = st[0, 2:3].view(torch.float32).item()
        ii, jj = np.meshgrid(np.arange(P), np.arange(m), indexing="ij")
        r = _philox_np(ii.ravel(), jj.ravel(), 0, seed)
        u = [(v >> np.uint64(8)).astype(np.float64) / 16777216.0 for v in r]
        theta, phi = 2 * np.pi * u[0], np.arccos(1 - 2 * u[1])
        g = np.sqrt(-2 * np.log(1 - u[2])) * np.cos(2 * np.pi * u[3])
        d = np.stack([np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)], -1) * np.abs(g * rho)[:, None]
        cand = s0.cpu().double().numpy()[ii.ravel()] + d                                  # [P*m, 3]
        a_c = small.point_alpha(torch.from_numpy(cand).float().to(dev)).cpu().numpy().reshape(P, m)
        clear = np.abs(a_c - thresh) > 1e-5                                                # decisions not inside the rounding band
        passing = a_c > thresh
        s0n, s1n = s0.cpu().double().numpy(), s1.cpu().double().numpy()
        moved = (s1n != s0n).any(-1)
        dist = np.abs(cand.reshape(P, m, 3) - s1n[:, None]).max(-1)                        # which candidate the sample sits on
        chosen = dist.argmin(-1)
        for i in range(P):
            if not clear[i].all():
                continue
            n_checked += 1
            assert moved[i] == passing[i].any(), (seed, i)
            if moved[i]:
                assert dist[i, chosen[i]] < 5e-6 and passing[i, chosen[i]], (seed, i)
                k = int(passing[i].sum())
                if k >= 2:
                    counts[k][int(passing[i, :chosen[i]].sum())] += 1
    assert n_checked > 0.95 * 8 * P
    # chi-square against the uniform pick, 1e-4 critical values for k - 1 degrees of freedom
    crit = {2: 15.14, 3: 18.42, 4: 21.11, 5: 23.51}
    for k, c in counts.items():
        n = c.sum()
        if n < 200:
            continue
        chi2 = float(((c - n / k) ** 2 / (n / k)).sum())
        assert chi2 < crit[k], (k, c.tolist(), chi2)
    assert sum(c.sum() for c in counts.values()) > 2000


def test_concurrent_replays_reproduce_the_sequential_result(small, dev):
    """Four captured sampler launches (16 runs each) replayed side by side on their own streams, 30 rounds: every replay returns
    the samples, alphas and statistics of the same graph replayed alone, bit for bit.  Next to other work the workgroups of one
    launch start far apart in time; whatever a workgroup reads that another one of the same launch may be writing (the winner
    slots, the converged flag) must not make its threads -- or the workgroups of a run -- disagree."""
    rho, B, P = rho_of(util.ckpt("small")), 16, 593
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    graphs, outs, refs = [], [], []
    for g in range(4):
        with torch.cuda.stream(streams[g]):
            small.surface_sample_batched(B, P, rho, 4, 200, seed=(g + 1) << 40)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=streams[g]):
            out = small.surface_sample_batched(B, P, rho, 4, 200, seed=(g + 1) << 40)
        graphs.append(gr)
        outs.append(out)
    torch.cuda.synchronize()
    for g in range(4):
        graphs[g].replay()
        torch.cuda.synchronize()
        refs.append([t.clone() for t in outs[g]])
        eager = small.surface_sample_batched(B, P, rho, 4, 200, seed=(g + 1) << 40)
        assert all(torch.equal(x, y) for x, y in zip(eager, refs[g]))
    for rnd in range(30):
        for g in range(4):
            with torch.cuda.stream(streams[g]):
                graphs[g].replay()
        torch.cuda.synchronize()
        for g in range(4):
            for x, y, what in zip(outs[g], refs[g], ("samples", "alpha", "stats")):
                assert torch.equal(x, y), (rnd, g, what)
