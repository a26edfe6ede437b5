"""GPU parity: field lookups, Ref head, iso-cell emission and the march, HIP (through the C ABI) vs oracle/golden.

Tolerances (fp32 path; the kernels sum taps and channels in a different order than aten and use the GPU's
expf/log1pf/powf/sinf): absolute unless noted.
"""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

TOL_FEATURE = 3e-5     # density / appearance features (magnitude up to ~30 -> ~1e-6 relative)
TOL_ALPHA = 2e-6
TOL_RGB = 2e-5
TOL_UNIT = 2e-6        # unit vectors


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def tiny(dev):
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    return field_handle_from_ckpt(util.ckpt("tiny"), dev)


@pytest.fixture(scope="module")
def small(dev):
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    return field_handle_from_ckpt(util.ckpt("small"), dev)


def close(got, want, atol, rtol=0.0, what=""):
    got = torch.as_tensor(got).detach().cpu().float()
    want = torch.as_tensor(want).float()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    torch.testing.assert_close(got, want, atol=atol, rtol=rtol, equal_nan=True, msg=lambda m: f"{what}: {m}")


def test_g1_point_lookups(golden, tiny, dev):
    g = golden["g1_field_points"]
    x = golden.t("g1_field_points", "xyz").to(dev)
    xn = tiny.normalize_coord(x)
    close(xn, g["xn"], 1e-6, what="normalize_coord")
    xn_ref = golden.t("g1_field_points", "xn").to(dev)
    close(tiny.density_feature(xn_ref), g["density_feature"], TOL_FEATURE, 2e-6, "density_feature")
    close(tiny.app_feature(xn_ref), g["app_feature"], TOL_FEATURE, 2e-6, "app_feature")
    close(tiny.mask_sample(x), g["mask_value"], 1e-6, what="mask")
    # the boolean the kernels branch on must be identical
    assert np.array_equal(tiny.mask_sample(x).cpu().numpy() > 0, g["mask_value"] > 0)
    close(tiny.point_alpha(x, 1.0), g["alpha_len1"], TOL_ALPHA, 2e-6, "alpha(length=1)")
    close(tiny.point_alpha(x, float(g["step_size"])), g["alpha_len_step"], TOL_ALPHA, 2e-6, "alpha(length=step)")


def test_corner_bit_occupancy_is_the_sign_of_the_trilinear_value(golden, tiny, small, dev):
    """iff_mask_occupied (one byte of the corner-bit table: what the march, compute_alpha and the sampler branch on) against
    `sample_alpha(xyz) > 0` -- the reference's own values (G1) and iff_mask_sample's trilinear sum -- on the points where the two
    formulations could part: coordinates exactly ON mask texels (a zero high-corner weight), on faces / edges / corners of the
    volume, up to two texels outside it, NaN and infinities; and on random points around an occupancy boundary."""
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    x = golden.t("g1_field_points", "xyz").to(dev)
    assert np.array_equal(tiny.mask_occupied(x).cpu().numpy(), golden["g1_field_points"]["mask_value"] > 0)
    for h, ck in ((tiny, util.ckpt("tiny")), (small, util.ckpt("small")),
                  (field_handle_from_ckpt(util.ckpt("tiny", contraction_type="unisphere", density_shift=0.0, peak=6.0), dev), None)):
        ck = ck or util.ckpt("tiny", contraction_type="unisphere", density_shift=0.0, peak=6.0)
        lo, hi = ck["alphaMask.aabb"][0].double(), ck["alphaMask.aabb"][1].double()
        D, H, W = [int(v) for v in tuple(ck["alphaMask.shape"])[-3:]]
        gen = torch.Generator().manual_seed(77)
        pts = []
        # lattice: every texel coordinate of the mask (and half-way points, and one / two texels outside) on every axis
        ax = [torch.cat((torch.arange(-2, n + 2).double(), torch.arange(-1, n).double() + 0.5)) / max(n - 1, 1) for n in (W, H, D)]
        gx, gy, gz = torch.meshgrid(ax[0][::2], ax[1][::3], ax[2][::2], indexing="ij")
        pts.append(torch.stack((gx, gy, gz), -1).reshape(-1, 3) * (hi - lo) + lo)
        gx, gy, gz = torch.meshgrid(ax[0][1::3], ax[1][::2], ax[2][1::2], indexing="ij")
        pts.append(torch.stack((gx, gy, gz), -1).reshape(-1, 3) * (hi - lo) + lo)
        pts.append(lo + (hi - lo) * (torch.rand(20000, 3, generator=gen).double() * 1.3 - 0.15))
        p = torch.cat(pts).float()
        p[:6] = torch.tensor([[float("nan"), 0, 0], [0, float("inf"), 0], [0, 0, -float("inf")], [1e30, 0, 0], [0, -1e30, 0],
                              [float("nan")] * 3])
        p = p.to(dev)
        val = h.mask_sample(p)
        occ = h.mask_occupied(p)
        assert torch.equal(occ, val > 0), int((occ != (val > 0)).sum())
        assert 0.05 < float(occ.float().mean()) < 0.95           # both outcomes are exercised
        on_texel = int(((val > 0) & (val < 1e-6)).sum()) + int((val == 0).sum())
        assert on_texel > 100


def test_g11_unisphere(golden, dev):
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    g = golden["g11_unisphere"]
    h = field_handle_from_ckpt(util.ckpt("tiny", contraction_type="unisphere", density_shift=0.0, peak=6.0), dev)
    x = golden.t("g11_unisphere", "xyz").to(dev)
    close(h.normalize_coord(x), g["xn"], 2e-6, what="unisphere normalize")
    close(h.point_alpha(x), g["alpha"], 5e-6, 5e-6, "unisphere alpha")
    assert np.array_equal(h.mask_sample(x).cpu().numpy() > 0, g["mask_value"] > 0)


def test_g3_ref_head(golden, tiny, dev):
    g = golden["g3_ref_head"]
    rgb = tiny.ref_shade(golden.t("g3_ref_head", "dirs").to(dev), golden.t("g3_ref_head", "feat").to(dev))
    close(rgb, g["rgb"], TOL_RGB, what="Ref.forward")
    close(tiny.head_normals(golden.t("g3_ref_head", "feat").to(dev)), g["normals"], TOL_UNIT, what="compute_normals")


def test_g4_g5_emit(golden, small, dev):
    from iffnerf_amd.hip_field import isocell_emit
    g4, g5 = golden["g4_isocell"], golden["g5_emit"]
    iso = golden.t("g4_isocell", "iso")
    n4 = golden.t("g4_isocell", "normals").to(dev)
    ori, dirs = isocell_emit(iso, torch.zeros_like(n4), n4)
    rot = torch.from_numpy(g4["rotated"])
    want = rot / torch.linalg.norm(rot, dim=-1, keepdim=True)      # sampling.py:455-457 renormalises
    close(dirs.view(-1, 27, 3)[2:], want[2:], TOL_UNIT, what="rotate_isocell")
    assert torch.isnan(dirs.view(-1, 27, 3)[0]).all()                 # -normal == +z: the reference yields NaN too
    # normals and the 27-ray fan on fixed surface samples
    s = golden.t("g5_emit", "samples").to(dev)
    nrm = small.point_normals(s)
    close(nrm, g5["normals"], TOL_UNIT * 5, what="samples_points_normals")
    ori, dirs = isocell_emit(iso, s, golden.t("g5_emit", "normals").to(dev))
    close(ori, g5["ori"], 0.0, what="origins")
    close(dirs, g5["dirs"], TOL_UNIT, what="dirs")
    rgb, _, _, _, _, _ = small.march(torch.cat([golden.t("g5_emit", "ori"), golden.t("g5_emit", "dirs")], -1).to(dev), 0, 20)
    close(rgb, g5["rgb"], TOL_RGB, what="ray rgb")


def test_g2_march_point(golden, small, dev):
    g = golden["g2_march_point"]
    rays = golden.t("g2_march_point", "rays").to(dev)
    rgb, depth, acc, alpha, counts, S = small.march(rays, 0, 20, want_alpha=True, want_counts=True)
    assert S == 20
    close(alpha, g["alpha"], TOL_ALPHA * 5, 5e-6, "alpha")
    close(acc, g["acc"], 5e-6, what="acc")
    close(depth, g["depth"], 5e-6, what="depth")
    close(rgb, g["rgb"], TOL_RGB, what="rgb")
    # counters agree with the oracle's (valid, shaded) sample counts
    from oracle import field as ofield
    f = ofield.field_from_ckpt(util.ckpt("small"))
    want_counts = ofield.march(f, golden.t("g2_march_point", "rays"), "point", 20)[-1]
    assert torch.equal(counts.cpu().long(), want_counts.long())


def test_g10_march_slab(golden, dev):
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    g = golden["g10_march_slab"]
    ck = dict(util.ckpt("small"))
    ck["kwargs"] = dict(ck["kwargs"], near_far=[float(v) for v in g["near_far"]])
    h = field_handle_from_ckpt(ck, dev)
    rays = golden.t("g10_march_slab", "rays").to(dev)
    rgb, depth, acc, alpha, _, S = h.march(rays, 1, -1, want_alpha=True)
    assert S == int(g["n_samples"])
    close(acc, g["acc"], 2e-5, what="slab acc")
    close(depth, g["depth"], 1e-4, what="slab depth")
    close(alpha.sum(-1), g["alpha_sum"], 1e-4, 1e-5, "slab alpha sum")
    close(rgb, g["rgb"], 5e-5, what="slab rgb")
    close(h.march(rays, 1, -1, bg=(1.0, 1.0, 1.0))[0], g["rgb_white"], 5e-5, what="slab rgb white bg")


def test_march_edge_cases(small, dev):
    # empty batch, rays entirely outside, NaN directions (what rotate_isocell emits for normals along z)
    rgb, depth, acc, alpha, _, _ = small.march(torch.zeros(0, 6, device=dev), 0, 20)
    assert rgb.shape == (0, 3) and alpha.shape == (0, 20)
    rays = torch.tensor([[9.0, 9.0, 9.0, 0.0, 0.0, 1.0], [0.0, 0.0, 0.0, float("nan"), float("nan"), float("nan")]], device=dev)
    rgb, depth, acc, alpha, counts, _ = small.march(rays, 0, 20, want_counts=True)
    assert torch.equal(rgb.cpu(), torch.zeros(2, 3)) and torch.equal(acc.cpu(), torch.zeros(2))
    assert torch.equal(counts.cpu(), torch.zeros(2, 2, dtype=torch.int32))
    # ragged tile sizes (not a multiple of the 16-ray tile) give the same per-ray answers
    g = torch.Generator().manual_seed(5)
    r = torch.cat([torch.randn(37, 3, generator=g) * 0.3, torch.nn.functional.normalize(torch.randn(37, 3, generator=g), dim=-1)], -1).to(dev)
    full = small.march(r, 0, 20)[0]
    part = small.march(r[:21].contiguous(), 0, 20)[0]
    assert torch.equal(full[:21], part)
    with pytest.raises(RuntimeError):
        small.march(torch.zeros(4, 5, device=dev), 0, 20)
    with pytest.raises(RuntimeError):
        small.march(torch.zeros(4, 6), 0, 20)    # CPU tensor: no fallback


def test_determinism(small, dev):
    g = torch.Generator().manual_seed(9)
    r = torch.cat([torch.randn(5000, 3, generator=g) * 0.4, torch.nn.functional.normalize(torch.randn(5000, 3, generator=g), dim=-1)], -1).to(dev)
    a = small.march(r, 0, 20)
    b = small.march(r, 0, 20)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)


def test_one_lane_density_gather_is_bit_identical_to_the_four_lane_form(dev):
    """K4a with one lane per sample (taps computed once for all 16 density channels) against the four-lanes-per-sample form
    (iff_field_desc.density_lanes = 4): density_full reproduces the butterfly's summation order, so every output must be
    equal, in the point-centred and in the slab sampler."""
    from iffnerf_amd import synthetic
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    from iffnerf_amd.pipeline import PosePipeline
    ck = util.ckpt("small")
    pipe = PosePipeline.from_checkpoints(ck, synthetic.make_id_weights(seed=99), dev)
    one, four = field_handle_from_ckpt(ck, dev, density_lanes=1), field_handle_from_ckpt(ck, dev, density_lanes=4)
    for P in (75, 9, 1):
        ori, dirs, _ = pipe.emit(P, seed=17)
        rays = torch.cat((ori, dirs), -1).contiguous()
        for r, mode, S in ((rays, 0, 20), (rays[:50].contiguous(), 1, 0)):
            a = one.march(r, mode, S, want_alpha=True, want_counts=True)
            b = four.march(r, mode, S, want_alpha=True, want_counts=True)
            for x, y in zip(a[:5], b[:5]):
                assert torch.equal(x, y), (P, mode)


def test_ref_head_forms_return_the_same_bits(small, dev):
    """Ref.forward in the 8-lanes-per-ray form (bottleneck on the fp32 matrix cores: k_ref_shade_oct, the default for the reference's
    head shape) against the 16-lanes-per-ray vector form (k_ref_shade: a handle made with iff_field_desc.head_lanes = 16), plain and
    as the march's shade + blend step: EQUAL bits, on full, ragged and looped tile counts."""
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    small16 = field_handle_from_ckpt(util.ckpt("small"), dev, head_lanes=16)
    g = torch.Generator().manual_seed(5)
    for n in (1, 31, 32, 33, 1000, 40007):                                  # 40007 rays = 1251 tiles > the grid's 1024 workgroups
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        feat = (torch.randn(n, 27, generator=g) * 2.0).to(dev)
        want = small16.ref_shade(d, feat)
        got = small.ref_shade(d, feat)
        assert torch.equal(got, want), (n, float((got - want).abs().max()))
    gen = field_handle_from_ckpt(util.ckpt("small"), dev, density_lanes=1)   # the general march: K4a, K4b, shade + blend
    gen16 = field_handle_from_ckpt(util.ckpt("small"), dev, density_lanes=1, head_lanes=16)
    assert gen.march_plan(0, 20) == 0 and gen16.march_plan(0, 20) == 0
    for R in (540, 37):
        o = (torch.rand(R, 3, generator=g) - 0.5) * 2.0
        rays = torch.cat((o, torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)), -1).to(dev)
        for bg in ((0.0, 0.0, 0.0), (1.0, 0.5, 0.25)):
            want = gen16.march(rays, 0, 20, bg=bg)[0]
            assert torch.equal(gen.march(rays, 0, 20, bg=bg)[0], want), (R, bg)


def test_fan_march_equals_the_general_kernels(dev):
    """The fused fan kernel (k4f_fan_march: LDS-staged table patches per 27-ray tile, or its in-kernel gather path when a tile's
    samples do not fit one patch) against the general kernels K4a / K4b (iff_field_desc.density_lanes = 1 keeps those): the
    per-sample arithmetic is shared, so alpha, acc, depth and the (valid, shaded) counters must be EQUAL; the colours differ
    only by the order in which the twelve quarter sums of basis_mat are added."""
    from iffnerf_amd import synthetic
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    from iffnerf_amd.pipeline import PosePipeline
    g = torch.Generator().manual_seed(31)
    for which, over in (("small", {}), ("tiny", {}), ("small", dict(grid=(300, 280, 260), mask_res=(60, 56, 52))),
                        ("small", dict(contraction_type="unisphere", density_shift=0.0, density_offset=-10.0, peak=20.0,
                                       aabb=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), near_far=(0.01, 1.4), blob_sigma=0.30, mask_radius=0.62,
                                       step_ratio=0.25))):      # unisphere halves the grid in the step (tensorBase.py:361): 0.25 keeps 5 texels per 10 steps
        ck = util.ckpt(which, **over)
        pipe = PosePipeline.from_checkpoints(ck, synthetic.make_id_weights(seed=99), dev)
        fan, gen = field_handle_from_ckpt(ck, dev), field_handle_from_ckpt(ck, dev, density_lanes=1)
        head16 = field_handle_from_ckpt(ck, dev, head_lanes=16)               # fan kernel without its head phase + the 16-lane head launch
        cases = []
        for P in (75, 9, 1):                                                   # fans: the staged path
            ori, dirs, _ = pipe.emit(P, seed=17 + P)
            cases.append(torch.cat((ori, dirs), -1).contiguous())
        cases.append(cases[0][:27 * 3 + 11].contiguous())                      # a ragged last tile
        for R in (540, 37):                                                    # arbitrary rays: the gather path
            o = (torch.rand(R, 3, generator=g) - 0.5) * 2.0
            cases.append(torch.cat((o, torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)), -1).to(dev))
        mixed = torch.cat((cases[0][:54], cases[4][:27], cases[0][54:108]))    # staged and gathered tiles in one launch
        cases.append(mixed.contiguous())
        for rays in cases:
            a = fan.march(rays, 0, 20, want_alpha=True, want_counts=True)
            b = gen.march(rays, 0, 20, want_alpha=True, want_counts=True)
            for x, y, what in zip(a[1:5], b[1:5], ("depth", "acc", "alpha", "counts")):
                assert torch.equal(x, y), (which, over, tuple(rays.shape), what)
            close(a[0], b[0].cpu(), 2e-6, what=f"rgb {which} {tuple(rays.shape)}")
            fa, fb = fan.march_features(rays, 0, 20)[0], gen.march_features(rays, 0, 20)[0]
            close(fa, fb.cpu(), 2e-5, 2e-6, what="weighted features")
            assert torch.equal(fa[:, 27], fb[:, 27])
            # the Ref head fused into the fan kernel (plan 3: bottleneck on the fp32 matrix cores, four lanes per ray in two of the
            # tile's waves) against the separate head kernel (16 lanes per ray) on the fan kernel's own features, blended as
            # k_ref_shade blends: EQUAL colours
            assert fan.march_plan(0, 20) == 3 and head16.march_plan(0, 20) == 2 and gen.march_plan(0, 20) == 0
            for bg in ((0.0, 0.0, 0.0), (1.0, 0.5, 0.25)):
                fused = fan.march(rays, 0, 20, bg=bg)
                assert torch.equal(head16.march(rays, 0, 20, bg=bg)[0], fused[0])     # the fan kernel + the separate 16-lane head launch
                c = head16.ref_shade(rays[:, 3:6].contiguous(), fa[:, :27].contiguous())
                c = torch.where(fa[:, 27:28] != 0, c, torch.zeros_like(c))
                acc = fused[2].reshape(-1, 1)
                want = (c * acc + torch.tensor(bg, device=c.device) * (1.0 - acc)).clamp(0.0, 1.0)
                assert torch.equal(fused[0], want), (which, over, tuple(rays.shape), bg, float((fused[0] - want).abs().max()))


def test_eight_wave_fan_kernel_equals_the_other_forms(dev):
    """k4g_fan_march (eight waves per fan, patches by global -> LDS DMA into two buffers, sixteen lanes per ray; iff_field_desc.fan_waves
    = 8) against the general kernels and the four-wave fan kernel: the per-sample arithmetic is shared, so alpha, acc, depth and the
    (valid, shaded) counters are EQUAL to both; the colours differ only by the order in which a ray's sample chains and the quarter
    sums of basis_mat are added.  Both patch sizes: 12-texel boxes (aabb scenes, one pass per 48-channel plane) and 22-texel boxes
    (unisphere contraction at the reference's step_ratio 0.5: a fan spans up to 21 texels; 16-channel slices, twelve passes), staged
    fans, ragged tiles, arbitrary rays (the in-kernel gather path), fans at the table faces (zero-padded taps)."""
    from iffnerf_amd import synthetic
    from iffnerf_amd.hip_field import field_handle_from_ckpt
    from iffnerf_amd.pipeline import PosePipeline
    g = torch.Generator().manual_seed(32)
    uni = dict(contraction_type="unisphere", density_shift=0.0, density_offset=-10.0, peak=20.0, aabb=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)),
               near_far=(0.01, 1.4), blob_sigma=0.30, mask_radius=0.62)
    for which, over, side in (("small", {}, 12), ("tiny", {}, 12), ("small", dict(grid=(300, 280, 260), mask_res=(60, 56, 52)), 12),
                              ("small", dict(step_ratio=0.25, **uni), 12),
                              ("small", dict(grid=(96, 96, 96), mask_res=(40, 36, 44), step_ratio=0.5, **uni), 22),
                              ("small", dict(grid=(64, 64, 64), mask_res=(20, 18, 16), step_ratio=0.5, **uni), 22)):      # patches that run over the table edge (clamped rows / columns)
        # (unisphere: the step is the MEAN grid unit, so ten steps are ten texels only on a cubic grid -- configs/bicycle.txt: 640^3)
        ck = util.ckpt(which, **over)
        pipe = PosePipeline.from_checkpoints(ck, synthetic.make_id_weights(seed=99), dev)
        gen = field_handle_from_ckpt(ck, dev, density_lanes=1)
        f8 = field_handle_from_ckpt(ck, dev, fan_waves=8)
        f4 = field_handle_from_ckpt(ck, dev, fan_waves=4)
        h16 = field_handle_from_ckpt(ck, dev, head_lanes=16, fan_waves=8)         # the eight-wave kernel without its head phase + the head launch
        assert f8.march_plan(0, 20) == 3 and h16.march_plan(0, 20) == 2 and gen.march_plan(0, 20) == 0
        assert f4.march_plan(0, 20) == (3 if side == 12 else 0)                    # the four-wave kernel stages 12-texel boxes only
        cases = []
        for P in (75, 9, 1):
            ori, dirs, _ = pipe.emit(P, seed=23 + P)
            cases.append(torch.cat((ori, dirs), -1).contiguous())
        cases.append(cases[0][:27 * 3 + 11].contiguous())                      # a ragged last tile
        for R in (540, 37):                                                    # arbitrary rays: the gather path
            o = (torch.rand(R, 3, generator=g) - 0.5) * 2.0
            cases.append(torch.cat((o, torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)), -1).to(dev))
        cases.append(torch.cat((cases[0][:54], cases[4][:27], cases[0][54:108])).contiguous())     # staged and gathered tiles in one launch
        for rays in cases:
            a = f8.march(rays, 0, 20, want_alpha=True, want_counts=True)
            b = gen.march(rays, 0, 20, want_alpha=True, want_counts=True)
            for x, y, what in zip(a[1:5], b[1:5], ("depth", "acc", "alpha", "counts")):
                assert torch.equal(x, y), (which, over, tuple(rays.shape), what, int((x != y).sum()))
            close(a[0], b[0].cpu(), 2e-6, what=f"rgb {which} {tuple(rays.shape)}")
            fa, fb = f8.march_features(rays, 0, 20)[0], gen.march_features(rays, 0, 20)[0]
            close(fa, fb.cpu(), 2e-5, 2e-6, what="weighted features")
            assert torch.equal(fa[:, 27], fb[:, 27])
            if side == 12:
                c4 = f4.march(rays, 0, 20, want_alpha=True, want_counts=True)
                for x, y in zip(a[1:5], c4[1:5]):
                    assert torch.equal(x, y)
                close(a[0], c4[0].cpu(), 2e-6, what="rgb vs the four-wave kernel")
            for bg in ((0.0, 0.0, 0.0), (1.0, 0.5, 0.25)):
                fused = f8.march(rays, 0, 20, bg=bg)
                assert torch.equal(h16.march(rays, 0, 20, bg=bg)[0], fused[0])     # the fused head == the separate 16-lane head launch
            again = f8.march(rays, 0, 20, want_alpha=True, want_counts=True)       # and the launch is deterministic
            for x, y in zip(a[:5], again[:5]):
                assert torch.equal(x, y)
