"""GPU: the image side of stage C: token assembly kernel, mask select on the softmax rows, and the
image-in -> pose-out capture, against the oracle's / the mirrored module's compacting formulation."""
import pytest
import torch

from iffnerf_amd import synthetic
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _keep_pattern(q, g=16):
    gen = torch.Generator().manual_seed(100 + q)
    yy, xx = torch.meshgrid(torch.arange(g), torch.arange(g), indexing="ij")
    cy, cx, r = 4 + 2 * q, 9 - q, 5.5 + 0.5 * q
    blob = ((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r
    return blob | (torch.rand(g, g, generator=gen) > 0.93)


def test_token_assemble_matches_get_img_position_encoding(dev):
    from iffnerf_amd.image_frontend import token_assemble
    from oracle import identify as oid
    gen = torch.Generator().manual_seed(4)
    patch = torch.randn(3, 256, 384, generator=gen)
    keep = torch.stack([_keep_pattern(q) for q in range(3)])
    mask_grid = torch.where(keep, torch.rand(3, 16, 16, generator=gen) * 0.8 + 0.15, torch.rand(3, 16, 16, generator=gen) * 0.09)
    out, flags = token_assemble(patch.to(dev), (16, 16), mask_grid.to(dev), 0.1)
    assert out.shape == (3, 256, 398) and torch.equal(flags.cpu().bool(), keep.reshape(3, 256))
    assert torch.equal(out[..., :384].cpu(), patch)
    for q in range(3):
        want = oid.tokens_with_pe(patch[q], keep[q])                    # identification_module.py:149-160, compacting
        got = out[q].cpu()[keep[q].reshape(-1)]
        assert got.shape == want.shape
        assert torch.equal(got[:, :386], want[:, :386])                 # features and the two raw positions: exact
        torch.testing.assert_close(got[:, 386:], want[:, 386:], atol=1e-6, rtol=0)      # sin / cos: libm
    # no mask: every row kept; a non-square grid follows 'ij' indexing
    out2, flags2 = token_assemble(patch[:1, :12 * 20].contiguous().to(dev), (12, 20))
    assert bool(flags2.all()) and out2.shape == (1, 240, 398)
    pe = oid.image_position_encoding((12, 20), 3).reshape(240, 14)
    torch.testing.assert_close(out2[0, :, 384:].cpu(), pe, atol=1e-6, rtol=0)
    with pytest.raises(RuntimeError):
        token_assemble(patch, (16, 16))                                 # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        token_assemble(patch.to(dev), (16, 15))


def test_mask_select_on_the_softmax_rows_equals_dropping_the_rows(dev):
    """All 256 rows through the logits, dropped rows' statistics set to (+inf, 1): the scores are those of the compacted
    token tensor (what identification_module.py:157-160 hands to :165-167), and so are the top-100 and the pose."""
    from iffnerf_amd import hip_identify as H
    from iffnerf_amd.image_frontend import mask_token_rows, token_assemble
    from iffnerf_amd.pipeline import PosePipeline
    from oracle import identify as oid
    w = synthetic.make_id_weights(seed=99)
    pipe = PosePipeline.from_checkpoints(util.ckpt("small"), w, dev, model_up=(0.1, 0.2, 0.9))
    ori, dirs, rgb = pipe.emit(75, seed=3)
    rays = pipe.make_resident(ori, dirs, rgb)
    gen = torch.Generator().manual_seed(8)
    patch = torch.randn(2, 256, 384, generator=gen)
    keep = torch.stack([_keep_pattern(q) for q in range(2)])
    tokens, flags = token_assemble(patch.to(dev), (16, 16), keep.float().to(dev), 0.1)
    qf = pipe.idnet.q_fold(tokens.reshape(512, 398))
    logits, rmax, rsum = pipe.idnet.logits_from_cache(qf, rays.cache, ori.shape[0])
    mask_token_rows(flags, rmax, rsum)
    score = H.attn_colsum_batched(logits, rmax, rsum, 2, write_attention=False)
    for q in range(2):
        compact = tokens[q][flags[q].bool()].contiguous()
        M = compact.shape[0]
        assert 60 < M < 256
        want, _ = pipe.scores(compact, ori, dirs, rgb, materialize_map=False)
        torch.testing.assert_close(score[q], want, atol=0, rtol=2e-6)
        assert abs(float(score[q].sum()) - M) < 1e-2
        idx, val = H.topk(score[q], 100)
        assert idx.tolist() == H.topk(want, 100)[0].tolist()
        score_ref = oid.test_image(w, compact.cpu(), ori.cpu(), dirs.cpu(), rgb.cpu(), 100)[2]
        util.assert_topk_matches(idx.cpu(), score_ref, 100)


def test_kept_rows_first_and_the_launches_stop_at_their_count(dev):
    """iff_token_assemble_compact + iff_logits_from_cache_rows + iff_attn_colsum_rows: the static-shape route with the reference's
    work per image.  The assembly is a STABLE partition of iff_token_assemble's rows (kept rows in grid order: the reference's
    boolean index, identification_module.py:157-160); the kept rows' logits and statistics are the bits of the unbounded launch, the
    dropped rows are not touched; the scores are those of the compacted token tensor, the top-100 and the attention-map rows too.
    Edge cases: an image that keeps nothing, one that keeps everything, one with fewer kept rows than a wave's 32."""
    from iffnerf_amd import hip_identify as H
    from iffnerf_amd.image_frontend import mask_token_rows, token_assemble
    from iffnerf_amd.pipeline import PosePipeline
    w = synthetic.make_id_weights(seed=99)
    pipe = PosePipeline.from_checkpoints(util.ckpt("small"), w, dev, model_up=(0.1, 0.2, 0.9))
    ori, dirs, rgb = pipe.emit(75, seed=3)
    rays = pipe.make_resident(ori, dirs, rgb)
    N = ori.shape[0]
    gen = torch.Generator().manual_seed(8)
    Q = 6
    patch = torch.randn(Q, 256, 384, generator=gen)
    keep = torch.stack([_keep_pattern(q % 3) for q in range(Q)]).reshape(Q, 256)
    keep[3] = False                                              # keeps nothing
    keep[4] = True                                               # keeps everything
    keep[5] = False; keep[5, [7, 100, 101, 255]] = True          # four rows: less than one wave's 32 tokens
    mg = keep.float().to(dev)
    plain, flags = token_assemble(patch.to(dev), (16, 16), mg, 0.1)
    tok, flags_c, rows = token_assemble(patch.to(dev), (16, 16), mg, 0.1, compact=True)
    assert rows.dtype == torch.int32 and rows.tolist() == keep.sum(1).tolist() == flags.sum(1).tolist()
    for q in range(Q):
        n = int(rows[q])
        assert torch.equal(tok[q, :n], plain[q][flags[q].bool()]) and torch.equal(tok[q, n:], plain[q][~flags[q].bool()])
        assert flags_c[q].tolist() == [1] * n + [0] * (256 - n)
    # logits: kept rows = the unbounded launch's bits on the same (compacted) rows; dropped rows untouched; statistics (+inf, 1)
    qf = pipe.idnet.q_fold(tok.reshape(Q * 256, 398))
    full, fmax, fsum = pipe.idnet.logits_from_cache(qf, rays.cache, N)
    got, gmax, gsum = pipe.idnet.logits_from_cache(qf, rays.cache, N, rows=rows)
    for q in range(Q):
        n, lo = int(rows[q]), 256 * q
        assert torch.equal(got[lo:lo + n], full[lo:lo + n]) and torch.equal(gmax[lo:lo + n], fmax[lo:lo + n]) and torch.equal(gsum[lo:lo + n], fsum[lo:lo + n])
        assert bool(torch.isinf(gmax[lo + n:lo + 256]).all()) and bool((gsum[lo + n:lo + 256] == 1).all())
    # scores: the column pass over the kept rows only == every row with the dropped ones masked (another order of the same terms)
    score = H.attn_colsum_batched(got, gmax, gsum, Q, write_attention=False, rows=rows)
    mask_token_rows(flags_c, fmax, fsum)
    want = H.attn_colsum_batched(full, fmax, fsum, Q, write_attention=False)
    torch.testing.assert_close(score, want, atol=1e-9, rtol=2e-6)
    assert bool((score[3] == 0).all())                           # nothing kept: nothing summed
    for q in (0, 1, 2, 4, 5):
        assert abs(float(score[q].sum()) - int(rows[q])) < 1e-2
        util.assert_topk_matches(H.topk(score[q], 100)[0].cpu(), want[q].cpu(), 100)
    # poisoned buffers: the bounded column pass must not read a dropped row (NaN there would reach every score)
    poisoned = got.clone()
    for q in range(Q):
        poisoned[256 * q + int(rows[q]):256 * (q + 1)] = float("nan")
    assert torch.equal(H.attn_colsum_batched(poisoned, gmax, gsum, Q, write_attention=False, rows=rows), score)
    with pytest.raises(RuntimeError):
        pipe.idnet.logits_from_cache(qf, rays.cache, N, rows=rows.long())
    with pytest.raises(RuntimeError):
        H.attn_colsum_batched(got, gmax, gsum, Q, write_attention=False, rows=rows[:3])


def test_image_in_pose_out_capture(dev, monkeypatch):
    """ImageFrontEnd reproduces the mirrored module's image_processing (same torch ops + the assembly kernel), and the
    captured image -> pose graph equals the per-image drop-in calls (IdentificationModule.test_image + pose solve)."""
    import iffnerf_amd
    from iffnerf_amd import hip_identify as H
    from iffnerf_amd.image_frontend import ImageFrontEnd
    from iffnerf_amd.pipeline import CapturedImageQuery, PosePipeline
    from iffnerf_amd.pose_estimation import backbone as bb, identification_module as im
    net, grid, C = bb.create_standin_backbone(seed=5)
    net = net.to(dev)
    monkeypatch.setattr(im, "create_backbone", lambda **kw: (net, grid, C))
    mod = im.IdentificationModule("dino").to(dev).eval()
    w = synthetic.make_id_weights(seed=99)
    mod.load_state_dict({k: v for k, v in w.items()}, strict=False)
    pipe = PosePipeline.from_checkpoints(util.ckpt("small"), w, dev, model_up=(0.1, 0.2, 0.9))
    ori, dirs, rgb = pipe.emit(75, seed=3)
    rays = pipe.make_resident(ori, dirs, rgb)
    fe = ImageFrontEnd(net, grid, native_preprocess=False)          # the mirrored module's own torch ops: exact comparisons below
    gen = torch.Generator().manual_seed(2)
    Q, Hh, Ww = 3, 200, 260
    imgs = torch.rand(Q, Hh, Ww, 3, generator=gen).to(dev)
    yy, xx = torch.meshgrid(torch.arange(Hh), torch.arange(Ww), indexing="ij")
    masks = torch.stack([(((yy - 100) ** 2 / (60 + 10 * q) ** 2 + (xx - 130) ** 2 / (90 - 10 * q) ** 2) <= 1).float() for q in range(Q)]).to(dev)
    tokens, keep = fe.tokens(imgs, masks)
    c2w, idx, val = pipe.identify_images_resident(fe, imgs, masks, rays, 100)
    for q in range(Q):
        # one image at a time: the very ops of the mirrored module, so the tokens agree to the PE's libm difference ...
        t_pe, t = mod.image_processing(imgs[q], masks[q])
        tok1, keep1 = fe.tokens(imgs[q:q + 1], masks[q:q + 1])
        got = tok1[0][keep1[0].bool()]
        assert got.shape == t_pe.shape and 30 < got.shape[0] < 256
        assert torch.equal(got[:, :386], t_pe[:, :386])
        torch.testing.assert_close(got[:, 386:], t_pe[:, 386:], atol=1e-6, rtol=0)
        # ... and image -> pose equals IdentificationModule.test_image + the pose solve
        c1, i1, v1 = pipe.identify_images_resident(fe, imgs[q:q + 1], masks[q:q + 1], rays, 100)
        i2, v2, _, _ = mod.test_image(imgs[q], masks[q], ori, dirs, rgb, rays_to_output=100)
        assert len(set(i1[0].tolist()) & set(i2.tolist())) >= 98               # PE libm + row-sum order: near-ties may swap
        torch.testing.assert_close(v1[0], v2, atol=1e-6, rtol=1e-3)
        torch.testing.assert_close(c1[0], H.pose_from_topk(i2, v2, ori, dirs, pipe.model_up), atol=1e-4, rtol=0)
        # the batched front end runs the backbone's GEMMs at another batch size: same tokens to GEMM rounding, same answer
        assert torch.equal(keep[q], keep1[0])
        torch.testing.assert_close(tokens[q], tok1[0], atol=5e-4, rtol=0)
        assert len(set(idx[q].tolist()) & set(i1[0].tolist())) >= 90
        torch.testing.assert_close(c2w[q], c1[0], atol=5e-3, rtol=0)
    cq = CapturedImageQuery(pipe, fe, imgs.shape, rays, 100)
    for rep in range(2):
        cq.replay(imgs, masks)
        torch.cuda.synchronize()
        assert torch.equal(cq.idx, idx) and torch.equal(cq.val, val) and torch.equal(cq.c2w, c2w)
    # the bf16-autocast option of the backbone: same masks, token features within bf16's reach of the fp32 ones, and a
    # capturable graph of its own (a throughput option: it is not parity-equivalent and nothing else uses it)
    fe16 = ImageFrontEnd(net, grid, backbone_autocast=torch.bfloat16, native_preprocess=False)
    tok16, keep16 = fe16.tokens(imgs, masks)
    assert tok16.dtype == torch.float32 and torch.equal(keep16, keep) and torch.equal(tok16[..., 384:], tokens[..., 384:])
    rel = (tok16[..., :384] - tokens[..., :384]).norm() / tokens[..., :384].norm()
    assert 0 < float(rel) < 3e-2, float(rel)
    cq16 = CapturedImageQuery(pipe, fe16, imgs.shape, rays, 100)
    cq16.replay(imgs, masks)
    torch.cuda.synchronize()
    assert torch.isfinite(cq16.c2w).all() and cq16.idx.shape == idx.shape


def test_native_vit_matches_the_fp32_torch_module(dev):
    """iff_vit_forward against the fp32 torch module of the same weights: DINOv2 ViT-S/14's architecture with seeded stand-in weights
    (the published ones are not available offline; both state-dict key schemes are understood by ViTHandle).
    Default precision (IFF_VIT_FP32: split fp16 operands, three products per block, fp32 accumulate / residual / LayerNorm / softmax):
    the reference's accuracy class (identification_module.py:137-142 runs the backbone in fp32) -- tokens to 1e-4 of the largest
    token value, and stage C fed with either token set returns the SAME top-100 list (tests/util.py assert_topk_matches: only
    near-ties of the fp32 scores may swap).  precision="bf16" (the throughput option): bf16-operand tolerance."""
    from oracle import identify as oid
    from iffnerf_amd.hip_vit import ViTHandle, is_served_natively, restore_stock, serve_natively
    from iffnerf_amd.image_frontend import token_assemble
    from iffnerf_amd.pipeline import PosePipeline
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    net, grid, C = create_standin_backbone(seed=3)
    net = net.to(dev)
    gen = torch.Generator().manual_seed(12)
    for Q in (1, 5):
        x = torch.randn(Q, 3, 224, 224, generator=gen).to(dev)
        with torch.no_grad():
            want = net.forward_features(x)
        ref, scale = want["x_norm_patchtokens"], float(want["x_norm_patchtokens"].abs().max())
        # an fp64 evaluation of the same module says how far fp32 torch itself is from the exact tokens
        with torch.no_grad():
            exact = net.double().forward_features(x.double())["x_norm_patchtokens"]
            net.float()
        torch_err = float((ref.double() - exact).abs().max())
        vit = ViTHandle(net.state_dict(), dev)
        assert vit.precision == "fp32"
        tok, cls = vit.forward(x, want_cls=True)
        assert tok.shape == (Q, 256, 384) and cls.shape == (Q, 384)
        err = float((tok - ref).abs().max())
        err_exact = float((tok.double() - exact).abs().max())
        assert torch.isfinite(tok).all() and err <= 1e-4 * scale, (err, scale)
        assert err_exact <= max(4.0 * torch_err, 2e-5 * scale), (err_exact, torch_err)       # as close to the exact tokens as fp32 torch is
        assert float((cls - want["x_norm_clstoken"]).abs().max()) <= 1e-4 * scale
        fast = ViTHandle(net.state_dict(), dev, precision="bf16")
        tok_b, cls_b = fast.forward(x, want_cls=True)
        err_b = float((tok_b - ref).abs().max())
        cos = torch.nn.functional.cosine_similarity(tok_b.reshape(-1, 384), ref.reshape(-1, 384), dim=-1)
        assert torch.isfinite(tok_b).all() and err_b <= 4e-2 * scale and float(cos.min()) > 0.9995, (err_b, scale)     # bf16 operands: ~2^-8 per product, 12 blocks
        assert err < 0.02 * err_b
        # the compact key names of earlier table files load too (qkv / ls1 / fc1 without attn. / .gamma / mlp.) and give the same bits
        sd = {}
        for k, v in net.state_dict().items():
            k2 = k.replace(".attn.qkv.", ".qkv.").replace(".attn.proj.", ".proj.").replace(".mlp.fc", ".fc").replace("patch_embed.proj.", "patch_embed.")
            if k2.endswith(".gamma"):
                k2 = k2[:-len(".gamma")]
            sd[k2] = v
        assert "blocks.0.qkv.weight" in sd and "blocks.0.ls1" in sd and "patch_embed.weight" in sd
        assert torch.equal(ViTHandle(sd, dev).forward(x), tok)
    # served in place: the SAME module, its keys untouched, no-grad inference through the kernels, same dictionary as the module
    keys = list(net.state_dict().keys())
    assert serve_natively(net) is net and is_served_natively(net) and list(net.state_dict().keys()) == keys
    with torch.no_grad():
        out = net.forward_features(x)
    assert torch.equal(out["x_norm_patchtokens"], tok) and torch.equal(out["x_norm_clstoken"], cls)
    # under autograd on a trainable backbone the module's own torch forward runs (and back-propagates)
    xs = x[:1].clone().requires_grad_(True)
    assert net.forward_features(xs)["x_norm_patchtokens"].requires_grad
    # a parameter changed in place: the handle is rebuilt from the new weights
    with torch.no_grad():
        net.norm.weight.mul_(2.0)
        out2 = net.forward_features(x)
        net.norm.weight.mul_(0.5)
    assert not torch.equal(out2["x_norm_patchtokens"], tok)
    assert restore_stock(net) is net and not is_served_natively(net) and "forward_features" not in net.__dict__
    # stage C on the golden-size ray set: the native fp32-class tokens select the top-100 LIST the fp32 torch tokens select
    pipe = PosePipeline.from_checkpoints(util.ckpt("small"), synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))
    ori, dirs, rgb = pipe.emit(300, seed=9)
    w = synthetic.make_id_weights(seed=99)
    for q in range(min(Q, 3)):
        ta, _ = token_assemble(tok[q:q + 1], grid)
        tb, _ = token_assemble(ref[q:q + 1].contiguous(), grid)
        tc, _ = token_assemble(tok_b[q:q + 1], grid)
        _, ia, _ = pipe.identify(ta[0], ori, dirs, rgb, k=100, materialize_map=False)
        score_ref = oid.test_image(w, tb[0].cpu(), ori.cpu(), dirs.cpu(), rgb.cpu(), 100)[2]        # the oracle on the fp32 torch tokens
        assert util.assert_topk_matches(ia.cpu(), score_ref, 100, rel_tie=1e-4) <= 4          # token differences of 1e-5 move scores by as much
        _, ic, _ = pipe.identify(tc[0], ori, dirs, rgb, k=100, materialize_map=False)
        assert len(set(ic.tolist()) & set(ia.tolist())) >= 90                                    # the bf16 option: most of the same rays
    with pytest.raises(RuntimeError):
        vit.forward(x.cpu())
    with pytest.raises(RuntimeError):
        vit.forward(torch.zeros(1, 3, 200, 224, device=dev))


def test_vit_gemm_forms_return_the_same_bits(dev):
    """A batch of 24 images or more takes the wide GEMM kernels (csrc/vit_kernels.hip gemm()): register-staged 128 x 128 tiles
    (form 1) or tiles filled by LDS-DMA in three shapes (forms 2-4; 0 = the library's choice).  Every form adds a row's products
    in the same order, so all of them -- and the 64-token kernels that serve the same images in smaller batches -- return the
    SAME tokens bit for bit, in both precisions; and they are the fp32 torch module's tokens to 1e-4."""
    from iffnerf_amd.hip_vit import ViTHandle
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    net, grid, C = create_standin_backbone(seed=4)
    net = net.to(dev)
    Q = 25                                                   # 6 425 tokens: the last 128- and 256-token tiles are partly empty
    x = torch.randn(Q, 3, 224, 224, generator=torch.Generator().manual_seed(31)).to(dev)
    with torch.no_grad():
        ref = net.forward_features(x)["x_norm_patchtokens"]
    scale = float(ref.abs().max())
    for prec in ("fp32", "bf16"):
        outs = {}
        for form in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            vit = ViTHandle(net.state_dict(), dev, precision=prec, gemm_form=form)
            tok, cls = vit.forward(x, want_cls=True)
            outs[form] = (tok, cls)
            assert torch.isfinite(tok).all()
        for form in (0, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            assert torch.equal(outs[form][0], outs[1][0]) and torch.equal(outs[form][1], outs[1][1]), (prec, form)
        small = ViTHandle(net.state_dict(), dev, precision=prec)
        parts = torch.cat([small.forward(x[i:i + 5]) for i in range(0, Q, 5)])             # the same images, five per call: 64-token tiles
        assert torch.equal(parts, outs[0][0]), prec
        err = float((outs[0][0] - ref).abs().max())
        assert err <= (1e-4 if prec == "fp32" else 4e-2) * scale, (prec, err, scale)
    with pytest.raises(RuntimeError, match="gemm_form"):
        ViTHandle(net.state_dict(), dev, gemm_form=99)


def test_vit_dma_gemm_reproduces_itself_under_concurrency(dev):
    """The batch-of-32 forward (operand tiles by LDS-DMA into two buffers: a read that overtook its DMA, or a refill that overtook a read,
    would show as a rare wrong tile) 40 times on four streams at once, both precisions: every output equals the first bit for bit, and
    the first equals the register-staged kernels'."""
    from iffnerf_amd.hip_vit import ViTHandle
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    net, grid, C = create_standin_backbone(seed=6)
    net = net.to(dev)
    x = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(77)).to(dev)
    for prec in ("fp32", "bf16"):
        want = ViTHandle(net.state_dict(), dev, precision=prec, gemm_form=1).forward(x).clone()
        vits = [ViTHandle(net.state_dict(), dev, precision=prec) for _ in range(4)]          # a handle owns its workspace: one per stream
        streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
        torch.cuda.synchronize(dev)
        outs = []
        for rep in range(10):
            for v, st in zip(vits, streams):
                with torch.cuda.stream(st):
                    outs.append(v.forward(x).clone())
        torch.cuda.synchronize(dev)
        bad = [i for i, o in enumerate(outs) if not torch.equal(o, want)]
        assert not bad, (prec, bad[:8], len(outs))


def test_native_vit_small_activations(dev):
    """The fp32 class splits an operand into two fp16 pieces; for |v| < 2^-3 the low piece is an fp16 SUBNORMAL (csrc/vit_kernels.hip
    split_h).  Move every matrix product's activations there -- LayerNorm gains and biases times 2^-7, the weights that consume them
    times 2^7, exact rescalings that leave the network's function unchanged -- and the tokens must still agree with fp32 torch on the
    same (rescaled) module to 1e-4: the matrix cores multiply fp16 subnormals, they do not flush them."""
    from iffnerf_amd.hip_vit import ViTHandle
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    net, grid, C = create_standin_backbone(seed=4)
    net = net.to(dev)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        plain = net.forward_features(x)["x_norm_patchtokens"]
        for b in net.blocks:
            for ln, lin in ((b.norm1, b.attn.qkv), (b.norm2, b.mlp.fc1)):
                ln.weight.mul_(2.0 ** -7), ln.bias.mul_(2.0 ** -7), lin.weight.mul_(2.0 ** 7)
        ref = net.forward_features(x)["x_norm_patchtokens"]
        small = float(net.blocks[0].norm1(torch.randn(4, 384, device=dev)).abs().max())
    assert small < 2.0 ** -3                                   # the GEMM inputs really sit in the subnormal-lo range
    torch.testing.assert_close(ref, plain, atol=2e-5 * float(plain.abs().max()), rtol=0)          # same function (power-of-two rescaling)
    tok = ViTHandle(net.state_dict(), dev).forward(x)
    scale = float(ref.abs().max())
    assert torch.isfinite(tok).all() and float((tok - ref).abs().max()) <= 1e-4 * scale


def test_native_backbone_serves_other_grids(dev):
    """The installed forward_features serves every input the hub module serves: another multiple-of-14 size gets its own handle
    (position table interpolated to that grid), a size that is no multiple of the patch or beyond the kernels' token budget runs the
    module's own forward on the GPU, a CPU tensor raises (no CPU path)."""
    from iffnerf_amd.hip_vit import serve_natively
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    net, grid, C = create_standin_backbone(seed=6)
    net = serve_natively(net.to(dev), grid)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for hw in ((224, 224), (196, 168), (112, 224)):
            x = torch.randn(2, 3, *hw, generator=gen).to(dev)
            got = net.forward_features(x)["x_norm_patchtokens"]
            want = type(net).forward_features(net, x)["x_norm_patchtokens"]
            assert got.shape == want.shape == (2, (hw[0] // 14) * (hw[1] // 14), 384)
            assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()) and not torch.equal(got, want)      # native, not stock
        assert len(net.__dict__["forward_features"]._handles) == 3
        big = torch.randn(1, 3, 252, 252, generator=gen).to(dev)                  # 325 tokens: beyond the kernels' budget -> the module's own forward
        assert torch.equal(net.forward_features(big)["x_norm_patchtokens"], type(net).forward_features(net, big)["x_norm_patchtokens"])
        with pytest.raises(RuntimeError, match="GPU"):
            net.forward_features(torch.zeros(1, 3, 224, 224))


def test_native_resize_crop_normalize_matches_the_torch_formulation(dev):
    """iff_image_resize_crop (one kernel: antialiased bicubic / bilinear resize of the shorter edge, centre crop, normalisation,
    channels-first) against the mirrored module's F.interpolate(antialias=True) + crop + normalise -- the torchvision transforms of
    identification_module.py:36-61 as this package restates them (torchvision itself is absent from the image: that boundary stays
    'parity unpinned')."""
    from iffnerf_amd.image_frontend import ImageFrontEnd, resize_crop
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    from iffnerf_amd.pose_estimation.identification_module import (IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD, _center_crop,
                                                                   _resize_short_edge)
    gen = torch.Generator().manual_seed(8)
    mean = torch.tensor(IMAGENET_DEFAULT_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_DEFAULT_STD, device=dev).view(1, 3, 1, 1)
    for (H, W) in ((800, 800), (600, 900), (1080, 1920), (300, 260), (224, 224), (150, 170)):
        imgs = torch.rand(2, H, W, 3, generator=gen).to(dev)
        want = (_center_crop(_resize_short_edge(imgs.permute(0, 3, 1, 2), 256, "bicubic"), 224) - mean) / std
        got = resize_crop(imgs, 256, 224, True, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)
        assert got.shape == want.shape == (2, 3, 224, 224)
        torch.testing.assert_close(got, want, atol=2e-5, rtol=0)
        masks = (torch.rand(2, H, W, generator=gen) > 0.4).float().to(dev)
        m_want = _center_crop(_resize_short_edge(masks[:, None], 256, "bilinear"), 224)
        m_got = resize_crop(masks[..., None], 256, 224, False)
        torch.testing.assert_close(m_got, m_want, atol=2e-6, rtol=0)
        g_want = _resize_short_edge(m_want, 16, "bilinear")
        g_got = resize_crop(m_got.permute(0, 2, 3, 1), 16, None, False)
        torch.testing.assert_close(g_got, g_want, atol=2e-6, rtol=0)
    # the front end with either preprocessing: same keep flags, tokens equal to the fp32 backbone's sensitivity
    net, grid, _ = create_standin_backbone(seed=5)
    net = net.to(dev)
    imgs = torch.rand(3, 800, 800, 3, generator=gen).to(dev)
    yy, xx = torch.meshgrid(torch.arange(800), torch.arange(800), indexing="ij")
    masks = (((yy - 380) ** 2 + (xx - 420) ** 2) <= 300 ** 2).float().to(dev)[None].expand(3, -1, -1).contiguous()
    ta, ka = ImageFrontEnd(net, grid, native_preprocess=True).tokens(imgs, masks)
    tb, kb = ImageFrontEnd(net, grid, native_preprocess=False).tokens(imgs, masks)
    assert torch.equal(ka, kb) and 50 < int(ka[0].sum()) < 256
    torch.testing.assert_close(ta, tb, atol=2e-4, rtol=0)
    with pytest.raises(RuntimeError):
        resize_crop(torch.rand(1, 4000, 4000, 3, device=dev), 256, 224, True)          # scale 15.6: beyond the kernel's 32 taps


@pytest.mark.gpu
def test_native_vit_other_token_counts(dev):
    """The attention kernel of the fp32 class passes the keys through LDS in four pieces (key blocks 0-4, then 5-8, of K and of V^T)
    and masks what lies beyond the T tokens of the grid: grids whose tokens end inside the first piece (8 x 8: T = 65), exactly at its
    end region (12 x 13: T = 157), inside the second (14 x 14: T = 197) and near the limit of 288 (16 x 17: T = 273) against the fp32
    torch module on the same weights, both precisions."""
    from iffnerf_amd.hip_vit import ViTHandle
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    net, _, _ = create_standin_backbone(seed=5)
    net = net.to(dev)
    gen = torch.Generator().manual_seed(4)
    for gh, gw in ((8, 8), (12, 13), (14, 14), (16, 17)):
        x = torch.randn(3, 3, 14 * gh, 14 * gw, generator=gen).to(dev)
        with torch.no_grad():
            want = net.forward_features(x)
        ref, scale = want["x_norm_patchtokens"], float(want["x_norm_patchtokens"].abs().max())
        tok, cls = ViTHandle(net.state_dict(), dev, grid=(gh, gw)).forward(x, want_cls=True)
        assert tok.shape == (3, gh * gw, 384) and torch.isfinite(tok).all()
        assert float((tok - ref).abs().max()) <= 1e-4 * scale, (gh, gw, float((tok - ref).abs().max()), scale)
        assert float((cls - want["x_norm_clstoken"]).abs().max()) <= 1e-4 * scale
        tok_b = ViTHandle(net.state_dict(), dev, grid=(gh, gw), precision="bf16").forward(x)
        assert torch.isfinite(tok_b).all() and float((tok_b - ref).abs().max()) <= 4e-2 * scale, (gh, gw)
