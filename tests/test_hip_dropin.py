"""GPU: the mirrored reference API end to end (module paths, classes and call order of train_eval_pose_est.py).

load_model -> explore_model -> IdentificationModule.test_image -> test_pose_estimation, with a fake backbone standing in
for DINOv2 exactly as the golden harness did when it ran the real reference (tests/golden/make_golden.py, G8).
"""
import numpy as np
import pytest
import torch

from iffnerf_amd import synthetic
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _install():
    import iffnerf_amd
    iffnerf_amd.install(force=True)


def test_reference_module_paths_run_on_the_gpu(golden, dev, tmp_path, monkeypatch):
    _install()
    from models.tensoRF import TensorVMSplit                       # noqa: F401  (the reference's import lines)
    from pose_estimation.model_utils import explore_model, load_model
    from pose_estimation import identification_module as im
    from pose_estimation.test import test_pose_estimation
    import renderer

    ck = util.ckpt("small")
    path = tmp_path / "tensorf_small_VM.th"
    torch.save(ck, str(path))
    model = load_model(str(path), dev)
    assert not any(p.requires_grad for p in model.parameters())

    # field API parity through the module (same checks as test_hip_field, one call each)
    g2 = golden["g2_march_point"]
    rays = golden.t("g2_march_point", "rays").to(dev)
    rgb, depth, acc, alpha, z_vals, dists = model(rays, N_samples=20, sample_func=model.sample_point_color)
    torch.testing.assert_close(rgb.cpu(), torch.from_numpy(g2["rgb"]), atol=2e-5, rtol=0)
    torch.testing.assert_close(alpha.cpu(), torch.from_numpy(g2["alpha"]), atol=1e-5, rtol=5e-6)
    assert torch.equal(z_vals.cpu(), torch.from_numpy(g2["z_vals"])) and torch.equal(dists.cpu(), torch.from_numpy(g2["dists"]))
    g10 = golden["g10_march_slab"]
    model.near_far = [float(v) for v in g10["near_far"]]
    model.invalidate_tables()
    rgb_s, _, depth_s, _, _ = renderer.OctreeRender_trilinear_fast(golden.t("g10_march_slab", "rays"), model, device=dev)
    torch.testing.assert_close(rgb_s.cpu(), torch.from_numpy(g10["rgb"]), atol=5e-5, rtol=0)
    torch.testing.assert_close(depth_s.cpu(), torch.from_numpy(g10["depth"]), atol=1e-4, rtol=0)
    x = golden.t("g5_emit", "samples").to(dev)
    from oracle import field as ofield
    want_alpha = ofield.compute_alpha(ofield.field_from_ckpt(ck), x.cpu())
    torch.testing.assert_close(model.compute_alpha(x).cpu(), want_alpha, atol=2e-6, rtol=2e-6)
    with pytest.raises(RuntimeError):
        model(rays, sample_func=lambda *a, **k: None)

    # stage A+B through explore_model: shapes, determinism under torch.manual_seed, colours consistent with the march
    torch.manual_seed(66)
    o, d, c = explore_model(model, gen_points=75)
    assert o.shape == d.shape == c.shape == (2025, 3) and o.is_cuda
    torch.manual_seed(66)
    o2, d2, c2 = explore_model(model, gen_points=75)
    assert torch.equal(o, o2) and torch.equal(d, d2) and torch.equal(c, c2)
    assert torch.equal(o.view(75, 27, 3)[:, 0], o.view(75, 27, 3)[:, 26])          # 27 rays share each origin
    assert torch.allclose(torch.linalg.norm(d, dim=-1), torch.ones(2025, device=dev), atol=1e-5)

    # stage C through the module, on the golden rays, with the golden harness's fake backbone
    tok8 = golden.t("g8_end_to_end", "tokens").to(dev)

    class FakeBackbone(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = 0

        def forward_features(self, x):
            self.calls += 1
            return {"x_norm_patchtokens": (tok8 * (1.0 + 0.05 * self.calls))[None]}

    monkeypatch.setattr(im, "create_backbone", lambda type="dino", pretrained=False, **k: (FakeBackbone(), (16, 16), 384))
    idm = im.IdentificationModule("dino")
    idm.load_state_dict({**idm.state_dict(), **synthetic.make_id_weights(seed=99)})
    idm = idm.to(dev).eval()
    # the golden harness used identity image transforms on 16x16 inputs; do the same here
    idm.transformations = lambda x: x
    idm.mask_transformations = lambda x: x

    class Dataset:
        pass

    g8 = golden["g8_end_to_end"]
    ds = Dataset()
    ds.all_rgbs = golden.t("g8_end_to_end", "imgs").clone()
    ds.K = torch.eye(3)[None]
    ds.all_rays = torch.zeros(2, 4, 6)
    ds.poses = golden.t("g8_end_to_end", "poses").clone()
    ro, rd, rc = (golden.t("g6_identify", k).to(dev) for k in ("ori", "dirs", "rgb"))
    res, te, ae, _, _ = test_pose_estimation(ds, idm, ro, rd, rc, golden.t("g8_end_to_end", "model_up").to(dev))
    pred = torch.tensor([r["pred_c2w"] for r in res])
    torch.testing.assert_close(pred, torch.from_numpy(g8["pred_c2w"]), atol=1e-4, rtol=0)
    assert abs(te - float(g8["avg_translation_error"])) < 1e-4 and abs(ae - float(g8["avg_angular_error"])) < 1e-2
    # inerf_refinement=True (reference test.py:196-211) runs the iNeRF loop on every image; 3 steps here, only the plumbing
    # (test_inerf_refinement_loop checks that it converges)
    import pose_estimation.test as pt
    monkeypatch.setattr(pt, "INERF_ITERS", 3)
    monkeypatch.setattr(pt, "INERF_BATCH", 128)                      # the fixture's images are 16 x 16
    ds.K = torch.tensor([[[20.0, 0.0, 8.0], [0.0, 20.0, 8.0], [0.0, 0.0, 1.0]]])
    with pytest.raises(RuntimeError, match="nerf_model"):
        test_pose_estimation(ds, idm, ro, rd, rc, golden.t("g8_end_to_end", "model_up").to(dev), inerf_refinement=True)
    res_i, _, _, _, _ = test_pose_estimation(ds, idm, ro, rd, rc, golden.t("g8_end_to_end", "model_up").to(dev),
                                             inerf_refinement=True, nerf_model=model)
    moved = torch.tensor([r["pred_c2w"] for r in res_i])
    assert torch.isfinite(moved).all() and 0 < float((moved - pred).abs().max()) < 0.2        # three lr=0.02 Adam steps
    # test_image returns the reference's 4-tuple; the attention map rows are softmaxes over the rays
    idx, val, scores, amap = idm.test_image(ds.all_rgbs[0, ..., :3].to(dev), ds.all_rgbs[0, ..., 3].to(dev), ro, rd, rc)
    assert idx.shape == (100,) and idx.dtype == torch.int64 and scores.shape == (2025,) and amap.shape[1] == 2025
    assert torch.allclose(amap.sum(-1), torch.ones(amap.shape[0], device=dev), atol=1e-4)
    assert torch.equal(scores[idx], val)
    # SURVEY 8(b) grad-mode contract: with trainable parameters under grad mode (pose_estimation/train.py:97-119) the module
    # evaluates the same formulas in differentiable torch ops on the GPU -- same scores as the HIP path, gradients that agree
    # with autograd through the oracle's op chain
    from oracle import identify as oid
    img, msk = ds.all_rgbs[0, ..., :3].to(dev), ds.all_rgbs[0, ..., 3].to(dev)
    fake = idm.image_preprocessing_net

    def same_tokens():               # the fake backbone scales its tokens by its call count: pin it for the comparisons below
        fake.calls = 10
    for prm in list(idm.attention.parameters()) + list(idm.ray_preprocessor.parameters()):
        prm.requires_grad_(True)
    torch.manual_seed(0)
    same_tokens()
    scores, amap_t, feats, used = idm(img, msk, ro, -rd, rc, rays_to_test=600)
    assert scores.requires_grad and amap_t.shape[1] == 600 and used.shape == (600,)
    wgt = torch.linspace(0.0, 1.0, 600, device=dev)
    (scores * wgt).sum().backward()
    grads = {n: p.grad.detach().cpu() for n, p in idm.named_parameters() if p.grad is not None}
    assert set(grads) == {f"ray_preprocessor.{k}" for k in idm.ray_preprocessor.state_dict()} | {f"attention.{k}" for k in idm.attention.state_dict()}
    wcpu = {k: v.clone().requires_grad_(True) for k, v in synthetic.make_id_weights(seed=99).items()}
    same_tokens()
    tokens_pe, _ = idm.image_processing(img, msk)
    u = used.cpu()
    att = oid.attention_map(wcpu, tokens_pe.detach().cpu(), oid.ray_encode(wcpu, ro.cpu()[u], -rd.cpu()[u], rc.cpu()[u]))
    (att.sum(0) * wgt.cpu()).sum().backward()
    torch.testing.assert_close(scores.detach().cpu(), att.sum(0).detach(), atol=1e-6, rtol=2e-4)
    gscale = max(float(v.grad.abs().max()) for v in wcpu.values())
    for k, gcpu in ((k, v.grad) for k, v in wcpu.items()):
        # (k_proj.bias has an analytically zero gradient -- a per-row constant cancels in the softmax -- hence the absolute term)
        torch.testing.assert_close(grads[k], gcpu, atol=2e-5 * float(gcpu.abs().max()) + 2e-5 * gscale, rtol=1e-3)
    # and the two formulations agree on the forward (HIP under no_grad vs torch ops under grad) on the full ray set
    for prm in idm.parameters():
        prm.grad = None
    same_tokens()
    s_torch = idm.run_attention(img, msk, ro, -rd, rc)[0]
    assert s_torch.requires_grad
    with torch.no_grad():
        same_tokens()
        s_hip2 = idm.run_attention(img, msk, ro, -rd, rc)[0]
    torch.testing.assert_close(s_torch.detach(), s_hip2, atol=1e-7, rtol=3e-4)
    # an optimizer step between two validations (pose_estimation/train.py:126,145,188,222): the parameters change IN PLACE, no
    # _apply / load_state_dict in between -- the no-grad (HIP) path must follow the new weights, not a stale kernel handle
    handle_before = idm._idnet()
    same_tokens()
    (idm.run_attention(img, msk, ro, -rd, rc)[0] * torch.linspace(0.0, 1.0, ro.shape[0], device=dev)).sum().backward()
    with torch.no_grad():                   # what optimizer.step() does: in-place updates (here 1 % of each tensor's scale)
        for p_ in idm.parameters():
            if p_.grad is not None:
                p_.add_(p_.grad, alpha=-0.01 * float(p_.abs().max()) / (float(p_.grad.abs().max()) + 1e-30))
    with torch.no_grad():
        same_tokens()
        s_hip3 = idm.run_attention(img, msk, ro, -rd, rc)[0]
    assert idm._idnet() is not handle_before, "the kernel handle survived an in-place parameter update"
    same_tokens()
    s_torch3 = idm.run_attention(img, msk, ro, -rd, rc)[0]
    assert float((s_hip3 - s_hip2).abs().max()) > 1e-5                       # the step did move the scores
    torch.testing.assert_close(s_torch3.detach(), s_hip3, atol=1e-7, rtol=3e-4)
    # CPU tensors are refused on the training branch too
    with pytest.raises(RuntimeError, match="no CPU path"):
        idm.ray_preprocessor(ro.cpu(), rd.cpu(), rc.cpu())


def test_captured_query_replays_like_eager(dev):
    """hipGraph capture of the cold query: every replay equals the eager run with the same effective seed."""
    from iffnerf_amd.pipeline import PosePipeline
    ck = util.ckpt("small")
    pipe = PosePipeline.from_checkpoints(ck, synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))
    tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
    eager = {s: [t.clone() for t in pipe.query(tok, 75, seed=100 + s, k=100)] for s in range(1, 5)}
    cq = pipe.capture_query(tok.shape, 75, seed=100, k=100)
    cq.tokens.copy_(tok)
    torch.cuda.synchronize()
    for s in range(1, 5):
        cq.replay()
        torch.cuda.synchronize()
        assert int(cq.counter.item()) == s
        assert torch.equal(cq.c2w, eager[s][0]) and torch.equal(cq.idx, eager[s][1]) and torch.equal(cq.val, eager[s][2])
    # two captured queries on two streams do not disturb each other
    cq2 = pipe.capture_query(tok.shape, 75, seed=100, k=100)
    cq2.tokens.copy_(tok)
    cq.counter.zero_(); cq2.counter.fill_(2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.cuda.stream(s1):
        cq.replay()
    with torch.cuda.stream(s2):
        cq2.replay()
    torch.cuda.synchronize()
    assert torch.equal(cq.c2w, eager[1][0]) and torch.equal(cq2.c2w, eager[3][0])


@pytest.mark.parametrize("config,rounds", [("truck32k", 40), ("bicycle64k", 15)])
def test_concurrent_captured_steps_reproduce_the_eager_path(dev, config, rounds):
    """Four captured cold steps replayed concurrently on four streams (how bench.py and a serving loop run them) at a BASELINE
    config's full size: every replay equals the eager path on the same seed counter bit for bit.  This is the case in which the fan
    kernel's packed-fp32 tap combination returned wrong sums for sixteen lanes of a wave once in ~30 launches while a workgroup of
    the trunk kernel shared the CU (fan_march_kernels.hip, lerp_plane_q): at that rate the 160 truck32k checks below hold 3-6 events.
    bicycle64k runs the general march kernels (K4a / K4b / the Ref head launch) next to the trunk."""
    from iffnerf_amd.pipeline import PosePipeline, CapturedBatchQuery
    wl = synthetic.WORKLOADS[config]
    pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(config), synthetic.make_id_weights(seed=99), dev)
    B, P = wl["queries"], wl["gen_points"]
    tokens = torch.stack([synthetic.make_tokens(256, 384, seed=7 + q) for q in range(B)]).to(dev)
    seeds = [1000 + 7919 * i for i in range(4)]
    graphs = [CapturedBatchQuery(pipe, tokens.shape, P, seed=seeds[i], k=100) for i in range(4)]
    for g in graphs:
        g.tokens.copy_(tokens)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    torch.cuda.synchronize(dev)
    for r in range(rounds):
        for _ in range(3):
            for i, g in enumerate(graphs):
                with torch.cuda.stream(streams[i]):
                    g.replay()
        torch.cuda.synchronize(dev)
        for i, g in enumerate(graphs):
            g.check()
            c2w, idx, val = pipe.query_batch(tokens, P, seeds[i], 100, seed_offset=g.counter)
            assert torch.equal(idx, g.idx) and torch.equal(val, g.val) and torch.equal(c2w, g.c2w), (r, i, float((val - g.val).abs().max()))


@pytest.mark.parametrize("config,rounds", [("truck32k", 150), ("lego16k", 100), ("bicycle64k", 40), ("lego540k", 20)])
def test_concurrent_stage_outputs_reproduce_the_eager_kernels(dev, config, rounds):
    """The stage-level form of the check above (what found the packed-fp32 fault: scripts/replay_vs_eager_stages.py ONLY=trunk): four
    captured graphs in flight, each the cold emission of a batch (sampler, normals, emit, the fan march) FOLLOWED BY the trunk launch
    on static rays, so that the tail of every march runs next to another graph's fp16-MFMA workgroups; after every round each graph's
    ray colours, depth, opacity and its logits are recomputed eagerly from the graph's OWN surface samples and compared bit for bit.
    A final pose or a top-100 list can hide a colour that is off by 1e-3 -- this cannot.  600 / 400 steps: the compiler's packed
    fp32 build showed 14 events per 2 000.  Every workload the bench can run is soaked: bicycle64k marches on the eight-wave fan kernel
    (patches by global -> LDS DMA: a transfer still in flight when a buffer is read would show here), lego540k at the reference's
    default 540 000 rays per query."""
    from iffnerf_amd.hip_field import isocell_emit
    from iffnerf_amd.pipeline import PosePipeline
    wl = synthetic.WORKLOADS[config]
    pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(config), synthetic.make_id_weights(seed=99), dev)
    B, P = wl["queries"], wl["gen_points"]
    tokens = torch.stack([synthetic.make_tokens(256, 384, seed=7 + q) for q in range(B)]).to(dev)
    M, C = tokens.shape[1:]
    o0, d0, c0 = pipe.emit(P, seed=5)
    static = [x.repeat(B, 1).contiguous() for x in (o0, d0, c0)]
    qf = pipe.idnet.q_fold(tokens.reshape(B * M, C))

    def emit(samples):
        normals = pipe.field.point_normals(samples)
        ori, dirs, rays = isocell_emit(pipe.cells, samples, normals, want_rays6=True)
        rgb, depth, acc = pipe.field.march(rays, 0, 20, want_alpha=False)[:3]
        return {"ori": ori, "dirs": dirs, "rgb": rgb, "depth": depth, "acc": acc}

    def step(seed, counter):
        samples, _, _ = pipe.field.surface_sample_batched(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=seed, seed_offset=counter)
        out = emit(samples.reshape(B * P, 3))
        out["samples"] = samples
        out["logits"] = pipe.idnet.ray_logits_folded_batched(qf, static[0], static[1], static[2], B)[0]
        return out

    logits_ref = pipe.idnet.ray_logits_folded_batched(qf, static[0], static[1], static[2], B)[0].clone()
    graphs = []
    for i in range(4):
        counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                step(1000 + 7919 * i, counter)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            counter += 1
            out = step(1000 + 7919 * i, counter)
        graphs.append((g, out))
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    torch.cuda.synchronize(dev)
    for r in range(rounds):
        for _ in range(3):
            for i, (g, _) in enumerate(graphs):
                with torch.cuda.stream(streams[i]):
                    g.replay()
        torch.cuda.synchronize(dev)
        for i, (_, out) in enumerate(graphs):
            eager = emit(out["samples"].reshape(B * P, 3))
            for k, v in eager.items():
                assert torch.equal(v.nan_to_num(7.0), out[k].nan_to_num(7.0)), (r, i, k, int((v != out[k]).sum()), float((v - out[k]).abs().nan_to_num(0.0).max()))
            assert torch.equal(out["logits"], logits_ref), (r, i, "logits")


def test_concurrent_warm_and_image_graphs_reproduce_the_eager_path(dev):
    """The other two graph forms of the bench -- a batch of query images against resident rays, and image in -> pose out (resize /
    crop, native ViT-S/14, token assembly, stage C) -- replayed four at a time: every replay equals the same call run eagerly."""
    from iffnerf_amd.hip_vit import serve_natively
    from iffnerf_amd.image_frontend import ImageFrontEnd
    from iffnerf_amd.pipeline import PosePipeline, CapturedImageQuery
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    wl = synthetic.WORKLOADS["lego16k"]
    pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt("lego16k"), synthetic.make_id_weights(seed=99), dev)
    ori, dirs, rgb = pipe.emit(wl["gen_points"], seed=42)
    resident = pipe.make_resident(ori, dirs, rgb)
    net, grid, _ = create_standin_backbone(seed=0)
    fe = ImageFrontEnd(serve_natively(net.to(dev), grid), grid)
    gen = torch.Generator().manual_seed(3)
    Q, forms = 16, []
    for i in range(4):                                           # warm: tokens in
        tok = torch.stack([synthetic.make_tokens(256, 384, seed=100 * i + q) for q in range(Q)]).to(dev)
        for _ in range(2):
            pipe.identify_resident(tok, resident, 100)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = pipe.identify_resident(tok, resident, 100)
        forms.append((g, out, lambda tok=tok: pipe.identify_resident(tok, resident, 100), tok))
    for i in range(4):                                           # image in -> pose out
        imgs = torch.rand(Q, 800, 800, 3, generator=gen).to(dev)
        masks = (torch.rand(Q, 800, 800, generator=gen) > 0.2).float().to(dev)
        cq = CapturedImageQuery(pipe, fe, imgs.shape, resident, 100)
        cq.imgs.copy_(imgs), cq.masks.copy_(masks)
        forms.append((cq.graph, (cq.c2w, cq.idx, cq.val), lambda imgs=imgs, masks=masks: pipe.identify_images_resident(fe, imgs, masks, resident, 100), cq))
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    torch.cuda.synchronize(dev)
    for group in (forms[:4], forms[4:], forms[2:6]):             # four warm, four image, two of each together
        for r in range(8):
            for _ in range(3):
                for i, f in enumerate(group):
                    with torch.cuda.stream(streams[i]):
                        f[0].replay()
            torch.cuda.synchronize(dev)
            for i, f in enumerate(group):
                for a, b in zip(f[2](), f[1]):
                    assert torch.equal(a, b), (r, i)


def test_batched_cold_queries_equal_single_queries(dev):
    """query_batch: B cold queries per set of launches (batched sampler, grid.y = query in the encoder/logits launch).
    Query b must equal the single-query path with seed + b * SAMPLER_SEED_STRIDE bit for bit; also as a captured graph."""
    from iffnerf_amd.hip_field import SAMPLER_SEED_STRIDE
    from iffnerf_amd.pipeline import PosePipeline
    ck = util.ckpt("small")
    pipe = PosePipeline.from_checkpoints(ck, synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))
    for M in (256, 137):
        tok = torch.stack([synthetic.make_tokens(M, 384, seed=7 + b) for b in range(3)]).to(dev)
        c2w, idx, val = pipe.query_batch(tok, 75, seed=900, k=100)
        assert c2w.shape == (3, 4, 4) and idx.shape == (3, 100)
        for b in range(3):
            w_c2w, w_idx, w_val = pipe.query(tok[b], 75, seed=(900 + b * SAMPLER_SEED_STRIDE) % 2 ** 64, k=100)
            assert torch.equal(idx[b], w_idx) and torch.equal(val[b], w_val) and torch.equal(c2w[b], w_c2w), (M, b)
    assert not torch.equal(idx[0], idx[1])                      # different draws, different tokens
    tok = torch.stack([synthetic.make_tokens(256, 384, seed=7 + b) for b in range(4)]).to(dev)
    eager = {r: [t.clone() for t in pipe.query_batch(tok, 75, seed=900 + r, k=100)] for r in (1, 2)}
    cq = pipe.capture_query_batch(tok.shape, 75, seed=900, k=100)
    cq.tokens.copy_(tok)
    torch.cuda.synchronize()
    for r in (1, 2):
        cq.replay()
        torch.cuda.synchronize()
        assert torch.equal(cq.idx, eager[r][1]) and torch.equal(cq.c2w, eager[r][0])


def test_warm_batch_equals_per_image_identification(dev):
    """identify_batch: Q query images against one resident ray set in one set of launches == identify per image."""
    from iffnerf_amd.pipeline import PosePipeline
    pipe = PosePipeline.from_checkpoints(util.ckpt("small"), synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))
    ori, dirs, rgb = pipe.emit(75, seed=3)
    for M in (256, 137):
        tok = torch.stack([synthetic.make_tokens(M, 384, seed=20 + q) for q in range(5)]).to(dev)
        c2w, idx, val = pipe.identify_batch(tok, ori, dirs, rgb, 100)
        for q in range(5):
            w_c2w, w_idx, w_val = pipe.identify(tok[q], ori, dirs, rgb, 100, materialize_map=False)
            assert torch.equal(idx[q], w_idx) and torch.equal(val[q], w_val) and torch.equal(c2w[q], w_c2w), (M, q)


def test_resident_rays_cache_in_every_gemm_mode(dev):
    """The per-model encoder cache (iff_ray_cache_build / iff_logits_from_cache): bit-identical to the uncached fused call
    under F16X2; under the other arithmetic modes the cache is the fp32 activation and the logits product a separate launch --
    same scores to fp32 rounding, same top-100."""
    from iffnerf_amd import hip_identify as H
    from iffnerf_amd.pipeline import PosePipeline
    w = synthetic.make_id_weights(seed=99)
    tok = torch.stack([synthetic.make_tokens(137, 384, seed=60 + q) for q in range(3)]).to(dev)
    for mode in (H.GEMM_F16X2, H.GEMM_BF16X3, H.GEMM_F32):
        pipe = PosePipeline.from_checkpoints(util.ckpt("small"), w, dev, model_up=(0.1, 0.2, 0.9), gemm_mode=mode)
        ori, dirs, rgb = pipe.emit(75, seed=3)
        rays = pipe.make_resident(ori, dirs, rgb)
        c2w, idx, val = pipe.identify_resident(tok, rays, 100)
        for q in range(3):
            w_c2w, w_idx, w_val = pipe.identify(tok[q], ori, dirs, rgb, 100, materialize_map=False)
            if mode == H.GEMM_F16X2:
                assert torch.equal(idx[q], w_idx) and torch.equal(val[q], w_val) and torch.equal(c2w[q], w_c2w), q
            else:
                assert idx[q].tolist() == w_idx.tolist(), (mode, q)
                torch.testing.assert_close(val[q], w_val, atol=1e-7, rtol=1e-4)
                torch.testing.assert_close(c2w[q], w_c2w, atol=1e-5, rtol=0)
    # empty batch of rays: nothing to cache, nothing crashes
    assert pipe.idnet.build_ray_cache(ori[:0], dirs[:0], rgb[:0]).numel() >= 0


def test_evaluation_loop_over_the_slab_march(dev):
    """renderer.evaluation (renderer.py:28-140) on a duck-typed dataset of pinhole views: PSNR list, view subsampling by
    N_vis, RGBA ground truth on white, chunk-size independence -- against the oracle's slab march of the same rays."""
    import types
    from iffnerf_amd import renderer as R
    from iffnerf_amd.pose_estimation import model_utils as mu
    from oracle import field as ofield
    ck = dict(util.ckpt("small"))
    ck["kwargs"] = dict(ck["kwargs"], near_far=[0.5, 6.0])
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.th")
        torch.save(ck, path)
        model = mu.load_model(path, dev)
    f = ofield.field_from_ckpt(ck)
    W, H, n = 20, 24, 4
    gen = torch.Generator().manual_seed(3)
    views = []
    for v in range(n):
        ang = 2 * 3.14159 * v / n
        o = torch.tensor([2.6 * torch.cos(torch.tensor(ang)), 2.6 * torch.sin(torch.tensor(ang)), 0.4 + 0.2 * v])
        fwd = -o / o.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])); right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        ys, xs = torch.meshgrid(torch.linspace(0.45, -0.45, H), torch.linspace(-0.4, 0.4, W), indexing="ij")
        d = fwd[None, None] + xs[..., None] * right + ys[..., None] * up
        d = d / d.norm(dim=-1, keepdim=True)
        views.append(torch.cat((o.expand(H, W, 3), d), -1).reshape(H * W, 6))
    all_rays = torch.stack(views)
    want = torch.stack([ofield.march(f, r, "slab", -1, white_bg=True)[0] for r in all_rays])       # [n, H*W, 3]
    assert float(want.std()) > 0.05                                                                   # the views do see the object
    alpha = (torch.rand(n, H, W, 1, generator=gen) > 0.3).float()
    noise = 0.02 * torch.randn(n, H, W, 3, generator=gen)
    rgba = torch.cat(((want.view(n, H, W, 3) + noise).clamp(0, 1), alpha), -1)
    ds = types.SimpleNamespace(all_rays=all_rays, all_rgbs=rgba, img_wh=(W, H), near_far=[0.5, 6.0])
    args = types.SimpleNamespace(test_batch_size=-1, batch_size=4096, n_iters=1)
    psnrs = R.evaluation(ds, model, args, R.OctreeRender_trilinear_fast, N_vis=-1, white_bg=True, compute_extra_metrics=False, device=dev)
    assert len(psnrs) == n
    for v in range(n):
        gt = (rgba[v, ..., :3] * rgba[v, ..., 3:] + (1 - rgba[v, ..., 3:])).clamp(0, 1)
        ref = -10.0 * torch.log10(torch.mean((want[v].view(H, W, 3).clamp(0, 1) - gt) ** 2))
        assert abs(psnrs[v] - float(ref)) < 1e-2, (v, psnrs[v], float(ref))
    two, info = R.evaluation(ds, model, args, R.OctreeRender_trilinear_fast, N_vis=2, white_bg=True, compute_extra_metrics=False,
                             device=dev, return_result=True)
    assert len(two) == 2 and abs(two[1] - psnrs[2]) < 1e-9 and abs(info["avg_psnr"] - sum(two) / 2) < 1e-9
    # the renderer alone: pieces of any size give the same image
    old = R.MAX_RAYS_PER_LAUNCH
    try:
        a = R.OctreeRender_trilinear_fast(all_rays[0], model, white_bg=True, device=dev)[0]
        R.MAX_RAYS_PER_LAUNCH = 100
        b = R.OctreeRender_trilinear_fast(all_rays[0], model, chunk=7, white_bg=True, device=dev)[0]
    finally:
        R.MAX_RAYS_PER_LAUNCH = old
    assert torch.equal(a, b)
    torch.testing.assert_close(a.cpu(), want[0], atol=5e-5, rtol=0)
    with pytest.raises(RuntimeError, match="SSIM"):
        R.evaluation(ds, model, args, R.OctreeRender_trilinear_fast, device=dev)


def _load_small(dev, near_far):
    import os, tempfile
    from iffnerf_amd.pose_estimation import model_utils as mu
    ck = dict(util.ckpt("small"))
    ck["kwargs"] = dict(ck["kwargs"], near_far=list(near_far))
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.th")
        torch.save(ck, path)
        return mu.load_model(path, dev), ck


@pytest.mark.gpu
def test_ray_gradients_match_the_reference(golden, dev):
    """`model(rays_chunk, ...)` with rays that require grad (inerf/estimate_pose_inerf.py:164-176).  The forward
    is the HIP march, the backward `iff_march_grad`; d loss / d rays against the gradients autograd gave on the reference
    itself (fixture G12, slab and point-centred samplers).  fp32 with another summation order: 2e-5 of the largest
    gradient component, and rgb / acc to 1e-5."""
    g = golden["g12_march_grad"]
    model, _ = _load_small(dev, [float(v) for v in g["near_far"]])
    for tag in ("slab", "point"):
        rays = golden.t("g12_march_grad", f"{tag}_rays").to(dev).requires_grad_(True)
        kw = {} if tag == "slab" else dict(N_samples=20, sample_func=model.sample_point_color)
        rgb, depth, acc, alpha, z, dists = model(rays, bg_color=golden.t("g12_march_grad", "bg").to(dev), is_train=False, **kw)
        assert alpha is None and rgb.requires_grad and acc.requires_grad and not depth.requires_grad
        torch.testing.assert_close(rgb.detach().cpu(), golden.t("g12_march_grad", f"{tag}_rgb"), rtol=0, atol=1e-5)
        torch.testing.assert_close(acc.detach().cpu(), golden.t("g12_march_grad", f"{tag}_acc"), rtol=0, atol=1e-5)
        loss = (rgb * golden.t("g12_march_grad", f"{tag}_c_rgb").to(dev)).sum() + \
               (acc * golden.t("g12_march_grad", f"{tag}_c_acc").to(dev)).sum()
        (grad,) = torch.autograd.grad(loss, rays)
        want = golden.t("g12_march_grad", f"{tag}_grad")
        err = (grad.cpu() - want).abs().max() / want.abs().max()
        assert grad.shape == want.shape and float(err) < 2e-5, (tag, float(err))
        print(f"ray gradient vs reference ({tag}): max |d| / max |g| = {float(err):.2e}")
        # the same call without grad is the inference kernel path and returns the same picture
        with torch.no_grad():
            rgb0 = model(rays.detach(), bg_color=golden.t("g12_march_grad", "bg").to(dev), **kw)[0]
        torch.testing.assert_close(rgb0, rgb.detach(), rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_pose_refinement_descends_through_the_march(dev):
    """The shape of inerf/estimate_pose_inerf.py:104-176: a camera translation/rotation offset optimised with Adam through
    get-rays -> model(rays) -> MSE.  Target = the model's own render from the true pose; the loss must fall by 10x and the
    translation come back, using nothing but the HIP forward/backward pair for the march."""
    from oracle import field as ofield
    model, ck = _load_small(dev, [0.05, 6.0])
    gen = torch.Generator().manual_seed(11)
    cam = torch.tensor([2.4, 0.6, 0.9])
    fwd = -cam / cam.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])); right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    uv = (torch.rand(384, 2, generator=gen) - 0.5) * 0.7
    local = torch.nn.functional.normalize(torch.cat((uv, torch.ones(384, 1)), -1), dim=-1)          # camera-frame directions
    R0 = torch.stack((right, up, fwd), -1).to(dev)                                                     # camera -> world
    local, cam = local.to(dev), cam.to(dev)

    def rays_of(t, w):
        # small-angle rotation exp([w]x) ~ I + [w]x + [w]x^2/2 is enough for the 0.02 rad offsets used here
        K = torch.zeros(3, 3, device=dev)
        K = K.index_put((torch.tensor([0, 0, 1, 1, 2, 2]), torch.tensor([1, 2, 0, 2, 0, 1])),
                        torch.stack((-w[2], w[1], w[2], -w[0], -w[1], w[0])))
        Rw = (torch.eye(3, device=dev) + K + 0.5 * K @ K) @ R0
        d = torch.nn.functional.normalize(local @ Rw.T, dim=-1)
        return torch.cat(((cam + t).expand(384, 3), d, torch.full((384, 1), 1e-3, device=dev)), -1)

    with torch.no_grad():
        target = model(rays_of(torch.zeros(3, device=dev), torch.zeros(3, device=dev)))[0]
    assert float(target.std()) > 0.05
    t = torch.tensor([0.06, -0.05, 0.04], device=dev, requires_grad=True)
    w = torch.tensor([0.015, -0.02, 0.01], device=dev, requires_grad=True)
    opt = torch.optim.Adam([t, w], lr=4e-3)
    losses = []
    for _ in range(150):
        opt.zero_grad()
        rgb, _, opacity, _, _, _ = model(rays_of(t, w), is_train=False)
        loss = torch.mean((rgb - target) ** 2)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.1 * losses[0], (losses[0], losses[-1])
    # translation and small rotations trade off against each other in 384 rays; the offset must shrink, not vanish
    assert float(t.detach().norm()) < 0.6 * 0.0877, t


@pytest.mark.gpu
def test_inerf_refinement_loop(dev):
    """inerf.estimate_pose_inerf.pose_estimation (reference :23-195, 'random' pixel batches, Adam on the se(3) offset, MSE +
    soft Dice on the opacity) over the HIP march forward/backward: from a pose 0.1 off, the refined pose is several times
    closer to the pose the observation was rendered from."""
    from iffnerf_amd.inerf.estimate_pose_inerf import pose_estimation
    from iffnerf_amd import ray_utils as ru
    model, _ = _load_small(dev, [0.05, 6.0])
    H = W = 40
    K = torch.tensor([[60.0, 0.0, W / 2], [0.0, 60.0, H / 2], [0.0, 0.0, 1.0]], device=dev)
    cam = torch.tensor([2.3, 0.8, 1.0])
    fwd = -cam / cam.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0])); right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, cam
    c2w = c2w.to(dev)
    d, dx, dy = ru.get_ray_directions_Ks(H, W, K[None])
    unit = d / d.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        ro, rd, rad = ru.get_rays(unit, c2w, directions=d, dx=dx, dy=dy, keepdim=True)
        rays = torch.cat((ro[0], torch.nn.functional.normalize(rd[0], dim=-1), rad[0]), -1).reshape(-1, 7)
        rgb, _, acc, _, _, _ = model(rays, white_bg=False)
    assert 0.15 < float((acc > 0.5).float().mean()) < 0.95                 # the object covers part of the frame
    rgba = torch.cat((rgb, (acc > 0.5).float()[:, None]), -1).reshape(H, W, 4).cpu().numpy()
    start = c2w.clone()
    start[:3, 3] += torch.tensor([0.07, -0.06, 0.05], device=dev)
    np.random.seed(3)
    torch.manual_seed(3)
    import time
    from iffnerf_amd.inerf import estimate_pose_inerf as E
    runs = {}
    for mode, after in (("captured", 3), ("eager", None)):
        E.CAPTURE_AFTER = after
        try:
            np.random.seed(3)
            torch.manual_seed(3)
            t0 = time.time()
            runs[mode] = pose_estimation(start, rgba, K, model, sampling_strategy="random", batch_size=512, n_iters=250,
                                         lrate=0.01, color_bkgd_aug="random", dice_loss=True, print_progress=False, device=dev)
            print(f"iNeRF loop ({mode}): {250 / (time.time() - t0):.0f} iterations/s at 512 rays x {model.nSamples} samples")
        finally:
            E.CAPTURE_AFTER = 3
    loss, refined, trace = runs["captured"]
    # the captured iteration is the eager one (same kernels, same random streams): the two trajectories agree to rounding
    # (atomics-free kernels; Adam's capturable step keeps its counters on the device in both)
    assert float((refined - runs["eager"][1]).abs().max()) < 1e-4 and abs(loss - runs["eager"][0]) < 1e-5
    assert refined.shape == (4, 4) and refined.device.type == "cpu" and len(trace) == 250
    before = float((start[:3, 3] - c2w[:3, 3]).norm())
    after = float((refined[:3, 3] - c2w[:3, 3].cpu()).norm())
    print(f"iNeRF refinement: translation error {before:.4f} -> {after:.4f}, final rgb loss {loss:.2e}")
    assert after < 0.7 * before, (before, after)            # 40 x 40 pixels leave a translation / rotation trade-off
    with pytest.raises(RuntimeError, match="OpenCV"):
        pose_estimation(start, rgba, K, model, device=dev)
