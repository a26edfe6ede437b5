"""GPU: the evaluation loop behind the reference's own call signatures ("in the caller").

``test_pose_estimation(dataset, id_module, rays_ori, rays_dirs, rays_rgb, model_up)`` exactly as train_eval_pose_est.py:131-149
calls it: batches of images through one set of launches (captured hipGraphs, the tail batch padded) must return, bit for bit, what the
image-by-image route (``IdentificationModule.test_image`` + pose solve + error metrics per image, reference test.py:66-247) returns.
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


class _Dataset:
    pass


def _module(dev, monkeypatch, seed=5):
    from iffnerf_amd.pose_estimation import backbone as bb, identification_module as im
    net, grid, C = bb.create_standin_backbone(seed=seed, native=True)
    monkeypatch.setattr(im, "create_backbone", lambda **kw: (net, grid, C))
    mod = im.IdentificationModule("dino")
    mod.load_state_dict(synthetic.make_id_weights(seed=99), strict=False)
    return mod.to(dev).eval()


def _dataset(n, H, W, C, seed):
    gen = torch.Generator().manual_seed(seed)
    ds = _Dataset()
    rgb = torch.rand(n, H, W, 3, generator=gen)
    if C == 4:
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        alpha = torch.stack([((((yy - H / 2) / (0.30 * H + 3 * (q % 5))) ** 2 + ((xx - W / 2) / (0.42 * W - 4 * (q % 7))) ** 2) <= 1).float()
                             for q in range(n)])
        alpha = (alpha * (0.25 + 0.75 * torch.rand(n, H, W, generator=gen))).clamp(0, 1)      # soft alpha: the composite really blends
        rgb = torch.cat((rgb, alpha[..., None]), dim=-1)
    ds.all_rgbs = rgb
    ds.K = torch.eye(3)[None]
    ds.all_rays = torch.zeros(n, 4, 6)
    poses = torch.eye(4).repeat(n, 1, 1)
    poses[:, :3, :3] = torch.linalg.qr(torch.randn(n, 3, 3, generator=gen)).Q
    poses[:, :3, 3] = torch.randn(n, 3, generator=gen)
    ds.poses = poses
    return ds


def _rays(dev):
    from iffnerf_amd.pipeline import PosePipeline
    pipe = PosePipeline.from_checkpoints(util.ckpt("small"), synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))
    return pipe.emit(75, seed=3)


def test_batched_eval_loop_equals_image_by_image(dev, monkeypatch, capsys):
    """11 RGBA images, EVAL_BATCH = 4: two captured batches on two alternating graphs, a tail batch of 3 padded to 4 through the first graph again -- against
    the same call forced image by image.  Every field of every result record is EQUAL."""
    import iffnerf_amd.pose_estimation.test as pt
    mod = _module(dev, monkeypatch)
    ro, rd, rc = _rays(dev)
    up = torch.tensor([0.1, 0.2, 0.9], device=dev)
    monkeypatch.setattr(pt, "EVAL_BATCH", 4)
    for C, n in ((4, 11), (3, 5)):
        ds = _dataset(n, 200, 260, C, seed=40 + C)
        assert pt._batchable(ds, mod, ro, False)
        res_b, te_b, ae_b, _, _ = pt.test_pose_estimation(ds, mod, ro, rd, rc, up, sequence_id="s")
        sess = mod._ray_session
        assert sess is not None and len(sess.graphs) >= 1
        res_b2 = pt.test_pose_estimation(ds, mod, ro, rd, rc, up, sequence_id="s")[0]          # graphs and cache reused
        assert mod._ray_session is sess and res_b2 == res_b
        with monkeypatch.context() as m:
            m.setattr(pt, "_batchable", lambda *a, **k: False)
            res_i, te_i, ae_i, _, _ = pt.test_pose_estimation(ds, mod, ro, rd, rc, up, sequence_id="s")
        assert mod._ray_session is sess                       # the image-by-image route uses the same cached encoder
        assert len(res_b) == len(res_i) == n
        for a, b in zip(res_b, res_i):
            assert a.keys() == b.keys()
            for key in a:
                va, vb = np.asarray(a[key]), np.asarray(b[key])
                if va.dtype.kind == "f":
                    assert np.array_equal(va, vb, equal_nan=True), (C, a["frame_id"], key, va, vb)
                else:
                    assert a[key] == b[key], (C, a["frame_id"], key)
        assert te_b == te_i and ae_b == ae_i
        assert all(np.isfinite(r["loss"]) and 0 < r["loss"] < 1 for r in res_b)
        assert len({tuple(np.asarray(r["pred_c2w"]).ravel().tolist()) for r in res_b}) == n       # the images really differ
    # ... and test_image's own outputs are those of the batch's launches
    ds = _dataset(6, 200, 260, 4, seed=44)
    sess = mod.ray_session(ro, rd, rc)
    imgs = ds.all_rgbs.to(dev)
    from iffnerf_amd import hip_identify as H
    tokens, keep, rows = mod.static_tokens(imgs, None, compact=True)           # kept rows first: what the loop and test_image run
    score, _ = mod.scores_static(tokens, keep, sess, want_map=False, rows=rows)
    idx_b, val_b = H.topk_batched(score, 100)
    for q in range(6):
        obs = imgs[q]
        i1, v1, s1, _ = mod.test_image(obs[..., :3] * obs[..., -1:] + (1 - obs[..., -1:]), obs[..., -1], ro, rd, rc)
        assert torch.equal(i1, idx_b[q]) and torch.equal(v1, val_b[q]) and torch.equal(s1, score[q]), q
    assert 30 < int(keep[0].sum()) == int(rows[0]) < 256
    # the route without the compaction (every row computed, dropped rows masked on the statistics) sums the same terms in another order
    tokens_p, keep_p = mod.static_tokens(imgs, None)
    score_p, _ = mod.scores_static(tokens_p, keep_p, sess, want_map=False)
    torch.testing.assert_close(score, score_p, atol=1e-9, rtol=2e-6)


def test_attention_map_is_lazy_and_right(dev, monkeypatch):
    """test_image returns the reference's 4-tuple; the [M,N] map is not computed unless read, and when read it is the softmax over
    the rays of the selected token rows (identification_module.py:157-166): rows sum to 1, column sums are the scores."""
    from iffnerf_amd.pose_estimation.identification_module import LazyAttentionMap
    mod = _module(dev, monkeypatch)
    ro, rd, rc = _rays(dev)
    ds = _dataset(1, 200, 260, 4, seed=7)
    obs = ds.all_rgbs[0].to(dev)
    img, mask = obs[..., :3] * obs[..., -1:] + (1 - obs[..., -1:]), obs[..., -1]
    idx, val, scores, amap = mod.test_image(img, mask, ro, rd, rc, rays_to_output=100)
    assert isinstance(amap, LazyAttentionMap) and not amap.is_materialized
    assert idx.shape == (100,) and idx.dtype == torch.int64 and scores.shape == (ro.shape[0],) and torch.equal(scores[idx], val)
    M = amap.shape[0]                                    # the shape (all pose_estimation/test.py:121 reads of the map) costs a 4-byte read
    assert not amap.is_materialized and amap.shape == (M, ro.shape[0]) and 30 < M < 256
    torch.testing.assert_close(amap.sum(-1), torch.ones(M, device=dev), atol=1e-4, rtol=0)      # first use as a tensor: computed now
    assert amap.is_materialized and tuple(amap.materialize().shape) == (M, ro.shape[0])
    torch.testing.assert_close(torch.sum(amap, dim=0), scores, atol=1e-6, rtol=1e-4)            # a torch function on the lazy object
    assert torch.is_tensor(amap * 2.0) and torch.is_tensor(amap[0]) and len(amap) == M
    # the compacting route (run_attention: boolean row select before the logits, as the reference) gives the same map
    score_c, amap_c, _ = mod.run_attention(img, mask, ro, rd, rc)
    assert amap_c.shape == (M, ro.shape[0])
    torch.testing.assert_close(amap.materialize(), amap_c, atol=1e-7, rtol=2e-4)
    torch.testing.assert_close(scores, score_c, atol=1e-7, rtol=2e-4)


def test_ray_session_follows_rays_and_weights(dev, monkeypatch):
    """The encoder cache is keyed on the identity + in-place version of the ray tensors and on the weights' versions."""
    mod = _module(dev, monkeypatch)
    ro, rd, rc = _rays(dev)
    s1 = mod.ray_session(ro, rd, rc)
    assert mod.ray_session(ro, rd, rc) is s1
    assert mod.ray_session(ro.clone(), rd, rc) is not s1            # equal values, another tensor: not trusted
    s2 = mod.ray_session(ro, rd, rc)
    rc.mul_(0.5)                                                    # in-place edit of the colours: the cache is stale
    s3 = mod.ray_session(ro, rd, rc)
    assert s3 is not s2
    ds = _dataset(1, 64, 64, 4, seed=1)
    obs = ds.all_rgbs[0].to(dev)
    a = mod.test_image(obs[..., :3], obs[..., 3], ro, rd, rc)[2].clone()
    with torch.no_grad():
        mod.attention.q_proj.weight.mul_(1.01)                      # what an optimizer step does
    b = mod.test_image(obs[..., :3], obs[..., 3], ro, rd, rc)[2]
    assert mod._ray_session is not s3 and float((a - b).abs().max()) > 0
    # a copy of the module starts without device handles and builds its own
    import copy
    twin = copy.deepcopy(mod)
    assert twin._net is None and twin._ray_session is None
    c = twin.test_image(obs[..., :3], obs[..., 3], ro, rd, rc)[2]
    assert torch.equal(b, c)
    # writes torch's version counter does not see.  A `.data` swap moves the storage: noticed through the key's address
    s4 = mod.ray_session(ro, rd, rc)
    rc.data = rc.data.clone()
    assert mod.ray_session(ro, rd, rc) is not s4
    # ... a write into the SAME storage behind torch's back (a ctypes kernel, a hipGraph replay into static buffers) is not: the caller says so
    new = rc.flip(0).contiguous()
    want = mod.test_image(obs[..., :3], obs[..., 3], ro, rd, new)[2].clone()         # the scores of the colours about to be written
    s5 = mod.ray_session(ro, rd, rc)
    v = rc._version
    rc.untyped_storage().copy_(new.untyped_storage())                                # storage-level copy: `rc`'s version counter does not move
    torch.cuda.synchronize()
    if rc._version == v:                                                             # (where a torch does bump it, the key catches the write by itself)
        assert mod.ray_session(ro, rd, rc) is s5                                     # stale by construction ...
        mod.invalidate_ray_session()                                                 # ... until the writer says what it did
    got = mod.test_image(obs[..., :3], obs[..., 3], ro, rd, rc)[2]
    assert mod._ray_session is not s5 and torch.equal(got, want)


def test_eval_loop_recovers_from_a_call_that_died_between_submit_and_collect(dev, monkeypatch):
    """A captured batch that was submitted and never collected (an exception mid-loop: ADVICE round 5) must not leak into the next call:
    its slot is drained on entry, and the next call's records are its own."""
    import iffnerf_amd.pose_estimation.test as pt
    mod = _module(dev, monkeypatch)
    ro, rd, rc = _rays(dev)
    up = torch.tensor([0.1, 0.2, 0.9], device=dev)
    monkeypatch.setattr(pt, "EVAL_BATCH", 4)
    ds = _dataset(9, 120, 160, 4, seed=77)
    good = pt.test_pose_estimation(ds, mod, ro, rd, rc, up)[0]
    real = pt.CapturedEvalBatch.collect
    calls = {"n": 0}

    def dying(self):
        calls["n"] += 1
        if calls["n"] == 1:
            raise RuntimeError("host-side failure between submit and collect")
        return real(self)

    with monkeypatch.context() as m:
        m.setattr(pt.CapturedEvalBatch, "collect", dying)
        with pytest.raises(RuntimeError, match="between submit and collect"):
            pt.test_pose_estimation(ds, mod, ro, rd, rc, up)
    slots = next(iter(mod._ray_session.graphs.values()))
    assert any(s.pending is not None for s in slots)                    # the dead call left a batch behind ...
    smaller = _dataset(5, 120, 160, 4, seed=78)                         # ... whose indices would not even fit the next dataset
    again = pt.test_pose_estimation(smaller, mod, ro, rd, rc, up)[0]
    assert len(again) == 5 and all(s.pending is None for s in slots)
    with monkeypatch.context() as m:
        m.setattr(pt, "_batchable", lambda *a, **k: False)
        assert again == pt.test_pose_estimation(smaller, mod, ro, rd, rc, up)[0]
    assert pt.test_pose_estimation(ds, mod, ro, rd, rc, up)[0] == good


def test_rgba_resize_equals_composite_then_resize(dev):
    """iff_image_resize_crop_rgba: composite on white / alpha taken inside the resize == the torch composite of test.py:77-81
    followed by iff_image_resize_crop, bit for bit (odd sizes, both orientations, the 16 x 16 token-grid mask included)."""
    from iffnerf_amd.image_frontend import RESIZE_ALPHA, RESIZE_RGB_ON_WHITE, resize_crop, resize_crop_rgba
    from iffnerf_amd.pose_estimation.identification_module import IMAGENET_DEFAULT_MEAN as MEAN, IMAGENET_DEFAULT_STD as STD
    gen = torch.Generator().manual_seed(9)
    for (Q, H, W) in ((3, 200, 260), (2, 333, 257), (1, 800, 800)):
        rgba = torch.rand(Q, H, W, 4, generator=gen).to(dev)
        rgb = rgba[..., :3] * rgba[..., -1:] + (1 - rgba[..., -1:])
        assert torch.equal(resize_crop_rgba(rgba, 256, 224, RESIZE_RGB_ON_WHITE, True, MEAN, STD), resize_crop(rgb, 256, 224, True, MEAN, STD))
        assert torch.equal(resize_crop_rgba(rgba, 256, 224, RESIZE_ALPHA, False), resize_crop(rgba[..., 3:], 256, 224, False))
    with pytest.raises(RuntimeError, match="RGBA"):
        resize_crop_rgba(rgba[..., :3], 256, 224, RESIZE_ALPHA, False)


def test_pose_errors_kernel_matches_the_reference_formulas(dev):
    """iff_pose_errors against errors.py:3-9 / test.py:213-241 evaluated with torch ops in float64."""
    from iffnerf_amd import hip_identify as H
    gen = torch.Generator().manual_seed(3)
    Q, k = 37, 100
    gt = torch.eye(4).repeat(Q, 1, 1)
    gt[:, :3, :3] = torch.linalg.qr(torch.randn(Q, 3, 3, generator=gen)).Q
    gt[:, :3, 3] = torch.randn(Q, 3, generator=gen)
    pred = gt.clone()
    pred[:, :3, :3] = torch.linalg.qr(gt[:, :3, :3] + 0.2 * torch.randn(Q, 3, 3, generator=gen)).Q
    pred[:, :3, 3] += 0.3 * torch.randn(Q, 3, generator=gen)
    pred[0] = gt[0]                                               # identical pose: zero errors (acos at 1)
    parts = torch.rand(Q, 8 + k, generator=gen)
    parts[:, 8:][torch.rand(Q, k, generator=gen) < 0.3] = -1.0    # filtered rays
    out = H.pose_errors(pred.to(dev), gt.to(dev), parts.to(dev)).cpu()
    for q in range(Q):
        w = parts[q, 8:]
        kept = w[w >= 0].double()
        assert out[q, 3] == kept.numel()
        assert abs(float(out[q, 0]) - float(kept.mean())) < 1e-6
        assert abs(float(out[q, 1]) - float((gt[q, :3, 3].double() - pred[q, :3, 3].double()).norm())) < 1e-6
        cos = ((gt[q, :3, :3].double() @ torch.linalg.inv(pred[q, :3, :3].double())).trace() - 1) / 2
        want = float(torch.rad2deg(torch.arccos(cos.clamp(-1, 1))))
        assert abs(float(out[q, 2]) - want) < 2e-2 + 1e-4 * want, (q, float(out[q, 2]), want)   # acos near 1 amplifies fp32 rounding
    assert float(out[0, 1]) == 0.0 and float(out[0, 2]) < 0.05
    no_parts = H.pose_errors(pred.to(dev), gt.to(dev)).cpu()
    assert torch.isnan(no_parts[:, 0]).all() and torch.equal(no_parts[:, 1:3], out[:, 1:3])
