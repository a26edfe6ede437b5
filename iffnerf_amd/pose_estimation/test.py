"""Mirror of ``pose_estimation/test.py``: ``test_pose_estimation`` with the reference's signature and result format.

The reference loops over the images (:66-247): RGBA -> RGB on white, ``id_module.test_image`` (which re-runs the ray encoder on
rays that never change, writes the [M,N] attention map and boolean-indexes the tokens), ~40 host-driven tensor ops for the pose and
three ``.item()`` reads.  Here the same call

    test_pose_estimation(dataset, id_module, rays_ori, rays_dirs, rays_rgb, model_up, ...)

serves the images in BATCHES through one set of launches per batch -- composite + resize / crop / normalise
(``iff_image_resize_crop_rgba``), DINOv2's forward (``iff_vit_forward``), token assembly, folded query projection, logits against
the ray encoder's cached output (``iff_ray_cache_build`` once per (weights, ray set), ``iff_logits_from_cache`` per batch), softmax
/ column sums / top-100, the closed-form pose (``iff_pose_from_topk_batched``) and the error metrics (``iff_pose_errors``) -- with
batches replayed as captured hipGraphs on four alternating streams (the resize launches stay outside the graph and read the
batch where the dataset holds it; a dataset's tail is padded with copies of its last image) and ONE device->host read per batch.  Every number equals,
bit for bit, what the image-by-image route below returns (tests/test_hip_eval_loop.py): both run the same kernels, whose per-row
arithmetic does not depend on the batch.  Image by image is the route of everything a batch cannot serve: an arbitrary backbone
module (``IdentificationModule.serves_batches``), replaced preprocessing, the iNeRF refinement of reference :196-211
(``inerf_refinement=True``: 800 Adam steps per image through the HIP slab march and its HIP backward, ``iffnerf_amd/inerf``).
``loss_fn`` (the validation calls of pose_estimation/train.py:145-153,188-196 pass the training loss) also runs image by image:
the loss is the CALLER's module, called per image exactly as reference :113-127 calls it, on scores that come from the HIP path.
``save=True`` writes to a path inside the author's home directory (reference :185-188) and is refused.
"""
from __future__ import annotations

import time
from statistics import mean

import torch

from .errors import compute_angular_error, compute_translation_error  # noqa: F401  (the reference's import line, :8)

INERF_ITERS = 800      # reference test.py:204
INERF_BATCH = 1024     # pose_estimation's default batch_size (inerf/estimate_pose_inerf.py:31), which test.py:196-209 leaves alone
EVAL_BATCH = 32        # images per batch of the batched route (one captured hipGraph per full batch)
EVAL_SLOTS = 4         # captured batches in flight, each on its own stream.  Same-box medians of five calls, images/s at 2 -> 4 slots
#                        (scripts/time_dropin_slots.py): 128 images 10 060-10 180 -> 10 900-10 990; 200 images (six batches + a padded tail
#                        of eight) 7 230-8 950 -> 10 140-10 170; 540 000 rays, 68 / 136 images 2 026 / 1 888 -> 2 154 / 2 198
LOGITS_BUDGET_BYTES = 9 << 30     # a batch's [B * 256, N] fp32 logits stay below this (540 000 rays: 17 images per batch; measured
#                                   images/s at 7 / 17 / 32 per batch: 1 660 / 1 830 / 1 830, scripts/time_dropin_540k.py) and below
#                                   a sixteenth of the HBM that is free when the graphs are made (four slots, each a graph pool + its
#                                   warm-up's blocks: at most half of it)
TOPK = 100             # rays_to_output of reference :90


def estimate_pose(id_module, obs_img, mask_img, rays_ori, rays_dirs, rays_rgb, model_up, rays_to_output=TOPK):
    """One query image -> (c2w [4,4] on the GPU, solver internals, top-k indices, top-k values, scores, attention map (lazy))."""
    from .. import hip_identify as H
    idx, weights, scores, attention_map = id_module.test_image(obs_img, mask_img, rays_ori, rays_dirs, rays_rgb,
                                                   rays_to_output=rays_to_output)
    c2w, parts = H.pose_from_topk(idx, weights, rays_ori, rays_dirs, model_up, want_parts=True)
    return c2w, parts, idx, weights, scores, attention_map


def _record(sequence_id, img_idx, summary_row, c2w_rows, gt_rows, scores_loss=-1.0, recall=-1.0):
    """One entry of the reference's result list (:234-246).  ``summary_row`` = (loss, translation error, angular error, kept)."""
    return {"sequence_id": sequence_id, "category_name": "id_net", "frame_id": img_idx,
            "loss": summary_row[0], "scores_loss": scores_loss, "recall": recall, "total_optimization_time_in_ms": 0.0,
            "pred_c2w": c2w_rows, "gt_c2w": gt_rows}


def _as_4x4(poses):
    """Ground-truth poses [..,3,4] (the reference only ever reads ``pose[:3, :]``, :213-232) or [..,4,4] -> [..,4,4]."""
    if poses.shape[-2:] == (4, 4):
        return poses
    if poses.shape[-2:] != (3, 4):
        raise RuntimeError(f"test_pose_estimation: dataset.poses must be [..,3,4] or [..,4,4], got {tuple(poses.shape)}")
    last = torch.zeros(poses.shape[:-2] + (1, 4), dtype=poses.dtype, device=poses.device)
    last[..., 0, 3] = 1.0
    return torch.cat((poses, last), dim=-2)


def _score_loss(loss_fn, id_module, scores, idx, weights, attention_map, pose, intrinsic, rays_ori, rays_dirs, model_up):
    """Reference :110-127: the caller's loss on this image's scores + the reference's "recall" (the positions of the 100 largest
    top-k weights looked up among the ray indices, exactly as :125-127 writes it).  Two host reads, as in the reference."""
    avg_score, _ = loss_fn(scores, pose, intrinsic, rays_ori, rays_dirs, attention_map.shape[-2], id_module.backbone_wh,
                           model_up=model_up)
    target_idx = torch.topk(weights, k=min(TOPK, weights.shape[0])).indices
    recall = torch.count_nonzero(torch.isin(target_idx, idx)).item() / target_idx.shape[0]
    return avg_score.item(), recall


def _eval_from_tokens(id_module, session, tokens, keep, rows, gt_poses, model_up, k):
    from .. import hip_identify as H
    score, _ = id_module.scores_static(tokens, keep, session, want_map=False, rows=rows)
    idx, val = H.topk_batched(score, k)
    c2w, parts = H.pose_from_topk_batched(idx, val, session.ori, session.dirs, model_up, want_parts=True)
    return c2w, H.pose_errors(c2w, gt_poses, parts)


def eval_batch(id_module, session, images, gt_poses, model_up, k=TOPK):
    """images [B,H,W,4] (RGBA) or [B,H,W,3], gt_poses [B,4,4] -> (c2w [B,4,4], summary [B,4]) on the device, nothing read back.
    The body of reference :71-232 for B images: the launches of ``IdentificationModule.test_image`` + pose solve + error metrics."""
    tokens, keep, rows = id_module.static_tokens(images, None, compact=True)
    return _eval_from_tokens(id_module, session, tokens, keep, rows, gt_poses, model_up, k)


class CapturedEvalBatch:
    """``eval_batch`` for a fixed batch shape, in two parts on the slot's own stream.  EAGER: the resize launches, the only ones that
    read the full-size images -- they take a device-resident batch where it lies (a slice of the dataset's tensor: no copy of
    B x H x W x 4 floats into a static buffer; 328 MB per 32 images of 800 x 800) and write the graph's static inputs, the
    normalised 224 x 224 crops and alpha planes.  CAPTURED (one hipGraph): backbone, tokens, logits against the cached encoder
    output, softmax statistics, column sums, top-k, pose solve, error metrics.  ``submit`` runs both and starts the read-back into
    pinned host memory; ``collect`` waits for it.  The launches are those of ``eval_batch``, in the same order per buffer."""

    def __init__(self, id_module, session, shape, model_up, k=TOPK):
        dev = session.ori.device
        self.fe = id_module.frontend()
        self.stream = torch.cuda.Stream(device=dev)
        self.shape = tuple(shape)
        self.staging = None                               # a host-resident (or non-fp32) dataset passes through here
        images = torch.zeros(shape, dtype=torch.float32, device=dev)
        images[..., -1] = 1.0 if shape[-1] == 4 else 0.0
        self.gt = torch.eye(4, device=dev).repeat(shape[0], 1, 1)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):            # warm-up outside capture (handles, workspaces, allocator)
            self.xin, self.alpha = self.fe.preprocess(images)
            for _ in range(2):
                self._tail(id_module, session, model_up, k)
        torch.cuda.synchronize(dev)
        del images
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.c2w, self.summary = self._tail(id_module, session, model_up, k)
        self.host_c2w = torch.empty(self.c2w.shape, dtype=torch.float32, pin_memory=True)
        self.host_summary = torch.empty(self.summary.shape, dtype=torch.float32, pin_memory=True)
        self.done = torch.cuda.Event()
        self.pending = None

    def _tail(self, id_module, session, model_up, k):
        tokens, keep, rows = self.fe.tokens_from_preprocessed(self.xin, self.alpha, compact=True)
        return _eval_from_tokens(id_module, session, tokens, keep, rows, self.gt, model_up, k)

    def submit(self, images, gt, tag):
        """``images`` [b,H,W,C], ``gt`` [b,4,4] with b <= B: a batch shorter than the captured shape (a dataset's tail) is padded with
        copies of its last image -- an image's result does not depend on what else is in its batch -- and ``collect`` returns its b rows."""
        b = images.shape[0]
        with torch.cuda.stream(self.stream):
            if b < self.shape[0] or not (images.is_cuda and images.dtype == torch.float32 and images.is_contiguous()):
                if self.staging is None:
                    self.staging = torch.empty(self.shape, dtype=torch.float32, device=self.xin.device)
                self.staging[:b].copy_(images, non_blocking=True)
                if b < self.shape[0]:
                    self.staging[b:].copy_(self.staging[b - 1:b].expand(self.shape[0] - b, -1, -1, -1))
                images = self.staging
            self.fe.preprocess(images, out=(self.xin, self.alpha))
            self.gt[:b].copy_(gt, non_blocking=True)
            if b < self.shape[0]:
                self.gt[b:].copy_(torch.eye(4, device=self.gt.device).expand(self.shape[0] - b, 4, 4))
            self.graph.replay()
            self.host_c2w.copy_(self.c2w, non_blocking=True)
            self.host_summary.copy_(self.summary, non_blocking=True)
            self.done.record(self.stream)
        self.pending = (tag, b)

    def collect(self):
        self.done.synchronize()
        (tag, b), self.pending = self.pending, None
        return tag, self.host_c2w[:b].clone(), self.host_summary[:b].clone()


def _batched_route(dataset, id_module, rays_ori, rays_dirs, rays_rgb, model_up, sequence_id):
    """Reference :66-247 over the whole dataset, a batch of images per set of launches."""
    device = rays_ori.device
    n, H_, W_, C_ = dataset.all_rgbs.shape
    session = id_module.ray_session(rays_ori, rays_dirs, rays_rgb)
    budget = min(LOGITS_BUDGET_BYTES, torch.cuda.mem_get_info(device)[0] // (4 * EVAL_SLOTS)) if not session.graphs else session.logits_budget
    session.logits_budget = budget                      # (the batch size of graphs that exist stands: their memory is no longer "free")
    B = max(1, min(EVAL_BATCH, budget // (256 * 4 * max(session.n_rays, 1))))
    up = tuple(float(v) for v in torch.as_tensor(model_up).detach().cpu().reshape(-1).tolist())
    out = [None] * n

    def harvest(first, c2w, summary):
        c2w, summary = c2w.tolist(), summary.tolist()
        for j in range(len(c2w)):
            out[first + j] = (summary[j], c2w[j])

    n_full = n // B
    if n_full:
        # a captured batch holds the addresses of the backbone's kernel tables too: keyed on its parameters' identity + version
        backbone_key = tuple((p.data_ptr(), p._version) for p in id_module.image_preprocessing_net.parameters())
        key = (B, H_, W_, C_, up, backbone_key)
        if key not in session.graphs:
            session.graphs.clear()                       # at most one batch shape's graphs (and their logits buffers) stay alive
            session.graphs[key] = [CapturedEvalBatch(id_module, session, (B, H_, W_, C_), up) for _ in range(min(EVAL_SLOTS, n_full))]
        slots = session.graphs[key]
        cur = torch.cuda.current_stream(device)
        for s in slots:
            if s.pending is not None:                    # an earlier call died between submit and collect: its batch is not ours
                s.stream.synchronize()
                s.pending = None
            s.stream.wait_stream(cur)
        for b in range(n_full + (1 if n_full * B < n else 0)):          # the tail goes through a captured batch too, padded (submit)
            slot = slots[b % len(slots)]
            if slot.pending is not None:
                harvest(*slot.collect())
            slot.submit(dataset.all_rgbs[b * B:(b + 1) * B], _as_4x4(dataset.poses[b * B:(b + 1) * B]), b * B)
        for slot in slots:
            if slot.pending is not None:
                harvest(*slot.collect())
            cur.wait_stream(slot.stream)
    elif n:                                             # a dataset smaller than one batch: the same launches, eagerly
        lo = n_full * B
        imgs = dataset.all_rgbs[lo:].to(device=device, dtype=torch.float32, non_blocking=True)
        gt = _as_4x4(dataset.poses[lo:]).to(device=device, dtype=torch.float32, non_blocking=True)
        c2w, summary = eval_batch(id_module, session, imgs, gt, up)
        harvest(lo, c2w.cpu(), summary.cpu())
    return out


def _batchable(dataset, id_module, rays_ori, inerf_refinement) -> bool:
    rgbs = getattr(dataset, "all_rgbs", None)
    return (not inerf_refinement and getattr(id_module, "fold_heads", True) and torch.is_tensor(rgbs) and rgbs.dim() == 4
            and rgbs.shape[-1] in (3, 4) and rays_ori.is_cuda and id_module.serves_batches())


def test_pose_estimation(dataset, id_module, rays_ori, rays_dirs, rays_rgb, model_up, sequence_id="", loss_fn=None,
                         save=False, inerf_refinement=False, nerf_model=None, save_all=False, augmentation_parameters={}):
    if save:
        raise RuntimeError("test_pose_estimation(save=True) writes its dump to a path inside the reference author's home directory "
                           "(pose_estimation/test.py:185-188); not provided")
    if inerf_refinement and nerf_model is None:
        raise RuntimeError("test_pose_estimation(inerf_refinement=True) needs nerf_model (reference test.py:196-203)")
    from .. import hip_identify as H
    id_module.eval()
    device = rays_ori.device
    n_images = dataset.all_rgbs.shape[0]
    results = []
    start = time.time()
    scores_losses, recalls = [], []
    if loss_fn is None and _batchable(dataset, id_module, rays_ori, inerf_refinement):
        with torch.no_grad():
            rows = _batched_route(dataset, id_module, rays_ori, rays_dirs, rays_rgb, model_up, sequence_id)
        gt_rows = dataset.poses.detach().cpu().tolist()               # one read for the whole dataset, not one per image
        for img_idx, (summary, c2w) in enumerate(rows):
            results.append((summary, _record(sequence_id, img_idx, summary, c2w, gt_rows[img_idx])))
    else:
        if loss_fn is not None:
            up_unit = torch.as_tensor(model_up, dtype=torch.float32, device=device)
            up_unit = up_unit / torch.linalg.norm(up_unit, dim=-1, keepdim=True)          # reference :29: the loss gets the unit vector
            intrinsic = dataset.K.to(device, non_blocking=True)[0]
        for img_idx in range(n_images):
            pose_as_given = dataset.poses[img_idx].to(device, non_blocking=True)
            pose = _as_4x4(pose_as_given)
            obs = dataset.all_rgbs[img_idx].to(device, non_blocking=True)
            if obs.shape[-1] == 4:
                mask_img = obs[..., -1]
                obs = obs[..., :3] * obs[..., -1:] + (1 - obs[..., -1:])
            else:
                mask_img = torch.ones_like(obs[..., -1], dtype=torch.bool)
            c2w, parts, idx, weights, scores, attention_map = estimate_pose(id_module, obs, mask_img, rays_ori, rays_dirs, rays_rgb, model_up)
            scores_loss, recall = -1.0, -1.0
            if loss_fn is not None:
                with torch.no_grad():
                    scores_loss, recall = _score_loss(loss_fn, id_module, scores, idx, weights, attention_map, pose, intrinsic,
                                                      rays_ori, rays_dirs, up_unit)
            scores_losses.append(scores_loss)
            recalls.append(recall)
            if inerf_refinement:                                                               # reference :196-211
                from ..inerf.estimate_pose_inerf import pose_estimation
                rgba = torch.cat((obs, mask_img[..., None].to(obs.dtype)), dim=-1).cpu().numpy()
                with torch.enable_grad():
                    _, c2w, _ = pose_estimation(c2w, rgba, dataset.K.to(device)[0], nerf_model, device=c2w.device, n_iters=INERF_ITERS, batch_size=INERF_BATCH,
                                                print_progress=False, lrate=0.02, dice_loss=True, sampling_strategy="random")
                c2w = c2w.to(device)
            # error metrics (:213-232) and the "loss" entry (:241) in one launch, ONE read per image
            both = torch.cat((H.pose_errors(c2w, pose, parts).reshape(-1), c2w.detach().to(torch.float32).reshape(-1))).cpu().tolist()
            results.append((both[:4], _record(sequence_id, img_idx, both[:4], [both[4 + 4 * r:8 + 4 * r] for r in range(4)], pose_as_given.cpu().tolist(),
                                              scores_loss, recall)))
    per_image = (time.time() - start) / max(n_images, 1)
    avg_score, avg_recall = (mean(scores_losses), mean(recalls)) if scores_losses else (-1.0, -1.0)
    if loss_fn is not None:
        print("Average loss score: ", avg_score)
        print("Average Recall: ", avg_recall)
    print("Time per element: ", per_image)
    translation_errors = [s[1] for s, _ in results]
    angular_errors = [s[2] for s, _ in results]
    avg_t, avg_a = mean(translation_errors), mean(angular_errors)
    print("Translation Error: ", avg_t)
    print("Angular Error: ", avg_a)
    return [r for _, r in results], avg_t, avg_a, avg_score, avg_recall
