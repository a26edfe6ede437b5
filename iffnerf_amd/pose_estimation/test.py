"""Mirror of ``pose_estimation/test.py``: ``test_pose_estimation`` with the reference's signature and result format.

Per image (reference :67-247): RGBA -> RGB on white, ``id_module.test_image`` (stage C in HIP), then the closed-form pose
from the top-100 rays in ONE kernel (``iff_pose_from_topk``) instead of ~40 tiny host-driven tensor ops, then the
reference's error metrics, optionally after the iNeRF refinement of reference :196-211 (``inerf_refinement=True``: 800 Adam
steps through the HIP slab march and its HIP backward, ``iffnerf_amd/inerf``).  ``loss_fn`` / ``save`` belong to the training
and plotting code and are out of scope for the MI355X path.
"""
from __future__ import annotations

import time
from statistics import mean

import torch

from .errors import compute_angular_error, compute_translation_error


def estimate_pose(id_module, obs_img, mask_img, rays_ori, rays_dirs, rays_rgb, model_up, rays_to_output=100):
    """One query image -> (c2w [4,4] on the GPU, solver internals, top-k indices, top-k values, scores)."""
    from .. import hip_identify as H
    idx, weights, scores, _ = id_module.test_image(obs_img, mask_img, rays_ori, rays_dirs, rays_rgb,
                                                   rays_to_output=rays_to_output)
    c2w, parts = H.pose_from_topk(idx, weights, rays_ori, rays_dirs, model_up, want_parts=True)
    return c2w, parts, idx, weights, scores


INERF_ITERS = 800      # reference test.py:204
INERF_BATCH = 1024     # pose_estimation's default batch_size (inerf/estimate_pose_inerf.py:31), which test.py:196-209 leaves alone


def test_pose_estimation(dataset, id_module, rays_ori, rays_dirs, rays_rgb, model_up, sequence_id="", loss_fn=None,
                         save=False, inerf_refinement=False, nerf_model=None, save_all=False, augmentation_parameters={}):
    if loss_fn is not None or save:
        raise RuntimeError("test_pose_estimation: loss_fn / save belong to the training and plotting code paths, which "
                           "are out of scope for the MI355X hot path")
    if inerf_refinement and nerf_model is None:
        raise RuntimeError("test_pose_estimation(inerf_refinement=True) needs nerf_model (reference test.py:196-203)")
    id_module.eval()
    device = rays_ori.device
    n_images = dataset.all_rgbs.shape[0]
    translation_errors, angular_errors, results = [], [], []
    start = time.time()
    for img_idx in range(n_images):
        pose = dataset.poses[img_idx].to(device, non_blocking=True)
        obs = dataset.all_rgbs[img_idx].to(device, non_blocking=True)
        if obs.shape[-1] == 4:
            mask_img = obs[..., -1]
            obs = obs[..., :3] * obs[..., -1:] + (1 - obs[..., -1:])
        else:
            mask_img = torch.ones_like(obs[..., -1], dtype=torch.bool)
        c2w, parts, idx, weights, _ = estimate_pose(id_module, obs, mask_img, rays_ori, rays_dirs, rays_rgb, model_up)
        if inerf_refinement:                                                               # reference :196-211
            from ..inerf.estimate_pose_inerf import pose_estimation
            rgba = torch.cat((obs, mask_img[..., None].to(obs.dtype)), dim=-1).cpu().numpy()
            with torch.enable_grad():
                _, c2w, _ = pose_estimation(c2w, rgba, dataset.K.to(device)[0], nerf_model, device=c2w.device, n_iters=INERF_ITERS, batch_size=INERF_BATCH,
                                            print_progress=False, lrate=0.02, dice_loss=True, sampling_strategy="random")
            c2w = c2w.to(device)
        translation_errors.append(compute_translation_error(pose[:3, 3], c2w[:3, 3]).item())
        angular_errors.append(compute_angular_error(pose[:3, :3], c2w[:3, :3]).item())
        kept = parts[8:][parts[8:] >= 0]           # weights after exclusion of the rays that survived the origin filter
        results.append({
            "sequence_id": sequence_id, "category_name": "id_net", "frame_id": img_idx,
            "loss": kept.mean().item(), "scores_loss": -1.0, "recall": -1.0, "total_optimization_time_in_ms": 0.0,
            "pred_c2w": c2w.cpu().tolist(), "gt_c2w": pose.cpu().tolist(),
        })
    per_image = (time.time() - start) / max(n_images, 1)
    print("Time per element: ", per_image)
    avg_t, avg_a = mean(translation_errors), mean(angular_errors)
    print("Translation Error: ", avg_t)
    print("Angular Error: ", avg_a)
    return results, avg_t, avg_a, -1.0, -1.0
