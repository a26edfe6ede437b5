"""Oracle: the score loss the reference's validation calls hand to ``test_pose_estimation``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, on torch-CPU (it runs wherever its inputs live),
  pose_estimation/loss.py:4-41    best_one_to_one_rays_selector: the distance of the true camera centre to every ray -> 1 - tanh
  pose_estimation/loss.py:107-147 DistanceBasedScoreLoss.forward (reweight_method "none": the constructor default and what
                                  pose_estimation/train.py builds)
The pixel projection of loss.py:43-92 feeds only ``is_inside``, which forward discards (:123-127): not restated.
Pinned by tests/golden/g15_score_loss.npz (tests/test_oracle_golden.py, tolerance 0).  pose_estimation/test.py:113-122 calls it as
``loss_fn(pred_scores, pose, K, rays_ori, rays_dirs, attention_map.shape[-2], backbone_wh, model_up=...)``.
"""
from __future__ import annotations

import torch


def target_score(camera_pose: torch.Tensor, rays_ori: torch.Tensor, rays_dir: torch.Tensor, tanh_denominator: float = 1.0):
    """loss.py:13-37: camera centre = [0 0 0 1] @ pose[:3,:].T; closest point of each ray (clamped at its origin) to it."""
    centre = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=camera_pose.dtype, device=camera_pose.device).reshape(1, 4) @ camera_pose[:3, :].T
    to_centre = centre - rays_ori
    along = torch.bmm(to_centre.view(-1, 1, 3), rays_dir.view(-1, 3, 1))[..., 0]
    closest = torch.where(along < 0, rays_ori, rays_ori + torch.multiply(along, rays_dir))
    distance = torch.linalg.norm(closest - centre, dim=-1)
    return 1 - torch.tanh(distance / tanh_denominator)


class DistanceBasedScoreLoss(torch.nn.Module):
    """loss.py:97-147 with the default constructor arguments."""

    def forward(self, pred_score, camera_pose, camera_intrinsic, rays_ori, rays_dir, total_number_of_features, backbone_wh,
                model_up=None, obs_img_shape=(800, 800)):
        with torch.no_grad():
            combined = target_score(camera_pose, rays_ori, rays_dir)
            combined = torch.multiply(combined, total_number_of_features / combined.sum())      # :137-141
        return torch.square(pred_score - combined).mean(), combined                               # :143-147
