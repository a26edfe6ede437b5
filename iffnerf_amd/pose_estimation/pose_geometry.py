"""Mirror of the three ``pose_estimation/pose_geometry.py`` helpers the per-image pose solve uses.

On the MI355X path the whole solve (unique-origin filter, least squares, exclusion, look-at) is ONE kernel,
``iff_pose_from_topk`` (see ``pose_estimation/test.py`` in this package).  These functions keep the reference's names
for callers that use a piece in isolation; they are a handful of 3x3 host-side tensor ops, not a compute path.
"""
from __future__ import annotations

from typing import Optional

import torch


def compute_line_intersection_impl2(points, directions, weights: Optional[torch.Tensor] = None, return_residuals=False):
    """Least-squares point closest to N lines (reference :42-95): solve [sum (I - d d^T)] p = sum (I - d d^T) o."""
    if return_residuals:
        raise RuntimeError("return_residuals is unused on the path (and broken upstream: linalg.solve has no residuals)")
    eye = torch.eye(directions.shape[-1], dtype=points.dtype, device=points.device)
    proj = eye - directions.unsqueeze(2) * directions.unsqueeze(1)
    rhs = proj @ points.unsqueeze(2)
    if weights is not None:
        w = weights.view(-1, 1, 1)
        lhs_sum, rhs_sum = (proj * w).sum(0), (rhs * w).sum(0)
    else:
        lhs_sum, rhs_sum = proj.sum(0), rhs.sum(0)
    if torch.linalg.det(lhs_sum) < 1.e-7:
        return torch.full((3,), float("nan"), dtype=lhs_sum.dtype, device=lhs_sum.device)
    return torch.linalg.solve(lhs_sum, rhs_sum)[:, 0]


def exclude_negatives(camera_optical_center, sample_points, dirs):
    """True for rays whose direction points towards the centre (reference :199-204)."""
    return ((camera_optical_center[None] - sample_points) * dirs).sum(-1) > 0


def make_rotation_mat(direction: torch.Tensor, up: torch.Tensor):
    """World-to-camera rotation with rows (x = up x dir, y = dir x x, dir), x and y normalised (reference :175-196).
    Built on the host like the reference does (it fills a CPU ``torch.eye(3)``)."""
    direction, up = direction.detach().cpu(), up.detach().cpu()
    x = torch.linalg.cross(up, direction)
    x = x / x.norm()
    y = torch.linalg.cross(direction, x)
    y = y / y.norm()
    return torch.stack((x, y, direction))
