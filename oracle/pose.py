"""Oracle: closed-form camera pose from the top-k rays (consumer of stage C).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, on torch-CPU,
  pose_estimation/test.py:133-174,192-194 (per-image body after test_image)
  pose_estimation/pose_geometry.py:42-95 (LS line intersection), :175-196, :199-204
  pose_estimation/errors.py:3-9
including the reference's quirks: the scalar-wise ``isin`` origin filter and the second
solve that ignores the exclusion weights.
"""
from __future__ import annotations

import torch


def line_intersection(points: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """pose_geometry.py:42-95 with weights=None."""
    proj = torch.eye(3, dtype=points.dtype) - dirs[:, :, None] * dirs[:, None, :]
    R = torch.sum(proj, dim=0)
    q = torch.sum(proj @ points[:, :, None], dim=0)
    if torch.linalg.det(R) < 1.e-7:
        return torch.tensor([float("nan")] * 3, dtype=R.dtype)
    return torch.linalg.solve(R, q)[:, 0]


def in_front(centre, origins, dirs):
    """pose_geometry.py:199-204."""
    v = centre[None] - origins
    return torch.bmm(v.view(-1, 1, 3), dirs.view(-1, 3, 1))[..., 0, 0] > 0


def look_rotation(direction, up):
    """pose_geometry.py:175-196: rows x=up x dir, y=dir x x, dir."""
    xa = torch.cross(up, direction, dim=-1)
    xa = xa / torch.linalg.norm(xa, dim=-1, keepdim=True)
    ya = torch.cross(direction, xa, dim=-1)
    ya = ya / torch.linalg.norm(ya, dim=-1, keepdim=True)
    return torch.stack((xa, ya, direction), dim=0)


def unique_origin_mask(origins: torch.Tensor) -> torch.Tensor:
    """test.py:133-136: keep a ray if any coordinate matches a scalar of a once-seen origin."""
    uniq, counts = torch.unique(origins, return_counts=True, dim=0)
    return torch.isin(origins, uniq[counts == 1], assume_unique=True).any(dim=1)


def pose_from_topk(idx, values, rays_o, rays_d, model_up, return_parts: bool = False):
    """test.py:133-174,192-194.  ``model_up`` is normalised here as test.py:29 does."""
    up = model_up / torch.linalg.norm(model_up, dim=-1, keepdim=True)
    keep = unique_origin_mask(rays_o[idx])
    idx, w = idx[keep], values[keep]
    o, d = rays_o[idx], rays_d[idx]
    w = w / torch.sum(w)
    c = line_intersection(o, d)
    w = w * in_front(c, o, d)
    w = w / torch.sum(w)
    c = line_intersection(o, d)
    watch = torch.sum(d * w[:, None], dim=0)
    watch = watch / torch.linalg.norm(watch, dim=-1, keepdim=True)
    c2w = torch.eye(4, dtype=rays_o.dtype)
    rot = look_rotation(-watch, up)
    if torch.linalg.det(rot) < 1.0e-7:
        rot = torch.eye(3)
    c2w[:3, :3] = torch.linalg.inv(rot)
    c2w[:3, -1] = c
    if torch.isnan(c2w).any():
        c2w = torch.eye(4, dtype=rays_o.dtype)
    if return_parts:
        return c2w, dict(keep=keep, centre=c, weights=w, watch=watch)
    return c2w


def translation_error(t1, t2):
    """errors.py:3-4."""
    return torch.linalg.norm(t1 - t2)


def angular_error_deg(r_gt, r_est):
    """errors.py:7-9."""
    c = (torch.trace(r_gt @ torch.linalg.inv(r_est)) - 1) / 2
    return torch.rad2deg(torch.arccos(torch.clamp(c, min=-1, max=1)))
