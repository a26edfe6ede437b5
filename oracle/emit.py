"""Oracle: surface sampling and iso-cell ray emission (stage A of the path).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, on torch-CPU,
  pose_estimation/isocell.py:6-68,131-171
  pose_estimation/sampling.py:35-67,78-116,131-213,229-251,442-488,509-541
  pose_estimation/model_utils.py:22-33
The stochastic sampler draws from the global torch RNG in the same order and with the same
shapes as the reference, so under one ``torch.manual_seed`` both produce the same stream.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
import torch

from . import field as fld


# ----------------------------------------------------------------------------- iso-cells
def isocell_dirs(n_target: int = 27, n0: int = 3, dtype=torch.float32) -> torch.Tensor:
    """isocell.py:6-68 with isrand=-1: deterministic cell centres on the +z hemisphere."""
    n = int(math.ceil(math.sqrt(n_target / n0)))
    total = int(n0 * n ** 2)
    ring = torch.arange(1, n + 1, dtype=torch.int64)
    per_ring = n0 * (2 * ring - 1)
    radius = torch.repeat_interleave(ring, per_ring, dim=0) * (1 / n)
    dth = 2 * math.pi / per_ring.to(dtype)
    _ = torch.rand(1, dtype=dtype)  # isocell.py:22 draws one value even when isrand == -1
    first = torch.cat((torch.zeros(1, dtype=per_ring.dtype), torch.cumsum(per_ring, 0)[:-1]), 0)
    k = (torch.arange(0, total, dtype=torch.int64) - torch.repeat_interleave(first, per_ring, 0)).to(dtype)
    dth_e = dth[torch.repeat_interleave(torch.arange(0, n, dtype=torch.int64), per_ring, 0)]
    theta = (0 + k * dth_e) + dth_e / 2
    radius = radius - (1 / n) / 2
    xr = radius * torch.cos(theta)
    yr = radius * torch.sin(theta)
    zr = torch.real(torch.sqrt(1 - torch.square(xr.to(torch.complex64)) - torch.square(yr.to(torch.complex64))))
    return torch.column_stack([xr, yr, zr])


def _skew(v: torch.Tensor) -> torch.Tensor:
    """isocell.py:131-141."""
    m = torch.zeros((*v.shape[:-1], 3, 3), dtype=v.dtype)
    m[..., 0, 1] = -v[..., 2]
    m[..., 0, 2] = v[..., 1]
    m[..., 1, 0] = v[..., 2]
    m[..., 1, 2] = -v[..., 0]
    m[..., 2, 0] = -v[..., 1]
    m[..., 2, 1] = v[..., 0]
    return m


def rotate_isocell(cells: torch.Tensor, normal: torch.Tensor) -> torch.Tensor:
    """isocell.py:144-171: rotation taking +z to -normal, applied to every cell -> [P,C,3]."""
    tgt = -normal
    C, P = cells.shape[0], tgt.shape[0]
    b = torch.divide(tgt, torch.linalg.norm(tgt, dim=-1, keepdim=True))[:, None]
    b = torch.broadcast_to(b, (P, C, 3)).reshape(-1, 3)
    a = torch.tensor([[0., 0., 1.]], dtype=cells.dtype)[None, :].expand(-1, C, -1)
    a = torch.broadcast_to(a, (P, C, 3)).reshape(-1, 3)
    v = torch.linalg.cross(a, b, dim=-1)
    c = torch.bmm(a.view(-1, 1, 3), b.view(-1, 3, 1))[..., 0, 0]
    s = torch.linalg.norm(v, dim=-1)
    k = _skew(v)
    rot = torch.eye(3, dtype=cells.dtype)[None] + k + torch.multiply(torch.bmm(k, k), ((1 - c) / (s ** 2))[..., None, None])
    src = torch.broadcast_to(cells[None, :], (P, C, 3)).reshape(-1, 3)
    out = torch.bmm(src.view(-1, 1, 3), torch.transpose(rot, -1, -2))
    return out.reshape(P, C, 3)


# ----------------------------------------------------------------------------- surface sampler
def jitter_candidates(samples: torch.Tensor, rho, m: int) -> torch.Tensor:
    """sampling.py:35-67: m candidates per sample, uniform direction, |N(0,rho)| radius."""
    n = samples.shape[0]
    theta = 2 * math.pi * torch.rand(n, m, dtype=samples.dtype)
    phi = torch.arccos(1 - 2 * torch.rand(n, m, dtype=samples.dtype))
    x = torch.sin(phi) * torch.cos(theta)
    y = torch.sin(phi) * torch.sin(theta)
    z = torch.cos(phi)
    r = torch.abs(torch.normal(0.0, rho, size=(n, m), dtype=samples.dtype))
    return samples[:, None] + torch.multiply(torch.stack((x, y, z), dim=-1), r[..., None])


def seeds_from_mask(f: fld.Field, n: int) -> torch.Tensor:
    """sampling.py:78-116: uniform points inside occupied mask voxels."""
    grid = f.mask_volume[0, 0]
    D, H, W = grid.shape
    idx = torch.stack(torch.meshgrid(torch.arange(0, D, dtype=torch.float32),
                                     torch.arange(0, H, dtype=torch.float32),
                                     torch.arange(0, W, dtype=torch.float32), indexing="ij"), dim=-1)
    idx = idx[..., [2, 1, 0]]
    occ = idx[grid.to(torch.bool)]
    pick = torch.randint(0, occ.shape[0], size=(n,), dtype=torch.long)
    s = occ[pick]
    s = s + torch.rand(*s.shape, dtype=s.dtype)
    shape_xyz = torch.tensor(grid.shape, dtype=torch.float32)[[2, 1, 0]]
    size = f.mask_aabb[1] - f.mask_aabb[0]
    return torch.divide(torch.multiply(size, s), shape_xyz - 1.0) + f.mask_aabb[0]


def uniform_in_aabb(f: fld.Field, n: int) -> torch.Tensor:
    """sampling.py:119-128."""
    return torch.multiply(torch.rand(n, 3, dtype=f.aabb.dtype), f.aabb[1] - f.aabb[0]) + f.aabb[0]


def sampling_epoch(f: fld.Field, samples, alpha_old, rho, max_iterations=100):
    """sampling.py:143-213.  Returns samples, alpha, iterations run, samples left invalid."""
    budget = samples.shape[0] * 5
    todo = torch.ones(samples.shape[0], dtype=torch.bool)
    ids = torch.arange(samples.shape[0], dtype=torch.long)
    thresh = torch.quantile(alpha_old, q=0.6)
    left = torch.count_nonzero(todo)
    it = 0
    while left != 0 and it < max_iterations:
        m = budget // left
        cand = jitter_candidates(samples[todo], rho, int(m))
        a_new = fld.compute_alpha(f, cand.view(-1, 3)).view(*cand.shape[:-1])
        order = torch.argsort(a_new, dim=-1, descending=True)
        hits = torch.argwhere(torch.take_along_dim(a_new, order, dim=-1) > thresh)
        last = torch.full(a_new.shape[:1], -1, dtype=hits.dtype)
        last.scatter_reduce_(0, hits[:, 0], hits[:, 1], reduce="amax", include_self=False)
        ok = last != -1
        n_ok = (torch.count_nonzero(ok).item(),)
        sel = torch.multiply(torch.rand(*n_ok, dtype=torch.float32), last[ok] + 0.99).to(torch.long)
        src = torch.take_along_dim(order[ok], sel[:, None], dim=-1)
        a_sel = torch.take_along_dim(a_new[ok], src, dim=-1)[:, 0]
        p_sel = torch.take_along_dim(cand[ok], src[..., None, :], dim=-2)[:, 0]
        tgt = ids[todo][ok]
        samples[tgt] = p_sel
        alpha_old[tgt] = a_sel
        todo[tgt] = False
        left = torch.count_nonzero(todo)
        it += 1
    return samples, alpha_old, it, left


def surface_samples(f: fld.Field, gen_points: int = 8000, n_iteration: int = 4,
                    max_resampling_iterations: int = 200, return_stats: bool = False):
    """sampling.py:509-532 (the debug history copies draw one rand block first, kept here)."""
    _ = uniform_in_aabb(f, gen_points)  # sampling.py:512-514 consumes RNG; result unused
    if f.mask_volume is not None:
        s = seeds_from_mask(f, gen_points)
        g = torch.tensor(f.grid, dtype=torch.long)
        rho = (torch.max(g) * 0.1) * torch.max(f.aabb_size / g)
    else:
        s = uniform_in_aabb(f, gen_points)
        rho = torch.linalg.norm(f.aabb_size)
    a = fld.compute_alpha(f, s)
    stats = []
    for _i in range(n_iteration):
        thr = torch.quantile(a, q=0.6).item()
        s, a, it, left = sampling_epoch(f, s, a, rho, max_iterations=max_resampling_iterations)
        stats.append((thr, it, int(left)))
    if return_stats:
        return s, a, stats
    return s


def point_normals(f: fld.Field, samples: torch.Tensor) -> torch.Tensor:
    """sampling.py:535-541."""
    return fld.compute_normals(f.head, fld.app_feature(f, fld.normalize_coord(f, samples)))


def emit_rays(f: fld.Field, samples: torch.Tensor, normals: torch.Tensor,
              rays_per_chunk: int = 10240, n_cells: int = 27):
    """sampling.py:442-488 + 237-251: 27 rays per surface point, coloured by the short march."""
    cells = isocell_dirs(n_cells, dtype=samples.dtype)
    dirs = rotate_isocell(cells, normals)
    dirs = torch.divide(dirs, torch.linalg.norm(dirs, dim=-1, keepdim=True))
    ori = torch.broadcast_to(samples[:, None], dirs.shape)
    step = rays_per_chunk // cells.shape[0]
    rgbs = []
    for lo in range(0, ori.shape[0], step):
        o = ori[lo:lo + step].reshape(-1, 3)
        d = dirs[lo:lo + step].reshape(-1, 3)
        rgb = fld.march(f, torch.cat((o, d), dim=-1).view(-1, 6), mode="point", n_samples=20)[0]
        rgbs.append(rgb.view(-1, 3))
    return ori.reshape(-1, 3), dirs.reshape(-1, 3), torch.cat(rgbs, 0).reshape(-1, 3)


def explore_model(f: fld.Field, gen_points: int = 20000) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """model_utils.py:22-33."""
    s = surface_samples(f, gen_points=gen_points, n_iteration=4, max_resampling_iterations=200)
    return emit_rays(f, s, point_normals(f, s))
