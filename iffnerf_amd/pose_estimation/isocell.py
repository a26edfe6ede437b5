"""Mirror of ``pose_estimation/isocell.py`` for the path: the deterministic iso-cell table and its rotation.

``isocell_distribution`` (reference :6-68) is evaluated in closed form on the host for the only configuration the path
uses (``isrand=-1``); ``rotate_isocell`` (reference :144-171) runs ``iff_isocell_emit`` on the GPU.
"""
from __future__ import annotations

import math

import torch


def isocell_distribution(ray_target, dtype, device, N0=3, isrand=-1, int_dtype=torch.int64):
    """Cell centres on the +z hemisphere: ring r (1..n) has N0(2r-1) cells at radius (r-1/2)/n, angles (k+1/2)*2pi/nc."""
    if isrand != -1:
        raise RuntimeError("isocell_distribution: only the deterministic layout (isrand=-1) is on the IFFNeRF path "
                           "(pose_estimation/sampling.py:229-234)")
    n = int(math.ceil(math.sqrt(ray_target / N0)))
    xs, ys = [], []
    for ring in range(1, n + 1):
        cells = N0 * (2 * ring - 1)
        dth = torch.tensor(2 * math.pi, dtype=dtype) / torch.tensor(float(cells), dtype=dtype)
        k = torch.arange(cells, dtype=dtype)
        theta = k * dth + dth / 2
        radius = torch.tensor(ring, dtype=torch.int64) * (1 / n) - (1 / n) / 2
        xs.append(radius * torch.cos(theta))
        ys.append(radius * torch.sin(theta))
    x, y = torch.cat(xs), torch.cat(ys)
    z = torch.sqrt(torch.clamp(1 - x * x - y * y, min=0.0))
    return torch.stack((x, y, z), dim=1).to(device=device, dtype=dtype)


def rotate_isocell(isocell_directions: torch.Tensor, normal: torch.Tensor):
    """[C,3] directions, [P,3] normals -> [P,C,3]: rotation taking +z to -normal applied to every direction.

    The kernel also renormalises (the reference does that in its caller, sampling.py:455-457); a second
    normalisation there is a no-op up to rounding."""
    from ..hip_field import isocell_emit
    _, dirs = isocell_emit(isocell_directions, torch.zeros_like(normal), normal)
    return dirs.view(normal.shape[0], isocell_directions.shape[0], 3)
