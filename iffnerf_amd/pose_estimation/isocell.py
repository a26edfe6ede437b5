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
    # one entry per cell: its ring (1-based), the ring's cell count and the cell's position inside the ring.
    # Tensor ops (not per-ring Python scalars) so the float32 roundings equal the reference's table bit for bit.
    ring = torch.cat([torch.full((N0 * (2 * r - 1),), r, dtype=int_dtype) for r in range(1, n + 1)])
    count = N0 * (2 * ring - 1)
    within = torch.cat([torch.arange(N0 * (2 * r - 1), dtype=int_dtype) for r in range(1, n + 1)]).to(dtype)
    step = 2 * math.pi / count.to(dtype)
    theta = (0 + within * step) + step / 2
    radius = ring * (1 / n) - (1 / n) / 2
    x, y = radius * torch.cos(theta), radius * torch.sin(theta)
    z = torch.real(torch.sqrt(1 - torch.square(x.to(torch.complex64)) - torch.square(y.to(torch.complex64))))
    return torch.stack((x, y, z), dim=1).to(device=device, dtype=dtype)


def rotate_isocell(isocell_directions: torch.Tensor, normal: torch.Tensor):
    """[C,3] directions, [P,3] normals -> [P,C,3]: rotation taking +z to -normal applied to every direction.

    The kernel also renormalises (the reference does that in its caller, sampling.py:455-457); a second
    normalisation there is a no-op up to rounding."""
    from ..hip_field import isocell_emit
    _, dirs = isocell_emit(isocell_directions, torch.zeros_like(normal), normal)
    return dirs.view(normal.shape[0], isocell_directions.shape[0], 3)
