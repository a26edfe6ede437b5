"""Oracle: ``F.grid_sample(..., mode='bilinear', padding_mode='zeros', align_corners=True)`` with explicit indices.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every table lookup of the path goes through this one third-party
operator (PyTorch aten, ``grid_sampler_2d`` / ``grid_sampler_3d``; the reference pins torch~=2.0.1 in
requirements_torch.txt:2, the authoring container has 2.10.0).  Its published semantics, restated in numpy so the
kernels' addressing (iffnerf_amd/csrc/iff_device.h: ``unnorm``, ``make_taps``, ``mask_value``) has a plain-text
specification that is itself checked against the real operator (tests/test_oracle_golden.py::test_gridsample_restatement):

  * a normalised coordinate c in [-1, 1] maps to the texel coordinate ((c + 1) / 2) * (size - 1);
  * i0 = floor(coordinate); the far tap i0 + 1 has weight (coordinate - i0), the near tap the complement;
  * a tap outside [0, size - 1] contributes 0 ("zeros" padding);
  * 2-D input [C,H,W] is read with (x, y) <-> (W, H); 3-D input [C,D,H,W] with (x, y, z) <-> (W, H, D);
  * the VM "line" lookup is the 2-D operator on a width-1 image at x = 0 (models/tensoRF.py:225): texel 0 gets weight 1.
"""
from __future__ import annotations

import numpy as np


def _taps(coord: np.ndarray, size: int):
    pos = ((coord.astype(np.float32) + np.float32(1)) / np.float32(2)) * np.float32(size - 1)
    lo = np.floor(pos)
    w_hi = (pos - lo).astype(np.float32)
    w_lo = (np.float32(1) - w_hi).astype(np.float32)
    lo = lo.astype(np.int64)
    return lo, w_lo, w_hi


def _fetch(img: np.ndarray, idx, sizes):
    """img [C, *sizes]; idx tuple of int arrays [n]; zero outside."""
    ok = np.ones(idx[0].shape, dtype=bool)
    for i, s in zip(idx, sizes):
        ok &= (i >= 0) & (i < s)
    safe = tuple(np.clip(i, 0, s - 1) for i, s in zip(idx, sizes))
    vals = img[(slice(None),) + safe]           # [C, n]
    return np.where(ok[None, :], vals, np.float32(0))


def grid_sample_2d(img: np.ndarray, xy: np.ndarray) -> np.ndarray:
    """img [C,H,W] float32, xy [n,2] normalised (x->W, y->H) -> [C,n]."""
    C, H, W = img.shape
    x0, wx0, wx1 = _taps(xy[:, 0], W)
    y0, wy0, wy1 = _taps(xy[:, 1], H)
    out = np.zeros((C, xy.shape[0]), dtype=np.float32)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            out += _fetch(img, (y0 + dy, x0 + dx), (H, W)) * (wy * wx)[None, :]
    return out


def grid_sample_3d(vol: np.ndarray, xyz: np.ndarray) -> np.ndarray:
    """vol [C,D,H,W] float32, xyz [n,3] normalised (x->W, y->H, z->D) -> [C,n]."""
    C, D, H, W = vol.shape
    x0, wx0, wx1 = _taps(xyz[:, 0], W)
    y0, wy0, wy1 = _taps(xyz[:, 1], H)
    z0, wz0, wz1 = _taps(xyz[:, 2], D)
    out = np.zeros((C, xyz.shape[0]), dtype=np.float32)
    for dz, wz in ((0, wz0), (1, wz1)):
        for dy, wy in ((0, wy0), (1, wy1)):
            for dx, wx in ((0, wx0), (1, wx1)):
                out += _fetch(vol, (z0 + dz, y0 + dy, x0 + dx), (D, H, W)) * (wz * wy * wx)[None, :]
    return out


def vm_density_feature(planes, lines, xn: np.ndarray) -> np.ndarray:
    """sum_i sum_c plane_i[c](xn[a], xn[b]) * line_i[c](xn[v]) with the reference's axis conventions
    (models/tensorBase.py:311-312, models/tensoRF.py:216-235).  planes[i] [C,G_b,G_a], lines[i] [C,G_v]."""
    mat, vec = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)
    out = np.zeros(xn.shape[0], dtype=np.float32)
    for i in range(3):
        p = grid_sample_2d(planes[i], xn[:, list(mat[i])])
        zeros = np.zeros(xn.shape[0], dtype=np.float32)
        l = grid_sample_2d(lines[i][:, :, None], np.stack([zeros, xn[:, vec[i]]], axis=1))
        out += (p * l).sum(0)
    return out
