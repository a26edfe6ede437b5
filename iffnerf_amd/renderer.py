"""Mirror of the reference's ``renderer.OctreeRender_trilinear_fast`` (renderer.py:12-25).

The reference loops over 4096-ray chunks; the march kernel has no such limit, so the whole batch is one
``iff_march_shade`` launch (``chunk`` is accepted and ignored -- results do not depend on it).
"""
from __future__ import annotations

import torch


def OctreeRender_trilinear_fast(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, bg_color=None, white_bg=None,
                                is_train=False, device="cuda"):
    if ndc_ray or is_train:
        raise RuntimeError("OctreeRender_trilinear_fast: ndc_ray / is_train are outside the inference path")
    rays = rays.to(device)
    rgb, depth, _, _, _, _ = tensorf.march(rays, point_centred=False, N_samples=N_samples, white_bg=bool(white_bg),
                                           bg_color=bg_color)
    return rgb, None, depth, None, None
