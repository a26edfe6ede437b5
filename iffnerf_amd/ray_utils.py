"""The two functions of the reference's ``ray_utils.py`` the iNeRF refinement calls (:28-58, :61-100): per-pixel camera-frame
directions from an intrinsic matrix, and their rotation into world rays with mip-NeRF pixel radii."""
import math

import torch


def get_ray_directions_Ks(H: int, W: int, K: torch.Tensor, use_pixel_centers=True):
    """K [n,3,3] -> (directions, dx, dy), each [n,H,W,3] = K^-1 (u, v, 1) at the pixel, its right and its lower neighbour."""
    c = 0.5 if use_pixel_centers else 0.0
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=K.device) + c,
                          torch.arange(W, dtype=torch.float32, device=K.device) + c, indexing="ij")
    one = torch.ones_like(u)
    pix = torch.stack((torch.stack((u, v, one), -1), torch.stack((u + 1, v, one), -1), torch.stack((u, v + 1, one), -1)))
    cam = torch.einsum("nij,shwj->nshwi", torch.inverse(K), pix)                      # [n,3,H,W,3]
    return cam[:, 0], cam[:, 1], cam[:, 2]


def get_rays(viewdirs, c2w, keepdim=False, directions=None, dx=None, dy=None):
    """Rotate camera-frame directions by c2w[..., :3, :3]; origins = c2w[..., :3, 3].  With dx, dy also the pixel radius
    0.5 (|dx - d| + |dy - d|) * 2 / sqrt(12) (:92-99)."""
    if viewdirs.shape[-1] != 3 or (dx is None) != (dy is None):
        raise RuntimeError("get_rays: viewdirs must be [...,3]; dx and dy come together")
    R = c2w[..., :3, :3]
    rot = lambda x: (x[..., None, :] * R).sum(-1)      # noqa: E731
    rays_d = rot(viewdirs)
    base = rot(directions) if directions is not None else rays_d
    rays_o = c2w[..., :3, 3].unsqueeze(-2).expand(rays_d.shape)
    if dx is not None:
        dx, dy = rot(dx), rot(dy)
    if not keepdim:
        rays_o, rays_d, base = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), base.reshape(-1, 3)
        if dx is not None:
            dx, dy = dx.reshape(-1, 3), dy.reshape(-1, 3)
    if dx is None:
        return rays_o, rays_d
    radii = (0.5 * (torch.linalg.norm(dx - base, dim=-1) + torch.linalg.norm(dy - base, dim=-1))[..., None]) * (2 / math.sqrt(12))
    return rays_o, rays_d, radii
