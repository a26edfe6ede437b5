"""Oracle: tensorial radiance field (VM-split) lookups, alpha, march and Ref shading.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional restatement on torch-CPU of
  models/tensorBase.py:23-35,50-83,354-368,389-397,494-536,623-638,750-917
  models/tensoRF.py:216-256
  models/ref.py:103-155, models/ref_utils.py:6-18,82-112, models/image.py:6-13
  utils.py:139-146
of the reference.  A field is a plain ``Field`` record built from the reference's checkpoint
dictionary (tensorBase.py:424-442), so the same fixtures drive oracle and product.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field as _dc_field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MAT_MODE = ((0, 1), (0, 2), (1, 2))  # tensorBase.py:311
VEC_MODE = (2, 1, 0)                 # tensorBase.py:312


@dataclass
class Field:
    aabb: torch.Tensor                      # [2,3]
    grid: Tuple[int, int, int]              # gridSize (x,y,z)
    density_plane: List[torch.Tensor]       # 3 x [1,Cd,G_b,G_a]
    density_line: List[torch.Tensor]        # 3 x [1,Cd,G_v,1]
    app_plane: List[torch.Tensor]           # 3 x [1,Ca,G_b,G_a]
    app_line: List[torch.Tensor]            # 3 x [1,Ca,G_v,1]
    basis: torch.Tensor                     # [app_dim, 3*Ca]
    head: Dict[str, torch.Tensor]           # Ref head weights (renderModule.*)
    mask_volume: Optional[torch.Tensor]     # [1,1,D,H,W] float {0,1}
    mask_aabb: Optional[torch.Tensor]
    density_shift: float = -10.0
    distance_scale: float = 25.0
    weight_thres: float = 1e-4
    step_ratio: float = 2.0
    fea2dense: str = "softplus"
    contraction: str = "aabb"
    near_far: Tuple[float, float] = (2.0, 6.0)
    step_size_bg: float = 0.1
    # derived (tensorBase.py:354-368)
    aabb_size: torch.Tensor = _dc_field(default=None)
    inv_aabb: torch.Tensor = _dc_field(default=None)
    step_size: torch.Tensor = _dc_field(default=None)
    n_samples: int = 0

    def __post_init__(self):
        self.aabb_size = self.aabb[1] - self.aabb[0]
        self.inv_aabb = 2.0 / self.aabb_size
        g = torch.tensor(self.grid, dtype=torch.long)
        if self.contraction == "unisphere":
            g = g * 0.5
        units = self.aabb_size / (g - 1)
        self.step_size = torch.mean(units) * self.step_ratio
        diag = torch.sqrt(torch.sum(torch.square(self.aabb_size)))
        self.n_samples = int((diag / self.step_size).item()) + 1


def field_from_ckpt(ckpt: dict) -> Field:
    """Checkpoint dictionary (tensorBase.py:424-442 layout) -> Field (cf. :444-458)."""
    kw = ckpt["kwargs"]
    sd = {k: torch.as_tensor(v).float() if torch.as_tensor(v).is_floating_point()
          else torch.as_tensor(v) for k, v in ckpt["state_dict"].items()}
    head = {k[len("renderModule."):]: v for k, v in sd.items() if k.startswith("renderModule.")}
    mask_volume = mask_aabb = None
    if "alphaMask.aabb" in ckpt:
        shape = tuple(int(s) for s in ckpt["alphaMask.shape"])
        n = int(np.prod(shape))
        bits = np.unpackbits(np.asarray(ckpt["alphaMask.mask"]))[:n].reshape(shape)
        mask_volume = torch.from_numpy(bits).float().view(1, 1, *shape[-3:])
        mask_aabb = torch.as_tensor(ckpt["alphaMask.aabb"]).float()
    return Field(
        aabb=torch.as_tensor(kw["aabb"]).float().cpu(),
        grid=tuple(int(g) for g in kw["gridSize"]),
        density_plane=[sd[f"density_plane.{i}"] for i in range(3)],
        density_line=[sd[f"density_line.{i}"] for i in range(3)],
        app_plane=[sd[f"app_plane.{i}"] for i in range(3)],
        app_line=[sd[f"app_line.{i}"] for i in range(3)],
        basis=sd["basis_mat.weight"],
        head=head,
        mask_volume=mask_volume,
        mask_aabb=mask_aabb,
        density_shift=float(kw.get("density_shift", -10)),
        distance_scale=float(kw.get("distance_scale", 25)),
        weight_thres=float(kw.get("rayMarch_weight_thres", 1e-4)),
        step_ratio=float(kw.get("step_ratio", 2.0)),
        fea2dense=kw.get("fea2denseAct", "softplus"),
        contraction=kw.get("contraction_type", "aabb"),
        near_far=tuple(float(v) for v in kw.get("near_far", (2.0, 6.0))),
    )


def field_as_float64(f: Field) -> Field:
    """The same field with every table, weight and bound held in float64 -- the REFEREE of the full-size tests (which of two fp32
    evaluations is nearer the exact value of the reference's formulas), not a restatement of anything the reference runs."""
    import dataclasses
    up = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t
    changes = {}
    for fl in dataclasses.fields(f):
        v = getattr(f, fl.name)
        if torch.is_tensor(v):
            changes[fl.name] = up(v)
        elif isinstance(v, list):
            changes[fl.name] = [up(t) for t in v]
        elif isinstance(v, dict):
            changes[fl.name] = {k: up(t) for k, t in v.items()}
    return dataclasses.replace(f, **changes)          # __post_init__ re-derives step_size and friends from the float64 aabb


# ----------------------------------------------------------------------------- coordinates
def contract_power(x: torch.Tensor, alpha: float = -1.5) -> torch.Tensor:
    """utils.py:139-146 (power_transformation)."""
    mag = torch.abs(x)
    na = math.fabs(alpha - 1)
    return torch.sign(x) * (na / alpha) * (torch.pow((mag / na) + 1.0, alpha) - 1.0)


def normalize_coord(f: Field, xyz: torch.Tensor) -> torch.Tensor:
    """tensorBase.py:389-397."""
    if f.contraction == "unisphere":
        centre = (f.aabb[0] + f.aabb[1]) / 2.0
        return contract_power(xyz - centre, alpha=-1.5)
    return (xyz - f.aabb[0]) * f.inv_aabb - 1


def mask_normalize(f: Field, xyz: torch.Tensor) -> torch.Tensor:
    """tensorBase.py:57-59,74-83 -- the mask keeps its own aabb and (1/size)*2 scale."""
    if f.contraction == "unisphere":
        centre = (f.mask_aabb[0] + f.mask_aabb[1]) / 2.0
        return contract_power(xyz - centre, alpha=-1.5)
    size = f.mask_aabb[1] - f.mask_aabb[0]
    inv = 1.0 / size * 2
    return (xyz - f.mask_aabb[0]) * inv - 1


def mask_sample(f: Field, xyz: torch.Tensor) -> torch.Tensor:
    """tensorBase.py:66-72: trilinear read of the {0,1} occupancy volume."""
    g = mask_normalize(f, xyz)
    return F.grid_sample(f.mask_volume, g.view(1, -1, 1, 1, 3), align_corners=True).view(-1)


# ----------------------------------------------------------------------------- VM lookups
def _vm_coords(xn: torch.Tensor):
    planes = torch.stack([xn[..., list(MAT_MODE[i])] for i in range(3)]).view(3, -1, 1, 2)
    lines = torch.stack([xn[..., VEC_MODE[i]] for i in range(3)])
    lines = torch.stack((torch.zeros_like(lines), lines), dim=-1).view(3, -1, 1, 2)
    return planes, lines


def density_feature(f: Field, xn: torch.Tensor) -> torch.Tensor:
    """tensoRF.py:216-235.  xn: normalised [n,3] -> [n]."""
    planes, lines = _vm_coords(xn)
    n = xn.shape[0]
    out = torch.zeros((n,), dtype=xn.dtype)
    for i in range(3):
        p = F.grid_sample(f.density_plane[i], planes[[i]], align_corners=True).view(-1, n)
        l = F.grid_sample(f.density_line[i], lines[[i]], align_corners=True).view(-1, n)
        out = out + torch.sum(p * l, dim=0)
    return out


def app_products(f: Field, xn: torch.Tensor) -> torch.Tensor:
    """The 3*Ca plane*line products of tensoRF.py:248-256, as [n, 3*Ca]."""
    planes, lines = _vm_coords(xn)
    n = xn.shape[0]
    ps, ls = [], []
    for i in range(3):
        ps.append(F.grid_sample(f.app_plane[i], planes[[i]], align_corners=True).view(-1, n))
        ls.append(F.grid_sample(f.app_line[i], lines[[i]], align_corners=True).view(-1, n))
    return (torch.cat(ps) * torch.cat(ls)).T


def app_feature(f: Field, xn: torch.Tensor) -> torch.Tensor:
    """tensoRF.py:237-256: basis_mat applied to the plane*line products -> [n, app_dim]."""
    return F.linear(app_products(f, xn), f.basis)


def feature2density(f: Field, feat: torch.Tensor) -> torch.Tensor:
    """tensorBase.py:750-754."""
    if f.fea2dense == "softplus":
        return F.softplus(feat + f.density_shift)
    return F.relu(feat)


def compute_alpha(f: Field, xyz: torch.Tensor, length: float = 1) -> torch.Tensor:
    """tensorBase.py:756-773."""
    if f.mask_volume is not None:
        keep = mask_sample(f, xyz) > 0
    else:
        keep = torch.ones_like(xyz[:, 0], dtype=torch.bool)
    sigma = torch.zeros(xyz.shape[:-1], dtype=xyz.dtype)
    if keep.any():
        sigma[keep] = feature2density(f, density_feature(f, normalize_coord(f, xyz[keep])))
    return 1 - torch.exp(-sigma * length).view(xyz.shape[:-1])


# ----------------------------------------------------------------------------- samplers
def sample_point_centred(f: Field, o: torch.Tensor, d: torch.Tensor, n_samples: int = 20):
    """tensorBase.py:623-638: n_samples positions centred on the ray origin."""
    before = n_samples // 2
    offs = (f.step_size * torch.arange(-before, n_samples - before, dtype=o.dtype)[None])
    pts = o[..., None, :] + d[..., None, :] * offs[..., None]
    outside = ((f.aabb[0] > pts) | (pts > f.aabb[1])).any(dim=-1)
    return pts, offs, ~outside


def sample_slab(f: Field, o: torch.Tensor, d: torch.Tensor, n_samples: int = -1):
    """tensorBase.py:494-536 with is_train=False, contraction 'aabb'."""
    n = n_samples if n_samples > 0 else f.n_samples
    near, far = f.near_far
    vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
    ra = (f.aabb[1] - o) / vec
    rb = (f.aabb[0] - o) / vec
    t0 = torch.minimum(ra, rb).amax(-1).clamp(min=near, max=far)
    z = t0[..., None] + torch.multiply(f.step_size, torch.arange(n, dtype=o.dtype))
    pts = o[..., None, :] + d[..., None, :] * z[..., None]
    outside = ((f.aabb[0] > pts) | (pts > f.aabb[1])).any(dim=-1)
    return pts, z, ~outside


def alpha_compositing(sigma: torch.Tensor, dist: torch.Tensor):
    """tensorBase.py:23-35 (raw2alpha)."""
    alpha = 1.0 - torch.exp(-sigma * dist)
    trans = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1, dtype=alpha.dtype), 1.0 - alpha + 1e-10], -1), -1)
    return alpha, alpha * trans[:, :-1], trans[:, -1:]


# ----------------------------------------------------------------------------- Ref head
def srgb(lin: torch.Tensor) -> torch.Tensor:
    """models/image.py:6-13."""
    eps = torch.finfo(lin.dtype).eps
    lo = 323 / 25 * lin
    hi = (211 * torch.clamp(lin, min=eps) ** (5 / 12) - 11) / 200
    return torch.where(lin <= 0.0031308, lo, hi)


def integrated_dir_enc(head: Dict[str, torch.Tensor], dirs: torch.Tensor, kinv: torch.Tensor):
    """models/ref_utils.py:82-112 -> [n,19,2]."""
    ml = head["dir_enc_fn.ml_array"]
    mat = head["dir_enc_fn.mat"]
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    vmz = torch.pow(z, torch.arange(mat.shape[0], dtype=z.dtype)[None, :])
    vmxy = torch.pow((x + 1j * y), ml[0, :])
    harm = vmxy * torch.matmul(vmz, mat)
    att = 0.5 * ml[1, :] * (ml[1, :] + 1)
    return torch.view_as_real(harm * torch.exp(-att * kinv))


def ref_normals_raw(head, feat):
    """normal_mlp of models/ref.py:85-89: Linear -> normalise -> negate."""
    n = F.linear(feat, head["normal_mlp.0.weight"], head["normal_mlp.0.bias"])
    return F.normalize(n, p=2, dim=-1) * -1


def compute_normals(head, feat):
    """models/ref.py:154-155."""
    return -ref_normals_raw(head, feat)


def ref_shade(head: Dict[str, torch.Tensor], viewdirs: torch.Tensor, feat: torch.Tensor):
    """models/ref.py:103-152 with normals=None."""
    nrm = ref_normals_raw(head, feat)
    tint = torch.sigmoid(F.linear(feat, head["tint_color_mlp.0.weight"], head["tint_color_mlp.0.bias"]))
    rough = F.softplus(F.linear(feat, head["roughness_mlp.0.weight"], head["roughness_mlp.0.bias"]) + -1.0)
    bott = F.linear(feat, head["bottleneck_mlp.weight"], head["bottleneck_mlp.bias"])
    v = -viewdirs
    ndv = torch.bmm(nrm.view(-1, 1, 3), v.view(-1, 3, 1))[..., 0]
    refl = torch.multiply(2.0 * ndv, nrm) - v                      # ref_utils.py:18
    enc = integrated_dir_enc(head, refl, rough)
    dot = torch.bmm(nrm.view(-1, 1, 3), viewdirs.view(-1, 3, 1))[..., 0]
    x = torch.cat([bott, enc.view(enc.shape[0], -1), dot], dim=-1)
    spec = torch.sigmoid(F.linear(x, head["specular_mlp.0.weight"], head["specular_mlp.0.bias"]))
    diff = torch.sigmoid(
        F.linear(feat, head["diffuse_color_mlp.0.weight"], head["diffuse_color_mlp.0.bias"]) + -math.log(3.0))
    rgb = torch.clip(srgb(tint * spec + diff), 0.0, 1.0)
    return rgb * (1 + 2 * 0.001) - 0.001


# ----------------------------------------------------------------------------- march
def march(f: Field, rays: torch.Tensor, mode: str = "point", n_samples: int = -1,
          white_bg: bool = False, bg_color: Optional[torch.Tensor] = None):
    """TensorBase.forward, tensorBase.py:775-917 (is_train=False, ndc_ray=False).

    mode "point": sample_func=sample_point_color; mode "slab": default sample_ray.
    Returns rgb, depth, acc, alpha, z_vals, dists plus per-ray (n_valid, n_app) counts.
    """
    o, d = rays[:, :3], rays[:, 3:6]
    if mode == "point":
        pts, z, valid = sample_point_centred(f, o, d, n_samples if n_samples > 0 else 20)
    else:
        pts, z, valid = sample_slab(f, o, d, n_samples)
    dists = torch.cat((z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])), dim=-1)
    if f.mask_volume is not None:
        m = mask_sample(f, pts[valid]) > 0
        bad = ~valid
        bad[valid] |= ~m
        valid = ~bad
    sigma = torch.zeros(pts.shape[:-1], dtype=pts.dtype)
    if valid.any():
        pts = normalize_coord(f, pts)
        sigma[valid] = feature2density(f, density_feature(f, pts[valid]))
    alpha, weight, _ = alpha_compositing(sigma, dists * f.distance_scale)
    shade = weight > f.weight_thres
    feats = torch.zeros((*pts.shape[:2], f.basis.shape[0]), dtype=pts.dtype)
    if shade.any():
        feats[shade] = app_feature(f, pts[shade])
    consider = shade.any(dim=-1)
    acc = torch.sum(weight, -1)
    ray_feat = torch.sum(weight[..., None] * feats, -2)
    rgb = torch.zeros((d.shape[0], 3), dtype=d.dtype)
    rgb[consider] = ref_shade(f.head, d[consider], ray_feat[consider])
    if bg_color is None:
        bg_color = torch.ones(3, dtype=d.dtype) if white_bg else torch.zeros(3, dtype=d.dtype)
    rgb = (rgb * acc[..., None] + bg_color * (1.0 - acc[..., None])).clamp(0, 1)
    depth = torch.sum(weight * z, -1)
    depth = depth + (1.0 - acc) * rays[..., -1]
    counts = torch.stack((valid.sum(-1), shade.sum(-1)), dim=-1)
    return rgb, depth, acc, alpha, z, dists, counts
