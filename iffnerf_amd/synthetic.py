"""Seeded synthetic inputs with the shapes of the reference's checkpoints.

No pretrained TensoRF checkpoint, DINOv2 weights or dataset is available offline (SURVEY.md
section 8c/8d), so tests, ``bench.py`` and the golden generator all draw their inputs here:

* ``make_field_ckpt``   -- a dictionary in the layout ``TensorBase.save`` writes
                           (reference models/tensorBase.py:424-442): ``kwargs``, ``state_dict``
                           with the VM planes/lines, ``basis_mat`` and the Ref head, and the
                           bit-packed occupancy mask.  The density field is a noisy blob so that
                           a closed surface exists for the surface sampler.
* ``make_id_weights``   -- the ``id_module.th`` ``model_state_dict`` entries on the path
                           (ray_preprocessor.*, attention.*).
* ``make_tokens``       -- stand-in for the DINOv2 patch tokens + the 14-channel position code.

Everything is numpy-seeded and device independent; nothing here touches the GPU.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import numpy as np
import torch

MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)


# ----------------------------------------------------------------------------- IDE constants
def ide_ml_pairs(deg_view: int = 4) -> np.ndarray:
    """(m, l) pairs of the integrated directional encoding: l = 1,2,4,.., m = 0..l -> [2, 19]."""
    ms, ls = [], []
    for i in range(deg_view):
        l = 2 ** i
        for m in range(l + 1):
            ms.append(m)
            ls.append(l)
    return np.asarray([ms, ls], dtype=np.int64)


def _gen_binom(a: float, k: int) -> float:
    p = 1.0
    for j in range(k):
        p *= (a - j)
    return p / math.factorial(k)


def ide_coeff_matrix(deg_view: int = 4) -> np.ndarray:
    """z-polynomial coefficients of the spherical harmonics Y_l^m used by the IDE -> [l_max+1, 19].

    Entry [k, i] multiplies z^k for pair i (Ref-NeRF eq. 4-6; the reference builds the same table
    in models/ref_utils.py:23-80).
    """
    ml = ide_ml_pairs(deg_view)
    l_max = 2 ** (deg_view - 1)
    mat = np.zeros((l_max + 1, ml.shape[1]), dtype=np.float64)
    for i, (m, l) in enumerate(ml.T):
        m, l = int(m), int(l)
        norm = math.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * math.pi * math.factorial(l + m)))
        for k in range(l - m + 1):
            leg = ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m)
                   * _gen_binom(0.5 * (l + k + m - 1.0), l))
            mat[k, i] = norm * leg
    return mat.astype(np.float32)


# ----------------------------------------------------------------------------- field checkpoint
def _linear(rng, out_f, in_f, gain=1.0):
    b = gain / math.sqrt(in_f)
    return (rng.uniform(-b, b, size=(out_f, in_f)).astype(np.float32),
            rng.uniform(-b, b, size=(out_f,)).astype(np.float32))


def make_field_ckpt(grid: Sequence[int] = (300, 300, 300),
                    aabb=((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)),
                    mask_res: Sequence[int] = (180, 180, 180),
                    seed: int = 1234,
                    n_sigma: int = 16, n_app: int = 48, app_dim: int = 27, feature_c: int = 128,
                    step_ratio: float = 0.5, density_shift: float = -10.0, distance_scale: float = 25.0,
                    contraction_type: str = "aabb", near_far=(2.0, 6.0),
                    blob_sigma: float = 0.42, peak: float = 32.0, mask_radius: float = 0.9,
                    view_pe: int = 2, fea_pe: int = 2, density_offset: float = 0.0) -> dict:
    """Build a TensorVMSplit-shaped checkpoint dictionary (see module docstring).

    ``grid`` is gridSize (x, y, z); plane i is [1,C,G[b],G[a]] with (a,b)=MAT_MODE[i], line i is
    [1,C,G[v],1] with v=VEC_MODE[i] (reference models/tensoRF.py:160-170).
    Density feature ~= peak * exp(-r^2/(2 blob_sigma^2)) * (1+noise) in normalised coordinates, so
    softplus(feature + shift) crosses from ~0 to large around r ~ 0.6-0.65: a closed surface.
    ``density_offset`` (for configs with ``density_shift = 0``, reference configs/bicycle.txt:28): channel 0 of every
    density plane becomes the constant ``density_offset / 3`` and channel 0 of every line 1, i.e. the offset is added to
    the feature by the tables themselves.
    """
    rng = np.random.default_rng(seed)
    G = [int(g) for g in grid]
    ax = [np.linspace(-1.0, 1.0, g, dtype=np.float64) for g in G]
    sd: Dict[str, torch.Tensor] = {}
    amp = (peak / (3.0 * n_sigma)) ** 0.5
    for i in range(3):
        a, b = MAT_MODE[i]
        v = VEC_MODE[i]
        ga = np.exp(-ax[a] ** 2 / (2 * blob_sigma ** 2))
        gb = np.exp(-ax[b] ** 2 / (2 * blob_sigma ** 2))
        gv = np.exp(-ax[v] ** 2 / (2 * blob_sigma ** 2))
        plane = amp * gb[None, :, None] * ga[None, None, :] * (1.0 + 0.15 * rng.standard_normal((n_sigma, G[b], G[a])))
        line = amp * gv[None, :] * (1.0 + 0.15 * rng.standard_normal((n_sigma, G[v])))
        if density_offset != 0.0:
            plane[0], line[0] = density_offset / 3.0, 1.0
        sd[f"density_plane.{i}"] = torch.from_numpy(plane.astype(np.float32))[None]
        sd[f"density_line.{i}"] = torch.from_numpy(line.astype(np.float32))[None, :, :, None]
    for i in range(3):
        a, b = MAT_MODE[i]
        v = VEC_MODE[i]
        # smooth-ish appearance: low-frequency pattern + noise, O(1) products after basis_mat
        pa = rng.standard_normal((n_app, 1, 1)) * np.cos(2.5 * ax[a])[None, None, :] \
            + rng.standard_normal((n_app, 1, 1)) * np.sin(2.0 * ax[b])[None, :, None]
        plane = 0.6 * pa + 0.25 * rng.standard_normal((n_app, G[b], G[a]))
        line = 0.8 + 0.3 * rng.standard_normal((n_app, G[v]))
        sd[f"app_plane.{i}"] = torch.from_numpy(plane.astype(np.float32))[None]
        sd[f"app_line.{i}"] = torch.from_numpy(line.astype(np.float32))[None, :, :, None]
    w, _ = _linear(rng, app_dim, 3 * n_app, gain=1.5)
    sd["basis_mat.weight"] = torch.from_numpy(w)
    # Ref head (reference models/ref.py:48-101 parameter names)
    sd["renderModule.dir_enc_fn.ml_array"] = torch.from_numpy(ide_ml_pairs(4))
    sd["renderModule.dir_enc_fn.mat"] = torch.from_numpy(ide_coeff_matrix(4))
    for name, (o, i_) in {"diffuse_color_mlp.0": (3, app_dim), "tint_color_mlp.0": (3, app_dim),
                          "roughness_mlp.0": (1, app_dim), "bottleneck_mlp": (feature_c, app_dim),
                          "normal_mlp.0": (3, app_dim),
                          "specular_mlp.0": (3, feature_c + 19 * 2 + 1)}.items():
        w, b = _linear(rng, o, i_, gain=2.0)
        sd[f"renderModule.{name}.weight"] = torch.from_numpy(w)
        sd[f"renderModule.{name}.bias"] = torch.from_numpy(b)
    # occupancy mask: ball of radius mask_radius (normalised) on its own grid, [D,H,W] = (z,y,x)
    mx, my, mz = [np.linspace(-1.0, 1.0, int(r)) for r in mask_res]
    occ = (mz[:, None, None] ** 2 + my[None, :, None] ** 2 + mx[None, None, :] ** 2) <= mask_radius ** 2
    aabb_t = torch.tensor(aabb, dtype=torch.float32)
    kwargs = {
        "aabb": aabb_t, "gridSize": G, "density_n_comp": [n_sigma] * 3, "appearance_n_comp": [n_app] * 3,
        "app_dim": app_dim, "contraction_type": contraction_type, "density_shift": density_shift,
        "alphaMask_thres": 0.001, "distance_scale": distance_scale, "rayMarch_weight_thres": 0.0001,
        "fea2denseAct": "softplus", "near_far": list(near_far), "step_ratio": step_ratio,
        "shadingMode": "Ref", "pos_pe": 6, "view_pe": view_pe, "fea_pe": fea_pe, "featureC": feature_c,
    }
    return {
        "model_name": "TensorVMSplit", "kwargs": kwargs, "state_dict": sd,
        "alphaMask.shape": occ.shape, "alphaMask.mask": np.packbits(occ.reshape(-1)),
        "alphaMask.aabb": aabb_t.clone(),
    }


# ----------------------------------------------------------------------------- identification weights
def make_id_weights(seed: int = 99, feature_c: int = 256, fea: int = 384, logit_gain: float = 6.0
                    ) -> Dict[str, torch.Tensor]:
    """ray_preprocessor.* / attention.* tensors of ``id_module.th`` (random, trained-like logit scale)."""
    rng = np.random.default_rng(seed)
    in_c = 141
    sd = {}
    for name, (o, i_) in {"ray_preprocessor.mlp.0": (feature_c, in_c), "ray_preprocessor.mlp.2": (feature_c, feature_c),
                          "ray_preprocessor.mlp2.0": (feature_c, feature_c + in_c),
                          "ray_preprocessor.mlp2.2": (fea, feature_c)}.items():
        w, b = _linear(rng, o, i_, gain=1.7)
        sd[name + ".weight"], sd[name + ".bias"] = torch.from_numpy(w), torch.from_numpy(b)
    for name, (o, i_) in {"attention.q_proj": (fea, fea + 14), "attention.k_proj": (fea, fea)}.items():
        lim = math.sqrt(6.0 / (o + i_)) * logit_gain
        sd[name + ".weight"] = torch.from_numpy(rng.uniform(-lim, lim, size=(o, i_)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy((0.05 * rng.standard_normal(o)).astype(np.float32))
    return sd


def make_tokens(m: int = 256, fea: int = 384, seed: int = 7) -> torch.Tensor:
    """Stand-in image tokens [m, fea+14]: N(0,1) features + the deterministic 14-ch position code of
    the first m cells of the 16x16 grid (reference identification_module.py:76-99,149-154)."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((m, fea)).astype(np.float32)
    lin = np.linspace(-1.0, 1.0, 16, dtype=np.float32)
    pos = np.stack(np.meshgrid(lin, lin, indexing="ij"), axis=-1).reshape(-1, 2)
    bands = (2.0 ** np.arange(3)).astype(np.float32)
    p = (pos[..., None] * bands).reshape(pos.shape[0], -1)
    pe = np.concatenate([pos, np.sin(p), np.cos(p)], axis=-1).astype(np.float32)
    reps = int(math.ceil(m / pe.shape[0]))
    pe = np.tile(pe, (reps, 1))[:m]
    return torch.from_numpy(np.concatenate([feats, pe], axis=-1))


# ----------------------------------------------------------------------------- BASELINE.json workloads
def n_to_reso(n_voxels: int, aabb) -> list:
    """gridSize for a voxel budget, as reference utils.py:20-24 (N_to_reso) derives it from the scene box."""
    a = torch.as_tensor(aabb, dtype=torch.float32)
    size = a[1] - a[0]
    voxel = (size.prod() / n_voxels).pow(1 / 3)
    return (size / voxel).long().tolist()


# T&T-shaped scene box: the reference reads ``bbox.txt`` of the scene and scales it by 1.2 (dataLoader/tankstemple.py:
# 113-119); the file is not available offline, so this is a fixed non-cubic box of Truck-like proportions (x 1.2 applied)
TRUCK_AABB = ((-1.38, -0.90, -1.86), (1.44, 1.08, 1.98))

# name -> what BASELINE.json's ``configs`` describe.  ``field``: make_field_ckpt arguments; ``gen_points`` x 27 = rays;
# ``queries``: query images per step of bench.py; ``shared_rays``: all queries of a step see ONE emitted ray set (the
# reference's eval semantics, train_eval_pose_est.py:131-149) instead of one fresh ray set per query.
WORKLOADS = {
    # configs[1]: "lego 800x800, 16k candidate rays": configs/lego.txt (300^3, blender box, near_far dataLoader/blender.py:41)
    "lego16k": dict(
        field=dict(grid=(300, 300, 300), aabb=((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)), mask_res=(180, 180, 180), seed=1234,
                   step_ratio=0.5, peak=20.0, near_far=(2.0, 6.0)),
        # 32 cold queries per step: the sampler's chain of ~37 short launches per step is the same for any batch, and 16 -> 32 queries
        # per step is +9 % poses/s (64: +12 %) at twice (four times) the latency of a step
        gen_points=593, queries=32, shared_rays=False,
        describe="lego-shaped TensorVMSplit 300^3 (16/48 comps, 180^3 mask), gen_points=593 -> 16011 rays"),
    # configs[2]: "truck (Tanks&Temples) 1920x1080, 32k rays": configs/truck.txt (27e6 voxels over a non-cubic box,
    # near_far dataLoader/tankstemple.py:113)
    "truck32k": dict(
        field=dict(grid=tuple(n_to_reso(27_000_000, TRUCK_AABB)), aabb=TRUCK_AABB, mask_res=(176, 124, 240), seed=4321,
                   step_ratio=0.5, peak=20.0, near_far=(0.01, 6.0)),
        gen_points=1186, queries=16, shared_rays=False,
        describe="truck-shaped TensorVMSplit 27e6 voxels over a non-cubic T&T box, gen_points=1186 -> 32022 rays"),
    # configs[4]: "mip360 bicycle (unbounded scene), 64k rays": configs/bicycle.txt (640^3, density_shift 0), scene box
    # and near_far of dataLoader/mip360.py:215-219, contraction_type='unisphere' (models/tensorBase.py:389-396)
    "bicycle64k": dict(
        field=dict(grid=(640, 640, 640), aabb=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), mask_res=(256, 256, 256), seed=777,
                   step_ratio=0.5, peak=20.0, near_far=(0.01, 1.4), contraction_type="unisphere", density_shift=0.0,
                   density_offset=-10.0, blob_sigma=0.30, mask_radius=0.62),
        gen_points=2371, queries=8, shared_rays=False,
        describe="bicycle-shaped TensorVMSplit 640^3, unisphere contraction, density_shift 0, gen_points=2371 -> 64017 rays"),
    # the reference's DEFAULT operating point: explore_model(model, gen_points=20000) (pose_estimation/model_utils.py:22-24) ->
    # 540 000 rays on the lego-shaped model; 553 MB of logits per query image (identification_module.py:165)
    "lego540k": dict(
        field=dict(grid=(300, 300, 300), aabb=((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)), mask_res=(180, 180, 180), seed=1234,
                   step_ratio=0.5, peak=20.0, near_far=(2.0, 6.0)),
        gen_points=20000, queries=2, shared_rays=False,
        describe="lego-shaped TensorVMSplit 300^3, the reference's default gen_points=20000 -> 540000 rays"),
    # configs[3]: "lego, batch of 64 query images, rays sharded across the GPUs": one emitted ray set per step
    "lego_b64": dict(
        field=dict(grid=(300, 300, 300), aabb=((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)), mask_res=(180, 180, 180), seed=1234,
                   step_ratio=0.5, peak=20.0, near_far=(2.0, 6.0)),
        gen_points=593, queries=64, shared_rays=True,
        describe="lego-shaped TensorVMSplit 300^3, ONE emitted ray set (16011 rays) per step shared by a batch of 64 query images"),
}


def make_workload_ckpt(name: str) -> dict:
    return make_field_ckpt(**WORKLOADS[name]["field"])
