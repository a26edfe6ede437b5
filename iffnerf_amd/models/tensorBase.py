"""Mirror of the reference's ``models/tensorBase.py`` for the hot path, over libiffnerf_hip.

Keeps the reference's names and call signatures: ``TensorBase`` with ``forward``,
``compute_alpha``, ``normalize_coord``, ``sample_point_color``, ``sample_ray``, ``save``/``load``/``get_kwargs``;
``AlphaGridMask``; ``positional_encoding``; ``raw2alpha``.  Parameters keep the reference's shapes and
``state_dict`` keys, so ``.th`` checkpoints load unchanged; the kernels read a re-laid-out copy owned by a
``FieldHandle`` that is rebuilt when the parameters change (``load`` / ``load_state_dict`` / ``invalidate_tables``).

Inference only: the reference's training-time members (upsampling, shrink, alpha-mask update, TV/L1 losses,
NDC / zip-nerf samplers) are outside the path and are not provided.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..hip_field import FieldHandle, MARCH_POINT, MARCH_SLAB
from .ref import Ref


def derive_step(aabb: torch.Tensor, grid_size, step_ratio: float, contraction_type: str = "aabb"):
    """stepSize / nSamples exactly as reference models/tensorBase.py:354-368 derives them (same float ops)."""
    aabb = torch.as_tensor(aabb, dtype=torch.float32).cpu()
    size = aabb[1] - aabb[0]
    g = torch.tensor([int(v) for v in grid_size], dtype=torch.long)
    if contraction_type == "unisphere":
        g = g * 0.5
    step = torch.mean(size / (g - 1)) * step_ratio
    diag = torch.sqrt(torch.sum(torch.square(size)))
    return step, int((diag / step).item()) + 1


def positional_encoding(positions, freqs):
    """[sin(x 2^k)] (coordinate-major, octave-minor) followed by the cosines; reference tensorBase.py:14-20.
    Utility kept for API compatibility -- the ray encoder computes it inside its HIP kernel (k5_ray_input)."""
    bands = (2 ** torch.arange(freqs, device=positions.device)).to(positions.dtype)
    ang = (positions[..., None] * bands).flatten(-2)
    return torch.cat((ang.sin(), ang.cos()), dim=-1)


def raw2alpha(sigma, dist):
    """alpha, weights, final transmittance of one ray batch; reference tensorBase.py:23-35.
    Utility kept for API compatibility -- the march kernel (k4_march, phase 2) does this per ray on the GPU."""
    alpha = 1.0 - torch.exp(-sigma * dist)
    ones = torch.ones_like(alpha[:, :1])
    trans = torch.cumprod(torch.cat((ones, 1.0 - alpha + 1e-10), dim=-1), dim=-1)
    return alpha, alpha * trans[:, :-1], trans[:, -1:]


class _MarchFeatures(torch.autograd.Function):
    """(feat28, depth, acc) = march(rays) with dL/d(o, d) from ``iff_march_grad`` (csrc/march_grad_kernels.hip).  depth is
    not differentiable (reference :903-905 computes it under no_grad); a 7th ray column (radius) receives zero."""

    @staticmethod
    def forward(ctx, rays, handle, mode, n_samples):
        feat28, depth, acc, S = handle.march_features(rays, mode, n_samples)
        ctx.handle, ctx.mode, ctx.S = handle, mode, S
        ctx.save_for_backward(rays.detach())
        ctx.mark_non_differentiable(depth)
        return feat28, depth, acc

    @staticmethod
    def backward(ctx, g_feat, g_depth, g_acc):
        (rays,) = ctx.saved_tensors
        g_feat = torch.zeros(rays.shape[0], 28, device=rays.device) if g_feat is None else g_feat
        g_acc = torch.zeros(rays.shape[0], device=rays.device) if g_acc is None else g_acc
        g6 = ctx.handle.march_grad(rays, ctx.mode, ctx.S, g_feat, g_acc)
        if rays.shape[1] > 6:
            g6 = torch.cat((g6, torch.zeros(rays.shape[0], rays.shape[1] - 6, device=g6.device, dtype=g6.dtype)), -1)
        return g6.to(rays.dtype), None, None, None


class AlphaGridMask(torch.nn.Module):
    """Occupancy volume of reference tensorBase.py:50-83; ``sample_alpha`` runs ``iff_mask_sample``."""

    def __init__(self, device, aabb, alpha_volume, contraction_type="aabb"):
        super().__init__()
        self.device = device
        self.contraction_type = contraction_type
        self.aabb = aabb.to(self.device)
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invgridSize = 1.0 / self.aabbSize * 2
        self.alpha_volume = alpha_volume.view(1, 1, *alpha_volume.shape[-3:])
        self.gridSize = torch.tensor([alpha_volume.shape[-1], alpha_volume.shape[-2], alpha_volume.shape[-3]],
                                     dtype=torch.int64).to(self.device)
        self._handle: Optional[FieldHandle] = None

    def _h(self) -> FieldHandle:
        if self._handle is None:
            self._handle = FieldHandle.mask_only(self.alpha_volume[0, 0], self.aabb, self.alpha_volume.device,
                                                 unisphere=self.contraction_type == "unisphere")
        return self._handle

    def sample_alpha(self, xyz_sampled):
        return self._h().mask_sample(xyz_sampled)

    def normalize_coord(self, xyz_sampled):
        return self._h().normalize_coord(xyz_sampled)


class TensorBase(torch.nn.Module):
    def __init__(self, aabb, gridSize, device, density_n_comp=8, appearance_n_comp=24, app_dim=27,
                 shadingMode="MLP_PE", alphaMask=None, near_far=[2.0, 6.0], density_shift=-10, alphaMask_thres=0.001,
                 distance_scale=25, rayMarch_weight_thres=0.0001, pos_pe=6, view_pe=6, fea_pe=6, featureC=128,
                 step_ratio=2.0, fea2denseAct="softplus", contraction_type="aabb", step_size_bg=0.1):
        super().__init__()
        if shadingMode != "Ref":
            raise RuntimeError(f"shadingMode={shadingMode!r}: only 'Ref' (what every reference config sets, "
                               "configs/*.txt) is built for the MI355X path")
        if fea2denseAct not in ("softplus", "relu"):
            raise RuntimeError(f"fea2denseAct={fea2denseAct!r} unsupported")
        self.density_n_comp, self.app_n_comp, self.app_dim = density_n_comp, appearance_n_comp, app_dim
        self.aabb = torch.as_tensor(aabb, dtype=torch.float32).to(device)
        self.alphaMask = alphaMask
        self.device = device
        self.density_shift, self.alphaMask_thres = density_shift, alphaMask_thres
        self.distance_scale, self.rayMarch_weight_thres = distance_scale, rayMarch_weight_thres
        self.fea2denseAct = fea2denseAct
        self.near_far, self.step_ratio = near_far, step_ratio
        self.contraction_type, self.step_size_bg = contraction_type, step_size_bg
        self.matMode, self.vecMode, self.comp_w = [[0, 1], [0, 2], [1, 2]], [2, 1, 0], [1, 1, 1]
        self.update_stepSize(gridSize)
        self.init_svd_volume(gridSize[0], device)
        self.shadingMode, self.pos_pe, self.view_pe, self.fea_pe, self.featureC = shadingMode, pos_pe, view_pe, fea_pe, featureC
        self.renderModule = Ref(self.app_dim, viewpe=view_pe, feature_c=featureC).to(device)
        self._handle: Optional[FieldHandle] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_tables())

    # ------------------------------------------------------------------ geometry bookkeeping (host scalars)
    def update_stepSize(self, gridSize):
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        self.gridSize = torch.tensor([int(g) for g in gridSize], dtype=torch.long, device=self.device)
        step, n = derive_step(self.aabb, gridSize, self.step_ratio, self.contraction_type)
        grid = self.gridSize * 0.5 if self.contraction_type == "unisphere" else self.gridSize
        self.units = self.aabbSize / (grid - 1)
        self.stepSize = step.to(self.aabb.device)
        self.aabbDiag = torch.sqrt(torch.sum(torch.square(self.aabbSize)))
        self.nSamples = n
        near, far = self.near_far
        self.n_samples_bg = (far - near) / self.step_size_bg if self.contraction_type == "unisphere" else 0
        self.invalidate_tables()

    def init_svd_volume(self, res, device):
        raise NotImplementedError

    def get_kwargs(self):
        return {
            "aabb": self.aabb, "gridSize": self.gridSize.tolist(), "density_n_comp": self.density_n_comp,
            "appearance_n_comp": self.app_n_comp, "app_dim": self.app_dim, "contraction_type": self.contraction_type,
            "density_shift": self.density_shift, "alphaMask_thres": self.alphaMask_thres,
            "distance_scale": self.distance_scale, "rayMarch_weight_thres": self.rayMarch_weight_thres,
            "fea2denseAct": self.fea2denseAct, "near_far": self.near_far, "step_ratio": self.step_ratio,
            "shadingMode": self.shadingMode, "pos_pe": self.pos_pe, "view_pe": self.view_pe, "fea_pe": self.fea_pe,
            "featureC": self.featureC,
        }

    # ------------------------------------------------------------------ checkpoint I/O (reference :424-458 layout)
    def save(self, path):
        ckpt = {"model_name": type(self).__name__, "kwargs": self.get_kwargs(), "state_dict": self.state_dict()}
        if self.alphaMask is not None:
            vol = self.alphaMask.alpha_volume.bool().cpu().numpy()
            ckpt["alphaMask.shape"] = vol.shape
            ckpt["alphaMask.mask"] = np.packbits(vol.reshape(-1))
            ckpt["alphaMask.aabb"] = self.alphaMask.aabb.cpu()
        torch.save(ckpt, path)

    def load(self, ckpt):
        if "alphaMask.aabb" in ckpt.keys():
            shape = tuple(int(s) for s in ckpt["alphaMask.shape"])
            bits = np.unpackbits(np.asarray(ckpt["alphaMask.mask"]))[:int(np.prod(shape))].reshape(shape)
            self.alphaMask = AlphaGridMask(self.device, torch.as_tensor(ckpt["alphaMask.aabb"]).to(self.device),
                                           torch.from_numpy(bits).float().to(self.device),
                                           contraction_type=self.contraction_type)
        self.load_state_dict(ckpt["state_dict"])
        self.invalidate_tables()

    # ------------------------------------------------------------------ kernel tables
    def invalidate_tables(self):
        """Drop the re-laid-out copy; call after changing parameters in place."""
        if getattr(self, "_handle", None) is not None:
            self._handle.close()
        self._handle = None

    def _apply(self, fn, *a, **k):   # .to()/.cuda() move the parameters: the tables follow lazily
        out = super()._apply(fn, *a, **k)
        self.invalidate_tables()
        return out

    def field_handle(self) -> FieldHandle:
        if self._handle is None:
            dev = self.density_plane[0].device
            if dev.type != "cuda":
                raise RuntimeError("the model's parameters are on the CPU: move it to the GPU (model.to('cuda')); "
                                   "libiffnerf_hip has no CPU path")
            mask = self.alphaMask
            self._handle = FieldHandle(
                device=dev, grid=self.gridSize.tolist(), aabb=self.aabb,
                density_plane=[p[0] for p in self.density_plane], density_line=[l[0, :, :, 0] for l in self.density_line],
                app_plane=[p[0] for p in self.app_plane], app_line=[l[0, :, :, 0] for l in self.app_line],
                basis=self.basis_mat.weight, head=self.renderModule.head_tensors(),
                mask_volume=None if mask is None else mask.alpha_volume[0, 0],
                mask_aabb=None if mask is None else mask.aabb,
                density_shift=self.density_shift, distance_scale=self.distance_scale,
                weight_thres=self.rayMarch_weight_thres, step_size=float(self.stepSize), n_samples=self.nSamples,
                near_far=self.near_far, softplus=self.fea2denseAct == "softplus",
                unisphere=self.contraction_type == "unisphere")
        return self._handle

    def _no_grad_only(self, what):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError(f"{what}: the MI355X path is inference-only; freeze the field as load_model does "
                               "(pose_estimation/model_utils.py:12-13) or call under torch.no_grad()")

    # ------------------------------------------------------------------ lookups
    def normalize_coord(self, xyz_sampled):
        return self.field_handle().normalize_coord(xyz_sampled)

    def compute_densityfeature(self, xyz_sampled):
        self._no_grad_only("compute_densityfeature")
        return self.field_handle().density_feature(xyz_sampled)

    def compute_appfeature(self, xyz_sampled):
        self._no_grad_only("compute_appfeature")
        return self.field_handle().app_feature(xyz_sampled)

    def feature2density(self, density_features):
        if self.fea2denseAct == "softplus":
            return torch.nn.functional.softplus(density_features + self.density_shift)
        return torch.relu(density_features)

    def compute_alpha(self, xyz_locs, length=1):
        self._no_grad_only("compute_alpha")
        return self.field_handle().point_alpha(xyz_locs, float(length))

    # ------------------------------------------------------------------ samplers (positions only; API compatibility)
    def sample_point_color(self, rays_o, rays_d, radii, N_samples=20, **kwargs):
        """Positions of the point-centred sampler (reference :623-638).  The march kernel derives the same
        positions itself; this method only serves callers that want them as tensors."""
        lo = N_samples // 2
        step = (self.stepSize * torch.arange(-lo, N_samples - lo, dtype=rays_o.dtype, device=rays_o.device))[None]
        pts = rays_o[..., None, :] + rays_d[..., None, :] * step[..., None]
        inside = ~((self.aabb[0] > pts) | (pts > self.aabb[1])).any(dim=-1)
        return pts, step, inside

    def sample_ray(self, rays_o, rays_d, radii, is_train=True, N_samples=-1):
        """Positions of the slab sampler (reference :494-536, 'aabb' contraction), without training jitter."""
        if is_train:
            raise RuntimeError("sample_ray(is_train=True): the jittered training sampler is outside the inference path")
        if self.contraction_type == "unisphere":
            raise RuntimeError("sample_ray: the reference's unisphere branch is unfinished (tensorBase.py:511-525)")
        n = N_samples if N_samples > 0 else self.nSamples
        near, far = self.near_far
        vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
        t0 = torch.minimum((self.aabb[1] - rays_o) / vec, (self.aabb[0] - rays_o) / vec).amax(-1).clamp(min=near, max=far)
        z = t0[..., None] + self.stepSize * torch.arange(n, dtype=rays_o.dtype, device=rays_o.device)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., None]
        inside = ~((self.aabb[0] > pts) | (pts > self.aabb[1])).any(dim=-1)
        return pts, z, inside

    # ------------------------------------------------------------------ the march
    def march(self, rays_chunk, point_centred: bool, N_samples=-1, white_bg=False, bg_color=None, want_alpha=False,
              want_counts=False):
        """One ``iff_march_shade`` launch -> (rgb, depth, acc, alpha|None, counts|None, S)."""
        self._no_grad_only("forward")
        if bg_color is None:
            bg = (1.0, 1.0, 1.0) if white_bg else (0.0, 0.0, 0.0)
        else:
            bg = [float(v) for v in torch.as_tensor(bg_color).reshape(-1).tolist()]
        return self.field_handle().march(rays_chunk, MARCH_POINT if point_centred else MARCH_SLAB, N_samples, bg,
                                         want_alpha=want_alpha, want_counts=want_counts)

    def _forward_ray_grad(self, rays_chunk, point, N_samples, white_bg, bg_color):
        """``forward`` for rays that require grad (inerf/estimate_pose_inerf.py:164-176 optimises the camera pose through
        ``model(rays_chunk, ...)``): the march and its gradient w.r.t. (o, d) are HIP launches (``_MarchFeatures``); the Ref
        head -- a per-ray MLP -- and the background blend (reference :889-901) run as torch ops on the GPU, so autograd
        also reaches the view directions.  Field parameters stay frozen (load_model, model_utils.py:12-13)."""
        h = self.field_handle()
        mode = MARCH_POINT if point else MARCH_SLAB
        S = int(N_samples) if N_samples > 0 else (20 if point else h.n_samples_default)
        feat28, depth, acc = _MarchFeatures.apply(rays_chunk, h, mode, S)
        d = rays_chunk[:, 3:6]
        # rays_to_consider (:887) as a mask instead of an index list: no host synchronisation, so the whole iNeRF iteration
        # can be captured in a hipGraph; the head of an unshaded ray (F = 0) is finite and its gradient is masked out
        considered = (feat28[:, 27] > 0)[:, None]
        rgb = torch.where(considered, self.renderModule.forward_autograd(d, feat28[:, :27]), torch.zeros((), device=acc.device))
        if bg_color is None:
            bg_color = torch.ones(3, device=rgb.device) if white_bg else torch.zeros(3, device=rgb.device)
        rgb = (rgb * acc[..., None] + torch.as_tensor(bg_color, device=rgb.device) * (1.0 - acc[..., None])).clamp(0, 1)
        o = rays_chunk[:, :3]
        if point:
            z_vals = (self.stepSize * torch.arange(-(S // 2), S - S // 2, dtype=o.dtype, device=o.device))[None]
        else:
            z_vals = self.sample_ray(o, d, None, is_train=False, N_samples=S)[1]
        dists = torch.cat((z_vals[:, 1:] - z_vals[:, :-1], torch.zeros_like(z_vals[:, :1])), dim=-1)
        return rgb, depth, acc, None, z_vals, dists

    def forward(self, rays_chunk, white_bg=False, bg_color=None, is_train=False, ndc_ray=False, sample_func=None,
                N_samples=-1):
        """Reference :775-917.  Returns (rgb_map, depth_map, acc_map, alpha, z_vals, dists); with rays that require grad
        (``_forward_ray_grad``) ``alpha`` is None -- the per-sample alphas are not kept for the backward pass."""
        if is_train or ndc_ray:
            raise RuntimeError("forward(is_train/ndc_ray=True) is outside the inference path built for MI355X")
        if sample_func is None:
            point = False
        elif getattr(sample_func, "__func__", None) is TensorBase.sample_point_color and sample_func.__self__ is self:
            point = True
        else:
            raise RuntimeError("forward: sample_func must be None (slab sampler) or this model's sample_point_color")
        if torch.is_grad_enabled() and rays_chunk.requires_grad:
            self._no_grad_only("forward")
            return self._forward_ray_grad(rays_chunk, point, N_samples, white_bg, bg_color)
        rgb, depth, acc, alpha, _, S = self.march(rays_chunk, point, N_samples, white_bg, bg_color, want_alpha=True)
        o, d = rays_chunk[:, :3], rays_chunk[:, 3:6]
        if point:
            z_vals = (self.stepSize * torch.arange(-(S // 2), S - S // 2, dtype=o.dtype, device=o.device))[None]
        else:
            z_vals = self.sample_ray(o, d, None, is_train=False, N_samples=S)[1]
        dists = torch.cat((z_vals[:, 1:] - z_vals[:, :-1], torch.zeros_like(z_vals[:, :1])), dim=-1)
        return rgb, depth, acc, alpha, z_vals, dists
