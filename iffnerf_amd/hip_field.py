"""Python owner of an ``iff_field`` handle and thin tensor-in/tensor-out wrappers over the C ABI.

Device memory, streams and tensor allocation come from PyTorch-ROCm; all arithmetic happens in
libiffnerf_hip.so.  The nn.Module mirrors in ``iffnerf_amd.models`` hold one ``FieldHandle`` each.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check, dptr, fvec, stream_ptr

MARCH_POINT, MARCH_SLAB = 0, 1


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class FieldHandle:
    """Owns the re-laid-out tables of one TensorVMSplit + AlphaGridMask + Ref head on one GPU."""

    def __init__(self, *, device, grid: Sequence[int], aabb: torch.Tensor,
                 density_plane, density_line, app_plane, app_line, basis: torch.Tensor, head: Dict[str, torch.Tensor],
                 mask_volume: Optional[torch.Tensor], mask_aabb: Optional[torch.Tensor],
                 density_shift: float, distance_scale: float, weight_thres: float, step_size: float, n_samples: int,
                 near_far, softplus: bool = True, unisphere: bool = False, density_lanes: int = 0, head_lanes: int = 0,
                 sampler_persistent: bool = False, fan_waves: int = 0):
        self._h = None
        L = _lib.lib()
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"FieldHandle needs a GPU device (got {device}); libiffnerf_hip has no CPU path")
        self.device = device
        keep = []   # source tensors must outlive iff_field_create

        def dev(t, name):
            t = _f32c(t, device)
            keep.append(t)
            return dptr(t, name=name)

        d = _lib.FieldDesc()
        d.grid[:] = [int(g) for g in grid]
        aabb = torch.as_tensor(aabb, dtype=torch.float32).cpu().reshape(-1)
        d.aabb[:] = [float(v) for v in aabb]
        d.n_density = int(density_plane[0].shape[-3])
        d.n_app = int(app_plane[0].shape[-3])
        d.app_dim = int(basis.shape[0])
        d.feature_c = int(head["bottleneck_mlp.weight"].shape[0])
        for i in range(3):
            d.density_plane[i] = dev(density_plane[i], f"density_plane[{i}]")
            d.density_line[i] = dev(density_line[i], f"density_line[{i}]")
            d.app_plane[i] = dev(app_plane[i], f"app_plane[{i}]")
            d.app_line[i] = dev(app_line[i], f"app_line[{i}]")
        d.basis = dev(basis, "basis_mat")
        if mask_volume is not None:
            mv = mask_volume.reshape(mask_volume.shape[-3:])
            d.mask_volume = dev(mv, "alpha_volume")
            d.mask_dims[:] = [int(s) for s in mv.shape]
            ma = torch.as_tensor(mask_aabb, dtype=torch.float32).cpu().reshape(-1)
            d.mask_aabb[:] = [float(v) for v in ma]
        else:
            d.mask_volume = None
            d.mask_dims[:] = [0, 0, 0]
            d.mask_aabb[:] = [float(v) for v in aabb]
        d.density_shift, d.distance_scale, d.weight_thres = float(density_shift), float(distance_scale), float(weight_thres)
        d.step_size, d.n_samples = float(step_size), int(n_samples)
        d.near_far[:] = [float(near_far[0]), float(near_far[1])]
        d.softplus, d.unisphere = int(bool(softplus)), int(bool(unisphere))
        d.density_lanes = int(density_lanes)          # 0 = auto; 1 / 4 force a gather form (include/iffnerf_hip.h)
        d.head_lanes = int(head_lanes)                # 0 = auto; 16 = the vector form of the Ref head launches
        d.sampler_persistent = int(bool(sampler_persistent))      # the one-launch form of the surface sampler (parity tests)
        d.fan_waves = int(fan_waves)                  # 0 = auto; 4 / 8 name the fused fan kernel (include/iffnerf_hip.h)
        for field, key in (("normal", "normal_mlp.0"), ("tint", "tint_color_mlp.0"), ("rough", "roughness_mlp.0"),
                           ("diffuse", "diffuse_color_mlp.0"), ("bottleneck", "bottleneck_mlp"),
                           ("specular", "specular_mlp.0")):
            setattr(d, field + "_w", dev(head[key + ".weight"], key + ".weight"))
            setattr(d, field + "_b", dev(head[key + ".bias"], key + ".bias"))
        d.ide_mat = dev(head["dir_enc_fn.mat"], "dir_enc_fn.mat")
        self.app_dim, self.n_samples_default = d.app_dim, int(n_samples)
        out = C.c_void_p()
        with torch.cuda.device(device):
            check(L.iff_field_create(C.byref(d), stream_ptr(device), C.byref(out)), "iff_field_create")
        self._h = out
        self.table_bytes = int(L.iff_field_table_bytes(out))

    # ------------------------------------------------------------------ table files (include/iffnerf_hip.h iff_field_save / _load)
    def save(self, path: str) -> None:
        """Write the handle's tables, already in kernel layout, to ``path`` (a pre-laid-out counterpart of the ``.th``
        checkpoint: a serving process loads it with ``FieldHandle.from_file`` and skips TensorBase.load + the re-layout)."""
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_field_save(self._h, str(path).encode(), stream_ptr(self.device)), "iff_field_save")

    @classmethod
    def from_file(cls, path: str, device) -> "FieldHandle":
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"FieldHandle needs a GPU device (got {device}); libiffnerf_hip has no CPU path")
        L = _lib.lib()
        self = cls.__new__(cls)
        self._h, self.device = None, device
        out = C.c_void_p()
        with torch.cuda.device(device):
            check(L.iff_field_load(str(path).encode(), stream_ptr(device), C.byref(out)), "iff_field_load")
        self._h = out
        self.app_dim = 27
        self.n_samples_default = int(L.iff_march_default_samples(out, MARCH_SLAB))
        self.table_bytes = int(L.iff_field_table_bytes(out))
        return self

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().iff_field_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _pts(self, x: torch.Tensor, name: str) -> torch.Tensor:
        if x.shape[-1] != 3:
            raise RuntimeError(f"{name} must be [...,3] (got {tuple(x.shape)})")
        x = x.detach()
        if not x.is_cuda:
            raise RuntimeError(f"{name} must live on the GPU (got {x.device}); libiffnerf_hip has no CPU path")
        return x.to(torch.float32).reshape(-1, 3).contiguous()

    def _call(self, fn, x, out, *extra):
        with torch.cuda.device(self.device):
            check(fn(self._h, dptr(x), x.shape[0], *extra, dptr(out), stream_ptr(self.device)), fn.__name__)
        return out

    # ------------------------------------------------------------------ per-point lookups
    def normalize_coord(self, xyz):
        x = self._pts(xyz, "xyz")
        return self._call(_lib.lib().iff_normalize_coord, x, torch.empty_like(x)).reshape(xyz.shape)

    def mask_sample(self, xyz):
        x = self._pts(xyz, "xyz")
        return self._call(_lib.lib().iff_mask_sample, x, x.new_empty(x.shape[0]))

    def mask_occupied(self, xyz):
        """``sample_alpha(xyz) > 0`` (tensorBase.py:762-764) from the corner-bit table: bool [n]."""
        x = self._pts(xyz, "xyz")
        out = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_mask_occupied(self._h, dptr(x), x.shape[0], out.data_ptr(), stream_ptr(self.device)), "iff_mask_occupied")
        return out.bool()

    def density_feature(self, xn):
        x = self._pts(xn, "xyz_sampled")
        return self._call(_lib.lib().iff_density_feature, x, x.new_empty(x.shape[0]))

    def app_feature(self, xn):
        x = self._pts(xn, "xyz_sampled")
        return self._call(_lib.lib().iff_app_feature, x, x.new_empty(x.shape[0], self.app_dim))

    def point_alpha(self, xyz, length=1.0):
        x = self._pts(xyz, "xyz_locs")
        out = x.new_empty(x.shape[0])
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_point_alpha(self._h, dptr(x), x.shape[0], float(length), dptr(out), stream_ptr(self.device)),
                  "iff_point_alpha")
        return out.reshape(xyz.shape[:-1])

    def point_normals(self, xyz):
        x = self._pts(xyz, "samples")
        return self._call(_lib.lib().iff_point_normals, x, torch.empty_like(x))

    def ref_shade(self, viewdirs, features):
        d = self._pts(viewdirs, "viewdirs")
        f = features.detach().to(torch.float32).reshape(-1, self.app_dim).contiguous()
        out = torch.empty_like(d)
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_ref_shade(self._h, dptr(d), dptr(f, name="features"), d.shape[0], dptr(out),
                                           stream_ptr(self.device)), "iff_ref_shade")
        return out

    def head_normals(self, features):
        f = features.detach().to(torch.float32).reshape(-1, self.app_dim).contiguous()
        if not f.is_cuda:
            raise RuntimeError("features must live on the GPU; libiffnerf_hip has no CPU path")
        return self._call(_lib.lib().iff_ref_normals, f, f.new_empty(f.shape[0], 3))

    # ------------------------------------------------------------------ partial handles
    @classmethod
    def _dummy(cls, device, head=None, mask_volume=None, mask_aabb=None, unisphere=False):
        """Handle with 2x2x2 zero VM tables: carries only a Ref head and/or an occupancy mask."""
        z = torch.zeros
        if head is None:
            head = {"dir_enc_fn.mat": z(9, 19)}
            for k, (o, i) in {"normal_mlp.0": (3, 27), "tint_color_mlp.0": (3, 27), "roughness_mlp.0": (1, 27),
                              "diffuse_color_mlp.0": (3, 27), "bottleneck_mlp": (128, 27),
                              "specular_mlp.0": (3, 167)}.items():
                head[k + ".weight"], head[k + ".bias"] = z(o, i), z(o)
        aabb = mask_aabb if mask_aabb is not None else torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        return cls(device=device, grid=(2, 2, 2), aabb=aabb, density_plane=[z(4, 2, 2)] * 3, density_line=[z(4, 2)] * 3,
                   app_plane=[z(48, 2, 2)] * 3, app_line=[z(48, 2)] * 3, basis=z(27, 144), head=head,
                   mask_volume=mask_volume, mask_aabb=mask_aabb, density_shift=-10.0, distance_scale=25.0,
                   weight_thres=1e-4, step_size=1.0, n_samples=2, near_far=(2.0, 6.0), unisphere=unisphere)

    @classmethod
    def head_only(cls, head, device):
        return cls._dummy(device, head=head)

    @classmethod
    def mask_only(cls, mask_volume, mask_aabb, device, unisphere=False):
        return cls._dummy(device, mask_volume=mask_volume, mask_aabb=mask_aabb, unisphere=unisphere)

    # ------------------------------------------------------------------ march
    def march_plan(self, mode: int = MARCH_POINT, n_samples: int = -1) -> int:
        """``iff_march_plan``: 0 the general kernels, 2 the fused fan kernel + the Ref head launch, 3 the fan kernel with the head fused in."""
        return int(_lib.lib().iff_march_plan(self._h, int(mode), int(n_samples)))

    def fan_kernel(self, mode: int = MARCH_POINT, n_samples: int = -1):
        """``iff_march_fan_kernel``: (waves per fan, patch side) of the fused fan kernel that serves this march -- (4, 12): k4f_fan_march,
        (8, 12) / (8, 22): k4g_fan_march<12, 1> / <22, 3>, (0, 0): the general kernels."""
        w, side = C.c_int32(), C.c_int32()
        check(_lib.lib().iff_march_fan_kernel(self._h, int(mode), int(n_samples), C.byref(w), C.byref(side)), "iff_march_fan_kernel")
        return int(w.value), int(side.value)

    def fan_kernel_name(self, mode: int = MARCH_POINT, n_samples: int = -1) -> str:
        """The kernel as rocprofv3 prints it (profiles/*.csv)."""
        w, side = self.fan_kernel(mode, n_samples)
        plan = self.march_plan(mode, n_samples)
        if w == 4:
            return "k4f_fan_march<%d>" % plan
        if w == 8:
            return "k4g_fan_march<%d, %d, %d>" % (side, 1 if side == 12 else 3, plan)
        return ""

    def march(self, rays: torch.Tensor, mode: int, n_samples: int = -1, bg=(0.0, 0.0, 0.0), want_alpha: bool = True,
              want_counts: bool = False, stage_ms: Optional[list] = None):
        """stage_ms: pass an empty list to run the instrumented (synchronous) variant; it receives the 3 launch times."""
        if rays.dim() != 2 or rays.shape[-1] not in (6, 7):
            raise RuntimeError(f"rays_chunk must be [R,6] or [R,7] (got {tuple(rays.shape)})")
        if not rays.is_cuda:
            raise RuntimeError("rays must live on the GPU; libiffnerf_hip has no CPU path")
        r = rays.detach().to(torch.float32).contiguous()
        R = r.shape[0]
        S = n_samples if n_samples > 0 else (20 if mode == MARCH_POINT else self.n_samples_default)
        rgb = r.new_empty(R, 3)
        depth = r.new_empty(R)
        acc = r.new_empty(R)
        alpha = r.new_empty(R, S) if want_alpha else None
        counts = torch.empty(R, 2, dtype=torch.int32, device=r.device) if want_counts else None
        ws_bytes = int(_lib.lib().iff_march_workspace(self._h, R, mode, S)) if R > 0 else 0
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=r.device)   # weights + per-ray features
        with torch.cuda.device(self.device):
            if stage_ms is None:
                check(_lib.lib().iff_march_shade(self._h, dptr(r), r.shape[1], R, mode, S, fvec(bg), dptr(rgb), dptr(depth),
                                                 dptr(acc), dptr(alpha), dptr(counts, torch.int32), ws.data_ptr(),
                                                 ws.numel() * 4, stream_ptr(self.device)),
                      "iff_march_shade")
            else:
                ms = (C.c_float * 3)()
                check(_lib.lib().iff_march_shade_timed(self._h, dptr(r), r.shape[1], R, mode, S, fvec(bg), dptr(rgb),
                                                       dptr(depth), dptr(acc), dptr(alpha), dptr(counts, torch.int32),
                                                       ws.data_ptr(), ws.numel() * 4, ms, stream_ptr(self.device)),
                      "iff_march_shade_timed")
                stage_ms[:] = [float(v) for v in ms]
        return rgb, depth, acc, alpha, counts, S

    def march_features(self, rays: torch.Tensor, mode: int, n_samples: int = -1):
        """``iff_march_features``: the march up to the Ref head -> (feat28 [R,28], depth [R], acc [R], S)."""
        if rays.dim() != 2 or rays.shape[-1] not in (6, 7) or not rays.is_cuda:
            raise RuntimeError(f"rays_chunk must be a GPU tensor [R,6] or [R,7] (got {tuple(rays.shape)})")
        r = rays.detach().to(torch.float32).contiguous()
        R = r.shape[0]
        S = n_samples if n_samples > 0 else (20 if mode == MARCH_POINT else self.n_samples_default)
        feat, depth, acc = r.new_zeros(R, 28), r.new_empty(R), r.new_empty(R)
        ws_bytes = int(_lib.lib().iff_march_workspace(self._h, R, mode, S)) if R > 0 else 0
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=r.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_march_features(self._h, dptr(r), r.shape[1], R, mode, S, dptr(feat), dptr(depth), dptr(acc),
                                                ws.data_ptr(), ws.numel() * 4, stream_ptr(self.device)), "iff_march_features")
        return feat, depth, acc, S

    def march_grad(self, rays: torch.Tensor, mode: int, n_samples: int, g_feat28: torch.Tensor, g_acc: torch.Tensor):
        """``iff_march_grad``: dL/d(o, d) [R,6] from dL/dfeat28 [R,28] and dL/dacc [R]."""
        r = rays.detach().to(torch.float32).contiguous()
        R = r.shape[0]
        gf = g_feat28.detach().to(torch.float32).contiguous()
        ga = g_acc.detach().to(torch.float32).contiguous()
        if gf.shape != (R, 28) or ga.shape != (R,):
            raise RuntimeError(f"march_grad: gradients must be [R,28] and [R] (got {tuple(gf.shape)}, {tuple(ga.shape)})")
        out = r.new_zeros(R, 6)
        ws_bytes = int(_lib.lib().iff_march_grad_workspace(self._h, R, mode, n_samples)) if R > 0 else 0
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=r.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_march_grad(self._h, dptr(r), r.shape[1], R, mode, n_samples, dptr(gf), dptr(ga), dptr(out),
                                            ws.data_ptr(), ws.numel() * 4, stream_ptr(self.device)), "iff_march_grad")
        return out

    # ------------------------------------------------------------------ surface sampler
    def surface_sample(self, n_points: int, rho: float, n_epochs: int = 4, max_iterations: int = 200, seed: int = 0,
                       seed_offset: Optional[torch.Tensor] = None):
        """seed_offset: optional 1-element int64 device tensor added to ``seed`` on the device (hipGraph replays)."""
        L = _lib.lib()
        dev = self.device
        ws_bytes = int(L.iff_surface_sample_workspace(n_points))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        samples = torch.empty(n_points, 3, dtype=torch.float32, device=dev)
        alpha = torch.empty(n_points, dtype=torch.float32, device=dev)
        stats = torch.empty(max(n_epochs, 1), 4, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(L.iff_surface_sample(self._h, n_points, n_epochs, max_iterations, int(seed) & (2 ** 64 - 1),
                                       dptr(seed_offset, torch.int64, "seed_offset"), float(rho),
                                       dptr(samples), dptr(alpha), dptr(stats, torch.int32), ws.data_ptr(), ws_bytes,
                                       stream_ptr(dev)), "iff_surface_sample")
        return samples, alpha, stats


SAMPLER_SEED_STRIDE = 0x9E3779B97F4A7C15      # include/iffnerf_hip.h IFF_SAMPLER_SEED_STRIDE


def _surface_sample_batched(self, batch: int, n_points: int, rho: float, n_epochs: int = 4, max_iterations: int = 200, seed: int = 0,
                            seed_offset: Optional[torch.Tensor] = None):
    """``batch`` independent sampler runs in one launch -> samples [B,P,3], alpha [B,P], stats [B,n_epochs,4].
    Run b equals ``surface_sample`` with seed + b * SAMPLER_SEED_STRIDE (mod 2^64)."""
    L = _lib.lib()
    dev = self.device
    ws_bytes = int(L.iff_surface_sample_workspace(n_points)) * batch
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    samples = torch.empty(batch, n_points, 3, dtype=torch.float32, device=dev)
    alpha = torch.empty(batch, n_points, dtype=torch.float32, device=dev)
    stats = torch.empty(batch, max(n_epochs, 1), 4, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(L.iff_surface_sample_batched(self._h, batch, n_points, n_epochs, max_iterations, int(seed) & (2 ** 64 - 1),
                                           dptr(seed_offset, torch.int64, "seed_offset"), float(rho), dptr(samples), dptr(alpha),
                                           dptr(stats, torch.int32), ws.data_ptr(), ws_bytes, stream_ptr(dev)),
              "iff_surface_sample_batched")
    return samples, alpha, stats


def _sampler_residency(self, n_points: int, batch: int = 1):
    """(workgroups one sampler run uses inside a launch of ``batch`` runs, sampler workgroups the device holds at once)
    -- see iff_surface_sample_residency."""
    import ctypes as C
    w, c = C.c_int32(0), C.c_int32(0)
    with torch.cuda.device(self.device):
        check(_lib.lib().iff_surface_sample_residency(self._h, batch, n_points, C.byref(w), C.byref(c)),
              "iff_surface_sample_residency")
    return int(w.value), int(c.value)


FieldHandle.surface_sample_batched = _surface_sample_batched
FieldHandle.sampler_residency = _sampler_residency


def isocell_emit(cells: torch.Tensor, points: torch.Tensor, normals: torch.Tensor, want_rays6: bool = False):
    """rotate_isocell + renormalise + origin broadcast -> (ori [27P,3], dirs [27P,3]) (+ rays [27P,6] when asked)."""
    if not points.is_cuda:
        raise RuntimeError("points must live on the GPU; libiffnerf_hip has no CPU path")
    if cells.shape != (27, 3):
        raise RuntimeError(f"only the 27-direction iso-cell set is built (got {tuple(cells.shape)})")
    p = points.detach().to(torch.float32).reshape(-1, 3).contiguous()
    n = normals.detach().to(torch.float32).reshape(-1, 3).contiguous()
    P = p.shape[0]
    ori = p.new_empty(P * 27, 3)
    dirs = p.new_empty(P * 27, 3)
    rays6 = p.new_empty(P * 27, 6) if want_rays6 else None
    c = fvec(cells.detach().cpu().reshape(-1).tolist())
    with torch.cuda.device(p.device):
        check(_lib.lib().iff_isocell_emit(c, dptr(p), dptr(n), P, dptr(ori), dptr(dirs), dptr(rays6), stream_ptr(p.device)),
              "iff_isocell_emit")
    return (ori, dirs, rays6) if want_rays6 else (ori, dirs)


def field_handle_from_ckpt(ckpt: dict, device, density_lanes: int = 0, head_lanes: int = 0, sampler_persistent: bool = False,
                           fan_waves: int = 0) -> FieldHandle:
    """Build a handle straight from a checkpoint dictionary (TensorBase.save layout); used by tests and bench."""
    from .models.tensorBase import derive_step
    kw = ckpt["kwargs"]
    sd = ckpt["state_dict"]
    aabb = torch.as_tensor(kw["aabb"]).float().cpu()
    step, n_samples = derive_step(aabb, kw["gridSize"], kw.get("step_ratio", 2.0), kw.get("contraction_type", "aabb"))
    mask_volume = mask_aabb = None
    if "alphaMask.aabb" in ckpt:
        shape = tuple(int(s) for s in ckpt["alphaMask.shape"])
        bits = np.unpackbits(np.asarray(ckpt["alphaMask.mask"]))[:int(np.prod(shape))].reshape(shape)
        mask_volume = torch.from_numpy(bits).float()
        mask_aabb = torch.as_tensor(ckpt["alphaMask.aabb"]).float()
    head = {k[len("renderModule."):]: v for k, v in sd.items() if k.startswith("renderModule.")}
    return FieldHandle(
        device=device, grid=kw["gridSize"], aabb=aabb,
        density_plane=[sd[f"density_plane.{i}"][0] for i in range(3)],
        density_line=[sd[f"density_line.{i}"][0, :, :, 0] for i in range(3)],
        app_plane=[sd[f"app_plane.{i}"][0] for i in range(3)],
        app_line=[sd[f"app_line.{i}"][0, :, :, 0] for i in range(3)],
        basis=sd["basis_mat.weight"], head=head, mask_volume=mask_volume, mask_aabb=mask_aabb,
        density_shift=kw.get("density_shift", -10), distance_scale=kw.get("distance_scale", 25),
        weight_thres=kw.get("rayMarch_weight_thres", 1e-4), step_size=float(step), n_samples=n_samples,
        near_far=kw.get("near_far", (2.0, 6.0)), softplus=kw.get("fea2denseAct", "softplus") == "softplus",
        unisphere=kw.get("contraction_type", "aabb") == "unisphere", density_lanes=density_lanes, head_lanes=head_lanes,
        sampler_persistent=sampler_persistent, fan_waves=fan_waves)
