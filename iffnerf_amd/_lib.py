"""ctypes binding of libiffnerf_hip.so (include/iffnerf_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  ``lib()`` raises RuntimeError when the
shared object is missing (build it with ``python -m iffnerf_amd.build``), and every wrapper raises
RuntimeError -- the one exception type the reference driver survives (train_eval_pose_est.py:259-260) -- when a
call returns non-zero or is handed a tensor that is not resident on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IFF_LIB_PATH: a development build (`python -m iffnerf_amd.build --tag NAME -- <flags>` -> build/lib_NAME.so, or scripts/build_one_tu.sh)
# loaded INSTEAD of the in-tree library, so A/B runs never overwrite the product's .so; unset (the default, and what tests / bench /
# the driver use) -> the library next to this file.  An override is never silent: `lib()` warns once and bench.py prints `dev_library`.
PRODUCT_LIB_PATH = os.path.join(_HERE, "libiffnerf_hip.so")
LIB_PATH = os.environ.get("IFF_LIB_PATH") or PRODUCT_LIB_PATH
DEV_LIBRARY = os.path.abspath(LIB_PATH) != os.path.abspath(PRODUCT_LIB_PATH)
_lib = None
ABI_VERSION = 12         # include/iffnerf_hip.h IFF_ABI_VERSION this binding was written against

c_float_p = C.POINTER(C.c_float)


class FieldDesc(C.Structure):
    _fields_ = [
        ("grid", C.c_int32 * 3), ("aabb", C.c_float * 6),
        ("n_density", C.c_int32), ("n_app", C.c_int32), ("app_dim", C.c_int32), ("feature_c", C.c_int32),
        ("density_plane", C.c_void_p * 3), ("density_line", C.c_void_p * 3),
        ("app_plane", C.c_void_p * 3), ("app_line", C.c_void_p * 3),
        ("basis", C.c_void_p), ("mask_volume", C.c_void_p), ("mask_dims", C.c_int32 * 3), ("mask_aabb", C.c_float * 6),
        ("density_shift", C.c_float), ("distance_scale", C.c_float), ("weight_thres", C.c_float),
        ("step_size", C.c_float), ("n_samples", C.c_int32), ("near_far", C.c_float * 2),
        ("softplus", C.c_int32), ("unisphere", C.c_int32), ("density_lanes", C.c_int32), ("head_lanes", C.c_int32),
        ("sampler_persistent", C.c_int32), ("fan_waves", C.c_int32),
        ("normal_w", C.c_void_p), ("normal_b", C.c_void_p), ("tint_w", C.c_void_p), ("tint_b", C.c_void_p),
        ("rough_w", C.c_void_p), ("rough_b", C.c_void_p), ("diffuse_w", C.c_void_p), ("diffuse_b", C.c_void_p),
        ("bottleneck_w", C.c_void_p), ("bottleneck_b", C.c_void_p), ("specular_w", C.c_void_p), ("specular_b", C.c_void_p),
        ("ide_mat", C.c_void_p),
    ]


class IdNetDesc(C.Structure):
    _fields_ = [("feature_c", C.c_int32), ("fea", C.c_int32), ("img_fea", C.c_int32), ("gemm_mode", C.c_int32),
                ("trunk_variant", C.c_int32)] + [
        (n, C.c_void_p) for n in ("l1_w", "l1_b", "l2_w", "l2_b", "l3_w", "l3_b", "l4_w", "l4_b", "q_w", "q_b", "k_w", "k_b")]


class VitDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "depth", "heads", "mlp", "patch", "grid_h", "grid_w")] + [("ln_eps", C.c_float)] + [
        (n, C.c_void_p) for n in ("patch_w", "patch_b", "cls", "pos", "ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1",
                                  "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "norm_w", "norm_b")] + [("precision", C.c_int32), ("gemm_form", C.c_int32)]


VIT_FP32, VIT_BF16 = 0, 1          # include/iffnerf_hip.h IFF_VIT_FP32 / IFF_VIT_BF16


# name -> (restype, argtypes); must list every function include/iffnerf_hip.h declares (tests/test_abi.py checks)
_VP, _I32, _I64, _F, _SZ, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_uint64
SIGNATURES = {
    "iff_last_error": (C.c_char_p, []),
    "iff_abi_version": (C.c_int, []),
    "iff_field_create": (C.c_int, [C.POINTER(FieldDesc), _VP, C.POINTER(_VP)]),
    "iff_field_destroy": (None, [_VP]),
    "iff_field_table_bytes": (_SZ, [_VP]),
    "iff_field_save": (C.c_int, [_VP, C.c_char_p, _VP]),
    "iff_field_load": (C.c_int, [C.c_char_p, _VP, C.POINTER(_VP)]),
    "iff_idnet_save": (C.c_int, [_VP, C.c_char_p, _VP]),
    "iff_idnet_load": (C.c_int, [C.c_char_p, _VP, C.POINTER(_VP)]),
    "iff_normalize_coord": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_mask_sample": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_mask_occupied": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_merge_row_stats": (C.c_int, [_VP, _I32, _I64, _VP, _VP, _VP]),
    "iff_pack_candidates": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _I32, _I32, _I32, _I64, _VP, _VP]),
    "iff_merge_candidates": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP]),
    "iff_density_feature": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_app_feature": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_point_alpha": (C.c_int, [_VP, _VP, _I64, _F, _VP, _VP]),
    "iff_point_normals": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_ref_shade": (C.c_int, [_VP, _VP, _VP, _I64, _VP, _VP]),
    "iff_ref_normals": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_isocell_emit": (C.c_int, [c_float_p, _VP, _VP, _I64, _VP, _VP, _VP, _VP]),
    "iff_march_workspace": (_SZ, [_VP, _I64, _I32, _I32]),
    "iff_march_default_samples": (_I32, [_VP, _I32]),
    "iff_march_plan": (_I32, [_VP, _I32, _I32]),
    "iff_march_fan_kernel": (C.c_int, [_VP, _I32, _I32, _VP, _VP]),
    "iff_march_shade": (C.c_int, [_VP, _VP, _I32, _I64, _I32, _I32, c_float_p, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_march_features": (C.c_int, [_VP, _VP, _I32, _I64, _I32, _I32, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_march_grad_workspace": (_SZ, [_VP, _I64, _I32, _I32]),
    "iff_march_grad": (C.c_int, [_VP, _VP, _I32, _I64, _I32, _I32, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_march_shade_timed": (C.c_int, [_VP, _VP, _I32, _I64, _I32, _I32, c_float_p, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, c_float_p, _VP]),
    "iff_surface_sample_workspace": (_SZ, [_I64]),
    "iff_surface_sample": (C.c_int, [_VP, _I64, _I32, _I32, _U64, _VP, _F, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_surface_sample_residency": (C.c_int, [_VP, _I32, _I64, _VP, _VP]),
    "iff_surface_sample_batched": (C.c_int, [_VP, _I32, _I64, _I32, _I32, _U64, _VP, _F, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_idnet_create": (C.c_int, [C.POINTER(IdNetDesc), _VP, C.POINTER(_VP)]),
    "iff_idnet_destroy": (None, [_VP]),
    "iff_idnet_gemm_mode": (_I32, [_VP]),
    "iff_idnet_dims": (C.c_int, [_VP, _VP, _VP, _VP]),
    "iff_ray_encode_workspace": (_SZ, [_VP, _I64]),
    "iff_ray_encode": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _SZ, _VP]),
    "iff_k_proj": (C.c_int, [_VP, _VP, _I64, _VP, _VP]),
    "iff_q_proj_workspace": (_SZ, [_VP, _I32]),
    "iff_q_proj": (C.c_int, [_VP, _VP, _I32, _VP, _VP, _SZ, _VP]),
    "iff_attn_logits": (C.c_int, [_VP, _VP, _I32, _I64, _I32, _F, _VP, _VP, _VP, _I32, _VP]),
    "iff_ray_trunk_workspace": (_SZ, [_VP, _I64]),
    "iff_ray_trunk": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _VP, _VP, _SZ, _VP]),
    "iff_q_fold_width": (_I32, [_VP]),
    "iff_q_fold": (C.c_int, [_VP, _VP, _I32, _VP, _VP]),
    "iff_attn_logits_folded": (C.c_int, [_VP, _VP, _VP, _I32, _I64, _F, _VP, _VP, _VP, _VP]),
    "iff_ray_logits_folded_workspace": (_SZ, [_VP, _I64, _I32]),
    "iff_ray_logits_folded": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _VP, _I32, _F, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_ray_logits_folded_batched_workspace": (_SZ, [_VP, _I32, _I64, _I32]),
    "iff_ray_logits_folded_batched": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _I64, _VP, _I32, _F, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_ray_logits_folded_timed": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _I64, _VP, _I32, _F, _VP, _VP, _VP, _VP, _SZ, c_float_p, _VP]),
    "iff_vit_create": (C.c_int, [C.POINTER(VitDesc), _VP, C.POINTER(_VP)]),
    "iff_vit_destroy": (None, [_VP]),
    "iff_vit_workspace": (_SZ, [_VP, _I32]),
    "iff_vit_forward": (C.c_int, [_VP, _VP, _I32, _VP, _VP, _VP, _SZ, _VP]),
    "iff_image_resize_crop": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, c_float_p, c_float_p, _VP, _VP]),
    "iff_image_resize_crop_rgba": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, c_float_p, c_float_p, _VP, _VP]),
    "iff_token_assemble": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _VP, _F, c_float_p, c_float_p, _VP, _VP, _VP]),
    "iff_token_assemble_compact": (C.c_int, [_VP, _I32, _I32, _I32, _I32, _VP, _F, c_float_p, c_float_p, _VP, _VP, _VP, _VP]),
    "iff_mask_token_rows": (C.c_int, [_VP, _I64, _VP, _VP, _VP]),
    "iff_ray_cache_bytes": (_SZ, [_VP, _I64]),
    "iff_ray_cache_workspace": (_SZ, [_VP, _I64]),
    "iff_ray_cache_build": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _VP, _SZ, _VP, _SZ, _VP]),
    "iff_logits_from_cache_workspace": (_SZ, [_VP, _I64, _I32]),
    "iff_logits_from_cache": (C.c_int, [_VP, _VP, _I64, _VP, _I32, _F, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_logits_from_cache_rows": (C.c_int, [_VP, _VP, _I64, _VP, _I32, _VP, _F, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "iff_attn_colsum": (C.c_int, [_VP, _I32, _I64, _VP, _VP, _I32, _VP, _VP]),
    "iff_topk_workspace": (_SZ, [_I64, _I32]),
    "iff_topk": (C.c_int, [_VP, _I64, _I32, _VP, _VP, _VP, _SZ, _VP]),
    "iff_pose_from_topk": (C.c_int, [_VP, _VP, _I32, _VP, _VP, _I64, c_float_p, _VP, _VP, _VP]),
    "iff_attn_colsum_batched": (C.c_int, [_VP, _I32, _I32, _I64, _VP, _VP, _I32, _VP, _VP]),
    "iff_attn_colsum_rows": (C.c_int, [_VP, _I32, _I32, _I64, _VP, _VP, _VP, _I32, _VP, _VP]),
    "iff_topk_batched": (C.c_int, [_VP, _I32, _I64, _I32, _VP, _VP, _VP]),
    "iff_pose_from_topk_batched": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _I64, _I64, c_float_p, _VP, _VP, _VP]),
    "iff_pose_errors": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _VP]),
}


def lib():
    """Load (once) and return the C-ABI library; fail loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the product path and has no fallback. "
                "Build it with `python -m iffnerf_amd.build` (hipcc, gfx950).")
        import torch  # noqa: F401  (loads torch's HIP runtime first so both share one libamdhip64)
        if os.path.abspath(LIB_PATH) != os.path.abspath(PRODUCT_LIB_PATH):
            import warnings
            warnings.warn(f"libiffnerf_hip: IFF_LIB_PATH is set -- loading the development build {LIB_PATH} instead of the product library")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.iff_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {h.iff_abi_version()}, this binding expects {ABI_VERSION}: "
                               "rebuild it with `python -m iffnerf_amd.build`")
        _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().iff_last_error().decode(errors="replace")
        raise RuntimeError(f"libiffnerf_hip {what} failed ({rc}): {msg}")


def dptr(t: Optional[torch.Tensor], dtype=torch.float32, name: str = "tensor") -> Optional[int]:
    """Device pointer of a contiguous tensor on the GPU, or None for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got device {t.device}); libiffnerf_hip has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t.data_ptr()


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def fvec(values) -> "C.Array":
    vals = [float(v) for v in values]
    return (C.c_float * len(vals))(*vals)
