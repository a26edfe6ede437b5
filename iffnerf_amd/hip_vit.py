"""The image backbone on the matrix cores: DINOv2 ViT-S/14 ``forward_features`` through ``iff_vit_forward``.

``ViTHandle`` owns the weight slab of one network; ``serve_natively`` installs the native ``forward_features`` ON a backbone module
(DINOv2's ``dinov2_vits14`` from ``torch.hub`` -- reference pose_estimation/backbone.py:12-14 -- or the seeded stand-in of this
package's backbone.py) so that ``forward_features(x)["x_norm_patchtokens"]`` (what pose_estimation/identification_module.py:141
reads) runs in libiffnerf_hip under ``torch.no_grad`` while the module -- class, parameters, ``state_dict`` keys -- stays what the
reference's ``create_backbone`` returns; with autograd enabled on a trainable backbone the module's own torch forward runs.
State-dict keys of both sources are understood (``blocks.i.attn.qkv.*``, ``blocks.i.ls1.gamma``, ``blocks.i.mlp.fc1.*`` for
DINOv2; ``blocks.i.qkv.*``, ``blocks.i.ls1``, ``blocks.i.fc1.*`` for the stand-in).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, dptr, stream_ptr


def interpolate_pos_embed(pos: torch.Tensor, gh: int, gw: int, patch: int = 14, differentiable: bool = False) -> torch.Tensor:
    """DINOv2's ``interpolate_pos_encoding`` for an input of gh x gw patches: ``pos`` [1, 1 + n, D] -> [1 + gh*gw, D].  The class
    position is kept; the patch grid is resized bicubically with the +0.1 scale-factor offset of the original implementation."""
    pos = pos.float() if differentiable else pos.detach().float()
    n = pos.shape[1] - 1
    if n == gh * gw:
        return pos[0].contiguous()
    side = int(round(math.sqrt(n)))
    if side * side != n:
        raise RuntimeError(f"position embedding has {n} patch positions: not a square grid")
    grid = pos[0, 1:].reshape(1, side, side, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((gh + 0.1) / side, (gw + 0.1) / side), mode="bicubic")
    if grid.shape[-2:] != (gh, gw):
        raise RuntimeError("interpolated position grid has the wrong size")
    return torch.cat((pos[0, :1], grid.permute(0, 2, 3, 1).reshape(gh * gw, -1)), dim=0).contiguous()


PRECISIONS = {"fp32": _lib.VIT_FP32, "bf16": _lib.VIT_BF16}


def _pick(sd: Dict[str, torch.Tensor], *names):
    for n in names:
        if n in sd:
            return sd[n]
    raise RuntimeError(f"backbone state_dict has none of {names}")


DEFAULT_GEMM_FORM = 0     # iff_vit_desc.gemm_form of the handles a served backbone builds (0: the library's choice; development scripts set others for A/B runs)


class ViTHandle:
    """One ViT-S/14's weights on one GPU (``iff_vit``)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device, grid=(16, 16), patch: int = 14, heads: int = 6, ln_eps: float = 1e-6,
                 precision: str = "fp32", gemm_form: int = 0):
        """``gemm_form`` (``iff_vit_desc.gemm_form``): 0 the library's choice of GEMM kernels; 1-4 name one (A/B runs, the bit-equality test).
        ``precision``: "fp32" (the default) -- the accuracy class of the reference's fp32 DINOv2: every matrix operand split exactly
        into two fp16 pieces, three MFMA products per block, fp32 accumulation (include/iffnerf_hip.h IFF_VIT_FP32); "bf16" -- bf16
        operands, ~2x faster, token features move by ~1e-2 relative (a throughput option)."""
        self._h = None
        if precision not in PRECISIONS:
            raise RuntimeError(f"precision must be one of {sorted(PRECISIONS)} (got {precision!r})")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"ViTHandle needs a GPU device (got {device}); libiffnerf_hip has no CPU path")
        sd = {k: v.detach() for k, v in state_dict.items()}
        depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        dim = int(_pick(sd, "cls_token").shape[-1])
        gh, gw = int(grid[0]), int(grid[1])

        def stack(*alts):
            return torch.stack([_pick(sd, *(a.format(i=i) for a in alts)).float() for i in range(depth)])

        t = {
            "patch_w": _pick(sd, "patch_embed.proj.weight", "patch_embed.weight").float().reshape(dim, -1),
            "patch_b": _pick(sd, "patch_embed.proj.bias", "patch_embed.bias").float(),
            "cls": _pick(sd, "cls_token").float().reshape(dim),
            "pos": interpolate_pos_embed(_pick(sd, "pos_embed"), gh, gw, patch),
            "ln1_w": stack("blocks.{i}.norm1.weight"), "ln1_b": stack("blocks.{i}.norm1.bias"),
            "qkv_w": stack("blocks.{i}.attn.qkv.weight", "blocks.{i}.qkv.weight"),
            "qkv_b": stack("blocks.{i}.attn.qkv.bias", "blocks.{i}.qkv.bias"),
            "proj_w": stack("blocks.{i}.attn.proj.weight", "blocks.{i}.proj.weight"),
            "proj_b": stack("blocks.{i}.attn.proj.bias", "blocks.{i}.proj.bias"),
            "ls1": stack("blocks.{i}.ls1.gamma", "blocks.{i}.ls1"),
            "ln2_w": stack("blocks.{i}.norm2.weight"), "ln2_b": stack("blocks.{i}.norm2.bias"),
            "fc1_w": stack("blocks.{i}.mlp.fc1.weight", "blocks.{i}.fc1.weight"),
            "fc1_b": stack("blocks.{i}.mlp.fc1.bias", "blocks.{i}.fc1.bias"),
            "fc2_w": stack("blocks.{i}.mlp.fc2.weight", "blocks.{i}.fc2.weight"),
            "fc2_b": stack("blocks.{i}.mlp.fc2.bias", "blocks.{i}.fc2.bias"),
            "ls2": stack("blocks.{i}.ls2.gamma", "blocks.{i}.ls2"),
            "norm_w": _pick(sd, "norm.weight").float(), "norm_b": _pick(sd, "norm.bias").float(),
        }
        if t["patch_w"].shape[1] != 3 * patch * patch:
            raise RuntimeError(f"patch embedding has {t['patch_w'].shape[1]} inputs per patch, expected 3 x {patch} x {patch}")
        d = _lib.VitDesc()
        d.dim, d.depth, d.heads, d.mlp, d.patch, d.grid_h, d.grid_w = dim, depth, int(heads), int(t["fc1_w"].shape[1]), int(patch), gh, gw
        d.ln_eps = float(ln_eps)
        d.precision = PRECISIONS[precision]
        d.gemm_form = int(gemm_form)
        self.precision = precision
        keep = []
        for name, v in t.items():
            v = v.to(device=device, dtype=torch.float32).contiguous()
            keep.append(v)
            setattr(d, name, dptr(v, name=name))
        self.device, self.dim, self.grid, self.patch, self.depth = device, dim, (gh, gw), int(patch), depth
        out = C.c_void_p()
        with torch.cuda.device(device):
            check(_lib.lib().iff_vit_create(C.byref(d), stream_ptr(device), C.byref(out)), "iff_vit_create")
        self._h = out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().iff_vit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, images: torch.Tensor, want_cls: bool = False):
        """images [Q,3,14 gh,14 gw] (resized / cropped / normalised) -> patch tokens [Q, gh*gw, dim] (+ class token [Q, dim])."""
        gh, gw = self.grid
        if images.dim() != 4 or images.shape[1] != 3 or tuple(images.shape[-2:]) != (gh * self.patch, gw * self.patch):
            raise RuntimeError(f"images must be [Q,3,{gh * self.patch},{gw * self.patch}] (got {tuple(images.shape)})")
        if not images.is_cuda:
            raise RuntimeError("images must live on the GPU; libiffnerf_hip has no CPU path")
        x = images.detach().to(torch.float32).contiguous()
        Q = x.shape[0]
        tok = x.new_empty(Q, gh * gw, self.dim)
        cls = x.new_empty(Q, self.dim) if want_cls else None
        L = _lib.lib()
        ws_bytes = int(L.iff_vit_workspace(self._h, Q))
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=x.device)
        with torch.cuda.device(self.device):
            check(L.iff_vit_forward(self._h, dptr(x), Q, dptr(tok), dptr(cls), ws.data_ptr(), ws.numel() * 4, stream_ptr(self.device)),
                  "iff_vit_forward")
        return (tok, cls) if want_cls else tok


class _NativeForwardFeatures:
    """The callable installed as a backbone module's ``forward_features`` by ``serve_natively``.  It holds the module it serves (the
    module holds it in turn: an ordinary reference cycle) and the ``iff_vit`` handle built from the module's CURRENT parameters."""

    def __init__(self, module: torch.nn.Module, grid, patch: int, precision: str = "fp32"):
        self.module, self.grid, self.patch, self.precision = module, (int(grid[0]), int(grid[1])), int(patch), precision
        self._handles: Dict[tuple, ViTHandle] = {}        # one iff_vit per input grid (the position table is interpolated per grid)
        self._key = None

    # a handle is a device resource of THIS process: copies and pickles of the module start without one
    def __getstate__(self):
        return {"module": self.module, "grid": self.grid, "patch": self.patch, "precision": self.precision, "_handles": {}, "_key": None}

    @property
    def _handle(self) -> Optional[ViTHandle]:
        return self._handles.get(self.grid)

    def close(self):
        for h in self._handles.values():
            h.close()
        self._handles = {}

    def stock(self, x, *args, **kwargs):
        """The module's own (class-level) ``forward_features``: stock torch ops."""
        return type(self.module).forward_features(self.module, x, *args, **kwargs)

    def _vit(self, device, grid=None) -> ViTHandle:
        grid = self.grid if grid is None else (int(grid[0]), int(grid[1]))
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.module.parameters())
        if self._key != key:
            self.close()
            self._key = key
        if grid not in self._handles:
            self._handles[grid] = ViTHandle(self.module.state_dict(), device, grid, self.patch, precision=self.precision, gemm_form=DEFAULT_GEMM_FORM)
        return self._handles[grid]

    MAX_TOKENS = 288          # csrc/api.hip iff_vit_create: 1 + gh * gw tokens per image at most

    def __call__(self, x, masks=None, *args, **kwargs):
        if not isinstance(x, torch.Tensor) or masks is not None or args or kwargs:
            return self.stock(x, masks, *args, **kwargs)          # token masking / list inputs: DINOv2's training-side forms
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.module.parameters())):
            return self.stock(x)
        if x.is_cuda and x.dim() == 4:
            # any input whose sides are multiples of the patch size is served, each grid by its own handle (the hub module accepts
            # them all: it interpolates its position table, and so does ViTHandle); grids beyond the kernels' token budget run the
            # module's own forward ON THE GPU.  CPU tensors raise: libiffnerf_hip has no CPU path (tests/test_abi.py).
            gh, gw = x.shape[-2] // self.patch, x.shape[-1] // self.patch
            if (gh * self.patch, gw * self.patch) != tuple(x.shape[-2:]) or 1 + gh * gw > self.MAX_TOKENS:
                return self.stock(x)
            tok, cls = self._vit(x.device, (gh, gw)).forward(x, want_cls=True)
            return {"x_norm_clstoken": cls, "x_norm_patchtokens": tok}
        tok, cls = self._vit(x.device).forward(x, want_cls=True)
        return {"x_norm_clstoken": cls, "x_norm_patchtokens": tok}


def serve_natively(module: torch.nn.Module, grid=(16, 16), patch: int = 14, precision: str = "fp32") -> torch.nn.Module:
    """Serve ``module.forward_features`` (what pose_estimation/identification_module.py:141 calls) from ``iff_vit_forward`` and
    return THE SAME module.  Nothing is wrapped: the module keeps its class, its parameters and its ``state_dict`` keys, so an
    ``IdentificationModule`` built on it saves and strict-loads ``image_preprocessing_net.<backbone key>`` exactly as the reference
    does (train_eval_pose_est.py:59-66, pose_estimation/train.py:226).  Only the instance attribute ``forward_features`` is
    installed: no-grad inference on a GPU tensor goes through the HIP kernels (the handle is rebuilt when a parameter moves or
    changes in place); with autograd enabled on a trainable backbone, with token masks or list inputs the module's own torch
    forward runs.  ``precision``: see ``ViTHandle`` ("fp32": the reference's accuracy class, the default; "bf16": throughput).
    ``restore_stock(module)`` removes it."""
    if precision not in PRECISIONS:
        raise RuntimeError(f"precision must be one of {sorted(PRECISIONS)} (got {precision!r})")
    if not hasattr(type(module), "forward_features"):
        raise RuntimeError(f"{type(module).__name__} has no forward_features: not a DINOv2-style backbone")
    restore_stock(module)
    object.__setattr__(module, "forward_features", _NativeForwardFeatures(module, grid, patch, precision))
    return module


def restore_stock(module: torch.nn.Module) -> torch.nn.Module:
    """Undo ``serve_natively``: ``forward_features`` is the class's method again."""
    cur = module.__dict__.get("forward_features")
    if isinstance(cur, _NativeForwardFeatures):
        cur.close()
        object.__delattr__(module, "forward_features")
    return module


def is_served_natively(module: torch.nn.Module) -> bool:
    return isinstance(module.__dict__.get("forward_features"), _NativeForwardFeatures)
