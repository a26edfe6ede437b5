"""Image side of stage C: query images -> token tensors, without a host round trip.

The reference's ``IdentificationModule.image_processing`` (pose_estimation/identification_module.py:130-160) resizes and
crops the image and its alpha mask, runs DINOv2 ViT-S/14, appends a 14-channel position code and selects the tokens whose
mask value exceeds 0.1 with a boolean index (a device->host sync per image, and a token count that changes per image).
Here the same steps run as one static-shape sequence that a hipGraph can hold together with stage C:

  preprocessing + backbone   stock PyTorch-ROCm ops (the backbone is third-party and stays a torch module; offline it is the
                             seeded stand-in of pose_estimation/backbone.py)
  token assembly             ``iff_token_assemble``: position code appended in one kernel, mask select -> keep flags
  mask select                ``iff_mask_token_rows`` on the softmax row statistics: dropped rows contribute exactly 0

``PosePipeline.identify_images_resident`` chains this with the cached-encoder logits, column sums, top-k and pose solve, so a
batch of query images goes image-in -> pose-out in one captured graph (``CapturedImageQuery``).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check, dptr, fvec, stream_ptr
from .pose_estimation.identification_module import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD, _center_crop, _resize_short_edge


def token_assemble(patch_tokens: torch.Tensor, grid, mask_grid: Optional[torch.Tensor] = None, mask_thres: float = 0.1, compact: bool = False):
    """patch_tokens [Q, gh*gw, C] (+ mask_grid [Q, gh*gw]) -> (tokens [Q, gh*gw, C+14], keep [Q, gh*gw] uint8).
    ``compact``: every image's kept rows first, in grid order (``iff_token_assemble_compact``) -> (tokens, keep, rows [Q] int32)."""
    gh, gw = int(grid[0]), int(grid[1])
    if not patch_tokens.is_cuda:
        raise RuntimeError("patch tokens must live on the GPU; libiffnerf_hip has no CPU path")
    t = patch_tokens.detach().to(torch.float32).contiguous()
    Q, G, Cc = t.shape
    if G != gh * gw:
        raise RuntimeError(f"patch tokens have {G} rows, the grid {gh}x{gw} has {gh * gw}")
    m = None
    if mask_grid is not None:
        m = mask_grid.detach().to(device=t.device, dtype=torch.float32).reshape(Q, G).contiguous()
    out = t.new_empty(Q, G, Cc + 14)
    keep = torch.empty(Q, G, dtype=torch.uint8, device=t.device)
    lin_h = fvec(torch.linspace(-1.0, 1.0, steps=gh, dtype=torch.float32).tolist())
    lin_w = fvec(torch.linspace(-1.0, 1.0, steps=gw, dtype=torch.float32).tolist())
    with torch.cuda.device(t.device):
        if compact:
            rows = torch.empty(Q, dtype=torch.int32, device=t.device)
            check(_lib.lib().iff_token_assemble_compact(dptr(t), Q, gh, gw, Cc, dptr(m), float(mask_thres), lin_h, lin_w, dptr(out),
                                                        keep.data_ptr(), rows.data_ptr(), stream_ptr(t.device)), "iff_token_assemble_compact")
            return out, keep, rows
        check(_lib.lib().iff_token_assemble(dptr(t), Q, gh, gw, Cc, dptr(m), float(mask_thres), lin_h, lin_w, dptr(out),
                                            keep.data_ptr(), stream_ptr(t.device)), "iff_token_assemble")
    return out, keep


def mask_token_rows(keep: torch.Tensor, row_max: torch.Tensor, row_sumexp: torch.Tensor) -> None:
    """In place: statistics of the rows with keep == 0 become (+inf, 1), so they add 0 to every column sum."""
    k = keep.reshape(-1)
    if k.numel() != row_max.numel() or k.numel() != row_sumexp.numel():
        raise RuntimeError("one keep flag per statistics row expected")
    with torch.cuda.device(row_max.device):
        check(_lib.lib().iff_mask_token_rows(k.data_ptr(), k.numel(), dptr(row_max), dptr(row_sumexp), stream_ptr(row_max.device)),
              "iff_mask_token_rows")


def _into(out, like: torch.Tensor, shape) -> torch.Tensor:
    """The caller's output buffer (a captured graph's static input), checked, or a fresh one."""
    if out is None:
        return like.new_empty(shape)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or out.device != like.device or not out.is_contiguous():
        raise RuntimeError(f"out must be a contiguous fp32 tensor of shape {tuple(shape)} on {like.device}")
    return out


def resize_crop(src: torch.Tensor, resize_size: int, crop_size: int, cubic: bool, mean=None, std=None, out=None) -> torch.Tensor:
    """``iff_image_resize_crop``: channels-last images [Q,H,W,C] -> the centre ``crop_size`` window of the antialiased resize whose
    shorter edge is ``resize_size`` (torchvision Resize + CenterCrop of identification_module.py:36-61), normalised, channels-first
    [Q,C,crop,crop] (written into ``out`` when given).  ``crop_size`` None keeps the whole resized image."""
    if not src.is_cuda:
        raise RuntimeError("images must live on the GPU; libiffnerf_hip has no CPU path")
    x = src.detach().to(torch.float32).contiguous()
    Q, H, W, Cc = x.shape
    if H <= W:
        rh, rw = resize_size, max(1, int(resize_size * W / H))
    else:
        rh, rw = max(1, int(resize_size * H / W)), resize_size
    ch, cw = (rh, rw) if crop_size is None else (crop_size, crop_size)
    top, left = int(round((rh - ch) / 2.0)), int(round((rw - cw) / 2.0))
    out = _into(out, x, (Q, Cc, ch, cw))
    m = None if mean is None else fvec(mean)
    s = None if std is None else fvec(std)
    with torch.cuda.device(x.device):
        check(_lib.lib().iff_image_resize_crop(dptr(x), Q, H, W, Cc, rh, rw, top, left, ch, cw, int(bool(cubic)), m, s, dptr(out),
                                               stream_ptr(x.device)), "iff_image_resize_crop")
    return out


RESIZE_RGB_ON_WHITE, RESIZE_ALPHA = 1, 2          # include/iffnerf_hip.h IFF_RESIZE_*


def resize_crop_rgba(src: torch.Tensor, resize_size: int, crop_size, mode: int, cubic: bool, mean=None, std=None, out=None) -> torch.Tensor:
    """``iff_image_resize_crop_rgba``: RGBA images [Q,H,W,4] -> the resized / cropped / normalised colour composited on white
    [Q,3,crop,crop] (``RESIZE_RGB_ON_WHITE``: pose_estimation/test.py:77-81 folded into the resize) or alpha channel [Q,1,crop,crop]
    (``RESIZE_ALPHA``) -- the images never exist as separate RGB / mask tensors."""
    if not src.is_cuda:
        raise RuntimeError("images must live on the GPU; libiffnerf_hip has no CPU path")
    x = src.detach().to(torch.float32).contiguous()
    Q, H, W, Cc = x.shape
    if Cc != 4:
        raise RuntimeError(f"RGBA images [Q,H,W,4] expected (got {Cc} channels)")
    if H <= W:
        rh, rw = resize_size, max(1, int(resize_size * W / H))
    else:
        rh, rw = max(1, int(resize_size * H / W)), resize_size
    ch, cw = (rh, rw) if crop_size is None else (crop_size, crop_size)
    top, left = int(round((rh - ch) / 2.0)), int(round((rw - cw) / 2.0))
    out = _into(out, x, (Q, 3 if mode == RESIZE_RGB_ON_WHITE else 1, ch, cw))
    m = None if mean is None else fvec(mean)
    s = None if std is None else fvec(std)
    with torch.cuda.device(x.device):
        check(_lib.lib().iff_image_resize_crop_rgba(dptr(x), Q, H, W, int(mode), rh, rw, top, left, ch, cw, int(bool(cubic)), m, s, dptr(out),
                                                    stream_ptr(x.device)), "iff_image_resize_crop_rgba")
    return out


class ImageFrontEnd:
    """Resize / crop / normalise + backbone + token assembly for a batch of query images, all on the device."""

    def __init__(self, backbone: torch.nn.Module, grid=(16, 16), resize_size: int = 256, crop_size: int = 224,
                 backbone_autocast: Optional[torch.dtype] = None, native_preprocess: bool = True):
        """``backbone_autocast``: None runs the backbone as it is (fp32: the reference's arithmetic); ``torch.bfloat16`` /
        ``torch.float16`` runs its matrix products under ``torch.autocast`` -- a throughput option of the third-party model, not
        parity-equivalent (token features move by ~1e-2 relative)."""
        self.backbone, self.grid, self.resize_size, self.crop_size = backbone, (int(grid[0]), int(grid[1])), resize_size, crop_size
        self.backbone_autocast = backbone_autocast
        self.native_preprocess = bool(native_preprocess)     # iff_image_resize_crop instead of F.interpolate + crop + normalise
        self._norm = {}      # device -> (mean, std): made once, outside any capture (a host->device copy cannot be captured)

    def _mean_std(self, x):
        key = (x.device, x.dtype)
        if key not in self._norm:
            self._norm[key] = (torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1),
                               torch.tensor(IMAGENET_DEFAULT_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1))
        return self._norm[key]

    @torch.no_grad()
    def tokens(self, imgs: torch.Tensor, masks: Optional[torch.Tensor] = None, compact: bool = False):
        """imgs [Q,H,W,3] in [0,1], masks [Q,H,W] (alpha) -> (tokens [Q, gh*gw, C+14], keep [Q, gh*gw]).
        The same arithmetic as identification_module.py:130-160 (``transformations`` / ``mask_transformations`` of the mirror)."""
        if self.native_preprocess:
            # resize + crop + normalise in one kernel (iff_image_resize_crop); the mask: resize + crop, then down to the token grid
            xin = resize_crop(imgs, self.resize_size, self.crop_size, True, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)
        else:
            x = _center_crop(_resize_short_edge(imgs.permute(0, 3, 1, 2), self.resize_size, "bicubic"), self.crop_size)
            mean, std = self._mean_std(x)
            xin = (x - mean) / std
        if self.backbone_autocast is None:
            feats = self.backbone.forward_features(xin)["x_norm_patchtokens"]
        else:
            with torch.autocast(device_type="cuda", dtype=self.backbone_autocast):
                feats = self.backbone.forward_features(xin)["x_norm_patchtokens"]
        mg = None
        if masks is not None:
            if self.native_preprocess:
                m = resize_crop(masks[..., None], self.resize_size, self.crop_size, False)                   # [Q,1,224,224] (converted to fp32 inside)
                mg = resize_crop(m.permute(0, 2, 3, 1), self.grid[0], None, False).reshape(masks.shape[0], -1)
            else:
                m = _center_crop(_resize_short_edge(masks[:, None] * 1.0, self.resize_size, "bilinear"), self.crop_size)
                mg = _resize_short_edge(m, self.grid[0], "bilinear").reshape(masks.shape[0], -1)
        return token_assemble(feats, self.grid, mg, 0.1, compact=compact)

    @torch.no_grad()
    def tokens_rgba(self, rgba: torch.Tensor, compact: bool = False):
        """RGBA query images [Q,H,W,4] as the evaluation loop holds them (pose_estimation/test.py:75-81) -> (tokens, keep): the
        composite on white and the alpha mask are taken inside the two resize launches.  Equals ``tokens(rgb * a + (1 - a), a)``
        bit for bit."""
        return self.tokens_from_preprocessed(*self.preprocess(rgba), compact=compact)

    # The two halves of ``tokens_rgba`` / ``tokens(imgs, None)``, for a caller that keeps the second half in a captured graph: the
    # first half is the only part that reads the full-size images, so run eagerly it takes them WHERE THEY LIE (a slice of the
    # dataset's own tensor) and the graph's static inputs are the 224 x 224 results, not a copy of 32 images of 800 x 800 x 4.
    @torch.no_grad()
    def preprocess(self, imgs: torch.Tensor, out=None):
        """imgs [Q,H,W,4] (RGBA) or [Q,H,W,3] -> (xin [Q,3,crop,crop] normalised, alpha [Q,1,crop,crop] or None), written into
        ``out = (xin, alpha)`` when given.  Native preprocessing only."""
        if not self.native_preprocess:
            raise RuntimeError("ImageFrontEnd.preprocess is the iff_image_resize_crop path (native_preprocess=True)")
        o_x, o_m = (None, None) if out is None else out
        if imgs.shape[-1] == 4:
            xin = resize_crop_rgba(imgs, self.resize_size, self.crop_size, RESIZE_RGB_ON_WHITE, True, IMAGENET_DEFAULT_MEAN,
                                   IMAGENET_DEFAULT_STD, out=o_x)
            return xin, resize_crop_rgba(imgs, self.resize_size, self.crop_size, RESIZE_ALPHA, False, out=o_m)
        return resize_crop(imgs, self.resize_size, self.crop_size, True, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD, out=o_x), None

    @torch.no_grad()
    def tokens_from_preprocessed(self, xin: torch.Tensor, alpha: Optional[torch.Tensor], compact: bool = False):
        """(xin, alpha) of ``preprocess`` -> (tokens [Q, gh*gw, C+14], keep [Q, gh*gw]) (+ rows [Q] with ``compact``: ``token_assemble``)."""
        if self.backbone_autocast is None:
            feats = self.backbone.forward_features(xin)["x_norm_patchtokens"]
        else:
            with torch.autocast(device_type="cuda", dtype=self.backbone_autocast):
                feats = self.backbone.forward_features(xin)["x_norm_patchtokens"]
        mg = None
        if alpha is not None:
            mg = resize_crop(alpha.permute(0, 2, 3, 1), self.grid[0], None, False).reshape(xin.shape[0], -1)
        return token_assemble(feats, self.grid, mg, 0.1, compact=compact)
