"""Mirror of ``pose_estimation/identification_module.py``: same class, constructor, attributes and method signatures.

The ray side (encoder, k_proj), q_proj, the softmax over the ray axis, the column-sum score and the top-k run in
libiffnerf_hip (fp32 MFMA).  The image side up to the token tensor (resize / crop / normalise, DINOv2) is the stock
PyTorch-ROCm module: it is outside the accelerated path.
``state_dict`` keys equal the reference's (``norm_mean``, ``norm_std``, ``image_preprocessing_net.*``,
``ray_preprocessor.mlp*.{0,2}.*``, ``attention.{q,k}_proj.*``), so ``id_module.th`` loads unchanged.
Grad mode: under ``torch.no_grad`` / with frozen parameters (the whole north-star path) stage C
runs in libiffnerf_hip; when autograd has to flow -- ``pose_estimation/train.py:97-119`` calls the module with trainable
parameters -- the ray encoder and the attention evaluate the same formulas in differentiable PyTorch-ROCm ops on the GPU
(ray_preprocessor.py / multihead_attention.py of this package), so ``train_id_module`` back-propagates unchanged.
"""
from __future__ import annotations

import weakref

import torch
import torch.nn.functional as F

from .backbone import create_backbone
from .multihead_attention import MultiHeadAttention
from .ray_preprocessor import RayPreprocessor

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def _resize_short_edge(x, size, mode):
    """torchvision ``Resize(size, antialias=True)`` on an NCHW tensor: shorter edge -> size, aspect kept."""
    h, w = x.shape[-2:]
    if h <= w:
        nh, nw = size, max(1, int(size * w / h))
    else:
        nh, nw = max(1, int(size * h / w)), size
    return F.interpolate(x, size=(nh, nw), mode=mode, align_corners=False, antialias=True)


def _center_crop(x, size):
    h, w = x.shape[-2:]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return x[..., top:top + size, left:left + size]


class IdentificationModule(torch.nn.Module):
    def __init__(self, backbone_type: str = "superpoint"):
        super().__init__()
        assert backbone_type in ["dino", "superpoint"]
        self.image_preprocessing_net, backbone_wh, img_num_features = create_backbone(type=backbone_type, pretrained=True)
        self.norm_mean = torch.nn.Parameter(torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=torch.float32), requires_grad=False)
        self.norm_std = torch.nn.Parameter(torch.tensor(IMAGENET_DEFAULT_STD, dtype=torch.float32), requires_grad=False)
        self.resize_size, self.crop_size = 256, 224
        self.backbone_wh = backbone_wh
        self.img_num_features = img_num_features
        self.ray_preprocessor = RayPreprocessor(featureC=256, fea_output=img_num_features)
        self.attention = MultiHeadAttention(img_num_features, img_num_features + 14, img_num_features, 1)
        me = weakref.ref(self)
        self.ray_preprocessor._owner = me
        self.attention._owner = me
        self._net = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_tables())

    # ------------------------------------------------------------------ preprocessing (out of the accelerated path)
    def transformations(self, nchw):
        x = _center_crop(_resize_short_edge(nchw, self.resize_size, "bicubic"), self.crop_size)
        mean = torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_DEFAULT_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std

    def mask_transformations(self, n1hw):
        x = _center_crop(_resize_short_edge(n1hw, self.resize_size, "bilinear"), self.crop_size)
        return _resize_short_edge(x, self.backbone_wh[0], "bilinear")

    @staticmethod
    def get_img_position_encoding(img_features_shape, freqs, dtype=torch.float32, device="cpu"):
        """[*shape, 2 + 4*freqs]: grid position in [-1,1]^2, then sin / cos of it at octaves 0..freqs-1 (reference :76-99)."""
        axes = [torch.linspace(-1.0, 1.0, steps=s, dtype=dtype, device=device) for s in img_features_shape]
        pos = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(axes))
        bands = (2 ** torch.arange(freqs, device=device)).to(dtype)
        ang = (pos[..., None] * bands).flatten(-2)
        return torch.cat((pos, ang.sin(), ang.cos()), dim=-1).reshape(*img_features_shape, -1)

    def image_processing(self, img, mask):
        """[H,W,3] image + [H,W] mask -> (tokens with position code [M, C+14], tokens [M, C]) (reference :130-160)."""
        norm_img = self.transformations(img[None].permute(0, 3, 1, 2))
        keep = self.mask_transformations(mask[None, None] * 1.0)[0, 0] > 0.1
        tokens = self.image_preprocessing_net.forward_features(norm_img)["x_norm_patchtokens"][0]
        gh, gw = self.backbone_wh
        tokens = tokens.reshape(gh, gw, self.img_num_features)
        pe = self.get_img_position_encoding((gh, gw), 3, dtype=img.dtype, device=img.device)
        full = torch.cat((tokens, pe.to(tokens.dtype)), dim=-1)
        return full[keep].view(-1, full.shape[-1]), tokens[keep].view(-1, tokens.shape[-1])

    # ------------------------------------------------------------------ kernel handle
    def invalidate_tables(self):
        if getattr(self, "_net", None) is not None:
            self._net.close()
        self._net = None
        self._net_key = None

    def _weights_key(self):
        """Identity + in-place version of every tensor the kernel handle was built from.  ``optimizer.step()`` updates the
        parameters in place between the validations of pose_estimation/train.py:126,145,188,222 without going through
        ``_apply`` or ``load_state_dict``: the version counters move, and the handle is rebuilt from the new weights."""
        ps = list(self.ray_preprocessor.parameters()) + list(self.attention.parameters())
        return tuple((p.data_ptr(), p._version) for p in ps)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_tables()
        return out

    def _idnet(self):
        key = self._weights_key()
        if self._net is not None and getattr(self, "_net_key", None) != key:
            self.invalidate_tables()
        if self._net is None:
            from ..hip_identify import IdNetHandle
            w = {"ray_preprocessor." + k: v for k, v in self.ray_preprocessor.state_dict().items()}
            w.update({"attention." + k: v for k, v in self.attention.state_dict().items()})
            dev = self.attention.q_proj.weight.device
            if dev.type != "cuda":
                raise RuntimeError("IdentificationModule is on the CPU: move it to the GPU; libiffnerf_hip has no CPU path")
            self._net = IdNetHandle(w, dev)
            self._net_key = key
        return self._net

    def _training_graph(self, *tensors) -> bool:
        """Autograd has to flow through stage C (trainable parameters under grad mode): use the differentiable formulation."""
        from .ray_preprocessor import needs_autograd
        return needs_autograd(self.ray_preprocessor, *tensors) or needs_autograd(self.attention, *tensors)

    # ------------------------------------------------------------------ stage C
    def attention_from_tokens(self, features_img_w_pe_flat, rays_ori, rays_dir, rays_rgb, materialize_map=True):
        """Token boundary -> (score [N], attention map [M,N] or None).  The ray encoder + k_proj are recomputed per
        call, exactly as the reference does per image (identification_module.py:164)."""
        from .. import hip_identify as H
        if self._training_graph(features_img_w_pe_flat, rays_ori, rays_dir, rays_rgb):
            # identification_module.py:164-167 in differentiable torch ops (the modules dispatch the same way themselves)
            attention_map = self.attention(features_img_w_pe_flat, self.ray_preprocessor(rays_ori, rays_dir, rays_rgb))
            return attention_map.sum(0), attention_map
        net = self._idnet()
        if getattr(self, "fold_heads", True):
            # mlp2.2, k_proj and q_proj folded into one token-side Linear; encoder + logits in one launch
            # (include/iffnerf_hip.h iff_ray_logits_folded)
            logits, rmax, rsum = net.ray_logits_folded(net.q_fold(features_img_w_pe_flat), rays_ori, rays_dir, rays_rgb)
        else:
            _, k = net.ray_encode(rays_ori, rays_dir, rays_rgb, want_features=False, want_k=True)
            logits, rmax, rsum = H.attn_logits(net.q_proj(features_img_w_pe_flat), k)
        score = H.attn_colsum(logits, rmax, rsum, write_attention=materialize_map)
        return score, (logits if materialize_map else None)

    def run_attention(self, img, mask, rays_ori, rays_dir, rays_rgb):
        tokens_pe, tokens = self.image_processing(img, mask)
        score, attention_map = self.attention_from_tokens(tokens_pe, rays_ori, rays_dir, rays_rgb)
        return score, attention_map, tokens

    def forward(self, img, mask, rays_ori, rays_dir, rays_rgb, rays_to_test: int = -1):
        used = torch.randperm(rays_ori.shape[0], device=img.device, dtype=torch.long)
        if rays_to_test != -1:
            used = used[:rays_to_test]
        scores, attention_map, tokens = self.run_attention(img, mask, rays_ori[used], rays_dir[used], rays_rgb[used])
        return scores, attention_map, tokens, used

    @torch.no_grad()
    def test_image(self, img, mask, rays_ori, rays_dir, rays_rgb, rays_to_output: int = 100):
        from .. import hip_identify as H
        scores, attention_map, _ = self.run_attention(img, mask, rays_ori, rays_dir, rays_rgb)
        indices, values = H.topk(scores, rays_to_output)
        return indices, values, scores, attention_map
