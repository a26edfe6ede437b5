"""Oracle: ray encoder, ray<->patch attention, column-sum score and top-k (stage C).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, on torch-CPU,
  models/tensorBase.py:14-20 (positional_encoding)
  pose_estimation/ray_preprocessor.py:29-39
  pose_estimation/multihead_attention.py:4-12,56-66
  pose_estimation/identification_module.py:76-99,162-168,193-209
Weights come as the reference's ``id_module.th`` state_dict keys.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


def freq_encode(x: torch.Tensor, n_freq: int) -> torch.Tensor:
    """tensorBase.py:14-20: [sin(x_j 2^k)] (j-major, k-minor) then the cosines."""
    bands = (2 ** torch.arange(n_freq).float())
    p = (x[..., None] * bands).reshape(x.shape[:-1] + (n_freq * x.shape[-1],))
    return torch.cat([torch.sin(p), torch.cos(p)], dim=-1)


def ray_input(o, d, rgb, pospe=8, viewpe=8, rgbpe=6):
    """ray_preprocessor.py:30-37: the 141-wide encoder input."""
    return torch.cat([o, d, rgb, freq_encode(o, pospe), freq_encode(d, viewpe), freq_encode(rgb, rgbpe)], dim=-1)


def ray_encode(w: Dict[str, torch.Tensor], o, d, rgb):
    """ray_preprocessor.py:29-39 -> [N, fea_output]."""
    x = ray_input(o, d, rgb)
    h = F.relu(F.linear(x, w["ray_preprocessor.mlp.0.weight"], w["ray_preprocessor.mlp.0.bias"]))
    h = F.relu(F.linear(h, w["ray_preprocessor.mlp.2.weight"], w["ray_preprocessor.mlp.2.bias"]))
    g = torch.cat((h, x), dim=-1)
    g = F.relu(F.linear(g, w["ray_preprocessor.mlp2.0.weight"], w["ray_preprocessor.mlp2.0.bias"]))
    return F.linear(g, w["ray_preprocessor.mlp2.2.weight"], w["ray_preprocessor.mlp2.2.bias"])


def attention_map(w: Dict[str, torch.Tensor], img_feat, ray_feat, return_parts: bool = False):
    """multihead_attention.py:56-66 + :4-12 with mask=None: softmax over the ray axis."""
    q = F.linear(img_feat, w["attention.q_proj.weight"], w["attention.q_proj.bias"])
    k = F.linear(ray_feat, w["attention.k_proj.weight"], w["attention.k_proj.bias"])
    logits = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size()[-1])
    att = F.softmax(logits, dim=-1)
    if return_parts:
        return att, logits, q, k
    return att


def run_attention(w, img_feat, o, d, rgb):
    """identification_module.py:162-168 from the image-token boundary on."""
    att = attention_map(w, img_feat, ray_encode(w, o, d, rgb))
    return torch.sum(att, dim=0), att


def test_image(w, img_feat, o, d, rgb, rays_to_output: int = 100):
    """identification_module.py:193-209."""
    score, att = run_attention(w, img_feat, o, d, rgb)
    top = torch.topk(score, k=rays_to_output)
    return top.indices, top.values, score, att


def image_position_encoding(shape=(16, 16), freqs=3, dtype=torch.float32):
    """identification_module.py:76-99 -> [*shape, 2 + 4*freqs]."""
    axes = [torch.linspace(-1.0, 1.0, steps=s, dtype=dtype) for s in shape]
    pos = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(shape))
    bands = (2 ** torch.arange(freqs).float())
    p = (pos[..., None] * bands).reshape(pos.shape[:-1] + (freqs * pos.shape[-1],))
    out = torch.cat([pos, torch.sin(p), torch.cos(p)], dim=-1)
    return out.reshape(*shape, out.shape[-1])


def tokens_with_pe(patch_tokens: torch.Tensor, keep_mask: torch.Tensor, grid=(16, 16)):
    """identification_module.py:137-160 after the backbone: [H*W,C] tokens -> [M, C+14]."""
    C = patch_tokens.shape[-1]
    feat = patch_tokens.reshape(grid[0], grid[1], C).permute(2, 0, 1)
    pe = image_position_encoding(feat.shape[-2:], 3, dtype=patch_tokens.dtype)
    full = torch.cat([feat, pe.permute(2, 0, 1)], dim=0).permute(1, 2, 0)
    return full[keep_mask].view(-1, full.shape[-1])
