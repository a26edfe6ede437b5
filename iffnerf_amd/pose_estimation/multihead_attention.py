"""Mirror of ``pose_estimation/multihead_attention.py``: q/k projections + softmax over the ray axis, in HIP.

``MultiHeadAttention`` keeps the reference's constructor and the ``q_proj`` / ``k_proj`` parameter names
(multihead_attention.py:31-54).  ``forward`` (reference :56-66) = ``iff_q_proj`` + ``iff_k_proj`` + ``iff_attn_logits`` +
``iff_attn_colsum`` and returns the attention map; the column-sum score computed on the way is kept on the module as
``last_score`` so ``IdentificationModule.run_attention`` does not reduce the map a second time.

Grad mode: as in ray_preprocessor.py -- when autograd has to flow (``pose_estimation/train.py``) the same
formula runs in differentiable PyTorch-ROCm ops on the GPU; inference never takes that branch.
"""
from __future__ import annotations

import math

import torch


def scaled_attention_product(q, k, mask=None):
    """softmax(q k^T / sqrt(d)) over the last axis (reference :4-12), mask=None only."""
    from .. import hip_identify as H
    if mask is not None:
        raise RuntimeError("scaled_attention_product: masks are not on the IFFNeRF path (run_attention passes None)")
    logits, rmax, rsum = H.attn_logits(q, k)
    H.attn_colsum(logits, rmax, rsum, write_attention=True)
    return logits


class MultiHeadAttention(torch.nn.Module):
    def __init__(self, ray_fea_size, img_fea_size, embed_dim, num_heads=1):
        super().__init__()
        if num_heads != 1:
            raise RuntimeError("MultiHeadAttention: the reference instantiates one head (identification_module.py:72-74)")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.q_proj = torch.nn.Linear(img_fea_size, embed_dim)
        self.k_proj = torch.nn.Linear(ray_fea_size, embed_dim)
        for lin in (self.q_proj, self.k_proj):      # "original Transformer initialisation" (reference :49-54)
            torch.nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)
        self._owner = None
        self.last_score = None

    def forward(self, img_features, ray_features, mask=None):
        from .. import hip_identify as H
        from .ray_preprocessor import needs_autograd
        if mask is not None:
            raise RuntimeError("MultiHeadAttention.forward: masks are not on the IFFNeRF path")
        if needs_autograd(self, img_features, ray_features):
            q, k = self.q_proj(img_features), self.k_proj(ray_features)                 # multihead_attention.py:60-61
            att = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.size()[-1]), dim=-1)    # :4-12
            self.last_score = att.sum(0)
            return att
        if self._owner is None:
            raise RuntimeError("MultiHeadAttention.forward runs through its IdentificationModule's kernel handle")
        net = self._owner()._idnet()
        logits, rmax, rsum = H.attn_logits(net.q_proj(img_features), net.k_proj(ray_features))
        self.last_score = H.attn_colsum(logits, rmax, rsum, write_attention=True)
        return logits
