"""Image backbone factory with the reference's name (pose_estimation/backbone.py:3-14).

DINOv2 ViT-S/14 is a third-party model; its published weights come from ``torch.hub`` (network or a warm hub cache).
Where neither exists (tests, bench.py on the GPU box) ``create_standin_backbone`` gives a seeded, randomly initialised
module with DINOv2's module tree, ``state_dict`` keys and interface (``forward_features(x)["x_norm_patchtokens"]`` ->
[B, 256, 384] for 224 x 224 inputs): same shapes, same FLOPs, same checkpoint keys; its features mean nothing.
"""
import torch
import torch.nn.functional as F


def create_backbone(type="dino", pretrained=False, filter_size=4, pool_only=True, _force_nonfinetuned=False, **kwargs):
    """The reference's factory (backbone.py:3-14): returns the hub module ITSELF -- same class, same parameters, same
    ``state_dict`` keys, so ``IdentificationModule.state_dict()`` carries ``image_preprocessing_net.<dinov2 key>`` and a
    reference-trained ``id_module.th`` strict-loads (train_eval_pose_est.py:59-66).  ``hip_vit.serve_natively`` installs the
    native ``forward_features`` on that module (no-grad inference through ``iff_vit_forward``; everything else is the module's
    own torch code) in the reference's fp32 accuracy class (``precision="bf16"`` in ``kwargs``: bf16 operands, a throughput
    option); ``native=False`` in ``kwargs`` leaves the module untouched, exactly as the reference returns it."""
    if type != "dino":
        raise RuntimeError("only the 'dino' backbone exists in the reference (backbone.py:11-14)")
    model = _hub_load("facebookresearch/dinov2", "dinov2_vits14")
    if kwargs.get("native", True):
        from ..hip_vit import serve_natively
        model = serve_natively(model, (16, 16), 14, precision=kwargs.get("precision", "fp32"))
    return model, (16, 16), 384


def _hub_load(repo: str, name: str):
    """``torch.hub.load`` (network or a warm hub cache); one seam for the tests, which have neither."""
    return torch.hub.load(repo, name)


class _LayerScale(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = torch.nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Attention(torch.nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv, self.proj = torch.nn.Linear(dim, 3 * dim), torch.nn.Linear(dim, dim)

    def forward(self, x):
        B, T, C = x.shape
        q, k, v = self.qkv(x).view(B, T, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        return self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, C))


class _Mlp(torch.nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = torch.nn.Linear(dim, hidden), torch.nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(torch.nn.Module):
    def __init__(self, dim, heads, mlp):
        super().__init__()
        self.norm1, self.attn, self.ls1 = torch.nn.LayerNorm(dim, eps=1e-6), _Attention(dim, heads), _LayerScale(dim)
        self.norm2, self.mlp, self.ls2 = torch.nn.LayerNorm(dim, eps=1e-6), _Mlp(dim, mlp), _LayerScale(dim)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchEmbed(torch.nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = torch.nn.Conv2d(3, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class SeededViTS14(torch.nn.Module):
    """ViT-S/14 with the module tree -- hence the ``state_dict`` keys and shapes -- of DINOv2's ``dinov2_vits14``: ``cls_token``,
    ``pos_embed`` [1, 1 + 37 x 37, 384] (the 518-pixel pretraining grid, resized bicubically to the input's grid as DINOv2's
    ``interpolate_pos_encoding`` does), ``mask_token``, ``patch_embed.proj``, 12 x ``blocks.i.{norm1, attn.qkv, attn.proj, ls1.gamma,
    norm2, mlp.fc1, mlp.fc2, ls2.gamma}``, ``norm``.  Randomly initialised from ``seed``: a stand-in for timing, plumbing and the
    checkpoint-key contract, not a feature extractor."""

    def __init__(self, seed: int = 0, dim: int = 384, depth: int = 12, heads: int = 6, patch: int = 14, pretrain_grid: int = 37):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.patch_size = patch
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = torch.nn.Parameter(torch.zeros(1, 1 + pretrain_grid * pretrain_grid, dim))
        self.mask_token = torch.nn.Parameter(torch.zeros(1, dim))
        self.patch_embed = _PatchEmbed(dim, patch)
        self.blocks = torch.nn.ModuleList(_Block(dim, heads, 4 * dim) for _ in range(depth))
        self.norm = torch.nn.LayerNorm(dim, eps=1e-6)
        self.head = torch.nn.Identity()
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=gen) * (0.02 if p.dim() == 3 else 1.0 / (p[0].numel() ** 0.5)))
            for b in self.blocks:
                b.ls1.gamma.fill_(0.1), b.ls2.gamma.fill_(0.1)

    def forward_features(self, x, masks=None):
        from ..hip_vit import interpolate_pos_embed
        gh, gw = x.shape[-2] // self.patch_size, x.shape[-1] // self.patch_size
        t = self.patch_embed(x)
        if masks is not None:
            t = torch.where(masks.unsqueeze(-1), self.mask_token.to(t.dtype).unsqueeze(0), t)
        pos = interpolate_pos_embed(self.pos_embed, gh, gw, self.patch_size, differentiable=True)
        t = torch.cat((self.cls_token.expand(t.shape[0], -1, -1), t), dim=1) + pos
        for b in self.blocks:
            t = b(t)
        t = self.norm(t)
        return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1:], "x_prenorm": t, "masks": masks}

    def forward(self, x):
        return self.head(self.forward_features(x)["x_norm_clstoken"])


def create_standin_backbone(seed: int = 0, native: bool = False, precision: str = "fp32"):
    """(module, (16, 16), 384) like ``create_backbone("dino")``, without the network.  ``native``: ``hip_vit.serve_natively`` applied."""
    m = SeededViTS14(seed).eval()
    if native:
        from ..hip_vit import serve_natively
        m = serve_natively(m, (16, 16), 14, precision=precision)
    return m, (16, 16), 384
