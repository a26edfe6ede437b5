"""Image backbone factory with the reference's name (pose_estimation/backbone.py:3-14).

DINOv2 ViT-S/14 is a third-party model outside the accelerated path; it stays a stock
PyTorch-ROCm module.  ``torch.hub`` needs network access or a warm hub cache; where neither exists (tests, bench.py on
the GPU box) ``create_standin_backbone`` gives a seeded, randomly initialised module of the same architecture and
interface (``forward_features(x)["x_norm_patchtokens"]`` -> [B, 256, 384] for 224 x 224 inputs): same shapes, same
FLOPs, same launch pattern -- what the image-side capture (iffnerf_amd/image_frontend.py) needs; its features mean nothing.
"""
import torch
import torch.nn.functional as F


def create_backbone(type="dino", pretrained=False, filter_size=4, pool_only=True, _force_nonfinetuned=False, **kwargs):
    """The reference's factory (backbone.py:3-14).  The hub module is wrapped in ``hip_vit.NativeViT``, which serves its no-grad
    ``forward_features`` from ``iff_vit_forward`` (bf16 matrix cores) and keeps the module itself -- parameters, training path --
    untouched (its state_dict keys gain the ``module.`` prefix of the wrapper; ``native=False`` in ``kwargs`` returns the stock
    module exactly as the reference does)."""
    if type != "dino":
        raise RuntimeError("only the 'dino' backbone exists in the reference (backbone.py:11-14)")
    model = torch.hub.load("facebookresearch/dinov2", "dinov2_vits14")
    if kwargs.get("native", True):
        from ..hip_vit import NativeViT
        model = NativeViT(model, (16, 16), 14)
    return model, (16, 16), 384


class _Block(torch.nn.Module):
    def __init__(self, dim, heads, mlp):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = torch.nn.LayerNorm(dim, eps=1e-6), torch.nn.LayerNorm(dim, eps=1e-6)
        self.qkv, self.proj = torch.nn.Linear(dim, 3 * dim), torch.nn.Linear(dim, dim)
        self.fc1, self.fc2 = torch.nn.Linear(dim, mlp), torch.nn.Linear(mlp, dim)
        self.ls1, self.ls2 = torch.nn.Parameter(torch.ones(dim)), torch.nn.Parameter(torch.ones(dim))     # LayerScale

    def forward(self, x):
        B, T, C = x.shape
        q, k, v = self.qkv(self.norm1(x)).view(B, T, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, C)
        x = x + self.ls1 * self.proj(a)
        return x + self.ls2 * self.fc2(F.gelu(self.fc1(self.norm2(x))))


class SeededViTS14(torch.nn.Module):
    """ViT-S/14 in DINOv2's shape: 14 x 14 patch embedding, class token, 12 blocks of width 384 (6 heads, MLP 1536,
    LayerScale), final LayerNorm.  Randomly initialised from ``seed``: a stand-in for timing and plumbing, not a feature
    extractor."""

    def __init__(self, seed: int = 0, dim: int = 384, depth: int = 12, heads: int = 6, patch: int = 14, grid: int = 16):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.patch_embed = torch.nn.Conv2d(3, dim, patch, patch)
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = torch.nn.Parameter(torch.zeros(1, 1 + grid * grid, dim))
        self.blocks = torch.nn.ModuleList(_Block(dim, heads, 4 * dim) for _ in range(depth))
        self.norm = torch.nn.LayerNorm(dim, eps=1e-6)
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() > 1:
                    p.copy_(torch.randn(p.shape, generator=gen) * (0.02 if p.dim() == 3 else 1.0 / (p[0].numel() ** 0.5)))
            for b in self.blocks:
                b.ls1.fill_(0.1), b.ls2.fill_(0.1)

    def forward_features(self, x):
        t = self.patch_embed(x).flatten(2).transpose(1, 2)
        t = torch.cat((self.cls_token.expand(t.shape[0], -1, -1), t), dim=1) + self.pos_embed
        for b in self.blocks:
            t = b(t)
        t = self.norm(t)
        return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1:]}


def create_standin_backbone(seed: int = 0, native: bool = False):
    """(module, (16, 16), 384) like ``create_backbone("dino")``, without the network.  ``native``: wrapped in ``hip_vit.NativeViT``."""
    m = SeededViTS14(seed).eval()
    if native:
        from ..hip_vit import NativeViT
        m = NativeViT(m, (16, 16), 14)
    return m, (16, 16), 384
