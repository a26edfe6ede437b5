"""Mirror of the reference's ``models/ref.py`` (Ref-NeRF shading head) over the HIP library.

Same constructor signature, same sub-module names and therefore the same ``state_dict`` keys as
reference models/ref.py:48-101, so checkpoints load unchanged.  ``forward`` (reference :103-152) and
``compute_normals`` (:154-155) run in libiffnerf_hip (kernels ``k_ref_shade`` / ``k2_point_app``); inside
``TensorBase.forward`` the head is fused into the march kernel and this module only carries the weights.
"""
from __future__ import annotations

import math

import torch

from .. import synthetic


class _Affine(torch.nn.Module):
    """Parameter-free ``x * mul + add`` stage; keeps the Sequential indices of the reference layout."""

    def __init__(self, mul: float = 1.0, add: float = 0.0):
        super().__init__()
        self.mul, self.add = mul, add

    def forward(self, x):
        return x * self.mul + self.add


class _UnitNorm(torch.nn.Module):
    def forward(self, x):
        return torch.nn.functional.normalize(x, p=2, dim=-1)


class IntegratedDirEnc(torch.nn.Module):
    """Holds the (m, l) table and z-polynomial coefficients (reference models/ref_utils.py:23-80)."""

    def __init__(self, deg_view: int):
        super().__init__()
        self.ml_array = torch.nn.Parameter(torch.from_numpy(synthetic.ide_ml_pairs(deg_view)), requires_grad=False)
        self.mat = torch.nn.Parameter(torch.from_numpy(synthetic.ide_coeff_matrix(deg_view)), requires_grad=False)


class Ref(torch.nn.Module):
    def __init__(self, in_channels, viewpe=6, feature_c=128, deg_view=4, predicted_normals=True,
                 rgb_premultiplier=1.0, rgb_bias=0.0):
        super().__init__()
        if deg_view != 4 or not predicted_normals or abs(rgb_premultiplier - 1.0) > 1e-7 or rgb_bias > 1e-7:
            raise RuntimeError("Ref: only the configuration the reference instantiates (deg_view=4, predicted normals, "
                               "no rgb premultiplier/bias; models/tensorBase.py:339-342) is built")
        self.dir_enc_fn = IntegratedDirEnc(deg_view)
        self.deg_view = deg_view
        self.rgb_padding = 0.001
        self.in_mlpC = (3 + 2 * viewpe * 3) + in_channels
        self.viewpe = viewpe
        self.predicted_normals = predicted_normals
        lin = torch.nn.Linear
        self.diffuse_color_mlp = torch.nn.Sequential(lin(in_channels, 3), _Affine(add=-math.log(3.0)), torch.nn.Sigmoid())
        self.tint_color_mlp = torch.nn.Sequential(lin(in_channels, 3), torch.nn.Sigmoid())
        self.roughness_mlp = torch.nn.Sequential(lin(in_channels, 1), _Affine(add=-1.0), torch.nn.Softplus())
        self.bottleneck_mlp = lin(in_channels, feature_c)
        self.normal_mlp = torch.nn.Sequential(lin(in_channels, 3), _UnitNorm(), _Affine(mul=-1.0))
        self.specular_mlp = torch.nn.Sequential(lin(feature_c + 19 * 2 + 1, 3), torch.nn.Sigmoid())
        self._handle = None

    # ------------------------------------------------------------------ weights -> kernel view
    def head_tensors(self):
        return {k: v for k, v in self.state_dict().items()}

    def _standalone_handle(self, device):
        """A handle that carries only this head (used when Ref is called outside a TensorBase)."""
        from ..hip_field import FieldHandle
        if self._handle is None or self._handle.device != torch.device(device):
            self._handle = FieldHandle.head_only(self.head_tensors(), device)
        return self._handle

    def invalidate_tables(self):
        self._handle = None

    def forward(self, pts, viewdirs, features, normals):
        if torch.is_grad_enabled() and (viewdirs.requires_grad or features.requires_grad) and normals is None:
            return self.forward_autograd(viewdirs, features), None
        if normals is not None:
            raise RuntimeError("Ref.forward: explicit normals are not on the IFFNeRF path (models/tensorBase.py:891-896 "
                               "passes None); only predicted normals are built")
        rgb = self._standalone_handle(viewdirs.device).ref_shade(viewdirs, features)
        return rgb, None

    def forward_autograd(self, viewdirs, features):
        """The head in differentiable torch ops on the tensors' device (reference :103-152 with ``normals=None``), for callers
        that back-propagate through the colours to the view directions and features (SURVEY.md 8b grad mode; the iNeRF
        refinement, inerf/estimate_pose_inerf.py:164-176).  Per ray, not per sample: a 27 -> 128 bottleneck and a few 27 -> 3
        heads.  The integrated directional encoding (ref_utils.py:82-112) is written in real arithmetic: (x + iy)^m by the
        angle-addition recurrence instead of a complex pow."""
        lin = torch.nn.functional.linear
        F = features
        n = -torch.nn.functional.normalize(lin(F, self.normal_mlp[0].weight, self.normal_mlp[0].bias), p=2, dim=-1)
        tint = torch.sigmoid(lin(F, self.tint_color_mlp[0].weight, self.tint_color_mlp[0].bias))
        rough = torch.nn.functional.softplus(lin(F, self.roughness_mlp[0].weight, self.roughness_mlp[0].bias) - 1.0)
        bott = lin(F, self.bottleneck_mlp.weight, self.bottleneck_mlp.bias)
        v = -viewdirs
        refl = 2.0 * (n * v).sum(-1, keepdim=True) * n - v                                   # ref_utils.py:6-18
        ml, mat = self.dir_enc_fn.ml_array, self.dir_enc_fn.mat                                # [2,19] (m, l), [9,19]
        x, y, z = refl[..., 0:1], refl[..., 1:2], refl[..., 2:3]
        re, im, zp = [torch.ones_like(x)], [torch.zeros_like(x)], [torch.ones_like(z)]
        for _ in range(2 ** (self.deg_view - 1)):        # largest m of the (m, l) table: l = 1, 2, .., 2^(deg_view-1), m <= l (no host read)
            re, im = re + [re[-1] * x - im[-1] * y], im + [re[-1] * y + im[-1] * x]
        for _ in range(mat.shape[0] - 1):
            zp.append(zp[-1] * z)
        m_idx = ml[0].long()
        zpoly = torch.cat(zp, -1) @ mat                                                        # [n,19]
        att = torch.exp(-(0.5 * ml[1] * (ml[1] + 1)).to(x.dtype) * rough)                    # [n,19]
        enc = torch.stack((torch.cat(re, -1)[..., m_idx] * zpoly * att, torch.cat(im, -1)[..., m_idx] * zpoly * att), -1)
        ndotv = (n * viewdirs).sum(-1, keepdim=True)
        spec = torch.sigmoid(lin(torch.cat([bott, enc.flatten(-2), ndotv], -1), self.specular_mlp[0].weight,
                                 self.specular_mlp[0].bias))
        diff = torch.sigmoid(lin(F, self.diffuse_color_mlp[0].weight, self.diffuse_color_mlp[0].bias) - math.log(3.0))
        lin_rgb = tint * spec + diff
        eps = torch.finfo(lin_rgb.dtype).eps                                                   # models/image.py:6-13
        srgb = torch.where(lin_rgb <= 0.0031308, 323.0 / 25.0 * lin_rgb,
                           (211.0 * torch.clamp(lin_rgb, min=eps) ** (5.0 / 12.0) - 11.0) / 200.0)
        return torch.clip(srgb, 0.0, 1.0) * (1.0 + 2.0 * self.rgb_padding) - self.rgb_padding

    def compute_normals(self, features: torch.Tensor):
        return self._standalone_handle(features.device).head_normals(features)
