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
        if normals is not None:
            raise RuntimeError("Ref.forward: explicit normals are not on the IFFNeRF path (models/tensorBase.py:891-896 "
                               "passes None); only predicted normals are built")
        rgb = self._standalone_handle(viewdirs.device).ref_shade(viewdirs, features)
        return rgb, None

    def compute_normals(self, features: torch.Tensor):
        return self._standalone_handle(features.device).head_normals(features)
