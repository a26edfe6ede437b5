"""Mirror of ``pose_estimation/ray_preprocessor.py``: same constructor, same parameter names, forward in HIP.

``mlp`` / ``mlp2`` keep the reference's Sequential layout (ray_preprocessor.py:9-25) so ``id_module.th`` loads
unchanged; ``forward`` (reference :29-39) runs ``iff_ray_encode`` (positional encoding + 4 fp32-accurate MFMA GEMMs).

Grad mode: when autograd has to flow -- grad enabled and a parameter or an input requires grad, i.e.
``pose_estimation/train.py:97-119`` training this module -- ``forward`` evaluates the same formula with differentiable
PyTorch-ROCm ops ON THE GPU (``needs_autograd`` / ``_forward_autograd``).  Inference (``torch.no_grad`` / frozen
parameters, everything the north-star path does) never takes that branch, and neither branch accepts CPU tensors.
"""
from __future__ import annotations

import torch


def needs_autograd(module: torch.nn.Module, *tensors) -> bool:
    """True when a backward pass may be asked for: grad mode on and a parameter of ``module`` or one of ``tensors`` requires
    grad.  Such calls run the differentiable formulation on the GPU; CPU tensors are refused on every path."""
    if not torch.is_grad_enabled():
        return False
    if not (any(p.requires_grad for p in module.parameters()) or any(torch.is_tensor(t) and t.requires_grad for t in tensors)):
        return False
    for t in list(module.parameters()) + [t for t in tensors if torch.is_tensor(t)]:
        if not t.is_cuda:
            raise RuntimeError(f"{type(module).__name__}: tensors must live on the GPU (got {t.device}); there is no CPU path")
    return True


class RayPreprocessor(torch.nn.Module):
    def __init__(self, viewpe=8, pospe=8, rgbpe=6, featureC=128, fea_output=128):
        super().__init__()
        if (viewpe, pospe, rgbpe) != (8, 8, 6):
            raise RuntimeError("RayPreprocessor: only viewpe=8, pospe=8, rgbpe=6 (the reference's defaults, the ones "
                               "IdentificationModule uses) are built")
        self.in_mlpC = 2 * viewpe * 3 + 3 + 2 * pospe * 3 + 3 + 2 * rgbpe * 3 + 3
        relu = torch.nn.ReLU
        self.mlp = torch.nn.Sequential(torch.nn.Linear(self.in_mlpC, featureC), relu(inplace=True),
                                       torch.nn.Linear(featureC, featureC), relu(inplace=True))
        self.mlp2 = torch.nn.Sequential(torch.nn.Linear(featureC + self.in_mlpC, featureC), relu(inplace=True),
                                        torch.nn.Linear(featureC, fea_output))
        self.viewpe, self.pospe, self.rgbpe = viewpe, pospe, rgbpe
        self._owner = None      # the IdentificationModule that holds the shared kernel handle

    def _forward_autograd(self, pts, viewdirs, rgb):
        """ray_preprocessor.py:29-39 in differentiable torch ops (training only; see the module docstring)."""
        from ..models.tensorBase import positional_encoding
        x = torch.cat([pts, viewdirs, rgb, positional_encoding(pts, self.pospe), positional_encoding(viewdirs, self.viewpe),
                       positional_encoding(rgb, self.rgbpe)], dim=-1)
        return self.mlp2(torch.cat((self.mlp(x), x), dim=-1))

    def forward(self, pts, viewdirs, rgb):
        if needs_autograd(self, pts, viewdirs, rgb):
            return self._forward_autograd(pts, viewdirs, rgb)
        if self._owner is None:
            raise RuntimeError("RayPreprocessor.forward runs through its IdentificationModule's kernel handle; "
                               "construct it via IdentificationModule (identification_module.py:66-68)")
        return self._owner()._idnet().ray_encode(pts, viewdirs, rgb, want_features=True)[0]
