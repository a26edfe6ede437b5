"""Mirror of ``pose_estimation/ray_preprocessor.py``: same constructor, same parameter names, forward in HIP.

``mlp`` / ``mlp2`` keep the reference's Sequential layout (ray_preprocessor.py:9-25) so ``id_module.th`` loads
unchanged; ``forward`` (reference :29-39) runs ``iff_ray_encode`` (positional encoding + 4 fp32-MFMA GEMMs).
"""
from __future__ import annotations

import torch


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

    def forward(self, pts, viewdirs, rgb):
        if self._owner is None:
            raise RuntimeError("RayPreprocessor.forward runs through its IdentificationModule's kernel handle; "
                               "construct it via IdentificationModule (identification_module.py:66-68)")
        return self._owner()._idnet().ray_encode(pts, viewdirs, rgb, want_features=True)[0]
