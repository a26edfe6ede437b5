"""Mirror of the reference's ``models/tensoRF.py`` for the hot path: ``TensorVMSplit`` only.

The VM-split field is the only variant the reference's configs select (``model_name = TensorVMSplit``,
configs/*.txt); ``TensorVM`` / ``TensorCP`` are outside the path.  Parameter names and
shapes follow reference models/tensoRF.py:155-170 so checkpoints load unchanged:
``density_plane.i [1,C,G_b,G_a]``, ``density_line.i [1,C,G_v,1]``, ``app_plane.i``, ``app_line.i``,
``basis_mat.weight [app_dim, 3 C_app]``.  All lookups run in libiffnerf_hip through ``TensorBase``.
"""
from __future__ import annotations

import torch

from .tensorBase import TensorBase


class TensorVMSplit(TensorBase):
    def __init__(self, aabb, gridSize, device, **kargs):
        super().__init__(aabb, gridSize, device, **kargs)

    def _vm_tables(self, n_component, grid, scale, device):
        planes, lines = [], []
        for i, v in enumerate(self.vecMode):
            a, b = self.matMode[i]
            planes.append(torch.nn.Parameter(scale * torch.randn(1, n_component[i], grid[b], grid[a])))
            lines.append(torch.nn.Parameter(scale * torch.randn(1, n_component[i], grid[v], 1)))
        return torch.nn.ParameterList(planes).to(device), torch.nn.ParameterList(lines).to(device)

    def init_svd_volume(self, res, device):
        grid = [int(g) for g in self.gridSize.tolist()]
        self.density_plane, self.density_line = self._vm_tables(self.density_n_comp, grid, 0.1, device)
        self.app_plane, self.app_line = self._vm_tables(self.app_n_comp, grid, 0.1, device)
        self.basis_mat = torch.nn.Linear(sum(self.app_n_comp), self.app_dim, bias=False).to(device)
