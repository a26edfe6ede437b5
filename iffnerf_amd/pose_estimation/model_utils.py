"""Mirror of ``pose_estimation/model_utils.py``: ``load_model`` and ``explore_model`` with the reference's signatures."""
from __future__ import annotations

import torch

from . import sampling
from ..models.tensoRF import TensorVMSplit

_MODELS = {"TensorVMSplit": TensorVMSplit}


def load_model(checkpoint_path, device):
    """Reference :4-14.  (The reference eval()s the class name; a lookup table does the same job here.)"""
    ckpt = torch.load(checkpoint_path, map_location=device, weights_only=False)
    name = ckpt["model_name"]
    if name not in _MODELS:
        raise RuntimeError(f"model_name {name!r}: only TensorVMSplit is built for the MI355X path")
    kwargs = dict(ckpt["kwargs"])
    kwargs.update({"device": device})
    tensorf = _MODELS[name](**kwargs)
    tensorf.load(ckpt)
    for param in tensorf.parameters():
        param.requires_grad = False
    return tensorf


def explore_model(model, gen_points: int = 20000):
    """Reference :22-33: surface samples -> normals -> 27 rays per sample with their rendered colour."""
    samples = sampling.iterative_surface_sampling_process(model, gen_points=gen_points, n_iteration=4,
                                                          max_resampling_iterations=200)
    normals = sampling.samples_points_normals(model, samples)
    return sampling.generate_all_possible_rays(samples, normals, model)
