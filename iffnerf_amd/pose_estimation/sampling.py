"""Mirror of ``pose_estimation/sampling.py`` for the path (stage A and the caller of stage B).

Same function names and signatures as the reference.  Everything runs in libiffnerf_hip:
the surface sampler is a chain of short launches behind ``iff_surface_sample`` (its persistent one-launch form is a handle option kept for
parity tests), normals ``iff_point_normals``, the 27-ray fan
``iff_isocell_emit``, ray colours one ``iff_march_shade`` launch (the reference's 10 240-ray chunking is a memory
workaround of its boolean-compaction formulation and has no effect on results; ``num_viewdirs_per_chunk`` is accepted
and ignored).  The sampler draws from a device-side counter-based generator seeded from torch's global generator, so
``torch.manual_seed`` still makes runs reproducible, though not stream-identical to the reference's CPU draws.
"""
from __future__ import annotations

import torch

from .isocell import isocell_distribution, rotate_isocell  # noqa: F401  (re-exported like the reference)


def has_valid_occupancy_grid(model):
    from ..models.tensorBase import AlphaGridMask
    return isinstance(getattr(model, "alphaMask", None), AlphaGridMask)


def sampling_isocell(dtype=torch.float32, device="cpu", num_targets=27):
    return isocell_distribution(num_targets, dtype, device, N0=3, isrand=-1, int_dtype=torch.int64)


def _jitter_scale(model) -> float:
    """rho of reference sampling.py:518-523."""
    if has_valid_occupancy_grid(model):
        g = model.gridSize.cpu()
        return float((torch.max(g) * 0.1) * torch.max(model.aabbSize.cpu() / g))
    return float(torch.linalg.norm(model.aabbSize.cpu()))


@torch.no_grad()
def iterative_surface_sampling_process(model, gen_points=8000, n_iteration=4, max_resampling_iterations=200,
                                       return_stats=False):
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())      # tie the device stream to torch.manual_seed
    samples, alpha, stats = model.field_handle().surface_sample(
        int(gen_points), _jitter_scale(model), n_epochs=n_iteration, max_iterations=max_resampling_iterations, seed=seed)
    from ..pipeline import check_sampler_stats
    check_sampler_stats(stats)          # one device->host read per call (the reference syncs twice per iteration)
    if return_stats:
        return samples, alpha, stats
    return samples


def samples_points_normals(model, samples):
    return model.field_handle().point_normals(samples)


def evaluate_viewdirs_color(point_sampling, viewdir, model, **kwargs):
    point_sampling = torch.broadcast_to(point_sampling, viewdir.shape)
    rays = torch.cat((point_sampling, viewdir), dim=-1).view(-1, 6)
    rgb = model.march(rays, point_centred=True, N_samples=20, **kwargs)[0]
    return rgb.view(*viewdir.shape)


def generate_all_possible_rays(point_sampling, point_normals, model, num_viewdirs_per_chunk=10240,
                               sample_isocell_targets=27):
    from ..hip_field import isocell_emit
    cells = sampling_isocell(dtype=point_sampling.dtype, device="cpu", num_targets=sample_isocell_targets)
    ori, dirs, rays = isocell_emit(cells, point_sampling, point_normals, want_rays6=True)
    rgb = model.march(rays, point_centred=True, N_samples=20)[0]
    return ori, dirs, rgb
