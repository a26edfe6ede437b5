"""Mirror of the reference's ``renderer.py``: ``OctreeRender_trilinear_fast`` (renderer.py:12-25) and ``evaluation``
(renderer.py:28-140), the full-length-march consumers of ``TensorBase.forward``.

The reference loops over 4096-ray chunks; the march kernel has no such limit, but its compositing weights [R, nSamples]
live in a caller-side workspace (≈1000 samples per ray for a 300^3 grid), so a batch is marched in pieces of at most
``MAX_RAYS_PER_LAUNCH`` rays -- results do not depend on the piece size (rays are independent).
"""
from __future__ import annotations

import os

import numpy as np
import torch

MAX_RAYS_PER_LAUNCH = 65536


def OctreeRender_trilinear_fast(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, bg_color=None, white_bg=None,
                                is_train=False, device="cuda"):
    if ndc_ray or is_train:
        raise RuntimeError("OctreeRender_trilinear_fast: ndc_ray / is_train are outside the inference path")
    step = min(max(int(chunk), 4096), MAX_RAYS_PER_LAUNCH)      # the [R, nSamples] weight workspace stays bounded whatever `chunk` says
    rgbs, depths = [], []
    for lo in range(0, rays.shape[0], step):
        rgb, depth, _, _, _, _ = tensorf.march(rays[lo:lo + step].to(device), point_centred=False, N_samples=N_samples,
                                               white_bg=bool(white_bg), bg_color=bg_color)
        rgbs.append(rgb)
        depths.append(depth)
    if not rgbs:
        e = torch.empty(0, device=device)
        return e.reshape(0, 3), None, e, None, None
    return torch.cat(rgbs), None, torch.cat(depths), None, None


@torch.no_grad()
def evaluation(test_dataset, tensorf, args, renderer, savePath=None, N_vis=5, prtx="", N_samples=-1, white_bg=False,
               ndc_ray=False, compute_extra_metrics=True, device="cuda", return_result=False):
    """renderer.py:28-140: render every ``N_vis``-th test view with ``renderer`` (the slab march), PSNR against the
    dataset's images (RGBA composited on the background the same way), optional PNG dumps.

    ``test_dataset`` needs ``all_rays [n, H*W, 6]``, ``all_rgbs`` (or ``all_rgba``) ``[n, H, W, 3|4]``, ``img_wh``,
    ``near_far`` (duck-typed like the reference's loaders).  SSIM / LPIPS (``compute_extra_metrics``; the reference's
    utils.py:42-110 needs lpips / cv2 networks) are outside the path: asking for them raises RuntimeError instead of
    silently skipping.  Images are written only when ``savePath`` is given and ``imageio`` is importable."""
    if compute_extra_metrics:
        raise RuntimeError("evaluation(compute_extra_metrics=True): SSIM / LPIPS are outside the MI355X path; pass "
                           "compute_extra_metrics=False (PSNR is computed exactly as renderer.py:84-99 does)")
    if ndc_ray:
        raise RuntimeError("evaluation: ndc_ray is outside the inference path")
    imageio = None
    if savePath is not None:
        try:
            import imageio  # noqa: F811
        except ImportError as exc:
            raise RuntimeError("evaluation(savePath=...) writes PNGs through imageio, which is not installed") from exc
        os.makedirs(savePath, exist_ok=True)
        os.makedirs(savePath + "/rgbd", exist_ok=True)
    test_batch_size = getattr(args, "test_batch_size", -1)
    if test_batch_size < 1:
        test_batch_size = getattr(args, "batch_size", 4096)
    all_rgbs = test_dataset.all_rgbs if hasattr(test_dataset, "all_rgbs") else test_dataset.all_rgba
    n_views = test_dataset.all_rays.shape[0]
    interval = 1 if N_vis < 0 else max(n_views // N_vis, 1)
    idxs = list(range(0, n_views, interval))
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    PSNRs, rgb_maps = [], []
    for idx, samples in enumerate(test_dataset.all_rays[0::interval]):
        W, H = test_dataset.img_wh
        rays = samples.view(-1, samples.shape[-1])
        rgb_map, _, depth_map, _, _ = renderer(rays, tensorf, chunk=test_batch_size, N_samples=N_samples, ndc_ray=ndc_ray,
                                               white_bg=white_bg, device=device)
        rgb_map = rgb_map.clamp(0.0, 1.0)
        rgb_map, depth_map = rgb_map.reshape(H, W, 3).cpu(), depth_map.reshape(H, W).cpu()
        rgb_save = (rgb_map.numpy() * 255).astype("uint8")
        rgb_maps.append(rgb_save)
        if imageio is not None:
            imageio.imwrite(f"{savePath}/{prtx}{idx:03d}.png", rgb_save)
        if len(all_rgbs):
            if hasattr(test_dataset, "interpolation") and test_dataset.interpolation[idxs[idx]].any():
                continue
            gt = all_rgbs[idxs[idx]].view(H, W, all_rgbs.shape[-1])
            if gt.shape[-1] > 3:
                bg = torch.ones(3) if white_bg else torch.zeros(3)
                gt = (gt[..., :3] * gt[..., -1:] + bg * (1.0 - gt[..., -1:])).clamp(0, 1)
            loss = torch.mean((rgb_map - gt) ** 2)
            PSNRs.append(-10.0 * np.log(loss.item()) / np.log(10.0))
    end.record()
    torch.cuda.synchronize()
    total_ms = start.elapsed_time(end)
    psnr = float(np.mean(np.asarray(PSNRs))) if PSNRs else float("nan")
    if PSNRs:
        if savePath is not None:
            np.savetxt(f"{savePath}/{prtx}mean.txt", np.asarray([psnr]))
        else:
            print(f"PSNR: {psnr} dB")
    if return_result:
        return PSNRs, {"total_test_time": total_ms, "avg_psnr": psnr}
    return PSNRs
