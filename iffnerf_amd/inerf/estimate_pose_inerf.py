"""inerf/estimate_pose_inerf.py:23-195 of the reference: refine a camera pose by gradient descent on the photometric error
of a random pixel batch rendered through the field.

Per iteration the reference builds all H*W rays and indexes the batch (:151-163); here only the batch's pixels are
transformed (same values), and from the fourth iteration on an iteration is one hipGraph replay.  ``model(rays_chunk, bg_color=..., is_train=False)`` is the HIP slab march with its HIP backward
(``TensorBase._forward_ray_grad``); everything around it is a handful of small torch ops on the GPU.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from ..ray_utils import get_ray_directions_Ks, get_rays
from .dice_loss import SoftDiceLossV2
from .inerf import CameraTransfer, img2mse


# Iterations after the first CAPTURE_AFTER run as replays of ONE captured hipGraph (gather of the pixel batch, rays, HIP march
# forward, Ref head, losses, autograd with the HIP march backward, Adam): the eager loop is ~150 kernel launches per
# iteration and launch-bound.  Set to None to keep every iteration eager (same arithmetic, same random streams).
CAPTURE_AFTER = 3


def pose_estimation(start_pose: torch.Tensor, obs_img: np.ndarray, cam_K: torch.Tensor, model,
                    sampling_strategy="interest_regions", lrate: float = 0.02, optimizer_type: str = "adam",
                    batch_size: int = 1024, kernel_size: int = 35, dil_iter: int = 1, color_bkgd_aug: str = "random",
                    device: str = "cuda", n_iters=1000, dice_loss=False, print_progress=True, target_camera_position=None):
    """-> (last rgb loss, refined c2w [4,4] on the CPU, list of per-iteration poses).  ``obs_img`` [H,W,4] RGBA in [0,1]."""
    if sampling_strategy != "random":
        raise RuntimeError(f"sampling_strategy={sampling_strategy!r} needs OpenCV key points (inerf/inerf.py:38-49); "
                           "only 'random' (pose_estimation/test.py:208) is available")
    if optimizer_type not in ("adam", "adamW"):
        raise ValueError("optimizer type is invalid")
    H, W = obs_img.shape[0], obs_img.shape[1]
    if batch_size > H * W:
        raise RuntimeError(f"batch_size {batch_size} exceeds the image's {H * W} pixels")
    dev = torch.device(device)
    start_pose = torch.as_tensor(start_pose, device=dev)
    cam_transf = CameraTransfer(start_pose).to(dev)
    opt_cls = torch.optim.Adam if optimizer_type == "adam" else torch.optim.AdamW
    on_gpu = dev.type == "cuda"
    lr_t = torch.tensor(float(lrate), device=dev)                     # a tensor: the captured step reads it from memory
    optimizer = opt_cls(params=cam_transf.parameters(), lr=lr_t, betas=(0.9, 0.999), capturable=on_gpu)
    dice = SoftDiceLossV2()
    K = torch.as_tensor(cam_K, dtype=torch.float32, device=dev)
    K = K[None] if K.dim() == 2 else K
    raw_dirs, dx, dy = get_ray_directions_Ks(H, W, K, use_pixel_centers=True)
    unit_dirs = raw_dirs / torch.linalg.norm(raw_dirs, dim=-1, keepdim=True)
    obs = torch.from_numpy(np.ascontiguousarray(obs_img)).to(dev)
    # static inputs / outputs of one iteration
    py = torch.zeros(batch_size, dtype=torch.int64, device=dev)
    px = torch.zeros(batch_size, dtype=torch.int64, device=dev)
    bkgd = torch.zeros(3, dtype=obs.dtype, device=dev)
    out_loss = torch.zeros((), dtype=torch.float32, device=dev)
    out_pose = torch.zeros(4, 4, dtype=start_pose.dtype, device=dev)

    def iteration():
        optimizer.zero_grad(set_to_none=True)
        target = obs[py, px]
        rgb_t, alpha_t = target[..., :3], target[..., 3:4]
        target_rgb = rgb_t * alpha_t + bkgd * (1.0 - alpha_t)
        pose = cam_transf()
        rays_o, rays_d, radii = get_rays(unit_dirs[0, py, px], pose, directions=raw_dirs[0, py, px], dx=dx[0, py, px],
                                         dy=dy[0, py, px], keepdim=True)
        rays_chunk = torch.cat((rays_o, F.normalize(rays_d, p=2, dim=-1), radii), dim=-1)
        rgb, _, opacity, _, _, _ = model(rays_chunk, bg_color=bkgd, is_train=False)
        rgb_loss = img2mse(rgb, target_rgb)
        loss = torch.clone(rgb_loss)
        if dice_loss:
            loss = loss + dice(torch.clamp(opacity, 1.0e-3, 1.0 - 1.0e-3)[..., None], alpha_t)[0]
        loss.backward()
        optimizer.step()
        out_loss.copy_(rgb_loss.detach())
        with torch.no_grad():
            out_pose.copy_(cam_transf())

    trace = torch.zeros(max(n_iters, 1), 4, 4, dtype=start_pose.dtype)
    if on_gpu:
        trace = trace.pin_memory()
    graph = None
    start = time.time()
    for k in range(n_iters):
        pick = np.random.choice(H * W, size=batch_size, replace=False)                  # :108-112
        py.copy_(torch.from_numpy(pick // W), non_blocking=True)
        px.copy_(torch.from_numpy(pick % W), non_blocking=True)
        if color_bkgd_aug == "white":
            bkgd.fill_(1.0)
        elif color_bkgd_aug == "random":
            bkgd.copy_(torch.rand(3, dtype=obs.dtype, device=dev))
        else:
            bkgd.zero_()
        if on_gpu and CAPTURE_AFTER is not None and k == CAPTURE_AFTER and graph is None:
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                iteration()
        if graph is not None:
            graph.replay()
        else:
            iteration()
        trace[k].copy_(out_pose, non_blocking=True)
        lr_t.fill_(lrate * (0.8 ** ((k + 1) / 100)))
        if ((k + 1) % 20 == 0 or k == 0) and print_progress:
            print(f"[{k}] Loss: {out_loss.item()}")
    if on_gpu:
        torch.cuda.synchronize(dev)
    if print_progress:
        print(f"Total optimization time: {time.time() - start:.02f} s")
    return out_loss.item(), out_pose.detach().cpu(), [trace[k].clone() for k in range(n_iters)]
