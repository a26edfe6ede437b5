"""Time iff_vit_forward (16 images per call) and the image preprocessing; dev aid."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd.hip_vit import ViTHandle
from iffnerf_amd.image_frontend import ImageFrontEnd
from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
dev = torch.device("cuda:0")
net, grid, _ = create_standin_backbone(seed=0)
net = net.to(dev)
vit = ViTHandle(net.state_dict(), dev)
Q = 16
x = torch.randn(Q, 3, 224, 224, device=dev)
imgs = torch.rand(Q, 800, 800, 3, device=dev)
fe = ImageFrontEnd(net, grid)
def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
from iffnerf_amd.pose_estimation.identification_module import _center_crop, _resize_short_edge
t_vit = timed(lambda: vit.forward(x))
t_pre = timed(lambda: _center_crop(_resize_short_edge(imgs.permute(0, 3, 1, 2), 256, "bicubic"), 224))
from iffnerf_amd.image_frontend import resize_crop
masks = (torch.rand(Q, 800, 800, device=dev) > 0.3).float()
t_nat = timed(lambda: resize_crop(imgs, 256, 224, True, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)))
t_natm = timed(lambda: resize_crop(resize_crop(masks[..., None], 256, 224, False).permute(0, 2, 3, 1), 16, None, False))
t_prem = timed(lambda: _resize_short_edge(_center_crop(_resize_short_edge(masks[:, None], 256, "bilinear"), 224), 16, "bilinear"))
print(json.dumps({"native_img_ms": round(t_nat, 4), "native_mask_ms": round(t_natm, 4), "torch_mask_ms": round(t_prem, 4)}))
print(json.dumps({"vit_ms_per_16": round(t_vit, 4), "tflops": round(16 * 12.2e9 / (t_vit * 1e-3) / 1e12, 1), "preprocess_ms_per_16": round(t_pre, 4)}))
