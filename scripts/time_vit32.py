"""iff_vit_forward alone, 32 images per call, one stream (hipEvents); dev aid for A/Bs of vit_kernels.hip (IFF_LIB_PATH)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd.hip_vit import ViTHandle
from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
dev = torch.device("cuda:0")
net, grid, _ = create_standin_backbone(seed=0)
sd = net.to(dev).state_dict()
forms = [int(f) for f in os.environ.get("FORMS", "0").split(",")]
precs = os.environ.get("PRECS", "fp32,bf16").split(",")
for prec, form in [(p, f) for p in precs for f in forms]:
    vit = ViTHandle(sd, dev, precision=prec, gemm_form=form)
    for Q in ((16, 32) if form == forms[0] and not os.environ.get("ONLY32") else (32,)):
        x = torch.randn(Q, 3, 224, 224, device=dev)
        for _ in range(3): vit.forward(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): vit.forward(x)
        b.record(); torch.cuda.synchronize()
        print(json.dumps({"lib": os.environ.get("IFF_LIB_PATH", "in-tree"), "precision": prec, "gemm_form": form, "images": Q, "vit_ms": round(a.elapsed_time(b) / 20, 4)}))
