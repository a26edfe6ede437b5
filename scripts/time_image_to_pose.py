"""Image in -> pose out with the native backbone: Q 800x800 RGBA queries per captured graph (QS=16,32), INFLIGHT graphs in flight (4).  Dev aid."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline, CapturedImageQuery
from iffnerf_amd.image_frontend import ImageFrontEnd
from iffnerf_amd import hip_vit
from iffnerf_amd.hip_vit import serve_natively
hip_vit.DEFAULT_GEMM_FORM = int(os.environ.get("FORM", "0"))
PREC = os.environ.get("PREC", "fp32")
from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["lego16k"]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt("lego16k"), synthetic.make_id_weights(seed=99), dev)
ori, dirs, rgb = pipe.emit(wl["gen_points"], seed=42)
resident = pipe.make_resident(ori, dirs, rgb)
net, grid, _ = create_standin_backbone(seed=0)
fe = ImageFrontEnd(serve_natively(net.to(dev), grid, precision=PREC), grid)
gen = torch.Generator().manual_seed(11)
NF = int(os.environ.get("INFLIGHT", "4"))
for Q in [int(x) for x in os.environ.get("QS", "16,32").split(",")]:
    imgs = torch.rand(Q, 800, 800, 3, generator=gen).to(dev)
    masks = (torch.rand(Q, 800, 800, generator=gen) > 0.2).float().to(dev)
    graphs = [CapturedImageQuery(pipe, fe, imgs.shape, resident, 100) for _ in range(NF)]
    for g in graphs:
        g.imgs.copy_(imgs), g.masks.copy_(masks)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
    torch.cuda.synchronize()
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % NF]):
                graphs[i % NF].replay()
    run(8); torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"form": hip_vit.DEFAULT_GEMM_FORM, "prec": PREC, "inflight": NF, "images_per_graph": Q, "image_to_pose_per_s": round(n * Q / dt, 1), "ms_per_graph": round(dt / n * 1e3, 4)}), flush=True)
    del graphs
