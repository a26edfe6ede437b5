"""The warm path (resident rays) and image -> pose as captured graphs replayed NF at a time against the same call run eagerly on the
same inputs: counts replays whose poses / top-k lists differ.  Dev aid (CONFIG=, ROUNDS=, WHAT=warm|image)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline, CapturedImageQuery
dev = torch.device("cuda:0")
CFG = os.environ.get("CONFIG", "lego16k")
ROUNDS = int(os.environ.get("ROUNDS", "100"))
NF = int(os.environ.get("INFLIGHT", "4"))
WHAT = os.environ.get("WHAT", "warm")
Q = int(os.environ.get("Q", "32"))
wl = synthetic.WORKLOADS[CFG]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(CFG), synthetic.make_id_weights(seed=99), dev)
ori, dirs, rgb = pipe.emit(wl["gen_points"], seed=42)
resident = pipe.make_resident(ori, dirs, rgb)
gen = torch.Generator().manual_seed(3)
graphs, keep_alive = [], []
if WHAT == "warm":
    toks = [torch.stack([synthetic.make_tokens(256, 384, seed=100 * i + q) for q in range(Q)]).to(dev) for i in range(NF)]
    for i in range(NF):
        for _ in range(2):
            pipe.identify_resident(toks[i], resident, 100)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = pipe.identify_resident(toks[i], resident, 100)
        graphs.append((g, out, lambda i=i: pipe.identify_resident(toks[i], resident, 100)))
else:
    from iffnerf_amd.image_frontend import ImageFrontEnd
    from iffnerf_amd.hip_vit import serve_natively
    from iffnerf_amd.pose_estimation.backbone import create_standin_backbone
    net, grid, _ = create_standin_backbone(seed=0)
    fe = ImageFrontEnd(serve_natively(net.to(dev), grid), grid)
    for i in range(NF):
        imgs = torch.rand(Q, 800, 800, 3, generator=gen).to(dev)
        masks = (torch.rand(Q, 800, 800, generator=gen) > 0.2).float().to(dev)
        cq = CapturedImageQuery(pipe, fe, imgs.shape, resident, 100)
        cq.imgs.copy_(imgs), cq.masks.copy_(masks)
        keep_alive.append(cq)                      # its static input buffers
        graphs.append((cq.graph, (cq.c2w, cq.idx, cq.val), lambda imgs=imgs, masks=masks: pipe.identify_images_resident(fe, imgs, masks, resident, 100)))
streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
torch.cuda.synchronize()
n_bad = 0
for r in range(ROUNDS):
    for rep in range(3):
        for i, (g, _, _) in enumerate(graphs):
            with torch.cuda.stream(streams[i]):
                g.replay()
    torch.cuda.synchronize()
    for i, (g, out, eager) in enumerate(graphs):
        e = eager()
        names = ("c2w", "idx", "val")
        diff = [n for n, a, b in zip(names, e, out) if not torch.equal(a, b)]
        if diff:
            n_bad += 1
            print(json.dumps({"round": r, "graph": i, "differ": diff, "val_maxdiff": float((e[2] - out[2]).abs().max()),
                              "queries": (e[2] != out[2]).any(dim=1).nonzero().flatten()[:8].tolist()}), flush=True)
print(json.dumps({"what": WHAT, "config": CFG, "checks": ROUNDS * NF, "mismatches": n_bad}))
