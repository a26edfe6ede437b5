"""Warm path (rays resident, encoder cached): Q query images per captured graph (QS=16,32,64), 4 graphs in flight -> poses/s, on
the workload CONFIG (lego16k); TRUNK_VARIANT=1..4 names the work split of the logits launch.  Dev aid."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
dev = torch.device("cuda:0")
CFG = os.environ.get("CONFIG", "lego16k")
wl = synthetic.WORKLOADS[CFG]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(CFG), synthetic.make_id_weights(seed=99), dev,
                                     trunk_variant=int(os.environ.get("TRUNK_VARIANT", "0")))      # iff_idnet_desc.trunk_variant (1..4)
ori, dirs, rgb = pipe.emit(wl["gen_points"], seed=42)
resident = pipe.make_resident(ori, dirs, rgb)
for Q in [int(x) for x in os.environ.get("QS", "16,32,64").split(",")]:
    tok = torch.stack([synthetic.make_tokens(256, 384, seed=100 + q) for q in range(Q)]).to(dev)
    for _ in range(2):
        pipe.identify_resident(tok, resident, 100)
    torch.cuda.synchronize()
    graphs, streams = [], [torch.cuda.Stream(device=dev) for _ in range(4)]
    for _ in range(4):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = pipe.identify_resident(tok, resident, 100)
        graphs.append((g, out))
    torch.cuda.synchronize()
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % 4]):
                graphs[i % 4][0].replay()
    run(8); torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"config": CFG, "trunk_variant": int(os.environ.get("TRUNK_VARIANT", "0")), "queries_per_graph": Q, "warm_poses_per_s": round(n * Q / dt, 1), "ms_per_graph": round(dt / n * 1e3, 4)}), flush=True)
