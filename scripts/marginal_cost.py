"""What each stage of a cold step costs in THROUGHPUT (4 steps in flight), not in its own duration: the step is captured with one
stage at a time replaced by a precomputed result and timed like bench.py times it.  Dev aid."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
from iffnerf_amd.pipeline import PosePipeline
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["lego16k"]
B, M, K, P = wl["queries"], 256, 100, wl["gen_points"]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt("lego16k"), synthetic.make_id_weights(seed=99), dev)
tokens = torch.stack([synthetic.make_tokens(M, 384, seed=7 + q) for q in range(B)]).to(dev)

def measure(tag, steps=120):
    graphs = [pipe.capture_query_batch(tokens.shape, P, seed=(g + 1) << 40, k=K) for g in range(4)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    for g in graphs: g.tokens.copy_(tokens)
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % 4]):
                graphs[i % 4].replay()
    run(12); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"variant": tag, "ms_per_step": round(dt / steps * 1e3, 4)}), flush=True)
    return dt / steps * 1e3

base = measure("full step")
orig = {}
def patch(obj, name, fn):
    orig[(obj, name)] = getattr(obj, name); setattr(obj, name, fn)
def restore():
    for (obj, name), fn in orig.items(): setattr(obj, name, fn)
    orig.clear()

# precomputed stand-ins (static tensors: nothing is launched for the stage)
qf0 = pipe.idnet.q_fold(tokens.reshape(B * M, -1)).clone()
patch(pipe.idnet, "q_fold", lambda t: qf0); measure("without q_fold (k_gemm_tokens)"); restore()
n = 27 * P
score0 = torch.rand(B, n, device=dev)
f_col = H.attn_colsum_batched
patch(H, "attn_colsum_batched", lambda *a, **k: score0); measure("without k6_colsum"); restore()
idx0, val0 = H.topk_batched(score0, K)
patch(H, "topk_batched", lambda s, k: (idx0, val0)); measure("without k7_topk"); restore()
c2w0 = torch.eye(4, device=dev).repeat(B, 1, 1)
patch(H, "pose_from_topk_batched", lambda *a, **k: c2w0); measure("without k_pose"); restore()
patch(H, "attn_colsum_batched", lambda *a, **k: score0); patch(H, "topk_batched", lambda s, k: (idx0, val0)); patch(H, "pose_from_topk_batched", lambda *a, **k: c2w0)
patch(pipe.idnet, "q_fold", lambda t: qf0); measure("without q_fold, colsum, topk, pose"); restore()
measure("full step again")
