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

side = None          # optional: (graphs, streams) replayed next to every step, on their own streams (nothing depends on them)
NF = int(os.environ.get("IN_FLIGHT", "4"))
def measure(tag, steps=120):
    graphs = [pipe.capture_query_batch(tokens.shape, P, seed=(g + 1) << 40, k=K) for g in range(NF)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
    for g in graphs: g.tokens.copy_(tokens)
    def run(n):
        for i in range(n):
            if side is not None:
                with torch.cuda.stream(side[1][i % len(side[0])]):
                    side[0][i % len(side[0])].replay()
            with torch.cuda.stream(streams[i % NF]):
                graphs[i % NF].replay()
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

n = 27 * P
score0 = torch.rand(B, n, device=dev)
f_col = H.attn_colsum_batched
patch(H, "attn_colsum_batched", lambda *a, **k: score0); measure("without k6_colsum"); restore()
idx0, val0 = H.topk_batched(score0, K)

c2w0 = torch.eye(4, device=dev).repeat(B, 1, 1)

patch(H, "attn_colsum_batched", lambda *a, **k: score0); patch(H, "topk_batched", lambda s, k: (idx0, val0)); patch(H, "pose_from_topk_batched", lambda *a, **k: c2w0)
patch(pipe.idnet, "q_fold", lambda t: qf0); measure("without q_fold, colsum, topk, pose"); restore()
samples0, _, stats0 = pipe.field.surface_sample_batched(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=5)
patch(pipe.field, "surface_sample_batched", lambda *a, **k: (samples0, None, stats0)); measure("without the sampler"); restore()
# the sampler free-running on four side streams (one launch per step, nothing waits for it): resource interference without the
# dependency of the step on its result
true_sampler = pipe.field.surface_sample_batched
sg, ss = [], [torch.cuda.Stream(device=dev) for _ in range(4)]
for g_ in range(4):
    with torch.cuda.stream(ss[g_]):
        true_sampler(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=77 + g_)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=ss[g_]):
        keep = true_sampler(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=77 + g_)
    sg.append(gr)
torch.cuda.synchronize()
patch(pipe.field, "surface_sample_batched", lambda *a, **k: (samples0, None, stats0)); side = (sg, ss)
measure("sampler off the chain (free-running on side streams)"); side = None; restore()
nrm0 = pipe.field.point_normals(samples0.reshape(-1, 3))

ori0, dirs0, rgb0 = pipe.emit_from_samples(samples0.reshape(-1, 3))
patch(pipe.field, "surface_sample_batched", lambda *a, **k: (samples0, None, stats0)); patch(pipe, "emit_from_samples", lambda *a, **k: (ori0, dirs0, rgb0))
measure("without stages A + B (sampler, normals, emit, march)"); restore()
lg = pipe.idnet.ray_logits_folded_batched(qf0, ori0, dirs0, rgb0, B)
patch(pipe.idnet, "ray_logits_folded_batched", lambda *a, **k: lg); measure("without the trunk (+ merge_stats)"); restore()
measure("full step again")
