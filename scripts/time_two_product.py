"""What a score-only path WITHOUT the logits round trip would cost, from the kernels that exist (profiles/NOTES.md, round 5): the fused
cold launch + column pass of today against encoder -> cache (k5_trunk_h<2>) + two logits products from the cache (k5_trunk_h<3>).  Dev aid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
from iffnerf_amd.pipeline import PosePipeline
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["lego16k"]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt("lego16k"), synthetic.make_id_weights(seed=99), dev)
B, P, M = 32, wl["gen_points"], 256
ori, dirs, rgb = pipe.emit(P, seed=3)
N = ori.shape[0]
O, D, C = (x.repeat(B, 1).contiguous() for x in (ori, dirs, rgb))
tok = torch.stack([synthetic.make_tokens(M, 384, seed=7 + q) for q in range(B)]).to(dev)
qf = pipe.idnet.q_fold(tok.reshape(B * M, -1))
def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)
out = {"queries": B, "rays_per_query": N}
lg = pipe.idnet.ray_logits_folded_batched(qf, O, D, C, B)
out["fused_cold_launch_us"] = timed(lambda: pipe.idnet.ray_logits_folded_batched(qf, O, D, C, B))
out["column_pass_us"] = timed(lambda: H.attn_colsum_batched(lg[0], lg[1], lg[2], B, write_attention=False))
out["encoder_to_cache_%d_rays_us" % (B * N)] = timed(lambda: pipe.idnet.build_ray_cache(O, D, C))
cache = pipe.idnet.build_ray_cache(ori, dirs, rgb)
out["logits_from_cache_us"] = timed(lambda: pipe.idnet.logits_from_cache(qf, cache, N))
out["two_product_form_at_least_us"] = round(out["encoder_to_cache_%d_rays_us" % (B * N)] + 2 * out["logits_from_cache_us"], 1)
out["today_us"] = round(out["fused_cold_launch_us"] + out["column_pass_us"], 1)
print(json.dumps(out))
