"""A/B of the trunk work splits: logits bit-identity against variant 1 (the eight-wave form), row statistics, timing.  Dev aid."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
dev = torch.device("cuda:0")
w = synthetic.make_id_weights(seed=99)
g = torch.Generator().manual_seed(3)
B, N, M = 16, 16011, 256
o = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
d = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).to(dev)
c = torch.rand(B, N, 3, generator=g).to(dev)
tok = torch.stack([synthetic.make_tokens(M, 384, seed=7 + q) for q in range(B)]).to(dev)
ref = None
for var in (1, 4):
    net = H.IdNetHandle(w, dev, gemm_mode=H.GEMM_F16X2, trunk_variant=var)
    qf = net.q_fold(tok.reshape(B * M, -1))
    out = net.ray_logits_folded_batched(qf, o.reshape(-1, 3), d.reshape(-1, 3), c.reshape(-1, 3), B)
    ms = []
    for _ in range(5):
        net.ray_logits_folded_batched(qf, o.reshape(-1, 3), d.reshape(-1, 3), c.reshape(-1, 3), B, trunk_ms=ms)
    # the cached path (MODE 3) against the fused one
    cache = net.build_ray_cache(o[0], d[0], c[0])
    lc = net.logits_from_cache(qf[:M], cache, N)
    l1 = net.ray_logits_folded(qf[:M], o[0], d[0], c[0])
    same_cached = all(torch.equal(x, y) for x, y in zip(lc, l1))
    if ref is None:
        ref = out
        print(json.dumps({"variant": var, "trunk_ms": round(sum(ms) / len(ms), 4), "cached_equals_fused": same_cached}))
    else:
        print(json.dumps({"variant": var, "trunk_ms": round(sum(ms) / len(ms), 4), "cached_equals_fused": same_cached,
                          "logits_bit_identical": bool(torch.equal(out[0], ref[0])),
                          "rowmax_equal": bool(torch.equal(out[1], ref[1])),
                          "rowsum_max_rel_diff": float(((out[2] - ref[2]).abs() / ref[2]).max())}))
