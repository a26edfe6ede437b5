"""Time the cached-encoder logits launch of the warm path (k5_trunk_h<3>: 16 query images x 256 tokens against one resident ray
set) alone; dev aid for same-box A/Bs (scripts/gpu_ab.sh style: swap the library, run this)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
dev = torch.device("cuda:0")
net = H.IdNetHandle(synthetic.make_id_weights(seed=99), dev, gemm_mode=H.GEMM_F16X2, trunk_variant=int(os.environ.get("TV", "0")))
g = torch.Generator().manual_seed(3)
Q, N, M = int(os.environ.get("Q", "16")), 16011, 256
o = (torch.rand(N, 3, generator=g) * 2 - 1).to(dev)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
c = torch.rand(N, 3, generator=g).to(dev)
tok = torch.stack([synthetic.make_tokens(M, 384, seed=7 + q) for q in range(Q)]).to(dev)
qf = net.q_fold(tok.reshape(Q * M, -1))
cache = net.build_ray_cache(o, d, c)
for _ in range(3):
    out = net.logits_from_cache(qf, cache, N)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    out = net.logits_from_cache(qf, cache, N)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"queries": Q, "logits_from_cache_ms": round(e0.elapsed_time(e1) / n, 4), "checksum": float(out[0].double().abs().sum())}))
