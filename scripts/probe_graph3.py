import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
from iffnerf_amd import synthetic
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
eager = {s: pipe.query(tok, 593, seed=s, k=100)[0].clone() for s in range(3, 9)}
cq = pipe.capture_query(tok.shape, 593, seed=0, k=100)
cq.tokens.copy_(tok)
for trial in range(6):
    cq.counter.fill_(4)
    torch.cuda.synchronize()
    r = cq.replay().clone(); torch.cuda.synchronize()
    match = [s for s, v in eager.items() if torch.equal(v, r)]
    print("trial", trial, "counter", int(cq.counter.item()), "matches eager seed", match, r[0, 3].item())
# sampler alone in a graph
off = torch.zeros(1, dtype=torch.int64, device=dev)
g = torch.cuda.CUDAGraph()
pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=0, seed_offset=off); torch.cuda.synchronize()
with torch.cuda.graph(g):
    s, a, st = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=0, seed_offset=off)
ref = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=0, seed_offset=off)[0].clone()
for trial in range(4):
    g.replay(); torch.cuda.synchronize()
    print("sampler replay equal to eager:", torch.equal(s, ref), st.cpu()[:, :2].tolist())
