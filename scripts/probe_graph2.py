import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
from iffnerf_amd import synthetic
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
off = torch.tensor([5], dtype=torch.int64, device=dev)
a = pipe.query(tok, 593, seed=0, k=100, seed_offset=off)[0]
b = pipe.query(tok, 593, seed=5, k=100)[0]
c = pipe.query(tok, 593, seed=5, k=100)[0]
print("offset==byvalue", torch.equal(a, b), "repeat", torch.equal(b, c))
s1 = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=0, seed_offset=off)[0]
s2 = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=5)[0]
print("samples equal", torch.equal(s1, s2))
cq = pipe.capture_query(tok.shape, 593, seed=0, k=100)
cq.tokens.copy_(tok)
cq.counter.fill_(4); r1 = cq.replay().clone(); torch.cuda.synchronize()
print("graph vs eager(5)", torch.equal(r1, b), r1[:3,3].tolist(), b[:3,3].tolist(), int(cq.counter.item()))
cq.counter.fill_(4); r2 = cq.replay().clone(); torch.cuda.synchronize()
print("graph repeat", torch.equal(r1, r2))
