import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
from iffnerf_amd import synthetic
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
eager = {s: pipe.query(tok, 593, seed=s, k=100)[0].clone() for s in range(0, 16)}
cq = pipe.capture_query(tok.shape, 593, seed=0, k=100)
cq.tokens.copy_(tok); torch.cuda.synchronize()
print("counter after construction", int(cq.counter.item()))
for t in range(8):
    r = cq.replay(); torch.cuda.synchronize()
    c = int(cq.counter.item())
    print("replay", t, "cnt", c, "== eager(cnt):", torch.equal(eager[c], r))
# back-to-back without syncs, then check the last
for t in range(5): cq.replay()
torch.cuda.synchronize(); c = int(cq.counter.item())
print("after 5 unsynced replays cnt", c, "== eager:", torch.equal(eager[c], cq.c2w))
