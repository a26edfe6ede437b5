import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
ck, idw, pipe = bench.build_inputs(dev)
F = pipe.field
eager = {s: F.surface_sample(593, pipe.rho, 4, 200, seed=s)[0].clone() for s in range(0, 12)}
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
F.surface_sample(593, pipe.rho, 4, 200, seed=0, seed_offset=cnt); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cnt.add_(1)
    s, a, st = F.surface_sample(593, pipe.rho, 4, 200, seed=0, seed_offset=cnt)
for t in range(5):
    cnt.fill_(4); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("replay", t, "cnt", int(cnt.item()), "matches eager seeds", [k for k, v in eager.items() if torch.equal(v, s)], st.cpu()[:, :2].tolist())
# variant: no fill between replays (counter just keeps incrementing)
for t in range(4):
    g.replay(); torch.cuda.synchronize()
    print("free-running", t, "cnt", int(cnt.item()), "matches", [k for k, v in eager.items() if torch.equal(v, s)])
