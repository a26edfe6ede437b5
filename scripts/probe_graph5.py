import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
from iffnerf_amd import synthetic
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
def check(name, fn, pre=None):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = fn()
    ref = None; oks = []
    for t in range(4):
        if pre: pre()
        g.replay(); torch.cuda.synchronize()
        cur = [o.clone() for o in outs]
        if ref is None: ref = cur
        oks.append(all(torch.equal(a, b) for a, b in zip(ref, cur)))
    print(name, oks)
check("query seed by value", lambda: list(pipe.query(tok, 593, 5, 100)))
off = torch.full((1,), 5, dtype=torch.int64, device=dev)
check("query seed offset const", lambda: list(pipe.query(tok, 593, 0, 100, seed_offset=off)))
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
def q_inc():
    cnt.add_(1)
    return list(pipe.query(tok, 593, 0, 100, seed_offset=cnt))
check("query with in-graph increment", q_inc, pre=lambda: cnt.fill_(4))
def e_inc():
    cnt.add_(1)
    return list(pipe.emit(593, 0, seed_offset=cnt))
check("emit with in-graph increment", e_inc, pre=lambda: cnt.fill_(4))
def s_inc():
    cnt.add_(1)
    return list(pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=0, seed_offset=cnt))[:2]
check("sampler with in-graph increment", s_inc, pre=lambda: cnt.fill_(4))
