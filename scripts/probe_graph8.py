import sys, torch
sys.path.insert(0, ".")
import bench
from iffnerf_amd import hip_identify as H
from iffnerf_amd.hip_field import isocell_emit
dev = torch.device("cuda:0")
from iffnerf_amd import synthetic
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
def check(name, fn):
    ref = [o.clone() for o in fn()]; torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = fn()
    res = []
    for t in range(3):
        g.replay(); torch.cuda.synchronize()
        res.append(all(torch.equal(a, b) for a, b in zip(ref, outs)))
    print(f"{name:28s} replay == eager: {res}")
F = pipe.field
check("sampler", lambda: list(F.surface_sample(593, pipe.rho, 4, 200, seed=5))[:2])
smp = F.surface_sample(593, pipe.rho, 4, 200, seed=5)[0].clone()
check("normals", lambda: [F.point_normals(smp)])
nrm = F.point_normals(smp).clone()
check("emit", lambda: list(isocell_emit(pipe.cells, smp, nrm, want_rays6=True)))
ori, dirs, rays = [t.clone() for t in isocell_emit(pipe.cells, smp, nrm, want_rays6=True)]
check("march", lambda: list(F.march(rays, 0, 20, want_alpha=False)[:3]))
rgb = F.march(rays, 0, 20, want_alpha=False)[0].clone()
check("pipe.emit", lambda: list(pipe.emit(593, 5)))
check("encode", lambda: [pipe.idnet.ray_encode(ori, dirs, rgb, False, True)[1]])
k = pipe.idnet.ray_encode(ori, dirs, rgb, False, True)[1].clone()
check("qproj", lambda: [pipe.idnet.q_proj(tok)])
q = pipe.idnet.q_proj(tok).clone()
check("logits", lambda: list(H.attn_logits(q, k)))
def colsum():
    l, m, s = H.attn_logits(q, k)
    return [H.attn_colsum(l, m, s, True), l]
check("colsum", colsum)
sc = colsum()[0].clone()
check("topk", lambda: list(H.topk(sc, 100)))
idx, val = [t.clone() for t in H.topk(sc, 100)]
check("pose", lambda: [H.pose_from_topk(idx, val, ori, dirs, (0., 0., 1.))])
check("identify", lambda: list(pipe.identify(tok, ori, dirs, rgb, 100)))
check("query", lambda: list(pipe.query(tok, 593, 5, 100)))
