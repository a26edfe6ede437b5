import sys, torch
sys.path.insert(0, ".")
import bench
from iffnerf_amd import hip_identify as H
dev = torch.device("cuda:0")
from iffnerf_amd import synthetic
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
def check(name, fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = fn()
    ref = None
    oks = []
    for t in range(4):
        g.replay(); torch.cuda.synchronize()
        cur = [o.clone() for o in outs]
        if ref is None: ref = cur
        oks.append(all(torch.equal(a, b) for a, b in zip(ref, cur)))
    print(name, oks)
check("emit", lambda: pipe.emit(593, 5))
ori, dirs, rgb = [t.clone() for t in pipe.emit(593, 5)]
check("encode", lambda: [t for t in pipe.idnet.ray_encode(ori, dirs, rgb, True, True)])
_, k = pipe.idnet.ray_encode(ori, dirs, rgb, False, True); k = k.clone()
check("qproj", lambda: [pipe.idnet.q_proj(tok)])
q = pipe.idnet.q_proj(tok).clone()
check("logits", lambda: list(H.attn_logits(q, k)))
def colsum():
    l, m, s = H.attn_logits(q, k)
    sc = H.attn_colsum(l, m, s, True)
    return [l, sc]
check("colsum", colsum)
sc = colsum()[1].clone()
check("topk", lambda: list(H.topk(sc, 100)))
idx, val = [t.clone() for t in H.topk(sc, 100)]
check("pose", lambda: [H.pose_from_topk(idx, val, ori, dirs, (0., 0., 1.))])
check("identify", lambda: list(pipe.identify(tok, ori, dirs, rgb, 100)))
