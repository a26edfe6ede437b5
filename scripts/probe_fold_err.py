"""dev probe: error of the folded / unfolded HIP logits and of the fp32 CPU oracle against an fp64 evaluation."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
from oracle import identify as oid
from tests import util
g = util.Golden()
dev = torch.device("cuda:0")
w = synthetic.make_id_weights(seed=99)
o, d, c = (g.t("g6_identify", k) for k in ("ori", "dirs", "rgb"))
tok = synthetic.make_tokens(256, 384, seed=int(g["g6_identify"]["tokens_seed"]))
w64 = {k: v.double() for k, v in w.items()}
_, truth, _, _ = oid.attention_map(w64, tok.double(), oid.ray_encode(w64, o.double(), d.double(), c.double()), return_parts=True)
_, cpu32, _, _ = oid.attention_map(w, tok, oid.ray_encode(w, o, d, c), return_parts=True)
net = H.IdNetHandle(w, dev)
od, dd, cd, td = o.to(dev), d.to(dev), c.to(dev), tok.to(dev)
_, k = net.ray_encode(od, dd, cd, want_features=False, want_k=True)
unf, _, _ = H.attn_logits(net.q_proj(td), k)
fol, _, _ = net.attn_logits_folded(net.q_fold(td), net.ray_trunk(od, dd, cd))
print("|logit| max", float(truth.abs().max()))
for name, t in (("cpu fp32 oracle", cpu32), ("hip unfolded", unf.cpu()), ("hip folded", fol.cpu())):
    e = (t.double() - truth).abs()
    print(f"{name:18s} max err {float(e.max()):.3e}  mean {float(e.mean()):.3e}")
