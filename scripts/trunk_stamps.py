"""Per-stage wave timelines of the fused trunk kernel (library built with -DTRUNK_STAMPS); dev aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iffnerf_amd import synthetic, hip_identify as H
dev = torch.device("cuda:0")
w = synthetic.make_id_weights(seed=99)
g = torch.Generator().manual_seed(3)
B, N, M = 16, 16011, 256
o = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
d = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1).to(dev)
c = torch.rand(B, N, 3, generator=g).to(dev)
tok = torch.stack([synthetic.make_tokens(M, 384, seed=7 + q) for q in range(B)]).to(dev)
net = H.IdNetHandle(w, dev, gemm_mode=H.GEMM_F16X2, trunk_variant=1)
qf = net.q_fold(tok.reshape(B * M, -1))
for _ in range(3):
    out = net.ray_logits_folded_batched(qf, o.reshape(-1, 3), d.reshape(-1, 3), c.reshape(-1, 3), B)
torch.cuda.synchronize()
ms = []
out = net.ray_logits_folded_batched(qf, o.reshape(-1, 3), d.reshape(-1, 3), c.reshape(-1, 3), B, trunk_ms=ms)
torch.cuda.synchronize()
lg = out[0].reshape(B, M, N)
n_full = N // 64
st = lg[:, :, : n_full * 64 : 64].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff    # [B, 256, tiles]
st = st.reshape(B, 8, 32, n_full)[:, :, :16, :]                  # [B, wave, k, tile]
st = np.moveaxis(st, 3, 1).reshape(-1, 8, 16)                      # [tile, wave, k]
t_first, t_last = st[:, :, 0].min(), st[:, :, 15].max()
print(f"launch: {ms[0]:.4f} ms by events; first start -> last end {int(t_last - t_first)} ticks = {(t_last - t_first) / ms[0] * 1e-6:.3f} GHz if s_memtime is the core clock")
# the counters of the eight XCDs are not aligned: split the tiles into clusters of start values (gaps > 1e6 ticks) and take each cluster's span
order = np.argsort(st[:, 0, 0])
s0, e0 = st[order, 0, 0], st[order, :, 15].max(axis=1)
cuts = np.flatnonzero(np.diff(s0) > 1_000_000) + 1
spans = [int(e.max() - b.min()) for b, e in zip(np.split(s0, cuts), np.split(e0, cuts))]
print("per-XCD spans (ticks):", spans, "tiles per cluster:", [len(b) for b in np.split(s0, cuts)])
print(f"  -> {np.median(spans) / ms[0] * 1e-6:.3f} GHz if s_memtime is the core clock and a cluster spans the whole launch")
dd = (st - st[:, :1, :1]) & 0xffffffff
names = ["start", "PE done", "M1 done", "bar", "V1 done", "bar", "M2 done", "bar", "V2 done", "bar", "M3 done", "bar", "V3 done", "bar", "ML done", "epilogue"]
med = np.median(dd, axis=0)
print("median stamp (clk since wave 0's start) and stage length, waves 0, 3, 7:")
for k, n in enumerate(names):
    print(f"  {k:2d} {n:10s}", [int(med[w_, k]) for w_ in (0, 3, 7)], " stage:", [int(med[w_, k] - med[w_, k - 1]) if k else 0 for w_ in (0, 3, 7)])
print("tile duration median", int(np.median(dd[:, :, 15].max(axis=1))), "p90", int(np.percentile(dd[:, :, 15].max(axis=1), 90)))
