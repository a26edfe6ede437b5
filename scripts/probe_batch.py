"""dev probe: do the throughput kernels get cheaper per ray at 4x the rays?  (rocprofv3 --kernel-trace, grouped by grid size)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from iffnerf_amd import synthetic
dev = torch.device("cuda:0")
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
for P in (593, 2372):
    samples = pipe.sample_surface(P, seed=1)
    for _ in range(20):
        ori, dirs, rgb = pipe.emit_from_samples(samples)
        pipe.logits(tok, ori, dirs, rgb)
    torch.cuda.synchronize()
print("done")
