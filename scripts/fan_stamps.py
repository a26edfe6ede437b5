"""Per-phase wave timelines of the fused fan kernel (library built with -DFAN_STAMPS); dev aid."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
from iffnerf_amd.hip_field import isocell_emit
wl = synthetic.WORKLOADS["lego16k"]
dev = torch.device("cuda:0")
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt("lego16k"), synthetic.make_id_weights(seed=99), dev)
samples, _, _ = pipe.field.surface_sample_batched(16, wl["gen_points"], pipe.rho, 4, 200, seed=5000)
samples = samples.reshape(-1, 3)
ori, dirs, rays = isocell_emit(pipe.cells, samples, pipe.field.point_normals(samples), want_rays6=True)
for _ in range(3):
    out = pipe.field.march(rays, 0, 20, want_alpha=True)
torch.cuda.synchronize()
st = out[3].view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 27 * 20)[:, :64].reshape(-1, 4, 16) & 0xffffffff
d = (st - st[:, :1, :1]) & 0xffffffff          # relative to wave 0's start stamp
names = ["start", "box+fetchD", "A", "stash+fetchA0", "B gather", "composite", "C0 in", "C0 out", "C1 in", "C1 out", "C2 in", "C2 out", "D in", "D out", "end"]
med = np.median(d[:, :, :15], axis=0)
print("median stamp (clk since the tile's start), waves 0..3:")
for k, n in enumerate(names):
    print(f"  {n:14s}", [int(x) for x in med[:, k]], " phase:", [int(x) for x in (med[:, k] - med[:, k - 1] if k else med[:, k])])
print("tile duration median", int(np.median(d[:, :, 14].max(axis=1))), "p90", int(np.percentile(d[:, :, 14].max(axis=1), 90)))
