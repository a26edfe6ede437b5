"""Per-phase wave timelines of the eight-wave fan kernel (a library whose fan8_march_kernels.hip is built with -DFAN_STAMPS:
bash scripts/build_one_tu.sh stamps8 fan8_march_kernels.hip -DFAN_STAMPS; IFF_LIB_PATH=build/lib_stamps8.so python scripts/fan8_stamps.py [config]); dev aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
from iffnerf_amd.hip_field import isocell_emit
cfg = sys.argv[1] if len(sys.argv) > 1 else "lego16k"
wl = synthetic.WORKLOADS[cfg]
dev = torch.device("cuda:0")
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(cfg), synthetic.make_id_weights(seed=99), dev, fan_waves=8)
samples, _, _ = pipe.field.surface_sample_batched(wl["queries"], wl["gen_points"], pipe.rho, 4, 200, seed=5000)
samples = samples.reshape(-1, 3)
ori, dirs, rays = isocell_emit(pipe.cells, samples, pipe.field.point_normals(samples), want_rays6=True)
for _ in range(3):
    out = pipe.field.march(rays, 0, 20, want_alpha=True)
torch.cuda.synchronize()
st = out[3].view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 27 * 20)[:, :128].reshape(-1, 8, 16) & 0xffffffff
d = (st - st[:, :1, :1]) & 0xffffffff          # relative to wave 0's start stamp
names = {0: "start", 1: "box+dma0", 2: "records", 4: "density", 5: "composite", 6: "app0 in", 7: "app1 in", 8: "app2 in", 9: "app3 in", 10: "app4.. in",
         11: "app out", 12: "D operands", 13: "D out", 14: "end"}
med = np.median(d, axis=0)
print("median stamp (clk since the tile's start), waves 0..7:")
prev = None
for k in sorted(names):
    if med[:, k].max() <= 0 and k not in (0,):
        continue
    print(f"  {names[k]:12s}", [int(x) for x in med[:, k]], " since prev:", [int(x) for x in (med[:, k] - med[:, prev])] if prev is not None else "")
    prev = k
dur = d[:, :, 14].max(axis=1)
print("tile duration median", int(np.median(dur)), "p90", int(np.percentile(dur, 90)), "tiles", d.shape[0])
# how many tiles overlap in time on a CU cannot be seen from here; the launch time / (tiles / 256 CUs) / tile duration says it
