"""Time the point-centred march of the bench workload (16 cold queries' rays per launch) per launch stage.
    [IFF_LIB_PATH=build/lib_x.so] python scripts/time_march.py [config]      (dev aid; not part of the product or the tests)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
from iffnerf_amd.hip_field import isocell_emit

cfg = sys.argv[1] if len(sys.argv) > 1 else "lego16k"
wl = synthetic.WORKLOADS[cfg]
dev = torch.device("cuda:0")
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(cfg), synthetic.make_id_weights(seed=99), dev,
                                     fan_waves=int(os.environ.get("FAN_WAVES", "0")))      # 4 / 8: name the fused fan kernel (iff_field_desc.fan_waves)
QB, P = wl["queries"] if not wl["shared_rays"] else 1, wl["gen_points"]
samples, _, _ = pipe.field.surface_sample_batched(QB, P, pipe.rho, 4, 200, seed=5000)
samples = samples.reshape(QB * P, 3)
if os.environ.get("SORT"):          # experiment: fans in a spatially coherent order (Morton code of a 2^k grid over the samples' extent)
    k = int(os.environ["SORT"])
    lo_, hi_ = samples.min(0).values, samples.max(0).values
    q = ((samples - lo_) / (hi_ - lo_ + 1e-9) * (1 << k)).long().clamp(0, (1 << k) - 1)
    key = torch.zeros(samples.shape[0], dtype=torch.long, device=samples.device)
    for b in range(k):
        for ax in range(3):
            key |= ((q[:, ax] >> b) & 1) << (3 * b + ax)
    if os.environ.get("SORT_PER_QUERY"):          # ... inside each query only (what a product could do: a query's rays stay together)
        key = key + (torch.arange(samples.shape[0], device=samples.device) // P << 40)
    samples = samples[torch.argsort(key)]
normals = pipe.field.point_normals(samples)
ori, dirs, rays = isocell_emit(pipe.cells, samples, normals, want_rays6=True)
tot = [0.0, 0.0, 0.0]
n = 20
for i in range(n + 3):
    ms = []
    out = pipe.field.march(rays, 0, 20, want_alpha=False, want_counts=True, stage_ms=ms)
    if i >= 3:
        tot = [a + b / n for a, b in zip(tot, ms)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    pipe.field.march(rays, 0, 20, want_alpha=False)
e0.record()
for _ in range(n):
    pipe.field.march(rays, 0, 20, want_alpha=False)
e1.record()
torch.cuda.synchronize()
c = out[4].double().sum(0)
print(json.dumps({"config": cfg, "fan_waves": int(os.environ.get("FAN_WAVES", "0")), "plan": pipe.field.march_plan(0, 20), "lib": os.environ.get("IFF_LIB_PATH", "in-tree"), "rays": rays.shape[0],
                  "stage_ms": [round(t, 4) for t in tot], "march_ms": round(e0.elapsed_time(e1) / n, 4),
                  "valid_per_ray": round(c[0].item() / rays.shape[0], 2), "shaded_per_ray": round(c[1].item() / rays.shape[0], 2),
                  "rgb_sum": float(out[0].double().sum())}))
