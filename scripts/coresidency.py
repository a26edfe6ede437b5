"""Do the fan march and the encoder / logits trunk speed each other up when they share the CUs?  (dev aid)
Times N launches of each alone, then N of each issued on two streams at the same time; `pair_ms` against `fan_ms + trunk_ms` is what
the dispatcher's own mixing of the two kernels' workgroups buys.   [IFF_LIB_PATH=build/lib_x.so] python scripts/coresidency.py [config]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
from iffnerf_amd.hip_field import isocell_emit

cfg = sys.argv[1] if len(sys.argv) > 1 else "lego16k"
wl = synthetic.WORKLOADS[cfg]
dev = torch.device("cuda:0")
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(cfg), synthetic.make_id_weights(seed=99), dev)
QB, P = wl["queries"], wl["gen_points"]
samples, _, _ = pipe.field.surface_sample_batched(QB, P, pipe.rho, 4, 200, seed=5000)
samples = samples.reshape(QB * P, 3)
ori, dirs, rays = isocell_emit(pipe.cells, samples, pipe.field.point_normals(samples), want_rays6=True)
rgb = pipe.field.march(rays, 0, 20, want_alpha=False)[0]
tokens = torch.stack([synthetic.make_tokens(256, 384, seed=7 + q) for q in range(QB)]).to(dev)
qf = pipe.idnet.q_fold(tokens.reshape(QB * 256, -1).contiguous())
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 20


def fan():
    pipe.field.march(rays, 0, 20, want_alpha=False)


def trunk():
    pipe.idnet.ray_logits_folded_batched(qf, ori, dirs, rgb, QB)


def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_event(e0); sb.wait_event(e0)
    for _ in range(N):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa); eb.record(sb)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N


for _ in range(2):
    timed(fan, trunk)
out = {"config": cfg, "lib": os.environ.get("IFF_LIB_PATH", "in-tree")}
for rep in range(2):
    out[f"fan_ms_{rep}"] = round(timed(fan, None), 4)
    out[f"trunk_ms_{rep}"] = round(timed(None, trunk), 4)
    out[f"pair_ms_{rep}"] = round(timed(fan, trunk), 4)
print(json.dumps(out))
