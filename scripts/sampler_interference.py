"""How much a free-running surface sampler (one launch per step on side streams, nothing waits for it) slows the march and the trunk
when each runs alone in the step loop.  Dev aid."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
from iffnerf_amd.pipeline import PosePipeline
from iffnerf_amd.hip_field import isocell_emit
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS["lego16k"]
B, M, K, P = wl["queries"], 256, 100, wl["gen_points"]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt("lego16k"), synthetic.make_id_weights(seed=99), dev)
tokens = torch.stack([synthetic.make_tokens(M, 384, seed=7 + q) for q in range(B)]).to(dev)
samples0, _, _ = pipe.field.surface_sample_batched(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=5)
s0 = samples0.reshape(-1, 3)
nrm0 = pipe.field.point_normals(s0)
ori0, dirs0, rays0 = isocell_emit(pipe.cells, s0, nrm0, want_rays6=True)
rgb0 = pipe.field.march(rays0, 0, 20, want_alpha=False)[0]
qf0 = pipe.idnet.q_fold(tokens.reshape(B * M, -1))

print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None, flush=True)
def capture(fn, n=4, priority=0):
    gs, ss = [], [torch.cuda.Stream(device=dev, priority=priority) for _ in range(n)]
    for i in range(n):
        with torch.cuda.stream(ss[i]):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=ss[i]):
            keep = fn(i)
        gs.append((g, keep))
    torch.cuda.synchronize()
    return gs, ss

def run(main, side, steps=200):
    def go(n):
        for i in range(n):
            if side is not None:
                with torch.cuda.stream(side[1][i % 4]):
                    side[0][i % 4][0].replay()
            with torch.cuda.stream(main[1][i % 4]):
                main[0][i % 4][0].replay()
    go(12); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(steps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

tiny = [torch.zeros(64, device=dev) for _ in range(4)]
def empty_chain(i, n=int(os.environ.get("CHAIN", "42"))):
    for _ in range(n):
        tiny[i].add_(1.0)
    return tiny[i]
dummy = capture(empty_chain)
MAIN_PRIO = int(os.environ.get("MAIN_PRIO", "0"))
SIDE_PRIO = int(os.environ.get("SIDE_PRIO", "0"))
sampler = capture(lambda i: pipe.field.surface_sample_batched(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=77 + i), priority=SIDE_PRIO)
march = capture(lambda i: pipe.field.march(rays0, 0, 20, want_alpha=False), priority=MAIN_PRIO)
trunk = capture(lambda i: pipe.idnet.ray_logits_folded_batched(qf0, ori0, dirs0, rgb0, B), priority=MAIN_PRIO)
for name, m in (("march (fan kernel + head)", march), ("trunk", trunk)):
    a, b, c = run(m, None), run(m, sampler), run(m, dummy)
    print(json.dumps({"kernel": name, "ms_per_step_alone": round(a, 4), "ms_per_step_next_to_samplers": round(b, 4), "slowdown": round(b / a, 3),
                      "ms_per_step_next_to_chains_of_empty_kernels": round(c, 4)}), flush=True)
print(json.dumps({"sampler_alone_ms_per_launch_4_in_flight": round(run(sampler, None), 4)}))
