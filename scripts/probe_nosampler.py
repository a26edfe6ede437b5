"""dev probe: in-flight throughput of the cold query with and without the surface sampler in the graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iffnerf_amd import synthetic, hip_identify as H
from iffnerf_amd.hip_field import isocell_emit
dev = torch.device("cuda:0")
ck, idw, pipe = bench.build_inputs(dev)
tok = synthetic.make_tokens(256, 384, seed=7).to(dev)
samples, _, _ = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=1)

def body(with_sampler, counter):
    if with_sampler:
        s, _, _ = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=1, seed_offset=counter)
    else:
        s = samples
    normals = pipe.field.point_normals(s)
    ori, dirs, rays = isocell_emit(pipe.cells, s, normals, want_rays6=True)
    rgb = pipe.field.march(rays, 0, 20, want_alpha=False)[0]
    return pipe.identify(tok, ori, dirs, rgb, 100, False)[0]

for mode in ("full", "no sampler", "sampler only"):
    F = 4
    graphs, streams, outs = [], [], []
    for g in range(F):
        counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                body(mode == "full", counter) if mode != "sampler only" else pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=1, seed_offset=counter)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            counter += 1
            if mode == "sampler only":
                outs.append(pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=1, seed_offset=counter)[0])
            else:
                outs.append(body(mode == "full", counter))
        graphs.append(gr); streams.append(torch.cuda.Stream(dev))
    torch.cuda.synchronize()
    for i in range(40):
        with torch.cuda.stream(streams[i % F]): graphs[i % F].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 400
    for i in range(n):
        with torch.cuda.stream(streams[i % F]): graphs[i % F].replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:14s}: {dt / n * 1e6:.1f} us per replay, {n / dt:.0f}/s")

