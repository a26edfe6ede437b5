"""dev probe: the march alone (emit once, march 30 times) -- for rocprofv3 --pmc passes on K4a/K4b/K4c."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from iffnerf_amd.hip_field import isocell_emit
dev = torch.device("cuda:0")
ck, idw, pipe = bench.build_inputs(dev)
samples, _, _ = pipe.field.surface_sample(593, pipe.rho, 4, 200, seed=1)
normals = pipe.field.point_normals(samples)
ori, dirs, rays = isocell_emit(pipe.cells, samples, normals, want_rays6=True)
for _ in range(int(os.environ.get("REPS", "30"))):
    pipe.field.march(rays, 0, 20, want_alpha=False)
torch.cuda.synchronize()
print("done")
