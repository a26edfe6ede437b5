"""dev probe: K4a with one lane vs four lanes per sample, 16 queries' rays per launch (timed by the instrumented march)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.hip_field import field_handle_from_ckpt, isocell_emit
from iffnerf_amd.pose_estimation.isocell import isocell_distribution
dev = torch.device("cuda:0")
ck = synthetic.make_workload_ckpt("lego16k")
cells = isocell_distribution(27, torch.float32, dev)
for lanes in (1, 4, 1, 4):
    fh = field_handle_from_ckpt(ck, dev, density_lanes=lanes)
    rays = []
    for q in range(16):
        s, _, _ = fh.surface_sample(593, 0.1 * 300 * 0.01, 4, 200, seed=q + 1)
        n = fh.point_normals(s)
        rays.append(isocell_emit(cells, s, n, want_rays6=True)[2])
    rays = torch.cat(rays)
    ms = []
    for _ in range(12):
        st = []
        fh.march(rays, 0, 20, want_alpha=False, stage_ms=st)
        ms.append(st)
    ms = torch.tensor(ms[2:]).median(0).values.tolist()
    print(f"density_lanes={lanes}: k4a {ms[0]:.4f} ms, k4b {ms[1]:.4f}, shade {ms[2]:.4f}  ({rays.shape[0]} rays)")
