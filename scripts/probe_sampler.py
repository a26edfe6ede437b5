"""Dev probe: where the surface sampler's time goes (not part of the product or the tests)."""
import sys, time, torch
sys.path.insert(0, ".")
from iffnerf_amd import synthetic
from iffnerf_amd.hip_field import field_handle_from_ckpt
from iffnerf_amd.pipeline import jitter_scale_from_kwargs
ck = synthetic.make_field_ckpt(grid=(300, 300, 300), mask_res=(180, 180, 180), seed=1234, step_ratio=0.5, peak=20.0)
h = field_handle_from_ckpt(ck, "cuda:0")
rho = jitter_scale_from_kwargs(ck["kwargs"])
def t(P, ne, mi, n=30):
    for _ in range(3): h.surface_sample(P, rho, ne, mi, seed=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): s, a, st = h.surface_sample(P, rho, ne, mi, seed=10 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, st.cpu()[:, :2].tolist()
for P in (593, 2371, 20000):
    for ne, mi in ((0, 200), (1, 0), (1, 1), (1, 200), (4, 200)):
        us, st = t(P, ne, mi)
        print(f"P={P} epochs={ne} max_it={mi}: {us:8.1f} us  stats(it,left)={st}")
