"""Dev probe: pose / topk kernel timing vs k (not part of the product or the tests)."""
import sys, time, torch
sys.path.insert(0, ".")
from iffnerf_amd import hip_identify as H
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
N = 16011
o = torch.randn(N // 27 + 1, 3, generator=g).repeat_interleave(27, 0)[:N].contiguous().to(dev)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(dev)
score = torch.rand(N, generator=g).to(dev)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for k in (4, 32, 100, 400, 1000):
    idx, val = H.topk(score, k)
    print(f"k={k}: topk {t(lambda: H.topk(score, k)):7.1f} us   pose {t(lambda: H.pose_from_topk(idx, val, o, d, (0., 0., 1.))):7.1f} us")
x = torch.empty(16, device=dev)
print("empty torch op", t(lambda: x.zero_()))
