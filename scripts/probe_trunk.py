"""dev probe: standalone time of the fused trunk (events around 50 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
dev = torch.device("cuda:0")
net = H.IdNetHandle(synthetic.make_id_weights(seed=99), dev)
N = 16011
g = torch.Generator().manual_seed(0)
o, d, c = (torch.randn(N, 3, generator=g).to(dev) for _ in range(3))
for _ in range(5):
    net.ray_trunk(o, d, c)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    net.ray_trunk(o, d, c)
b.record()
torch.cuda.synchronize()
print("probe", os.environ.get("IFF_TRUNK_PROBE", "0"), "ray_input_planes + trunk: %.1f us" % (a.elapsed_time(b) / 50 * 1e3))
# back-to-back launches (no host gaps): a hipGraph of 20 trunk calls
gph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(gph):
        for _ in range(20):
            net.ray_trunk(o, d, c)
torch.cuda.synchronize()
gph.replay(); torch.cuda.synchronize()
a.record()
for _ in range(5):
    gph.replay()
b.record()
torch.cuda.synchronize()
print("graph of 20: %.1f us per (ray_input_planes + trunk)" % (a.elapsed_time(b) / 100 * 1e3))

