import sys, torch, ctypes as C
sys.path.insert(0, ".")
import bench
from iffnerf_amd import _lib
from iffnerf_amd._lib import check, dptr, stream_ptr
dev = torch.device("cuda:0")
ck, idw, pipe = bench.build_inputs(dev)
F = pipe.field; L = _lib.lib(); P = 593
def run(keep):
    ws_bytes = int(L.iff_surface_sample_workspace(P))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    samples = torch.empty(P, 3, device=dev); alpha = torch.empty(P, device=dev)
    stats = torch.empty(4, 4, dtype=torch.int32, device=dev)
    check(L.iff_surface_sample(F._h, P, 4, 200, 5, None, float(pipe.rho), dptr(samples), dptr(alpha), dptr(stats, torch.int32), ws.data_ptr(), ws_bytes, stream_ptr(dev)), "x")
    return (samples, alpha, stats, ws) if keep else (samples, alpha)
for keep in (True, False):
    ref = [o.clone() for o in run(True)]; torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = run(keep)
    for t in range(3):
        g.replay(); torch.cuda.synchronize()
        msg = f"keep={keep} replay {t}: samples==eager {torch.equal(ref[0], outs[0])}"
        if keep:
            hdr = outs[3][:8].view(torch.int32).cpu().tolist()
            msg += f" barrier_count={hdr[0]} abort={hdr[1]} stats={outs[2].cpu()[:, :2].tolist()}"
        print(msg)
