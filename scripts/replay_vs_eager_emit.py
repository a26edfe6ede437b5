"""Stages A + B only (batched sampler, normals, fans, march) as captured graphs replayed NF at a time against the eager march of the
SAME samples: which rays' colours differ, if any.  Dev aid (CONFIG=, ROUNDS=, INFLIGHT=)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
dev = torch.device("cuda:0")
CFG = os.environ.get("CONFIG", "truck32k")
ROUNDS = int(os.environ.get("ROUNDS", "200"))
NF = int(os.environ.get("INFLIGHT", "4"))
wl = synthetic.WORKLOADS[CFG]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(CFG), synthetic.make_id_weights(seed=99), dev)
B, P = wl["queries"], wl["gen_points"]


class G:
    def __init__(self, seed):
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self.body(seed)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.counter += 1
            self.samples, self.ori, self.dirs, self.rgb = self.body(seed)

    def body(self, seed):
        samples, _, _ = pipe.field.surface_sample_batched(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=seed, seed_offset=self.counter)
        ori, dirs, rgb = pipe.emit_from_samples(samples.reshape(B * P, 3))
        return samples, ori, dirs, rgb


graphs = [G(1000 + 7919 * i) for i in range(NF)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
torch.cuda.synchronize()
n_bad = 0
for r in range(ROUNDS):
    for rep in range(3):
        for i, g in enumerate(graphs):
            with torch.cuda.stream(streams[i]):
                g.graph.replay()
    torch.cuda.synchronize()
    for i, g in enumerate(graphs):
        ori, dirs, rgb = pipe.emit_from_samples(g.samples.reshape(B * P, 3))
        if not (torch.equal(ori, g.ori) and torch.equal(dirs, g.dirs) and torch.equal(rgb, g.rgb)):
            n_bad += 1
            d = (rgb - g.rgb).abs().amax(dim=1)
            rows = d.nonzero().flatten()
            print(json.dumps({"round": r, "graph": i, "ori_eq": bool(torch.equal(ori, g.ori)), "dirs_eq": bool(torch.equal(dirs, g.dirs)),
                              "n_rays_differ": int(rows.numel()), "first": rows[:6].tolist(), "last": rows[-3:].tolist(),
                              "tiles": sorted(set((rows // 27).tolist()))[:12], "n_total": int(rgb.shape[0]), "maxdiff": float(d.max())}), flush=True)
print(json.dumps({"config": CFG, "checks": ROUNDS * NF, "mismatches": n_bad}))
