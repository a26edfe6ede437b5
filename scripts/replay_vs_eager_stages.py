"""query_batch stage by stage inside captured graphs (every intermediate kept alive), NF graphs replayed concurrently; after each
round every stage of every graph is recomputed eagerly FROM THE GRAPH'S OWN inputs of that stage and compared bit for bit.  Dev aid."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
from iffnerf_amd.pipeline import PosePipeline
dev = torch.device("cuda:0")
CFG = os.environ.get("CONFIG", "truck32k")
ROUNDS = int(os.environ.get("ROUNDS", "150"))
NF = int(os.environ.get("INFLIGHT", "4"))
KEEP = os.environ.get("KEEP", "1") == "1"
wl = synthetic.WORKLOADS[CFG]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(CFG), synthetic.make_id_weights(seed=99), dev, trunk_variant=int(os.environ.get("TV", "0")))
B, P = wl["queries"], wl["gen_points"]
tokens = torch.stack([synthetic.make_tokens(256, 384, seed=7 + q) for q in range(B)]).to(dev)
M, C = tokens.shape[1:]


def emit(samples):
    from iffnerf_amd.hip_field import isocell_emit
    normals = pipe.field.point_normals(samples)
    ori, dirs, rays = isocell_emit(pipe.cells, samples, normals, want_rays6=True)
    m = pipe.field.march(rays, 0, 20, want_alpha=False)
    feat = pipe.field.march_features(rays, 0, 20)[0] if os.environ.get("FEAT", "0") == "1" else m[0]
    return ori, dirs, m[0], m[1], m[2], feat


def stages(seed, counter):
    out = {}
    samples, _, _ = pipe.field.surface_sample_batched(B, P, pipe.rho, n_epochs=4, max_iterations=200, seed=seed, seed_offset=counter)
    out["samples"] = samples
    out["ori"], out["dirs"], out["rgb"], out["depth"], out["acc"], out["feat"] = emit(samples.reshape(B * P, 3))
    ONLY = os.environ.get("ONLY", "")          # which later stages run in the graph (on static inputs when their producer is left out)
    if ONLY:
        st = STATIC
        if "qfold" in ONLY:
            out["x_qf"] = pipe.idnet.q_fold(tokens.reshape(B * M, C))
        if "trunk" in ONLY:
            out["x_logits"], _, _ = pipe.idnet.ray_logits_folded_batched(st["qf"], st["ori"], st["dirs"], st["rgb"], B)
        if "colsum" in ONLY:
            out["x_score"] = H.attn_colsum_batched(st["logits"], st["rmax"], st["rsum"], B, write_attention=False)
        if "topk" in ONLY:
            out["x_idx"], _ = H.topk_batched(st["score"], 100)
        return out
    out["qf"] = pipe.idnet.q_fold(tokens.reshape(B * M, C))
    out["logits"], out["rmax"], out["rsum"] = pipe.idnet.ray_logits_folded_batched(out["qf"], out["ori"], out["dirs"], out["rgb"], B)
    out["score"] = H.attn_colsum_batched(out["logits"], out["rmax"], out["rsum"], B, write_attention=False)
    out["idx"], out["val"] = H.topk_batched(out["score"], 100)
    return out


STATIC = {}
if os.environ.get("ONLY", ""):
    _o, _d, _c = pipe.emit(P, seed=5)
    STATIC["ori"], STATIC["dirs"], STATIC["rgb"] = [x.repeat(B, 1).contiguous() for x in (_o, _d, _c)]
    STATIC["qf"] = pipe.idnet.q_fold(tokens.reshape(B * M, C))
    STATIC["logits"], STATIC["rmax"], STATIC["rsum"] = pipe.idnet.ray_logits_folded_batched(STATIC["qf"], STATIC["ori"], STATIC["dirs"], STATIC["rgb"], B)
    STATIC["score"] = H.attn_colsum_batched(STATIC["logits"], STATIC["rmax"], STATIC["rsum"], B, write_attention=False)
    torch.cuda.synchronize()


class G:
    def __init__(self, seed):
        self.seed = seed
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                stages(seed, self.counter)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.counter += 1
            self.out = stages(seed, self.counter)


from iffnerf_amd.hip_field import field_handle_from_ckpt, isocell_emit
gen = field_handle_from_ckpt(synthetic.make_workload_ckpt(CFG), dev, density_lanes=1)      # the general kernels: a third opinion
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
        o = g.out
        e = {}
        e["ori"], e["dirs"], e["rgb"], e["depth"], e["acc"], e["feat"] = emit(o["samples"].reshape(B * P, 3))
        if not os.environ.get("ONLY", ""):
            e["qf"] = pipe.idnet.q_fold(tokens.reshape(B * M, C))
            e["logits"], e["rmax"], e["rsum"] = pipe.idnet.ray_logits_folded_batched(o["qf"], o["ori"], o["dirs"], o["rgb"], B)
            e["score"] = H.attn_colsum_batched(o["logits"].clone(), o["rmax"], o["rsum"], B, write_attention=False)
            e["idx"], e["val"] = H.topk_batched(o["score"], 100)
        n_nan_replay = int(torch.isnan(o["rgb"]).any(dim=1).sum())
        n_nan_eager = int(torch.isnan(e["rgb"]).any(dim=1).sum())
        if n_nan_replay or n_nan_eager:
            print(json.dumps({"round": r, "graph": i, "nan_rays_replay": n_nan_replay, "nan_rays_eager": n_nan_eager,
                              "which": torch.isnan(o["rgb"]).any(dim=1).nonzero().flatten()[:8].tolist()}), flush=True)
        diff = [k for k in e if not torch.equal(e[k].nan_to_num(7.0), o[k].nan_to_num(7.0))]
        if diff:
            n_bad += 1
            rec = {"round": r, "graph": i, "differ": diff}
            for k in diff[:4]:
                d = (e[k].float() - o[k].float()).abs()
                nz = d.reshape(-1).nonzero().flatten()
                rec[k] = {"n": int(nz.numel()), "first": nz[:4].tolist(), "last": nz[-2:].tolist(), "shape": list(o[k].shape), "max": float(d.max())}
            if "rgb" in diff:
                rows = (e["rgb"] - o["rgb"]).abs().amax(dim=1).nonzero().flatten()
                normals = pipe.field.point_normals(o["samples"].reshape(B * P, 3))
                rays = isocell_emit(pipe.cells, o["samples"].reshape(B * P, 3), normals, want_rays6=True)[2]
                t0 = int(rows[0]) // 27 * 27
                ref = gen.march(rays[t0:t0 + 27].contiguous(), 0, 20, want_alpha=False)[0]
                from iffnerf_amd.models.tensorBase import derive_step
                kw = synthetic.make_workload_ckpt(CFG)["kwargs"]
                aabb = torch.as_tensor(kw["aabb"]).float().to(dev)
                step, _ = derive_step(aabb.cpu(), kw["gridSize"], kw.get("step_ratio", 2.0), kw.get("contraction_type", "aabb"))
                G3 = torch.tensor(kw["gridSize"], device=dev).float()
                rt = rays[t0:t0 + 27]
                ends = torch.cat([rt[:, :3] + rt[:, 3:6] * (float(step) * z) for z in (-10.0, 9.0)])
                x = ((ends - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0 + 1.0) / 2.0 * (G3 - 1.0)
                fl = x.floor()
                ext = (fl.amax(0) + 1 - fl.amin(0) + 1).tolist()
                rec["box_extent"] = ext
                rec["box_lo"] = fl.amin(0).tolist()
                rec["grid"] = kw["gridSize"]
                rec["tile"] = t0 // 27
                rec["rays_in_tile"] = [int(x) - t0 for x in rows.tolist()]
                rec["replay_vs_general"] = float((o["rgb"][t0:t0 + 27] - ref).abs().max())
                rec["eager_vs_general"] = float((e["rgb"][t0:t0 + 27] - ref).abs().max())
            if "feat" in diff and o["feat"].shape[1] == 28:
                d = (e["feat"] - o["feat"]).abs()
                rows = d.amax(dim=1).nonzero().flatten().tolist()
                rec["feat_rows"] = [(rw, rw % 27, [round(float(x), 7) for x in d[rw].tolist()]) for rw in rows[:2]]
                rec["feat_vals"] = [round(float(x), 5) for x in o["feat"][rows[0]].tolist()]
            print(json.dumps(rec), flush=True)
print(json.dumps({"config": CFG, "checks": ROUNDS * NF, "mismatches": n_bad}))
