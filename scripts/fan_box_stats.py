"""How many texels a fan's patches span per axis (what the staged fan kernels could at best save by staging the ACTUAL box instead of the
template's FP x FP per plane): for every fan of a bench workload the box of the fused kernels' phase 0 -- floor of the 54 ray end
points' texel coordinates, + 1 -- recomputed in torch from the emitted rays (iff_normalize_coord for the contraction).
    python scripts/fan_box_stats.py [config ...]          (dev aid / evidence for profiles/; not part of the product or the tests)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline

dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["lego16k", "bicycle64k"]):
    wl = synthetic.WORKLOADS[cfg]
    ck = synthetic.make_workload_ckpt(cfg)
    pipe = PosePipeline.from_checkpoints(ck, synthetic.make_id_weights(seed=99), dev)
    ori, dirs, _ = pipe.emit(wl["gen_points"], seed=42)
    grid = torch.tensor([int(g) for g in ck["kwargs"]["gridSize"]], device=dev, dtype=torch.float32)        # (x, y, z)
    from iffnerf_amd.models.tensorBase import derive_step
    kw = ck["kwargs"]
    step = float(derive_step(torch.as_tensor(kw["aabb"]).float(), kw["gridSize"], kw.get("step_ratio", 2.0), kw.get("contraction_type", "aabb"))[0])
    fan_side = pipe.field.fan_kernel(0, 20)[1]
    ext = []
    for s in (0, 19):
        p = ori + dirs * (step * (s - 10))
        xn = pipe.field.normalize_coord(p)
        x = (xn + 1.0) / 2.0 * (grid - 1.0)
        fl = torch.floor(torch.minimum(torch.maximum(x, torch.tensor(-1.0, device=dev)), grid)).long()
        ext.append((torch.clamp(fl, min=0).minimum((grid - 1).long()), torch.clamp(fl + 1, min=0).minimum((grid - 1).long())))
    lo = torch.minimum(ext[0][0], ext[1][0]).view(-1, 27, 3).min(1).values
    hi = torch.maximum(ext[0][1], ext[1][1]).view(-1, 27, 3).max(1).values
    side = (hi - lo + 1).float()                                 # [fans, 3]
    srt = side.sort(dim=1).values                                # per fan: shortest, middle, longest axis
    # bytes a fan stages: three planes (a x b texels) + three lines, 16 density + 48 appearance channels of 4 B
    def staged(sd):
        a, b, c = sd[:, 0], sd[:, 1], sd[:, 2]
        return ((a * b + a * c + b * c) + (a + b + c)) * 64 * 4
    full = torch.full_like(side, float(fan_side))
    print(json.dumps({"config": cfg, "fans": int(side.shape[0]), "fan_kernel_patch_side": int(fan_side), "step": step,
                      "mean_side_xyz": [round(v, 2) for v in side.mean(0).tolist()],
                      "mean_shortest_middle_longest": [round(v, 2) for v in srt.mean(0).tolist()],
                      "max_side": int(side.max()), "fans_that_fit": round(float((side.max(1).values <= fan_side).float().mean()), 4),
                      "staged_bytes_actual_box_over_template": round(float(staged(side).mean() / staged(full).mean()), 3)}))
