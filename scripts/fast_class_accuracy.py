"""What ONE fp16 product per block (IFF_GEMM_F16X1) costs in accuracy, next to the default (IFF_GEMM_F16X2: three products, the fp32 class):
logits, scores, top-100 lists and poses of both on the SAME rays and query tokens, the default as the reference (itself within 1e-4 of the
oracle: tests/test_hip_fullsize.py).     python scripts/fast_class_accuracy.py [config] [queries]      (GPU box; dev aid / evidence)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic, hip_identify as H
from iffnerf_amd.pipeline import PosePipeline

cfg = sys.argv[1] if len(sys.argv) > 1 else "lego16k"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
wl = synthetic.WORKLOADS[cfg]
ck, idw = synthetic.make_workload_ckpt(cfg), synthetic.make_id_weights(seed=99)
ref = PosePipeline.from_checkpoints(ck, idw, dev, model_up=(0.0, 0.0, 1.0))
fast = PosePipeline.from_checkpoints(ck, idw, dev, model_up=(0.0, 0.0, 1.0), gemm_mode=H.GEMM_F16X1)
assert fast.idnet.gemm_mode == H.GEMM_F16X1 and ref.idnet.gemm_mode == H.GEMM_F16X2
ori, dirs, rgb = ref.emit(wl["gen_points"], seed=42)
same_list = same_set = 0
overlap, max_logit_err, max_logit, pose_t, pose_r, score_rel = [], 0.0, 0.0, [], [], 0.0
for q in range(Q):
    tok = synthetic.make_tokens(256, 384, seed=1000 + q).to(dev)
    la, _, _ = ref.logits(tok, ori, dirs, rgb)
    lb, _, _ = fast.logits(tok, ori, dirs, rgb)
    max_logit_err = max(max_logit_err, float((la - lb).abs().max())); max_logit = max(max_logit, float(la.abs().max()))
    pa, ia, va = ref.identify(tok, ori, dirs, rgb, k=100, materialize_map=False)
    pb, ib, vb = fast.identify(tok, ori, dirs, rgb, k=100, materialize_map=False)
    same_list += int(torch.equal(ia, ib)); n = len(set(ia.tolist()) & set(ib.tolist())); overlap.append(n); same_set += int(n == 100)
    score_rel = max(score_rel, float(((va - vb).abs() / va.abs().clamp_min(1e-30)).max()))
    pose_t.append(float((pa[:3, 3] - pb[:3, 3]).norm()))
    R = pa[:3, :3].double() @ pb[:3, :3].double().T
    pose_r.append(float(torch.arccos(((torch.trace(R) - 1) / 2).clamp(-1, 1))))
print(json.dumps({"config": cfg, "rays": int(ori.shape[0]), "queries": Q, "reference": "IFF_GEMM_F16X2 (default, fp32 class)", "measured": "IFF_GEMM_F16X1",
                  "max_abs_logit_diff": round(max_logit_err, 5), "max_abs_logit": round(max_logit, 2),
                  "top100_lists_identical": same_list, "top100_same_100_rays": same_set, "top100_common_rays_min_mean": [min(overlap), round(sum(overlap) / Q, 2)],
                  "top100_score_rel_diff_max": round(score_rel, 5),
                  "pose_translation_diff_max_mean": [round(max(pose_t), 6), round(sum(pose_t) / Q, 6)],
                  "pose_rotation_diff_rad_max_mean": [round(max(pose_r), 6), round(sum(pose_r) / Q, 6)]}))
