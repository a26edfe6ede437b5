"""The reference's call path at its default 540 000 rays, by images per captured batch (LOGITS_BUDGET_BYTES of
iffnerf_amd/pose_estimation/test.py): python scripts/time_dropin_540k.py [budget GiB ...]   (run on the GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from iffnerf_amd import synthetic
from iffnerf_amd.pose_estimation import test as T
dev = torch.device("cuda:0")
ck = synthetic.make_workload_ckpt("lego540k")
idw = synthetic.make_id_weights(seed=99)
for gib in [float(v) for v in sys.argv[1:]] or [4.0, 9.0, 18.0]:
    T.LOGITS_BUDGET_BYTES = int(gib * (1 << 30))
    r = bench.dropin_rates(ck, idw, dev, 20000, 64)
    r["budget_GiB"] = gib
    r["images_per_batch"] = max(1, min(T.EVAL_BATCH, T.LOGITS_BUDGET_BYTES // (256 * 4 * 540000)))
    r["peak_GiB"] = round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)
    print(json.dumps(r), flush=True)
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats(dev)
