"""iff_logits_from_cache alone (k5_trunk_h<3, ...> + the statistics merge), hipEvents on one stream: Q x 256 token rows against a
cached ray set.  python scripts/time_cached_logits.py [config:Q ...]   (IFF_LIB_PATH names a development library)   Dev aid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
dev = torch.device("cuda:0")
for spec in sys.argv[1:] or ["lego16k:32", "lego540k:8"]:
    cfg, Q = spec.split(":"); Q = int(Q)
    wl = synthetic.WORKLOADS[cfg]
    pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(cfg), synthetic.make_id_weights(seed=99), dev,
                                         trunk_variant=int(os.environ.get("TRUNK_VARIANT", "0")))
    ori, dirs, rgb = pipe.emit(wl["gen_points"], seed=3)
    N = ori.shape[0]
    tok = torch.stack([synthetic.make_tokens(256, 384, seed=7 + q) for q in range(Q)]).to(dev)
    qf = pipe.idnet.q_fold(tok.reshape(Q * 256, -1))
    cache = pipe.idnet.build_ray_cache(ori, dirs, rgb)
    for _ in range(3): pipe.idnet.logits_from_cache(qf, cache, N)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20 if N < 100000 else 6
    a.record()
    for _ in range(n): pipe.idnet.logits_from_cache(qf, cache, N)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(json.dumps({"lib": os.environ.get("IFF_LIB_PATH", "in-tree"), "config": cfg, "queries": Q, "rays": N, "us": round(us, 1),
                      "issued_TFLOPs": round(Q * 256 * N * 256 * 2 * 3 / us / 1e6, 1)}), flush=True)
    del pipe, cache, qf
    torch.cuda.empty_cache()
