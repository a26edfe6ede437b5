"""The reference's call path (explore_model + test_pose_estimation) by the number of captured batches in flight (EVAL_SLOTS of
iffnerf_amd/pose_estimation/test.py) and the dataset size; median of five timed calls.
    [IMAGES=128,200,256] python scripts/time_dropin_slots.py [config] [slots ...]   (GPU box; dev aid)"""
import json, os, sys, tempfile, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import iffnerf_amd
from iffnerf_amd import synthetic
iffnerf_amd.install(force=True)
from pose_estimation.model_utils import explore_model, load_model
from pose_estimation import backbone as bb, identification_module as im
from pose_estimation import test as T
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "lego16k"
ck = synthetic.make_workload_ckpt(cfg)
idw = synthetic.make_id_weights(seed=99)
gp = synthetic.WORKLOADS[cfg]["gen_points"]
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "m.th"); torch.save(ck, path); model = load_model(path, dev)
rays = explore_model(model, gen_points=gp)
up = torch.tensor([0.0, 0.0, 1.0], device=dev)


class DS:
    pass


for slots in [int(v) for v in sys.argv[2:]] or [2, 4]:
    T.EVAL_SLOTS = slots
    for n in ([int(v) for v in os.environ["IMAGES"].split(",")] if os.environ.get("IMAGES") else ((128, 200, 256) if gp < 5000 else (68, 136))):
        hub = bb._hub_load
        bb._hub_load = lambda repo, name: bb.SeededViTS14(0)
        try:
            idm = im.IdentificationModule(backbone_type="dino")
        finally:
            bb._hub_load = hub
        idm.load_state_dict({**idm.state_dict(), **idw}); idm = idm.to(dev).eval()
        ds = DS()
        g = torch.Generator().manual_seed(17)
        rgba = torch.rand(n, 800, 800, 4, generator=g); rgba[..., 3] = (rgba[..., 3] > 0.2).float()
        ds.all_rgbs = rgba.to(dev); ds.K = torch.eye(3)[None]; ds.all_rays = torch.zeros(n, 1, 6); ds.poses = torch.eye(4).repeat(n, 1, 1)
        ts = []
        with bench._QuietStdout():
            for rep in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                T.test_pose_estimation(ds, idm, *rays, up)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(json.dumps({"config": cfg, "slots": slots, "images": n, "median_images_per_s": round(n / statistics.median(ts[1:]), 1),
                          "best": round(n / min(ts[1:]), 1), "worst": round(n / max(ts[1:]), 1)}), flush=True)
        del idm, ds
        torch.cuda.empty_cache()
