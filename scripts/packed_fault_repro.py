"""Stand-alone reproducer attempt for the packed-fp32 fault: scripts/micro/packed_fault.hip's victim kernel -- the fan
march's appearance loop on synthetic LDS contents, every sample evaluated twice and compared in the kernel -- runs one long workgroup
per CU while the product library's encoder / logits kernel (fp16 MFMA) is launched over and over on a second stream.
    python scripts/packed_fault_repro.py build/libpacked_fault_on.so [rounds] [with_trunk=1]      (dev aid; GPU box)
Prints the mismatch count (values whose two evaluations differ) and which sixteen-lane slots of the workgroup saw them."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
from iffnerf_amd.hip_field import isocell_emit

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
with_trunk = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
lib.run_victim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = "truck32k"
wl = synthetic.WORKLOADS[cfg]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(cfg), synthetic.make_id_weights(seed=99), dev)
QB, P = wl["queries"], wl["gen_points"]
samples, _, _ = pipe.field.surface_sample_batched(QB, P, pipe.rho, 4, 200, seed=5000)
samples = samples.reshape(QB * P, 3)
ori, dirs, rays = isocell_emit(pipe.cells, samples, pipe.field.point_normals(samples), want_rays6=True)
rgb = pipe.field.march(rays, 0, 20, want_alpha=False)[0]
tokens = torch.stack([synthetic.make_tokens(256, 384, seed=7 + q) for q in range(QB)]).to(dev)
qf = pipe.idnet.q_fold(tokens.reshape(QB * 256, -1).contiguous())
g = torch.Generator(device="cpu"); g.manual_seed(1)
pattern = (torch.randn(7488, generator=g) * 0.3).to(dev)
blocks = int(os.environ.get("BLOCKS", "256"))
mism = torch.zeros(17, dtype=torch.int32, device=dev)
sink = torch.zeros(blocks * 256, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "40"))
done = torch.cuda.Event()
for rep in range(reps):
    with torch.cuda.stream(sa):
        rc = lib.run_victim(ctypes.c_void_p(sa.cuda_stream), ctypes.c_void_p(pattern.data_ptr()), blocks, rounds, 8,
                            ctypes.c_void_p(mism.data_ptr()), ctypes.c_void_p(sink.data_ptr()))
        assert rc == 0, rc
    if with_trunk:
        with torch.cuda.stream(sb):
            for _ in range(int(os.environ.get("TRUNKS", "6"))):
                pipe.idnet.ray_logits_folded_batched(qf, ori, dirs, rgb, QB)
torch.cuda.synchronize()
m = mism.cpu().tolist()
print(json.dumps({"lib": os.path.basename(sys.argv[1]), "with_trunk": with_trunk, "launches": reps, "rounds": rounds,
                  "values_checked": reps * blocks * 256 * rounds * 12, "mismatches": m[0], "threads_by_16_lane_slot": m[1:]}))
