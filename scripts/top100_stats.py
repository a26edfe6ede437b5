"""How often is the library's top-100 ray list the oracle's, bit for bit (identification_module.py:207)?  A RATE instead of an anecdote:
Q query token sets x 3 sampler seeds per BASELINE workload through the default arithmetic (2xFP16 split products, folded heads,
the per-model encoder cache -- the bits of the cold fused launch, tests/test_hip_dropin.py) against the oracle (the reference's fp32
op chain on the host) on the SAME rays.  Every list must be the oracle's up to near-tie pairs (tests/util.py assert_topk_matches,
2e-5 relative); for every list that differs an fp64 evaluation of the same op chain referees the disputed pairs.
    python scripts/top100_stats.py <config> [queries_per_seed]   ->  gpurun_out/top100_stats_<config>.json   (GPU box; the oracle is the checker)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
from oracle import identify as oid
from tests import util

cfg = sys.argv[1] if len(sys.argv) > 1 else "lego16k"
QS = int(sys.argv[2]) if len(sys.argv) > 2 else 171
SEEDS = (55176280, 1234567, 987654321)
TIE = 2e-5
K = 100
torch.set_num_threads(min(os.cpu_count() or 1, 16))
dev = torch.device("cuda:0")
idw = synthetic.make_id_weights(seed=99)
idw64 = {k: v.double() for k, v in idw.items()}
wl = synthetic.WORKLOADS[cfg]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(cfg), idw, dev)
out = {"config": cfg, "rays": 27 * wl["gen_points"], "queries": 0, "sampler_seeds": list(SEEDS), "identical_lists": 0,
       "lists_differing_by_near_tie_pairs_only": 0, "lists_whose_oracle_top101_has_a_near_tie_pair": 0,
       "positions_differing_total": 0, "disputed_pairs_fp64_agrees_with_hip": 0, "disputed_pairs_fp64_agrees_with_oracle": 0,
       "disputed_pairs_fp64_tied": 0, "near_tie_relative": TIE, "arithmetic": pipe.idnet.gemm_description()}
t_start = time.time()
for si, seed in enumerate(SEEDS):
    ori, dirs, rgb = pipe.emit(wl["gen_points"], seed=seed)
    resident = pipe.make_resident(ori, dirs, rgb)
    o, d, c = ori.cpu(), dirs.cpu(), rgb.cpu()
    rf = oid.ray_encode(idw, o, d, c)
    rf64 = None
    for q0 in range(0, QS, 32):
        nq = min(32, QS - q0)
        tok = torch.stack([synthetic.make_tokens(256, 384, seed=10000 * (si + 1) + q0 + q) for q in range(nq)])
        idx = pipe.identify_resident(tok.to(dev), resident, K)[1].cpu()
        for q in range(nq):
            want = oid.attention_map(idw, tok[q], rf).sum(0)
            diff = util.assert_topk_matches(idx[q], want, K, rel_tie=TIE)          # raises on anything but near-tie swaps
            out["queries"] += 1
            out["identical_lists"] += int(diff == 0)
            out["lists_differing_by_near_tie_pairs_only"] += int(diff != 0)
            out["positions_differing_total"] += diff
            out["lists_whose_oracle_top101_has_a_near_tie_pair"] += int(util.near_tie_pairs(want, K, TIE) > 0)
            if diff:
                if rf64 is None:
                    rf64 = oid.ray_encode(idw64, o.double(), d.double(), c.double())
                s64 = oid.attention_map(idw64, tok[q].double(), rf64).sum(0)
                h, r, t = util.fp64_referee(idx[q], want, s64, K)
                out["disputed_pairs_fp64_agrees_with_hip"] += h
                out["disputed_pairs_fp64_agrees_with_oracle"] += r
                out["disputed_pairs_fp64_tied"] += t
        print(cfg, "seed", si, "queries", out["queries"], "identical", out["identical_lists"], "%.0f s" % (time.time() - t_start), flush=True)
out["identical_fraction"] = round(out["identical_lists"] / max(out["queries"], 1), 4)
out["near_tie_list_fraction"] = round(out["lists_whose_oracle_top101_has_a_near_tie_pair"] / max(out["queries"], 1), 4)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/top100_stats_%s.json" % cfg, "w"), indent=1)
print(json.dumps(out))
