"""Captured cold batches replayed four at a time (as bench.py runs them) against the eager path on the same seed counter, many
times: counts the replays whose poses / top-k lists differ from the eager result and says in what.  Dev aid (CONFIG=, ROUNDS=)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline, CapturedBatchQuery
dev = torch.device("cuda:0")
CFG = os.environ.get("CONFIG", "truck32k")
ROUNDS = int(os.environ.get("ROUNDS", "200"))
NF = int(os.environ.get("INFLIGHT", "4"))
wl = synthetic.WORKLOADS[CFG]
pipe = PosePipeline.from_checkpoints(synthetic.make_workload_ckpt(CFG), synthetic.make_id_weights(seed=99), dev)
B, P = wl["queries"], wl["gen_points"]
tokens = torch.stack([synthetic.make_tokens(256, 384, seed=7 + q) for q in range(B)]).to(dev)
seeds = [1000 + 7919 * i for i in range(NF)]
graphs = [CapturedBatchQuery(pipe, tokens.shape, P, seed=seeds[i], k=100) for i in range(NF)]
for g in graphs:
    g.tokens.copy_(tokens)
streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]
torch.cuda.synchronize()
bad = []
for r in range(ROUNDS):
    for rep in range(int(os.environ.get("REPS", "3"))):          # several replays back to back, the last one is checked
        for i, g in enumerate(graphs):
            with torch.cuda.stream(streams[i]):
                g.replay()
    torch.cuda.synchronize()
    for i, g in enumerate(graphs):
        g.check()
        c2w, idx, val = pipe.query_batch(tokens, P, seeds[i], 100, seed_offset=g.counter)
        if not (torch.equal(c2w, g.c2w) and torch.equal(idx, g.idx) and torch.equal(val, g.val)):
            qs = [q for q in range(B) if not (torch.equal(idx[q], g.idx[q]) and torch.equal(val[q], g.val[q]) and torch.equal(c2w[q], g.c2w[q]))]
            q = qs[0]
            same_set = torch.equal(idx[q].sort().values, g.idx[q].sort().values)
            n_common = len(set(idx[q].tolist()) & set(g.idx[q].tolist()))
            bad.append({"round": r, "graph": i, "counter": int(g.counter.item()), "queries": qs, "idx_equal": bool(torch.equal(idx[q], g.idx[q])),
                        "same_set": bool(same_set), "common": n_common, "val_maxdiff": float((val[q] - g.val[q]).abs().max()),
                        "c2w_maxdiff": float((c2w[q] - g.c2w[q]).abs().max())})
            print(json.dumps(bad[-1]), flush=True)
print(json.dumps({"config": CFG, "checks": ROUNDS * NF, "mismatches": len(bad)}))
