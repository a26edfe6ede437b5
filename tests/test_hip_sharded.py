"""GPU: the ray-sharded batch-of-queries path (PosePipeline.query_sharded and its captured form).

There is one GPU on the test box, so the two-rank exchange is covered three ways: (1) one rank, no process group: the
sharded code path must reproduce the plain per-query path; (2) two ranks EMULATED on one GPU -- each rank's segments run
one after the other and the gathered messages are stacked by hand, exactly what the all_gathers deliver -- against the
unsharded result; (3) a real RCCL process group of world size 1 under the captured segments, two batches in flight.
The exchange/merge code itself also runs under world_size-2 gloo in tests/test_distributed_gloo.py.
"""
import os
import subprocess
import sys

import pytest
import torch

from iffnerf_amd import synthetic
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def pipe(dev):
    from iffnerf_amd.pipeline import PosePipeline
    return PosePipeline.from_checkpoints(util.ckpt("small"), synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))


def _tokens(dev, Q=3, M=64):
    return torch.stack([synthetic.make_tokens(M, 384, seed=7 + q) for q in range(Q)]).to(dev)


def test_one_rank_equals_the_per_query_path(pipe, dev):
    tok = _tokens(dev)
    poses, val, idx = pipe.query_sharded(tok, 75, seed=31, k=100)
    for q in range(tok.shape[0]):
        c2w, i, v = pipe.query(tok[q], 75, seed=31, k=100)
        assert torch.equal(idx[q], i) and torch.equal(val[q], v)
        assert torch.equal(poses[q], c2w)


def test_two_emulated_ranks_equal_one(pipe, dev):
    from iffnerf_amd import distributed as D
    tok = _tokens(dev)
    Q, k, P = tok.shape[0], 100, 75
    want_pose, want_val, want_idx = pipe.query_sharded(tok, P, seed=77, k=k)
    for ws in (2, 3, 8):                    # 8 ranks: shards of 9 / 10 points = 243 / 270 rays each
        seg1 = [pipe.shard_local_logits(tok, P, 77, r, ws) for r in range(ws)]
        assert sum(s[0].shape[0] for s in seg1) == P * 27
        stats_all = torch.stack([s[3] for s in seg1])
        cands = []
        for r, (ori, dirs, logits, _) in enumerate(seg1):
            lo, _ = D.shard_points(P, r, ws)
            cands.append(pipe.shard_local_candidates(logits, stats_all, ori, dirs, Q, k, lo * 27, materialize_map=False))
        poses, val, idx = pipe.shard_global_poses(torch.stack(cands), k)
        assert torch.equal(idx, want_idx), f"{ws} shards: global top-k indices differ from the unsharded run"
        torch.testing.assert_close(val, want_val, atol=1e-7, rtol=2e-5)
        torch.testing.assert_close(poses, want_pose, atol=1e-5, rtol=0)


def test_few_rays_per_rank_pads_candidates(pipe, dev):
    """A shard with fewer than k rays: its message is padded with -inf / sentinel indices that never win."""
    from iffnerf_amd import distributed as D
    tok = _tokens(dev, Q=2)
    P, k, ws = 9, 100, 3                       # 243 rays in total, 81 per rank < k
    want_pose, want_val, want_idx = pipe.query_sharded(tok, P, seed=5, k=k)
    seg1 = [pipe.shard_local_logits(tok, P, 5, r, ws) for r in range(ws)]
    stats_all = torch.stack([s[3] for s in seg1])
    cands = torch.stack([pipe.shard_local_candidates(s[2], stats_all, s[0], s[1], 2, k, D.shard_points(P, r, ws)[0] * 27, False)
                         for r, s in enumerate(seg1)])
    poses, val, idx = pipe.shard_global_poses(cands, k)
    assert torch.equal(idx, want_idx) and int(idx.max()) < P * 27
    torch.testing.assert_close(poses, want_pose, atol=1e-5, rtol=0)


def test_captured_segments_replay_like_eager(pipe, dev):
    tok = _tokens(dev)
    eager = {s: [t.clone() for t in pipe.query_sharded(tok, 75, seed=200 + s, k=100)] for s in (1, 2, 3)}
    cq = pipe.capture_query_sharded(tok.shape, 75, seed=200, k=100)
    cq.tokens.copy_(tok)
    torch.cuda.synchronize()
    for s in (1, 2, 3):
        cq.replay()
        torch.cuda.synchronize()
        assert torch.equal(cq.idx, eager[s][2]) and torch.equal(cq.val, eager[s][1]) and torch.equal(cq.poses, eager[s][0])


_RCCL_WORLD1 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from iffnerf_amd import synthetic
from iffnerf_amd.pipeline import PosePipeline
from tests import util
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
pipe = PosePipeline.from_checkpoints(util.ckpt("small"), synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))
tok = torch.stack([synthetic.make_tokens(64, 384, seed=7 + q) for q in range(3)]).to(dev)
eager = {s: [t.clone() for t in pipe.query_sharded(tok, 75, seed=300 + s, k=100)] for s in (1, 2, 3)}
cqs = [pipe.capture_query_sharded(tok.shape, 75, seed=300, k=100) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in cqs]
for c in cqs:
    c.tokens.copy_(tok)
cqs[1].counter.fill_(1)          # instance 0 plays seeds 301, 302; instance 1 plays 302, 303
torch.cuda.synchronize()
for rnd in range(2):
    for c, s in zip(cqs, streams):
        with torch.cuda.stream(s):
            c.replay()
    torch.cuda.synchronize()
    for j, c in enumerate(cqs):
        want = eager[1 + rnd + j]
        assert torch.equal(c.idx, want[2]) and torch.equal(c.poses, want[0]), (rnd, j)
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def test_captured_segments_with_a_real_rccl_group():
    """World size 1 over the nccl (= RCCL) backend: the collectives between the captured segments are real RCCL calls."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1, ROOT, str(port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_rccl_rehearsal_keeps_the_single_graph_rate():
    """The N > 1 code path of bench.py at world size 1 (--force-sharded: four captured segments, three real RCCL all_gathers per
    step, the merges as HIP launches inside the segments, steps issued skewed) against the single-graph path, same box, same run:
    what the host does between two collectives must not cost throughput.  The ratio goes to gpurun_out/rccl_rehearsal.json."""
    import json
    import socket
    rates = {}
    # (alternating, three runs of each form, best of three against best of three: a box drifts by a few per cent within a minute)
    for rep in range(3):
        for name, extra in ((f"single_graph_{rep}", []), (f"sharded_ws1_{rep}", ["--force-sharded"])):
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "120", "--warmup", "15", "--batch", "16", "--no-cpu-baseline",
                                "--no-instrument", *extra], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
            line = json.loads(r.stdout.strip().splitlines()[-1])
            rates[name] = line["value"]
            if extra:
                assert line["config"]["rccl_world_size"] == 1 and line["config"]["collective_backend"] == "nccl"
                rates["collective_us"] = line["config"]["collective_us"]
    base = max(rates[f"single_graph_{rep}"] for rep in range(3))
    rates["ratio"] = max(rates[f"sharded_ws1_{rep}"] for rep in range(3)) / base
    rates["ratio_of_medians"] = sorted(rates[f"sharded_ws1_{rep}"] for rep in range(3))[1] / sorted(rates[f"single_graph_{rep}"] for rep in range(3))[1]
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "rccl_rehearsal.json"), "w") as fh:
            json.dump(rates, fh, indent=1)
    except OSError:
        pass
    # measured 0.97-1.01 on a quiet box.  Best of three interleaved runs against best of three takes the box's drift out (two identical
    # runs have been seen 7 % apart within a minute): a host-side exchange that costs throughput shows in the BEST sharded run too
    # (a serialised step is 0.5), so the bar sits at 0.93
    assert rates["ratio"] >= 0.93, rates


def test_large_ray_sets_keep_the_invariants(dev):
    """BASELINE.json's larger configurations (32 k and 64 k candidate rays): size-independent properties of the path --
    every attention row sums to one so the scores sum to M, top-k is sorted with valid distinct indices, the pose is a
    rigid transform, and three emulated shards reproduce the unsharded top-k."""
    from iffnerf_amd import distributed as D, hip_identify as H
    from iffnerf_amd.pipeline import PosePipeline
    pipe = PosePipeline.from_checkpoints(util.ckpt("small"), synthetic.make_id_weights(seed=99), dev, model_up=(0.0, 0.0, 1.0))
    tok = _tokens(dev, Q=2, M=256)
    for P in (1186, 2400):                                   # 32 022 and 64 800 rays
        ori, dirs, rgb = pipe.emit(P, seed=9)
        assert ori.shape == (27 * P, 3) and int(pipe.last_sampler_stats[0, 3]) >= 0, "sampler timed out"
        assert torch.isfinite(rgb).all() and float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 + 1e-5
        logits, rmax, rsum = pipe.logits(tok[0], ori, dirs, rgb)
        assert torch.equal(rmax, logits.max(-1).values)
        score = H.attn_colsum(logits, rmax, rsum, write_attention=True)
        assert abs(float(score.double().sum()) - 256.0) < 2e-2
        torch.testing.assert_close(logits.sum(-1), torch.ones(256, device=dev), atol=2e-4, rtol=0)     # now the attention map
        idx, val = H.topk(score, 100)
        assert torch.equal(val, torch.sort(val, descending=True).values) and idx.unique().numel() == 100
        assert int(idx.min()) >= 0 and int(idx.max()) < 27 * P and torch.equal(score[idx], val)
        c2w = H.pose_from_topk(idx, val, ori, dirs, pipe.model_up).cpu().double()
        R = c2w[:3, :3]
        assert torch.allclose(R @ R.T, torch.eye(3, dtype=torch.float64), atol=1e-5) and torch.equal(c2w[3].float(), torch.tensor([0., 0., 0., 1.]))
        want_pose, want_val, want_idx = pipe.query_sharded(tok, P, seed=9, k=100)
        seg1 = [pipe.shard_local_logits(tok, P, 9, r, 3) for r in range(3)]
        stats_all = torch.stack([s[3] for s in seg1])
        cands = torch.stack([pipe.shard_local_candidates(s[2], stats_all, s[0], s[1], 2, 100, D.shard_points(P, r, 3)[0] * 27, False)
                             for r, s in enumerate(seg1)])
        poses, val3, idx3 = pipe.shard_global_poses(cands, 100)
        assert torch.equal(idx3, want_idx)
        torch.testing.assert_close(poses, want_pose, atol=1e-5, rtol=0)


def test_batched_cold_queries_sharded_over_emulated_ranks_equal_one_gpu(pipe, dev):
    """query_batch_sharded's four segments with G = 2, 3 and 8 ranks emulated on one GPU (the three all_gathers replaced by
    stacking the per-rank messages): the G*B cold queries are those of query_batch on one GPU -- same random streams, same
    rays, hence the same top-100 lists and poses."""
    from iffnerf_amd import distributed as D
    P, k, M, B = 75, 100, 64, 2
    for G in (2, 3, 8):                     # 8: shards of 9 / 10 points (243 / 270 rays), candidates padded from 100 per rank
        tok = torch.stack([synthetic.make_tokens(M, 384, seed=40 + q) for q in range(G * B)]).to(dev)
        want_c2w, want_idx, want_val = pipe.query_batch(tok, P, seed=1234, k=k)
        msgs = torch.stack([pipe.batch_shard_draw(tok[r * B:(r + 1) * B].contiguous(), P, 1234, r) for r in range(G)])
        seg2 = [pipe.batch_shard_local_logits(msgs, B, M, P, r, G) for r in range(G)]
        assert sum(s[0].shape[0] for s in seg2) == G * B * P * 27
        stats_all = torch.stack([s[3] for s in seg2])
        cand_all = torch.stack([pipe.batch_shard_local_candidates(s[2], stats_all, s[0], s[1], G * B, k, D.shard_points(P, r, G)[0] * 27)
                                for r, s in enumerate(seg2)])
        for r in range(G):
            poses, val, idx = pipe.batch_shard_global_poses(cand_all, k, r, B)
            assert torch.equal(idx, want_idx[r * B:(r + 1) * B]), (G, r)
            torch.testing.assert_close(val, want_val[r * B:(r + 1) * B], atol=0, rtol=1e-5)
            torch.testing.assert_close(poses, want_c2w[r * B:(r + 1) * B], atol=1e-5, rtol=0)
    # one rank, no process group: the sharded path IS the batch path
    tok = torch.stack([synthetic.make_tokens(M, 384, seed=40 + q) for q in range(3)]).to(dev)
    a = pipe.query_batch_sharded(tok, P, seed=77, k=k)
    b = pipe.query_batch(tok, P, seed=77, k=k)
    assert torch.equal(a[2], b[1]) and torch.equal(a[0], b[0])


_GLOO_RANKS = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, WS = int(sys.argv[3]), int(sys.argv[5])
B, M, P, k = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), 100
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK=str(rank), WORLD_SIZE=str(WS))
from iffnerf_amd import synthetic, distributed as D
from iffnerf_amd.pipeline import PosePipeline
from tests import util
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")          # several ranks on ONE GPU: RCCL refuses that, gloo stages the messages through the host
pipe = PosePipeline.from_checkpoints(util.ckpt("small"), synthetic.make_id_weights(seed=99), dev, model_up=(0.1, 0.2, 0.9))
tok_all = torch.stack([synthetic.make_tokens(M, 384, seed=40 + q) for q in range(WS * B)]).to(dev)
mine = tok_all[rank * B:(rank + 1) * B].contiguous()
out = {}
out["eager"] = [t.cpu() for t in pipe.query_batch_sharded(mine, P, seed=1234, k=k)]
cq = pipe.capture_query_batch_sharded(mine.shape, P, seed=1234, k=k)
cq.tokens.copy_(mine)
for rep in (1, 2):
    cq.replay(); torch.cuda.synchronize(); cq.check()
    out[f"replay{rep}"] = [cq.poses.cpu().clone(), cq.val.cpu().clone(), cq.idx.cpu().clone()]
# two slots issued skewed by one segment, the way bench.py drives them: head(i + 1) before tail(i), each slot on its own stream
cq2 = pipe.capture_query_batch_sharded(mine.shape, P, seed=1234, k=k)
cq2.tokens.copy_(mine)
slots, streams = [cq, cq2], [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
def part(i, head):
    with torch.cuda.stream(streams[i % 2]):
        slots[i % 2].replay_head() if head else slots[i % 2].replay_tail()
D.TRACE = []
part(0, True)
for i in range(4):
    if i + 1 < 4:
        part(i + 1, True)
    part(i, False)
torch.cuda.synchronize(); cq.check(); cq2.check()
out["issue_order"], D.TRACE = D.TRACE, None
out["msg_bytes"] = cq.msg.numel() * cq.msg.element_size()
out["skew_slot0"] = [cq.poses.cpu().clone(), cq.val.cpu().clone(), cq.idx.cpu().clone()]        # its 4th replay (counter 4)
out["skew_slot1"] = [cq2.poses.cpu().clone(), cq2.val.cpu().clone(), cq2.idx.cpu().clone()]     # its 2nd replay (counter 2)
# the shared-ray-set form (BASELINE configs[3]) through the same transport
out["shared"] = [t.cpu() for t in pipe.query_sharded(tok_all, P, seed=55, k=k)]
if rank == 0:
    want = {"eager": pipe.query_batch(tok_all, P, seed=1234, k=k)}
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    for rep in (1, 2):
        ctr += 1
        want[f"replay{rep}"] = pipe.query_batch(tok_all, P, seed=1234, k=k, seed_offset=ctr)
    want["skew_slot1"] = want["replay2"]
    ctr += 2
    want["skew_slot0"] = pipe.query_batch(tok_all, P, seed=1234, k=k, seed_offset=ctr)
    out["want"] = {key: [t.cpu() for t in v] for key, v in want.items()}
    out["want_shared"] = [torch.stack([pipe.query(tok_all[q], P, seed=55, k=k)[j] for q in range(WS * B)]).cpu() for j in range(3)]
torch.save(out, sys.argv[4])
dist.barrier()
dist.destroy_process_group()
print("GLOO_RANKS_OK")
"""


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("ws,B,M,P", [(2, 2, 64, 75), (4, 16, 256, 593)])
def test_real_ranks_on_one_gpu_over_gloo(tmp_path, ws, B, M, P):
    """`ws` PROCESSES (ranks of a gloo group, all on the one GPU of the box) run the sharded paths end to end -- eager, and as
    captured segments with the collectives between them: every rank's poses / top-100 are those of the one-GPU batch path.  Only
    the transport differs from the 8-GPU run (gloo through host memory instead of RCCL over xGMI).  (4, 16, 256, 593) is the bench's
    own shape per rank -- 16 cold queries of 256 tokens, 16 011 rays each, and 64 images against one ray set (BASELINE configs[3]) --
    on four ranks: the box admits six processes of ours on its GPU, this one included, so eight real ranks are not possible here.
    The skewed schedule (the head of step i + 1 before the tail of step i, bench.py) is checked where it is decided, in the HOST's
    issue order: the first all_gather of step i + 1 (points + folded queries: 4.6 MB per rank at the bench's shape) goes out before
    the two small ones of step i."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, "-c", _GLOO_RANKS, ROOT, str(port), str(r), str(tmp_path / f"r{r}.pt"), str(ws), str(B), str(M), str(P)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(ws)]
    outs = [p.communicate(timeout=1000) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "GLOO_RANKS_OK" in so, so[-2000:] + se[-4000:]
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(ws)]
    def same_lists(i, idx, val, what):
        """The rank's top-100 lists are the one-GPU batch's, up to NEAR TIES: the shards' statistics are merged rank by rank, so a
        score can differ from the one-GPU sum in its last bit, and two rays whose scores agree to 2e-5 relative may swap places (or
        the 100th and 101st may trade the last place).  Nothing else may differ (the order of such pairs is
        open in any fp32 evaluation)."""
        K = idx.shape[1]
        for q in range(idx.shape[0]):
            bad = (i[q] != idx[q]).nonzero().flatten().tolist()
            assert len(bad) <= 4, (what, q, bad)
            for p_ in bad:
                near = [abs(float(val[q, p_] - val[q, n])) <= 2e-5 * abs(float(val[q, p_])) for n in (p_ - 1, p_ + 1) if 0 <= n < K]
                assert any(near) or p_ == K - 1, (what, q, p_, val[q, max(p_ - 1, 0):p_ + 2].tolist())

    for key in ("eager", "replay1", "replay2", "skew_slot0", "skew_slot1"):
        c2w, idx, val = r[0]["want"][key]
        for rank in range(ws):
            poses, v, i = r[rank][key]
            same_lists(i, idx[rank * B:(rank + 1) * B], val[rank * B:(rank + 1) * B], (key, rank))
            torch.testing.assert_close(v, val[rank * B:(rank + 1) * B], atol=0, rtol=2e-5)
            torch.testing.assert_close(poses, c2w[rank * B:(rank + 1) * B], atol=1e-4 if ws > 2 else 1e-5, rtol=0)
    w_c2w, w_idx, w_val = r[0]["want_shared"]
    for rank in range(ws):
        poses, v, i = r[rank]["shared"]
        assert torch.equal(i, r[0]["shared"][2]) and torch.equal(poses, r[0]["shared"][0])       # every rank holds the SAME merged result
        same_lists(i, w_idx, w_val, ("shared", rank))
        torch.testing.assert_close(poses, w_c2w, atol=1e-4 if ws > 2 else 1e-5, rtol=0)
    # the host's issue order of the four skewed steps: msg(0) | msg(1) stats(0) cand(0) | msg(2) stats(1) cand(1) | msg(3) stats(2) cand(2) | stats(3) cand(3)
    for rank in range(ws):
        sizes, big = [b for b, _ in r[rank]["issue_order"]], r[rank]["msg_bytes"]
        assert len(sizes) == 12 and sizes.count(big) == 4 and all(b < big // 8 for b in sizes if b != big), sizes
        assert [i for i, b in enumerate(sizes) if b == big] == [0, 1, 4, 7], sizes
    if (B, M, P) == (16, 256, 593):
        assert 4.4e6 < r[0]["msg_bytes"] < 4.8e6


def test_merge_kernels_equal_their_torch_statements(dev):
    """iff_merge_row_stats / iff_pack_candidates / iff_merge_candidates (one launch each inside the captured segments) against the
    torch formulation in iffnerf_amd/distributed.py that the CPU gloo tests exercise: identical top-k lists, values and payloads
    -- with exact score ties across ranks (lower global ray index first), ranks that hold fewer than k rays (padding), 1 to 8
    ranks -- and row statistics to the last bits of expf."""
    from iffnerf_amd import distributed as D
    from iffnerf_amd import hip_identify as H
    gen = torch.Generator().manual_seed(31)
    for G, Q, k, n_local in ((1, 3, 100, 400), (2, 5, 100, 237), (3, 4, 100, 60), (8, 6, 100, 75), (8, 2, 7, 3)):
        R = Q * 37
        stats = torch.stack((torch.randn(G, R, generator=gen) * 20, torch.rand(G, R, generator=gen) * 500 + 1), dim=-1).to(dev)
        stats[0, :5, 0] = stats[-1, :5, 0]                                      # equal row maxima on two ranks
        gm, gs = H.merge_row_stats(stats)
        gm_ref, gs_ref = D.merge_row_stats_gathered(stats)
        assert torch.equal(gm, gm_ref)
        torch.testing.assert_close(gs, gs_ref, rtol=2e-6, atol=0.0)
        # per rank: scores with planted exact ties across ranks, local top-kl through the product's own iff_topk_batched
        msgs, refs = [], []
        tie_vals = torch.rand(Q, 4, generator=gen) + 2.0
        for g in range(G):
            score = torch.rand(Q, n_local, generator=gen)
            if n_local >= 4:
                score[:, :4] = tie_vals                                         # every rank holds the same four top values
            score = score.to(dev)
            ori, dirs = torch.randn(n_local, 3, generator=gen).to(dev), torch.randn(n_local, 3, generator=gen).to(dev)
            kl = min(k, n_local)
            i, v = H.topk_batched(score, kl)
            msgs.append(H.pack_candidates(i, v, ori, dirs, k, g * n_local))
            lval = torch.full((Q, k), float("-inf"), device=dev)
            lidx = torch.full((Q, k), 2 ** 31 - 1, dtype=torch.int64, device=dev)
            pay = torch.zeros(Q, k, 6, device=dev)
            lval[:, :kl], lidx[:, :kl] = v, i + g * n_local
            pay[:, :kl, :3], pay[:, :kl, 3:] = ori[i], dirs[i]
            refs.append(D.pack_candidates(lval, lidx, pay))
            assert torch.equal(msgs[-1].view(torch.int32), refs[-1].view(torch.int32))
            # per-query ray sets ([Q, n, 3]) pack the same way
            m2 = H.pack_candidates(i, v, ori[None].expand(Q, -1, -1).contiguous(), dirs[None].expand(Q, -1, -1).contiguous(), k, g * n_local)
            assert torch.equal(m2.view(torch.int32), msgs[-1].view(torch.int32))
        cand = torch.stack(msgs)
        val, idx, wo, wd = H.merge_candidates(cand, k)
        v_ref, i_ref, p_ref = D.merge_topk_gathered(*D.unpack_candidates(cand), k)
        assert torch.equal(val, v_ref) and torch.equal(idx, i_ref)
        assert torch.equal(wo, p_ref[..., :3]) and torch.equal(wd, p_ref[..., 3:])
        if G > 1 and n_local >= 4:
            assert bool((idx[:, :G] == (torch.arange(G, device=dev) * n_local + idx[:, 0:1] % n_local)).all())     # a tie: ranks in order
        # a window of the queries (what a rank of the cold-batch path solves)
        if Q >= 3:
            v1, i1, o1, d1 = H.merge_candidates(cand, k, 1, 2)
            assert torch.equal(v1, val[1:3]) and torch.equal(i1, idx[1:3]) and torch.equal(o1, wo[1:3]) and torch.equal(d1, wd[1:3])
