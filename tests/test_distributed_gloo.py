"""CPU-only, world_size 2 over gloo: the ray-sharded exchange of iffnerf_amd/distributed.py.

Each rank takes a contiguous block of surface points (27-ray fans), computes its local logits / statistics with the
oracle (the GPU ranks use the HIP kernels for that part), and runs the product's exchange + merge code.  The merged
result must equal the single-process oracle on the full ray set: identical top-k indices, scores within fp32 rounding.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iffnerf_amd import distributed as D
from iffnerf_amd import synthetic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    g = torch.Generator().manual_seed(11)
    P = 75                                   # 2025 rays; not divisible by 2 -> ragged shards
    ori = (torch.randn(P, 3, generator=g) * 0.4).repeat_interleave(27, dim=0)
    dirs = torch.nn.functional.normalize(torch.randn(P * 27, 3, generator=g), dim=-1)
    rgb = torch.rand(P * 27, 3, generator=g)
    tokens = torch.stack([synthetic.make_tokens(64, 384, seed=s) for s in (7, 8, 9)])      # Q = 3 queries, M = 64
    return P, ori, dirs, rgb, tokens, synthetic.make_id_weights(seed=99)


def _worker(rank, world_size, port, out_dir):
    from oracle import identify as oid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        torch.set_num_threads(2)
        P, ori, dirs, rgb, tokens, w = _inputs()
        Q, M, _ = tokens.shape
        k = 100
        lo, hi = D.shard_points(P, rank, world_size)
        sl = slice(lo * 27, hi * 27)
        o, d, c = ori[sl], dirs[sl], rgb[sl]
        _, logits, _, _ = oid.attention_map(w, tokens.reshape(Q * M, -1), oid.ray_encode(w, o, d, c), return_parts=True)
        rmax = logits.max(-1).values
        rsum = torch.exp(logits - rmax[:, None]).sum(-1)
        gmax, gsum = D.merge_row_stats(rmax, rsum)
        att = torch.exp(logits - gmax[:, None]) / gsum[:, None]
        lval = torch.full((Q, k), float("-inf"))
        lidx = torch.full((Q, k), 2 ** 62, dtype=torch.int64)
        pay = torch.zeros(Q, k, 6)
        scores = []
        for q in range(Q):
            score = att[q * M:(q + 1) * M].sum(0)
            scores.append(score)
            v, i = torch.topk(score, min(k, score.shape[0]))
            lval[q, :len(v)], lidx[q, :len(v)] = v, i + lo * 27
            pay[q, :len(v), :3], pay[q, :len(v), 3:] = o[i], d[i]
        val, idx, pay_m = D.merge_topk(lval, lidx, pay, k)
        # the packed single-message form the captured segments use (pipeline.CapturedShardedQuery.replay): statistics
        # and candidates travel through all_gather_into_tensor, indices as int32 bits inside the f32 message
        stats_local = torch.stack((rmax, rsum), dim=-1).contiguous()
        stats_all = torch.empty((world_size,) + tuple(stats_local.shape))
        dist.all_gather_into_tensor(stats_all.view((-1,) + tuple(stats_local.shape[1:])), stats_local)
        gmax2, gsum2 = D.merge_row_stats_gathered(stats_all)
        assert torch.equal(gmax2, gmax) and torch.equal(gsum2, gsum)
        lidx32 = torch.where(lidx >= 2 ** 31, torch.full_like(lidx, 2 ** 31 - 1), lidx)      # pipeline's sentinel for empty slots
        cand = D.pack_candidates(lval, lidx32, pay)
        cand_all = torch.empty((world_size,) + tuple(cand.shape))
        dist.all_gather_into_tensor(cand_all.view((-1,) + tuple(cand.shape[1:])), cand)
        v2, i2, p2 = D.unpack_candidates(cand_all)
        val2, idx2, pay2 = D.merge_topk_gathered(v2, i2, p2, k)
        assert torch.equal(val2, val) and torch.equal(idx2, idx) and torch.equal(pay2, pay_m)
        pay = pay_m
        counts = [(D.shard_points(P, r, world_size)[1] - D.shard_points(P, r, world_size)[0]) * 27 for r in range(world_size)]
        full = D.gather_scores(torch.stack(scores), counts)
        torch.save({"val": val, "idx": idx, "pay": pay, "full": full}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_exchange_equals_single_process(tmp_path):
    from oracle import identify as oid
    world_size = 2
    port = _free_port()
    mp.spawn(_worker, args=(world_size, port, str(tmp_path)), nprocs=world_size, join=True)
    P, ori, dirs, rgb, tokens, w = _inputs()
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for key in ("val", "idx", "pay", "full"):
        assert torch.equal(r0[key], r1[key]), f"ranks disagree on {key}"
    for q in range(tokens.shape[0]):
        idx, val, score, _ = oid.test_image(w, tokens[q], ori, dirs, rgb, 100)
        torch.testing.assert_close(r0["full"][q], score, rtol=2e-4, atol=1e-9)
        assert r0["idx"][q].tolist() == idx.tolist(), "global top-100 must equal the single-process top-100"
        torch.testing.assert_close(r0["val"][q], val, rtol=2e-4, atol=1e-9)
        assert torch.equal(r0["pay"][q][:, :3], ori[idx]) and torch.equal(r0["pay"][q][:, 3:], dirs[idx])


def test_shard_points_partition():
    for n, ws in ((593, 8), (75, 2), (5, 8), (20000, 3)):
        blocks = [D.shard_points(n, r, ws) for r in range(ws)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_paths_are_identity():
    v = torch.rand(3, 5)
    i = torch.arange(15).reshape(3, 5)
    pay = torch.rand(3, 5, 6)
    val, idx, p = D.merge_topk(v, i, pay, 4)
    tv, ti = torch.topk(v, 4)
    assert torch.equal(val, tv) and torch.equal(idx, torch.gather(i, 1, ti))
    m, s = D.merge_row_stats(torch.tensor([1.0, 2.0]), torch.tensor([3.0, 4.0]))
    assert torch.equal(m, torch.tensor([1.0, 2.0])) and torch.equal(s, torch.tensor([3.0, 4.0]))
