"""Ray sharding across GPUs: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The reference is single-process.  Rays are independent everywhere except in the softmax over
the ray axis and the global top-k, so each rank keeps a contiguous block of the surface points (hence of 27-ray fans)
and the path needs exactly two small exchanges per batch of queries:

  1. per-row softmax statistics (max, sum-exp): ``all_gather`` of [Q*M, 2] floats per rank, merged in rank order
     (deterministic) -- after it every rank normalises its own logits columns with the GLOBAL statistics, so its
     column-sum scores are exactly the scores the single-GPU path gives those rays;
  2. per-query local top-k (value, global ray index, origin, direction): ``all_gather`` of [Q, k, 8] floats per rank,
     reduced to the global top-k with the single-GPU tie rule (higher value first, lower index on ties).

Batches of COLD queries (every query with its own freshly drawn ray set, ``PosePipeline.query_batch_sharded``) add one
exchange in front: each rank draws the surface points of its own B queries (the sampler is globally coupled per query) and
folds its own B token blocks, and one ``all_gather`` of [B*P*3 + B*M*272] floats per rank (4.5 MB at B = 16) hands every
rank the points and folded queries of all G*B queries, of which it then emits, marches and encodes its block.

The messages are a few KB to a few MB: latency-bound, no bandwidth tuning needed.  This module holds the exchange itself
(``all_gather_into`` over RCCL) and, as plain torch ops on small tensors, the STATEMENT of the merge arithmetic
(``merge_row_stats_gathered``, ``merge_topk_gathered``, ``pack_candidates``): device-agnostic, so the world_size-2 ``gloo``
tests on CPU check the exchange logic with the oracle as the local work.  On the GPU the pipeline runs each merge as ONE HIP
launch inside its captured segments (``iff_merge_row_stats``, ``iff_pack_candidates``, ``iff_merge_candidates``:
csrc/shard_kernels.hip, hip_identify.merge_* -- tests/test_hip_sharded.py holds them equal to the statements below), so between
two collectives nothing runs on the host but the collective's own issue.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_points(n_points: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of surface points owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(n_points, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _all_gather_stack(t: torch.Tensor, group=None) -> torch.Tensor:
    rank, ws = world(group)
    if ws == 1:
        return t[None]
    out = t.new_empty((ws,) + tuple(t.shape))
    all_gather_into(out, t, group)
    return out


TRACE = None     # tests set a list here: every all_gather_into then appends (bytes per rank, shape) in HOST ISSUE ORDER (the skewed schedule's check)


def all_gather_into(out: torch.Tensor, src: torch.Tensor, group=None) -> None:
    """``out`` [world, *src.shape] <- every rank's ``src``, in rank order.  With the ``nccl`` (= RCCL) backend this is one
    ``all_gather_into_tensor`` on device memory.  With ``gloo`` and device tensors (rehearsing several ranks on ONE GPU,
    where RCCL refuses two ranks per device) the message is staged through host memory; without a process group it is a copy."""
    if TRACE is not None:
        TRACE.append((src.numel() * src.element_size(), tuple(src.shape)))
    if not (dist.is_available() and dist.is_initialized()):
        out.copy_(src[None])
        return
    flat = out.view((-1,) + tuple(src.shape[1:])) if src.dim() > 0 else out.view(-1)
    if src.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(flat.shape, dtype=flat.dtype)
        dist.all_gather_into_tensor(host, src.detach().cpu().contiguous(), group=group)
        flat.copy_(host)
        return
    dist.all_gather_into_tensor(flat, src.contiguous(), group=group)


def merge_row_stats(row_max: torch.Tensor, row_sumexp: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Local (max_i, sum_j exp(l_ij - max_i)) over this rank's columns -> the same statistics over all columns."""
    return merge_row_stats_gathered(_all_gather_stack(torch.stack((row_max, row_sumexp), dim=-1), group))


def merge_row_stats_gathered(both: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``both`` [G, R, 2]: every rank's (max, sum-exp) per row, in rank order -> global (max [R], sum-exp [R])."""
    gmax = both[..., 0].max(dim=0).values
    gsum = torch.zeros_like(gmax)
    for r in range(both.shape[0]):                                                    # fixed rank order: reproducible
        gsum = gsum + both[r, :, 1] * torch.exp(both[r, :, 0] - gmax)
    return gmax, gsum


def merge_topk(local_val: torch.Tensor, local_idx: torch.Tensor, payload: torch.Tensor, k: int, group=None):
    """Per-rank candidates -> global top-k.

    local_val [Q,kl] scores (pad with -inf), local_idx [Q,kl] GLOBAL ray indices (int64), payload [Q,kl,C] per-ray data
    that must travel with the winners (origin, direction).  Returns (val [Q,k], idx [Q,k], payload [Q,k,C]) ordered by
    value descending, lower global index first on ties -- the order ``iff_topk`` produces on one GPU.
    """
    return merge_topk_gathered(_all_gather_stack(local_val, group), _all_gather_stack(local_idx, group),
                               _all_gather_stack(payload, group), k)


def merge_topk_gathered(vals: torch.Tensor, idxs: torch.Tensor, pays: torch.Tensor, k: int):
    """vals, idxs [G,Q,kl], pays [G,Q,kl,C] (every rank's candidates in rank order) -> global top-k as ``merge_topk``."""
    G, Q, kl = vals.shape
    vals = vals.permute(1, 0, 2).reshape(Q, G * kl)
    idxs = idxs.permute(1, 0, 2).reshape(Q, G * kl)
    pays = pays.permute(1, 0, 2, 3).reshape(Q, G * kl, -1)
    by_idx = torch.argsort(idxs, dim=1, stable=True)               # secondary key first ...
    v1 = torch.gather(vals, 1, by_idx)
    by_val = torch.argsort(v1, dim=1, descending=True, stable=True)    # ... then the stable primary sort
    order = torch.gather(by_idx, 1, by_val)[:, :k]
    return (torch.gather(vals, 1, order), torch.gather(idxs, 1, order),
            torch.gather(pays, 1, order[..., None].expand(-1, -1, pays.shape[-1])))


def gather_scores(local_score: torch.Tensor, counts, group=None) -> torch.Tensor:
    """Optional: the full score vector [Q, N_total] on every rank (what the reference returns as ``scores``)."""
    rank, ws = world(group)
    if ws == 1:
        return local_score
    width = max(counts)
    padded = local_score.new_zeros(local_score.shape[0], width)
    padded[:, :local_score.shape[1]] = local_score
    allp = _all_gather_stack(padded, group)
    return torch.cat([allp[r, :, :counts[r]] for r in range(ws)], dim=1)


def pack_candidates(val: torch.Tensor, idx: torch.Tensor, payload: torch.Tensor) -> torch.Tensor:
    """(val [Q,k] f32, idx [Q,k] int64 < 2^31, payload [Q,k,C] f32) -> one f32 message [Q,k,2+C] (index bits in slot 1)."""
    bits = idx.to(torch.int32).contiguous().view(torch.float32)
    return torch.cat((val[..., None], bits[..., None], payload), dim=-1).contiguous()


def unpack_candidates(msg: torch.Tensor):
    """Inverse of ``pack_candidates`` on a gathered message [..., k, 2+C] -> (val, idx int64, payload)."""
    val = msg[..., 0].contiguous()
    idx = msg[..., 1].contiguous().view(torch.int32).to(torch.int64)
    return val, idx, msg[..., 2:].contiguous()
