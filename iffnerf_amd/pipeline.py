"""The per-query hot path as one object: emission (A+B) and identification + pose (C) over the C ABI.

``PosePipeline`` is what ``bench.py``, ``__graft_entry__.smoke()`` and the multi-GPU path drive.  It is the same
sequence the mirrored reference API runs (``explore_model`` -> ``IdentificationModule.test_image`` ->
``test_pose_estimation`` body), without the nn.Module wrappers: handles in, device tensors out, no host
synchronisation anywhere in ``emit`` / ``identify`` / ``query``.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import hip_identify as H
from .hip_field import FieldHandle, field_handle_from_ckpt, isocell_emit
from .pose_estimation.isocell import isocell_distribution


def jitter_scale_from_kwargs(kw: dict, has_mask: bool = True) -> float:
    """rho of reference pose_estimation/sampling.py:518-523 from checkpoint kwargs: 10 % of the grid in voxel units when
    the model carries an occupancy mask, the aabb diagonal otherwise."""
    g = torch.tensor([int(v) for v in kw["gridSize"]], dtype=torch.long)
    aabb = torch.as_tensor(kw["aabb"]).float().cpu()
    if not has_mask:
        return float(torch.linalg.norm(aabb[1] - aabb[0]))
    return float((torch.max(g) * 0.1) * torch.max((aabb[1] - aabb[0]) / g))


def check_sampler_stats(stats: torch.Tensor) -> None:
    """Raise RuntimeError when a surface-sampler run reported an in-kernel barrier timeout (``stats[..., 0, 3] == -1``,
    include/iffnerf_hip.h iff_surface_sample): its samples are unconverged and every pose derived from them is invalid.
    One device->host read; call it outside timed / captured regions."""
    s = stats.reshape(-1, stats.shape[-2], 4)
    bad = (s[:, 0, 3] == -1).nonzero().flatten().tolist()
    if bad:
        raise RuntimeError(f"surface sampler: in-kernel grid barrier timed out in run(s) {bad} "
                           "(more sampler workgroups in flight than the device holds, or the device is shared)")


class PosePipeline:
    def __init__(self, field: FieldHandle, idnet: H.IdNetHandle, rho: float, model_up=(0.0, 0.0, 1.0), fold_heads: bool = True):
        self.field, self.idnet, self.rho = field, idnet, float(rho)
        self.fold_heads = bool(fold_heads)   # logits through iff_attn_logits_folded (include/iffnerf_hip.h) or the 5-GEMM chain
        self.model_up = torch.as_tensor(model_up, dtype=torch.float32).cpu()
        self.cells = isocell_distribution(27, torch.float32, "cpu")
        self.device = field.device

    @classmethod
    def from_checkpoints(cls, field_ckpt: dict, id_weights: Dict[str, torch.Tensor], device, model_up=(0.0, 0.0, 1.0),
                         fold_heads: bool = True, gemm_mode: Optional[int] = None, trunk_variant: int = 0, fan_waves: int = 0):
        if gemm_mode is None:
            # IFF_GEMM_F16X2 plans its power-of-two scales for ray origins within 64 scene units (include/iffnerf_hip.h); rays start
            # on the surface inside the scene box, so a larger box keeps the range-free 3xBF16 arithmetic
            extent = float(torch.as_tensor(field_ckpt["kwargs"]["aabb"]).abs().max())
            gemm_mode = H.GEMM_DEFAULT if extent <= H.F16_ORIGIN_BOUND else H.GEMM_BF16X3
        return cls(field_handle_from_ckpt(field_ckpt, device, fan_waves=fan_waves), H.IdNetHandle(id_weights, device, gemm_mode, trunk_variant),
                   jitter_scale_from_kwargs(field_ckpt["kwargs"], "alphaMask.aabb" in field_ckpt), model_up, fold_heads)

    def save_tables(self, field_path: str, idnet_path: str) -> None:
        """Both handles as pre-laid-out table files; ``from_table_files`` is the matching loader."""
        self.field.save(field_path)
        self.idnet.save(idnet_path)

    @classmethod
    def from_table_files(cls, field_path: str, idnet_path: str, device, rho: float, model_up=(0.0, 0.0, 1.0)):
        """Serving-side constructor: no ``.th`` checkpoint, no re-layout.  ``rho`` is the sampler's jitter scale
        (``jitter_scale_from_kwargs``: a property of the checkpoint's grid, not stored in the tables)."""
        return cls(FieldHandle.from_file(field_path, device), H.IdNetHandle.from_file(idnet_path, device), rho, model_up)

    def logits(self, tokens, ori, dirs, rgb):
        """tokens [M, C+14] x rays -> (logits [M,N], row_max [M], row_sumexp [M])."""
        if self.fold_heads:
            return self.idnet.ray_logits_folded(self.idnet.q_fold(tokens), ori, dirs, rgb)
        _, k = self.idnet.ray_encode(ori, dirs, rgb, want_features=False, want_k=True)
        return H.attn_logits(self.idnet.q_proj(tokens), k, gemm_mode=min(self.idnet.gemm_mode, 1))

    # ------------------------------------------------------------------ stage A + B  (explore_model)
    def emit(self, gen_points: int, seed: int, point_range: Optional[Tuple[int, int]] = None, seed_offset=None):
        """-> (ori [27P,3], dirs [27P,3], rgb [27P,3]).  ``point_range`` keeps a contiguous block of the surface points
        (ray sharding across GPUs: every rank draws the same samples from the same seed and colours only its block)."""
        return self.emit_from_samples(self.sample_surface(gen_points, seed, seed_offset), point_range)

    def sample_surface(self, gen_points: int, seed: int, seed_offset=None):
        """Stage A, first half: the iterative surface sampler (sampling.py:509-532) -> samples [P,3]."""
        samples, _, stats = self.field.surface_sample(gen_points, self.rho, n_epochs=4, max_iterations=200, seed=seed,
                                                      seed_offset=seed_offset)
        self.last_sampler_stats = stats
        return samples

    def emit_from_samples(self, samples, point_range: Optional[Tuple[int, int]] = None):
        """Normals, 27-ray fans and their colours (sampling.py:535-541, :442-488) for given surface points."""
        if point_range is not None:
            samples = samples[point_range[0]:point_range[1]].contiguous()
        normals = self.field.point_normals(samples)
        ori, dirs, rays = isocell_emit(self.cells, samples, normals, want_rays6=True)
        rgb = self.field.march(rays, 0, 20, want_alpha=False)[0]
        return ori, dirs, rgb

    # ------------------------------------------------------------------ stage C  (test_image + pose solve)
    def scores(self, tokens, ori, dirs, rgb, materialize_map: bool = True):
        logits, rmax, rsum = self.logits(tokens, ori, dirs, rgb)
        score = H.attn_colsum(logits, rmax, rsum, write_attention=materialize_map)
        return score, logits

    def identify(self, tokens, ori, dirs, rgb, k: int = 100, materialize_map: bool = True):
        score, _ = self.scores(tokens, ori, dirs, rgb, materialize_map)
        idx, val = H.topk(score, k)
        return H.pose_from_topk(idx, val, ori, dirs, self.model_up), idx, val

    def identify_batch(self, tokens, ori, dirs, rgb, k: int = 100):
        """Warm path for a batch of query images against ONE resident ray set (the reference's eval loop,
        train_eval_pose_est.py:131-149 + pose_estimation/test.py:67-91: rays emitted once per model, then every image):
        tokens [Q,M,C+14] -> (c2w [Q,4,4], idx [Q,k], val [Q,k]).  Query q equals ``identify(tokens[q], ...)``."""
        Q, M, C = tokens.shape
        logits, rmax, rsum = self.logits(tokens.reshape(Q * M, C), ori, dirs, rgb)
        score = H.attn_colsum_batched(logits, rmax, rsum, Q, write_attention=False)
        idx, val = H.topk_batched(score, k)
        return H.pose_from_topk_batched(idx, val, ori, dirs, self.model_up), idx, val

    # ------------------------------------------------------------------ resident rays: the encoder cached per model
    def make_resident(self, ori, dirs, rgb) -> "ResidentRays":
        """Rays of a model kept for many query images (the reference's eval loop emits once per model,
        train_eval_pose_est.py:131-149): the ray encoder runs here, once; ``identify_resident`` then serves any number of
        query batches without it.  Re-emission or new weights need a new ResidentRays."""
        if not self.fold_heads:
            raise RuntimeError("resident rays use the folded path (fold_heads=True)")
        return ResidentRays(ori, dirs, rgb, self.idnet.build_ray_cache(ori, dirs, rgb))

    def identify_resident(self, tokens, rays: "ResidentRays", k: int = 100):
        """tokens [Q,M,C+14] against resident rays -> (c2w [Q,4,4], idx [Q,k], val [Q,k]); query q equals
        ``identify(tokens[q], rays.ori, rays.dirs, rays.rgb)`` (bit for bit under F16X2)."""
        Q, M, C = tokens.shape
        qf = self.idnet.q_fold(tokens.reshape(Q * M, C))
        logits, rmax, rsum = self.idnet.logits_from_cache(qf, rays.cache, rays.ori.shape[0])
        score = H.attn_colsum_batched(logits, rmax, rsum, Q, write_attention=False)
        idx, val = H.topk_batched(score, k)
        return H.pose_from_topk_batched(idx, val, rays.ori, rays.dirs, self.model_up), idx, val

    def identify_images_resident(self, frontend, imgs, masks, rays: "ResidentRays", k: int = 100):
        """Image in -> pose out against resident rays: imgs [Q,H,W,3], masks [Q,H,W] -> (c2w [Q,4,4], idx, val).  Static shapes
        throughout (the mask select of identification_module.py:157-160 as kept rows first + a count per image on the device,
        image_frontend.py), so the whole call captures as ONE hipGraph (CapturedImageQuery).  The launches of
        ``IdentificationModule.test_image`` + the pose solve per image."""
        from .image_frontend import mask_token_rows
        tokens, keep, rows = frontend.tokens(imgs, masks, compact=True)
        Q, M, C = tokens.shape
        qf = self.idnet.q_fold(tokens.reshape(Q * M, C))
        if M == 256:
            logits, rmax, rsum = self.idnet.logits_from_cache(qf, rays.cache, rays.ori.shape[0], rows=rows)
            score = H.attn_colsum_batched(logits, rmax, rsum, Q, write_attention=False, rows=rows)
        else:
            logits, rmax, rsum = self.idnet.logits_from_cache(qf, rays.cache, rays.ori.shape[0])
            mask_token_rows(keep, rmax, rsum)
            score = H.attn_colsum_batched(logits, rmax, rsum, Q, write_attention=False)
        idx, val = H.topk_batched(score, k)
        return H.pose_from_topk_batched(idx, val, rays.ori, rays.dirs, self.model_up), idx, val

    def query(self, tokens, gen_points: int, seed: int, k: int = 100, seed_offset=None, materialize_map: bool = False):
        """Cold per-query path: emission + identification + pose -> (c2w, top-k idx, top-k val).  The attention map is not
        part of the result, so by default it is not written back (scores come straight from logits + row statistics)."""
        ori, dirs, rgb = self.emit(gen_points, seed, seed_offset=seed_offset)
        return self.identify(tokens, ori, dirs, rgb, k, materialize_map)

    # ------------------------------------------------------------------ a batch of cold queries in one set of launches
    def query_batch(self, tokens, gen_points: int, seed: int, k: int = 100, seed_offset=None, keep=None):
        """``tokens`` [B,M,C+14]: B cold queries, each with its OWN freshly drawn ray set, served by one launch per stage
        (batched sampler, one march over B*27P rays, one encoder/logits launch with grid.y = query, batched score / top-k
        / pose).  Query b equals ``query(tokens[b], gen_points, seed + b * SAMPLER_SEED_STRIDE)`` bit for bit.
        ``keep`` [B,M] (``image_frontend.token_assemble``): the mask select of identification_module.py:157-160 applied to the
        softmax rows.  -> (c2w [B,4,4], idx [B,k], val [B,k])."""
        B, M, C = tokens.shape
        if not self.fold_heads:
            raise RuntimeError("query_batch runs the folded path (fold_heads=True)")
        samples, _, stats = self.field.surface_sample_batched(B, gen_points, self.rho, n_epochs=4, max_iterations=200, seed=seed,
                                                              seed_offset=seed_offset)
        self.last_sampler_stats = stats
        ori, dirs, rgb = self.emit_from_samples(samples.reshape(B * gen_points, 3))          # query-major: [B * 27P, 3]
        qf = self.idnet.q_fold(tokens.reshape(B * M, C))
        logits, rmax, rsum = self.idnet.ray_logits_folded_batched(qf, ori, dirs, rgb, B)
        if keep is not None:
            from .image_frontend import mask_token_rows
            mask_token_rows(keep, rmax, rsum)
        score = H.attn_colsum_batched(logits, rmax, rsum, B, write_attention=False)
        idx, val = H.topk_batched(score, k)
        n = ori.shape[0] // B
        c2w = H.pose_from_topk_batched(idx, val, ori.view(B, n, 3), dirs.view(B, n, 3), self.model_up)
        return c2w, idx, val

    def max_steps_in_flight(self, gen_points: int, batch: int = 1) -> int:
        """How many query graphs (each running ``batch`` samplers) may be in flight at once.  The default sampler (a chain of short
        launches) sets no limit; the persistent form's workgroups (a handle made with sampler_persistent=True) meet at in-kernel barriers, so the
        samplers of ALL in-flight graphs must then fit on the device together (iff_surface_sample_residency)."""
        wgs, capacity = self.field.sampler_residency(gen_points, batch)
        return max(1, capacity // max(1, wgs * batch))

    def capture_query_batch(self, tokens_shape, gen_points: int, seed: int = 0, k: int = 100) -> "CapturedBatchQuery":
        return CapturedBatchQuery(self, tokens_shape, gen_points, seed, k)

    def capture_query(self, tokens_shape, gen_points: int, seed: int = 0, k: int = 100) -> "CapturedQuery":
        return CapturedQuery(self, tokens_shape, gen_points, seed, k)

    # ------------------------------------------------------------------ ray-sharded batch of queries (multi-GPU)
    # Three local segments with one small exchange between each pair (distributed.py): the eager ``query_sharded`` and the
    # captured ``CapturedShardedQuery`` run the same three functions.
    def shard_local_logits(self, tokens, gen_points: int, seed: int, rank: int, ws: int, seed_offset=None):
        """Segment 1: this rank's block of the ray set and its logits columns.
        -> (ori, dirs [n_local,3], logits [Q*M, n_local], stats [Q*M, 2] = local (row max, row sum-exp))."""
        from . import distributed as D
        Q, M, C = tokens.shape
        lo, hi = D.shard_points(gen_points, rank, ws)
        ori, dirs, rgb = self.emit(gen_points, seed, (lo, hi) if ws > 1 else None, seed_offset=seed_offset)
        logits, rmax, rsum = self.logits(tokens.reshape(Q * M, C), ori, dirs, rgb)
        return ori, dirs, logits, torch.stack((rmax, rsum), dim=-1)

    def shard_local_candidates(self, logits, stats_all, ori, dirs, Q: int, k: int, first_ray: int, materialize_map: bool = False):
        """Segment 2: global statistics -> this rank's score columns -> its top-k candidates per query, packed as one
        message [Q, k, 8] = (score, global ray index bits, origin, direction); unfilled slots hold -inf / 2^31-1."""
        gmax, gsum = H.merge_row_stats(stats_all)                        # one launch (iff_merge_row_stats)
        score = H.attn_colsum_batched(logits, gmax, gsum, Q, write_attention=materialize_map)
        i, v = H.topk_batched(score, min(k, ori.shape[0]))
        return H.pack_candidates(i, v, ori, dirs, k, first_ray)          # one launch (iff_pack_candidates)

    def shard_global_poses(self, cand_all, k: int, q0: int = 0, n_queries=None):
        """Segment 3: every rank's candidates [G, Qt, k, 8] -> (poses [Q,4,4], val [Q,k], global ray idx [Q,k]) of the queries
        q0 .. q0 + Q - 1 (default: all Qt)."""
        val, idx, wo, wd = H.merge_candidates(cand_all, k, q0, n_queries)        # one launch (iff_merge_candidates)
        Q = val.shape[0]
        # the winners arrive gathered: candidate j of a query IS row j.  One index tensor per shape, kept for the pipeline's lifetime
        # (captured graphs hold its address)
        cache = self.__dict__.setdefault("_arange_cache", {})
        key = (Q, k, str(val.device))
        if key not in cache:
            cache[key] = torch.arange(k, device=val.device).expand(Q, k).contiguous()
        poses = H.pose_from_topk_batched(cache[key], val, wo, wd, self.model_up)
        return poses, val, idx

    def query_sharded(self, tokens, gen_points: int, seed: int, k: int = 100, group=None, materialize_map: bool = False):
        """``tokens`` [Q,M,C+14]: Q query images against ONE emitted ray set whose surface points are sharded over the
        ranks of ``group``.  Every rank returns the Q poses [Q,4,4] (identical on all ranks) and the global top-k
        (values [Q,k], global ray indices [Q,k]).  With one rank this is ``query`` for each image."""
        from . import distributed as D
        rank, ws = D.world(group)
        Q = tokens.shape[0]
        lo, _ = D.shard_points(gen_points, rank, ws)
        ori, dirs, logits, stats = self.shard_local_logits(tokens, gen_points, seed, rank, ws)
        cand = self.shard_local_candidates(logits, D._all_gather_stack(stats, group), ori, dirs, Q, k, lo * 27, materialize_map)
        return self.shard_global_poses(D._all_gather_stack(cand, group), k)

    def capture_query_sharded(self, tokens_shape, gen_points: int, seed: int = 0, k: int = 100, group=None) -> "CapturedShardedQuery":
        return CapturedShardedQuery(self, tokens_shape, gen_points, seed, k, group)

    # ------------------------------------------------------------------ ray-sharded batches of COLD queries (multi-GPU)
    # Every rank owns B query images per step; EVERY query's freshly drawn ray set is sharded over all G ranks (contiguous
    # blocks of its surface points), so a step serves G*B cold queries and per-rank work does not depend on G.  Four local
    # segments with three small all_gathers between them.  Global query g = rank * B + b draws with
    # seed + g * SAMPLER_SEED_STRIDE: the G*B queries are exactly those of ``query_batch`` with G*B token blocks on one GPU.
    def batch_shard_draw(self, tokens_local, gen_points: int, seed: int, rank: int, seed_offset=None):
        """Segment 1: this rank's B surface-sampler runs (the sampler is globally coupled per query -- quantile,
        candidate budget -- so one rank draws all P points of a query) and its B folded query blocks, packed as ONE
        message [B*P*3 + B*M*qw] for the first all_gather."""
        from .hip_field import SAMPLER_SEED_STRIDE
        B, M, C = tokens_local.shape
        s = (int(seed) + rank * B * SAMPLER_SEED_STRIDE) % (1 << 64)
        samples, _, stats = self.field.surface_sample_batched(B, gen_points, self.rho, n_epochs=4, max_iterations=200, seed=s,
                                                              seed_offset=seed_offset)
        self.last_sampler_stats = stats
        qf = self.idnet.q_fold(tokens_local.reshape(B * M, C))
        return torch.cat((samples.reshape(-1), qf.reshape(-1)))

    def batch_shard_local_logits(self, msg_all, B: int, M: int, gen_points: int, rank: int, ws: int):
        """Segment 2: ``msg_all`` [G, L] (every rank's segment-1 message) -> this rank's block of EVERY query's ray set and
        its logits columns: (ori, dirs [G*B*n_local, 3] query-major, logits [G*B*M, n_local], stats [G*B*M, 2])."""
        from . import distributed as D
        G, P = msg_all.shape[0], gen_points
        ns = B * P * 3
        samples = msg_all[:, :ns].reshape(G * B, P, 3)
        qf = msg_all[:, ns:].reshape(G * B * M, -1).contiguous()
        lo, hi = D.shard_points(P, rank, ws)
        ori, dirs, rgb = self.emit_from_samples(samples[:, lo:hi].reshape(-1, 3).contiguous())
        logits, rmax, rsum = self.idnet.ray_logits_folded_batched(qf, ori, dirs, rgb, G * B)
        return ori, dirs, logits, torch.stack((rmax, rsum), dim=-1)

    def batch_shard_local_candidates(self, logits, stats_all, ori, dirs, QT: int, k: int, first_ray: int):
        """Segment 3: global statistics -> this rank's score columns -> its top-k candidates of every query [QT, k, 8]
        (as ``shard_local_candidates``, with one ray set per query)."""
        gmax, gsum = H.merge_row_stats(stats_all)
        score = H.attn_colsum_batched(logits, gmax, gsum, QT, write_attention=False)
        n_local = score.shape[1]
        i, v = H.topk_batched(score, min(k, n_local))
        return H.pack_candidates(i, v, ori.view(QT, n_local, 3), dirs.view(QT, n_local, 3), k, first_ray)

    def batch_shard_global_poses(self, cand_all, k: int, rank: int, B: int):
        """Segment 4: every rank's candidates [G, G*B, k, 8] -> the poses of THIS rank's B queries (poses [B,4,4], val, idx)."""
        return self.shard_global_poses(cand_all, k, rank * B, B)

    def query_batch_sharded(self, tokens_local, gen_points: int, seed: int, k: int = 100, group=None, seed_offset=None):
        """``tokens_local`` [B,M,C]: this rank's B cold queries.  Every rank returns the poses / top-k of its own queries;
        with one rank this is ``query_batch``."""
        from . import distributed as D
        rank, ws = D.world(group)
        B, M, _ = tokens_local.shape
        lo, _ = D.shard_points(gen_points, rank, ws)
        msg = self.batch_shard_draw(tokens_local, gen_points, seed, rank, seed_offset)
        ori, dirs, logits, stats = self.batch_shard_local_logits(D._all_gather_stack(msg, group), B, M, gen_points, rank, ws)
        cand = self.batch_shard_local_candidates(logits, D._all_gather_stack(stats, group), ori, dirs, ws * B, k, lo * 27)
        return self.batch_shard_global_poses(D._all_gather_stack(cand, group), k, rank, B)

    def capture_query_batch_sharded(self, tokens_shape, gen_points: int, seed: int = 0, k: int = 100, group=None) -> "CapturedShardedBatch":
        return CapturedShardedBatch(self, tokens_shape, gen_points, seed, k, group)


class ResidentRays:
    """A model's emitted rays plus the cached encoder output for them (PosePipeline.make_resident)."""

    def __init__(self, ori, dirs, rgb, cache):
        self.ori, self.dirs, self.rgb, self.cache = ori, dirs, rgb, cache


class CapturedQuery:
    """The cold query captured once as a hipGraph (torch.cuda.CUDAGraph) and replayed per query.

    Replaying removes the per-launch host work (about twenty ctypes calls per query) and lets two captured queries run
    on two streams, so the latency-bound sampler of one query overlaps the throughput-bound stages of the other.  Every
    replay draws a fresh sampler stream: a device-side counter is bumped inside the graph and added to the seed
    (``iff_surface_sample(..., seed_dev_opt)``).  Inputs/outputs live in static buffers: ``tokens`` in, ``c2w`` /
    ``idx`` / ``val`` out (valid after the replay's stream has been synchronised, until the next replay).
    """

    def __init__(self, pipe: PosePipeline, tokens_shape, gen_points: int, seed: int = 0, k: int = 100):
        dev = pipe.device
        self.pipe = pipe
        self.tokens = torch.zeros(tokens_shape, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                 # warm-up outside capture (lazy initialisations, allocator)
            for _ in range(2):
                pipe.query(self.tokens, gen_points, seed, k, seed_offset=self.counter)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.counter += 1
            self.c2w, self.idx, self.val = pipe.query(self.tokens, gen_points, seed, k, seed_offset=self.counter)
            self.sampler_stats = pipe.last_sampler_stats

    def check(self) -> None:
        """After a synchronised replay: raise if this instance's last sampler run timed out (see check_sampler_stats)."""
        check_sampler_stats(self.sampler_stats)

    def replay(self, tokens: Optional[torch.Tensor] = None):
        if tokens is not None:
            self.tokens.copy_(tokens, non_blocking=True)
        self.graph.replay()
        return self.c2w


class CapturedShardedQuery:
    """``PosePipeline.query_sharded`` as three captured hipGraph segments with the two RCCL all_gathers issued eagerly between
    them (collectives are never captured).  A replay costs the host three graph launches and two collective calls instead
    of about forty kernel launches, and several instances replayed round-robin on their own streams keep the latency-bound
    sampler of one batch under the throughput-bound stages of another.  All instances use ONE process group, so the
    collectives of all in-flight batches are issued in the same order on every rank.

    Static buffers: ``tokens`` [Q,M,C] in; ``poses`` [Q,4,4], ``val`` / ``idx`` [Q,k] out (valid once the replay's stream
    is synchronised, until the next replay of this instance).
    """

    def __init__(self, pipe: PosePipeline, tokens_shape, gen_points: int, seed: int = 0, k: int = 100, group=None):
        import torch.distributed as dist
        from . import distributed as D
        dev = pipe.device
        self.pipe, self.group, self.k = pipe, group, k
        self.rank, self.ws = D.world(group)
        self.collective = dist.is_available() and dist.is_initialized()
        Q = int(tokens_shape[0])
        lo, _ = D.shard_points(gen_points, self.rank, self.ws)
        self.tokens = torch.zeros(tokens_shape, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                 # warm-up outside capture (lazy initialisations, allocator, RCCL)
            for _ in range(2):
                pipe.query_sharded(self.tokens, gen_points, seed, k, group)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        opts = dict(capture_error_mode="thread_local")     # the RCCL watchdog thread must not invalidate a capture
        self.g1, self.g2, self.g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g1, **opts):
            self.counter += 1
            ori, dirs, logits, self.stats = pipe.shard_local_logits(self.tokens, gen_points, seed, self.rank, self.ws,
                                                                    seed_offset=self.counter)
            self.sampler_stats = pipe.last_sampler_stats
        self._seg1 = (ori, dirs, logits)      # g2 reads these blocks of g1's private pool: keep them allocated
        self.stats_all = self.stats.new_zeros((self.ws,) + tuple(self.stats.shape))
        with torch.cuda.graph(self.g2, **opts):
            self.cand = pipe.shard_local_candidates(logits, self.stats_all, ori, dirs, Q, k, lo * 27, materialize_map=False)
        self.cand_all = self.cand.new_zeros((self.ws,) + tuple(self.cand.shape))
        with torch.cuda.graph(self.g3, **opts):
            self.poses, self.val, self.idx = pipe.shard_global_poses(self.cand_all, k)
        self.c2w = self.poses

    def _gather(self, out, src):
        from . import distributed as D
        D.all_gather_into(out, src, self.group)

    def check(self) -> None:
        """After a synchronised replay: raise if this instance's last sampler run timed out (see check_sampler_stats)."""
        check_sampler_stats(self.sampler_stats)

    def replay(self, tokens: Optional[torch.Tensor] = None):
        if tokens is not None:
            self.tokens.copy_(tokens, non_blocking=True)
        self.g1.replay()
        self._gather(self.stats_all, self.stats)
        self.g2.replay()
        self._gather(self.cand_all, self.cand)
        self.g3.replay()
        return self.poses


class CapturedBatchQuery:
    """``PosePipeline.query_batch`` captured as one hipGraph: B cold queries per replay.  Static buffers: ``tokens`` [B,M,C]
    in; ``c2w`` [B,4,4], ``idx`` / ``val`` [B,k] out.  Every replay adds 1 to the device-side seed counter; query b of replay r
    draws with seed + r + b * SAMPLER_SEED_STRIDE, so no two queries ever share a random stream."""

    def __init__(self, pipe: PosePipeline, tokens_shape, gen_points: int, seed: int = 0, k: int = 100):
        dev = pipe.device
        self.pipe = pipe
        self.tokens = torch.zeros(tokens_shape, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                 # warm-up outside capture (lazy initialisations, allocator)
            for _ in range(2):
                pipe.query_batch(self.tokens, gen_points, seed, k, seed_offset=self.counter)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.counter += 1
            self.c2w, self.idx, self.val = pipe.query_batch(self.tokens, gen_points, seed, k, seed_offset=self.counter)
            self.sampler_stats = pipe.last_sampler_stats

    def check(self) -> None:
        """After a synchronised replay: raise if this instance's last sampler run timed out (see check_sampler_stats)."""
        check_sampler_stats(self.sampler_stats)

    def replay(self, tokens: Optional[torch.Tensor] = None):
        if tokens is not None:
            self.tokens.copy_(tokens, non_blocking=True)
        self.graph.replay()
        return self.c2w


class CapturedShardedBatch:
    """``PosePipeline.query_batch_sharded`` as four captured hipGraph segments with the three RCCL all_gathers issued eagerly
    between them (samples + folded queries | softmax statistics | top-k candidates).  Static buffers: ``tokens`` [B,M,C] in
    (this rank's queries); ``poses`` [B,4,4], ``val`` / ``idx`` [B,k] out."""

    def __init__(self, pipe: PosePipeline, tokens_shape, gen_points: int, seed: int = 0, k: int = 100, group=None):
        from . import distributed as D
        dev = pipe.device
        self.pipe, self.group = pipe, group
        self.rank, self.ws = D.world(group)
        B, M = int(tokens_shape[0]), int(tokens_shape[1])
        lo, _ = D.shard_points(gen_points, self.rank, self.ws)
        self.tokens = torch.zeros(tokens_shape, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                 # warm-up outside capture (lazy initialisations, allocator, RCCL)
            for _ in range(2):
                pipe.query_batch_sharded(self.tokens, gen_points, seed, k, group, seed_offset=self.counter)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        opts = dict(capture_error_mode="thread_local")     # the RCCL watchdog thread must not invalidate a capture
        self.g1, self.g2, self.g3, self.g4 = (torch.cuda.CUDAGraph() for _ in range(4))
        with torch.cuda.graph(self.g1, **opts):
            self.counter += 1
            self.msg = pipe.batch_shard_draw(self.tokens, gen_points, seed, self.rank, seed_offset=self.counter)
            self.sampler_stats = pipe.last_sampler_stats
        self.msg_all = self.msg.new_zeros((self.ws,) + tuple(self.msg.shape))
        with torch.cuda.graph(self.g2, **opts):
            ori, dirs, logits, self.stats = pipe.batch_shard_local_logits(self.msg_all, B, M, gen_points, self.rank, self.ws)
        self._seg2 = (ori, dirs, logits)              # g3 reads these blocks of g2's private pool: keep them allocated
        self.stats_all = self.stats.new_zeros((self.ws,) + tuple(self.stats.shape))
        with torch.cuda.graph(self.g3, **opts):
            self.cand = pipe.batch_shard_local_candidates(logits, self.stats_all, ori, dirs, self.ws * B, k, lo * 27)
        self.cand_all = self.cand.new_zeros((self.ws,) + tuple(self.cand.shape))
        with torch.cuda.graph(self.g4, **opts):
            self.poses, self.val, self.idx = pipe.batch_shard_global_poses(self.cand_all, k, self.rank, B)
        self.c2w = self.poses

    def check(self) -> None:
        check_sampler_stats(self.sampler_stats)

    def replay_head(self, tokens: Optional[torch.Tensor] = None) -> None:
        """Segment 1 (surface points of this rank's queries, folded queries) and its all_gather.  A serving loop issues the
        head of step i + 1 BEFORE the tail of step i (each on its own stream and slot): the collectives of one process group
        execute in issue order, so this puts step i + 1's first exchange ahead of step i's last two and lets its heavy
        segment start under step i's latency-bound tail (small kernels + two exchanges) instead of after it."""
        from . import distributed as D
        if tokens is not None:
            self.tokens.copy_(tokens, non_blocking=True)
        self.g1.replay()
        D.all_gather_into(self.msg_all, self.msg, self.group)

    def replay_tail(self):
        from . import distributed as D
        self.g2.replay()
        D.all_gather_into(self.stats_all, self.stats, self.group)
        self.g3.replay()
        D.all_gather_into(self.cand_all, self.cand, self.group)
        self.g4.replay()
        return self.poses

    def replay(self, tokens: Optional[torch.Tensor] = None):
        self.replay_head(tokens)
        return self.replay_tail()


class CapturedColdImageQuery:
    """Image in -> emission -> pose out: B RGBA query images [B,H,W,4], EACH against its own freshly drawn ray set (the north
    star's "per-query" path with the image side included) as one hipGraph -- ``ImageFrontEnd.tokens_rgba`` (composite, resize /
    crop, backbone, token assembly) then ``query_batch`` (surface sampler, march, encoder + logits, score, top-k, pose)."""

    def __init__(self, pipe: PosePipeline, frontend, imgs_shape, gen_points: int, seed: int = 0, k: int = 100):
        dev = pipe.device
        self.rgba = torch.zeros(imgs_shape, dtype=torch.float32, device=dev)
        self.rgba[..., 3] = 1.0
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)

        def run():
            tokens, keep = frontend.tokens_rgba(self.rgba)
            return pipe.query_batch(tokens, gen_points, seed, k, seed_offset=self.counter, keep=keep)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.counter += 1
            self.c2w, self.idx, self.val = run()
            self.sampler_stats = pipe.last_sampler_stats

    def check(self) -> None:
        check_sampler_stats(self.sampler_stats)

    def replay(self, rgba: Optional[torch.Tensor] = None):
        if rgba is not None:
            self.rgba.copy_(rgba, non_blocking=True)
        self.graph.replay()
        return self.c2w


class CapturedImageQuery:
    """``PosePipeline.identify_images_resident`` as one hipGraph: a batch of query images in, poses out.  Static buffers:
    ``imgs`` [Q,H,W,3], ``masks`` [Q,H,W] in; ``c2w`` [Q,4,4], ``idx`` / ``val`` [Q,k] out."""

    def __init__(self, pipe: PosePipeline, frontend, imgs_shape, rays: ResidentRays, k: int = 100):
        dev = pipe.device
        self.imgs = torch.zeros(imgs_shape, dtype=torch.float32, device=dev)
        self.masks = torch.ones(imgs_shape[:3], dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                 # warm-up outside capture (library handles, autotuning, allocator)
            for _ in range(3):
                pipe.identify_images_resident(frontend, self.imgs, self.masks, rays, k)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.c2w, self.idx, self.val = pipe.identify_images_resident(frontend, self.imgs, self.masks, rays, k)

    def replay(self, imgs: Optional[torch.Tensor] = None, masks: Optional[torch.Tensor] = None):
        if imgs is not None:
            self.imgs.copy_(imgs, non_blocking=True)
        if masks is not None:
            self.masks.copy_(masks, non_blocking=True)
        self.graph.replay()
        return self.c2w
