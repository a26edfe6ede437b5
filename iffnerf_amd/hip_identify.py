"""Python owner of an ``iff_idnet`` handle (ray encoder + q/k projections) and wrappers for stage C.

Tensor allocation and streams come from PyTorch-ROCm; the arithmetic (fp32 MFMA GEMMs, softmax statistics,
column sums, top-k, pose solve) happens in libiffnerf_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check, dptr, fvec, stream_ptr

GEMM_F32, GEMM_BF16X3, GEMM_BF16X3_LAYERED, GEMM_F16X2, GEMM_F16X1 = 0, 1, 2, 3, 4        # include/iffnerf_hip.h IFF_GEMM_*
# (GEMM_F16X1: ONE fp16 product per block -- a throughput class outside the reference's accuracy class; opt-in only)
GEMM_DEFAULT = GEMM_F16X2        # falls back to BF16X3 by itself when a network does not fit fp16's range
F16_ORIGIN_BOUND = 64.0          # |ray origin| the F16X2 scale plan covers (csrc/api.hip plan_f16_scales); PosePipeline picks BF16X3 beyond

_KEYS = (("l1", "ray_preprocessor.mlp.0"), ("l2", "ray_preprocessor.mlp.2"), ("l3", "ray_preprocessor.mlp2.0"),
         ("l4", "ray_preprocessor.mlp2.2"), ("q", "attention.q_proj"), ("k", "attention.k_proj"))


def _gpu(t: torch.Tensor, name: str, cols: Optional[int] = None) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); libiffnerf_hip has no CPU path")
    t = t.detach().to(torch.float32)
    if cols is not None:
        t = t.reshape(-1, cols)
    return t.contiguous()


class IdNetHandle:
    """Weights of RayPreprocessor + MultiHeadAttention, transposed/padded once for the MFMA GEMMs."""

    def __init__(self, weights: Dict[str, torch.Tensor], device, gemm_mode: Optional[int] = None, trunk_variant: int = 0):
        self._h = None
        self.gemm_mode = int(GEMM_DEFAULT if gemm_mode is None else gemm_mode)
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"IdNetHandle needs a GPU device (got {device}); libiffnerf_hip has no CPU path")
        self.device = device
        d = _lib.IdNetDesc()
        keep = []
        for short, key in _KEYS:
            for suffix, field in ((".weight", "_w"), (".bias", "_b")):
                t = weights[key + suffix].detach().to(device=device, dtype=torch.float32).contiguous()
                keep.append(t)
                setattr(d, short + field, t.data_ptr())
        self.feature_c = int(weights["ray_preprocessor.mlp.0.weight"].shape[0])
        self.fea = int(weights["ray_preprocessor.mlp2.2.weight"].shape[0])
        self.img_fea = int(weights["attention.q_proj.weight"].shape[1])
        if int(weights["ray_preprocessor.mlp.0.weight"].shape[1]) != 141:
            raise RuntimeError("ray encoder input width must be 141 (pospe=8, viewpe=8, rgbpe=6)")
        d.feature_c, d.fea, d.img_fea = self.feature_c, self.fea, self.img_fea
        d.gemm_mode = self.gemm_mode
        d.trunk_variant = int(trunk_variant)
        out = C.c_void_p()
        with torch.cuda.device(device):
            check(_lib.lib().iff_idnet_create(C.byref(d), stream_ptr(device), C.byref(out)), "iff_idnet_create")
        self._h = out
        self.requested_gemm_mode = self.gemm_mode
        self.gemm_mode = int(_lib.lib().iff_idnet_gemm_mode(out))       # F16X2 falls back to BF16X3 when fp16's range is too small

    # ------------------------------------------------------------------ table files (include/iffnerf_hip.h iff_idnet_save / _load)
    def save(self, path: str) -> None:
        """Write every Linear in its MFMA layouts (and the planned F16X2 scales) to ``path``; ``IdNetHandle.from_file`` loads it."""
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_idnet_save(self._h, str(path).encode(), stream_ptr(self.device)), "iff_idnet_save")

    @classmethod
    def from_file(cls, path: str, device) -> "IdNetHandle":
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"IdNetHandle needs a GPU device (got {device}); libiffnerf_hip has no CPU path")
        L = _lib.lib()
        self = cls.__new__(cls)
        self._h, self.device = None, device
        out = C.c_void_p()
        with torch.cuda.device(device):
            check(L.iff_idnet_load(str(path).encode(), stream_ptr(device), C.byref(out)), "iff_idnet_load")
        self._h = out
        dims = [C.c_int32() for _ in range(3)]
        check(L.iff_idnet_dims(out, *[C.byref(v) for v in dims]), "iff_idnet_dims")
        self.feature_c, self.fea, self.img_fea = (int(v.value) for v in dims)
        self.gemm_mode = self.requested_gemm_mode = int(L.iff_idnet_gemm_mode(out))
        return self

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().iff_idnet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ bookkeeping for bench.py's roofline
    def mfma_products(self) -> int:
        """MFMA products the matrix-product mode issues per fp32-accurate product (include/iffnerf_hip.h IFF_GEMM_*)."""
        return {GEMM_F32: 1, GEMM_BF16X3: 6, GEMM_BF16X3_LAYERED: 6, GEMM_F16X2: 3, GEMM_F16X1: 1}[self.gemm_mode]

    def gemm_description(self) -> str:
        return {GEMM_F32: "fp32-input MFMA (k-ordered fmaf chain)",
                GEMM_BF16X3: "3xBF16 split on the bf16 MFMA (6 products, fp32-accurate), fp32 accumulate",
                GEMM_BF16X3_LAYERED: "3xBF16 split on the bf16 MFMA, one launch per layer",
                GEMM_F16X2: "2xFP16 split on the fp16 MFMA (3 products, fp32-accurate), fp32 accumulate",
                GEMM_F16X1: "fp16 operands on the fp16 MFMA (1 product, 11 significant bits: NOT the fp32 class), fp32 accumulate",
                }[self.gemm_mode] + "; march and shading in fp32"

    def trunk_kernel_name(self) -> str:
        """Name of the fused encoder + logits kernel as rocprofv3 prints it (profiles/*.csv)."""
        return {GEMM_F16X2: "k5_trunk_h<1, 1, 2>", GEMM_F16X1: "k5_trunk_h<5, 1, 2>"}.get(self.gemm_mode, "k5_trunk<true, 1>")

    # ------------------------------------------------------------------ K5
    def ray_encode(self, o, d, rgb, want_features: bool = True, want_k: bool = False):
        """RayPreprocessor.forward (+ k_proj when want_k) -> (features|None, k|None)."""
        o, d, rgb = _gpu(o, "rays_ori", 3), _gpu(d, "rays_dir", 3), _gpu(rgb, "rays_rgb", 3)
        N = o.shape[0]
        if d.shape[0] != N or rgb.shape[0] != N:
            raise RuntimeError("rays_ori / rays_dir / rays_rgb must have the same number of rows")
        L = _lib.lib()
        feat = o.new_empty(N, self.fea) if want_features else None
        k = o.new_empty(N, self.fea) if want_k else None
        ws_bytes = int(L.iff_ray_encode_workspace(self._h, N))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=o.device)
        with torch.cuda.device(self.device):
            check(L.iff_ray_encode(self._h, dptr(o), dptr(d), dptr(rgb), N, dptr(feat), dptr(k), ws.data_ptr(), ws_bytes,
                                   stream_ptr(self.device)), "iff_ray_encode")
        return feat, k

    def ray_trunk(self, o, d, rgb):
        """The encoder up to its last ReLU (ray_preprocessor.py:29-38) -> h3 [N, feature_c], the ray side of the folded
        logits (``q_fold`` / ``attn_logits_folded``)."""
        o, d, rgb = _gpu(o, "rays_ori", 3), _gpu(d, "rays_dir", 3), _gpu(rgb, "rays_rgb", 3)
        N = o.shape[0]
        if d.shape[0] != N or rgb.shape[0] != N:
            raise RuntimeError("rays_ori / rays_dir / rays_rgb must have the same number of rows")
        L = _lib.lib()
        h3 = o.new_empty(N, self.feature_c)
        ws_bytes = int(L.iff_ray_trunk_workspace(self._h, N))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=o.device)
        with torch.cuda.device(self.device):
            check(L.iff_ray_trunk(self._h, dptr(o), dptr(d), dptr(rgb), N, dptr(h3), ws.data_ptr(), ws_bytes,
                                  stream_ptr(self.device)), "iff_ray_trunk")
        return h3

    def q_fold(self, img_features):
        """Token side of the folded logits: [M, img_fea] -> qf [M, feature_c + 16] (column feature_c = per-token constant)."""
        x = _gpu(img_features, "img_features", self.img_fea)
        M = x.shape[0]
        L = _lib.lib()
        qf = x.new_empty(M, int(L.iff_q_fold_width(self._h)))
        with torch.cuda.device(self.device):
            check(L.iff_q_fold(self._h, dptr(x), M, dptr(qf), stream_ptr(self.device)), "iff_q_fold")
        return qf

    def attn_logits_folded(self, qf, h3, want_stats: bool = True):
        """logits = q k^T / sqrt(fea) from the folded operands, plus the per-row softmax statistics."""
        qf, h3 = _gpu(qf, "qf"), _gpu(h3, "h3", self.feature_c)
        M, N = qf.shape[0], h3.shape[0]
        logits = qf.new_empty(M, N)
        rmax = qf.new_empty(M) if want_stats else None
        rsum = qf.new_empty(M) if want_stats else None
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_attn_logits_folded(self._h, dptr(qf), dptr(h3), M, N, float(math.sqrt(self.fea)), dptr(logits),
                                                    dptr(rmax), dptr(rsum), stream_ptr(self.device)), "iff_attn_logits_folded")
        return logits, rmax, rsum

    def ray_logits_folded(self, qf, o, d, rgb, want_stats: bool = True, trunk_ms: Optional[list] = None):
        """Rays + folded queries -> (logits [M,N], row_max, row_sumexp) in one call (``ray_trunk`` + ``attn_logits_folded``;
        one fused launch for a 256-wide encoder)."""
        qf = _gpu(qf, "qf")
        o, d, rgb = _gpu(o, "rays_ori", 3), _gpu(d, "rays_dir", 3), _gpu(rgb, "rays_rgb", 3)
        M, N = qf.shape[0], o.shape[0]
        if d.shape[0] != N or rgb.shape[0] != N:
            raise RuntimeError("rays_ori / rays_dir / rays_rgb must have the same number of rows")
        L = _lib.lib()
        logits = qf.new_empty(M, N)
        rmax = qf.new_empty(M) if want_stats else None
        rsum = qf.new_empty(M) if want_stats else None
        ws_bytes = int(L.iff_ray_logits_folded_workspace(self._h, N, M))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=o.device)
        with torch.cuda.device(self.device):
            if trunk_ms is not None:          # synchronous, instrumented variant (bench.py roofline)
                ms = (C.c_float * 1)()
                check(L.iff_ray_logits_folded_timed(self._h, 1, dptr(o), dptr(d), dptr(rgb), N, dptr(qf), M,
                                                    float(math.sqrt(self.fea)), dptr(logits), dptr(rmax), dptr(rsum),
                                                    ws.data_ptr(), ws_bytes, ms, stream_ptr(self.device)),
                      "iff_ray_logits_folded_timed")
                trunk_ms.append(float(ms[0]))
                return logits, rmax, rsum
            check(L.iff_ray_logits_folded(self._h, dptr(o), dptr(d), dptr(rgb), N, dptr(qf), M, float(math.sqrt(self.fea)),
                                          dptr(logits), dptr(rmax), dptr(rsum), ws.data_ptr(), ws_bytes,
                                          stream_ptr(self.device)), "iff_ray_logits_folded")
        return logits, rmax, rsum

    def ray_logits_folded_batched(self, qf, o, d, rgb, batch: int, want_stats: bool = True, trunk_ms: Optional[list] = None):
        """``batch`` queries, each with its own rays: qf [B*M, width], o/d/rgb [B*N, 3] (query-major) ->
        (logits [B*M, N] = B blocks of [M, N], row_max [B*M], row_sumexp [B*M]) in one launch."""
        qf = _gpu(qf, "qf")
        o, d, rgb = _gpu(o, "rays_ori", 3), _gpu(d, "rays_dir", 3), _gpu(rgb, "rays_rgb", 3)
        if batch < 1 or qf.shape[0] % batch or o.shape[0] % batch or d.shape[0] != o.shape[0] or rgb.shape[0] != o.shape[0]:
            raise RuntimeError("token rows and ray rows must both be multiples of the batch size")
        M, N = qf.shape[0] // batch, o.shape[0] // batch
        L = _lib.lib()
        logits = qf.new_empty(batch * M, N)
        rmax = qf.new_empty(batch * M) if want_stats else None
        rsum = qf.new_empty(batch * M) if want_stats else None
        ws_bytes = int(L.iff_ray_logits_folded_batched_workspace(self._h, batch, N, M))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=o.device)
        with torch.cuda.device(self.device):
            if trunk_ms is not None:          # synchronous, instrumented variant (bench.py roofline)
                ms = (C.c_float * 1)()
                check(L.iff_ray_logits_folded_timed(self._h, batch, dptr(o), dptr(d), dptr(rgb), N, dptr(qf), M,
                                                    float(math.sqrt(self.fea)), dptr(logits), dptr(rmax), dptr(rsum),
                                                    ws.data_ptr(), ws_bytes, ms, stream_ptr(self.device)),
                      "iff_ray_logits_folded_timed")
                trunk_ms.append(float(ms[0]))
                return logits, rmax, rsum
            check(L.iff_ray_logits_folded_batched(self._h, batch, dptr(o), dptr(d), dptr(rgb), N, dptr(qf), M,
                                                  float(math.sqrt(self.fea)), dptr(logits), dptr(rmax), dptr(rsum), ws.data_ptr(),
                                                  ws_bytes, stream_ptr(self.device)), "iff_ray_logits_folded_batched")
        return logits, rmax, rsum

    # ------------------------------------------------------------------ per-model encoder cache
    def build_ray_cache(self, o, d, rgb) -> torch.Tensor:
        """The encoder once per resident ray set -> opaque cache (uint8 tensor) for ``logits_from_cache``."""
        o, d, rgb = _gpu(o, "rays_ori", 3), _gpu(d, "rays_dir", 3), _gpu(rgb, "rays_rgb", 3)
        N = o.shape[0]
        if d.shape[0] != N or rgb.shape[0] != N:
            raise RuntimeError("rays_ori / rays_dir / rays_rgb must have the same number of rows")
        L = _lib.lib()
        nbytes = int(L.iff_ray_cache_bytes(self._h, N))
        cache = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=o.device)
        ws_bytes = int(L.iff_ray_cache_workspace(self._h, N))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=o.device)
        with torch.cuda.device(self.device):
            check(L.iff_ray_cache_build(self._h, dptr(o), dptr(d), dptr(rgb), N, cache.data_ptr(), nbytes, ws.data_ptr(), ws_bytes,
                                        stream_ptr(self.device)), "iff_ray_cache_build")
        return cache

    def logits_from_cache(self, qf, cache: torch.Tensor, n_rays: int, want_stats: bool = True, rows: Optional[torch.Tensor] = None):
        """Folded token rows qf [M, width] x cached rays -> (logits [M,N], row_max, row_sumexp): no encoder work.
        ``rows`` (int32 [M / 256], device): kept rows per 256-row block, which come first (``iff_token_assemble_compact``): the rows
        behind them are skipped (``iff_logits_from_cache_rows``) -- their logits are whatever the buffer held."""
        qf = _gpu(qf, "qf")
        M, N = qf.shape[0], int(n_rays)
        L = _lib.lib()
        logits = qf.new_empty(M, N)
        rmax = qf.new_empty(M) if want_stats else None
        rsum = qf.new_empty(M) if want_stats else None
        ws_bytes = int(L.iff_logits_from_cache_workspace(self._h, N, M))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=qf.device)
        with torch.cuda.device(self.device):
            if rows is None:
                check(L.iff_logits_from_cache(self._h, cache.data_ptr(), N, dptr(qf), M, float(math.sqrt(self.fea)), dptr(logits),
                                              dptr(rmax), dptr(rsum), ws.data_ptr(), ws_bytes, stream_ptr(self.device)),
                      "iff_logits_from_cache")
            else:
                if M % 256 or rows.dtype != torch.int32 or rows.numel() != M // 256 or rows.device != qf.device or not rows.is_contiguous():
                    raise RuntimeError(f"rows must be a contiguous int32 tensor of {M} / 256 counts on {qf.device} (M a multiple of 256)")
                check(L.iff_logits_from_cache_rows(self._h, cache.data_ptr(), N, dptr(qf), M, rows.data_ptr(), float(math.sqrt(self.fea)),
                                                   dptr(logits), dptr(rmax), dptr(rsum), ws.data_ptr(), ws_bytes, stream_ptr(self.device)),
                      "iff_logits_from_cache_rows")
        return logits, rmax, rsum

    def k_proj(self, ray_features):
        """k_proj alone, for MultiHeadAttention called with already-encoded rays."""
        x = _gpu(ray_features, "ray_features", self.fea)
        k = torch.empty_like(x)
        with torch.cuda.device(self.device):
            check(_lib.lib().iff_k_proj(self._h, dptr(x), x.shape[0], dptr(k), stream_ptr(self.device)), "iff_k_proj")
        return k

    def q_proj(self, img_features):
        x = _gpu(img_features, "img_features", self.img_fea)
        M = x.shape[0]
        L = _lib.lib()
        q = x.new_empty(M, self.fea)
        ws_bytes = int(L.iff_q_proj_workspace(self._h, M))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(self.device):
            check(L.iff_q_proj(self._h, dptr(x), M, dptr(q), ws.data_ptr(), ws_bytes, stream_ptr(self.device)), "iff_q_proj")
        return q


def attn_logits(q: torch.Tensor, k: torch.Tensor, want_stats: bool = True, gemm_mode: int = GEMM_BF16X3):
    """logits = q k^T / sqrt(d) and the per-row softmax statistics (max, sum exp)."""
    q, k = _gpu(q, "q"), _gpu(k, "k")
    M, D = q.shape
    N = k.shape[0]
    logits = q.new_empty(M, N)
    rmax = q.new_empty(M) if want_stats else None
    rsum = q.new_empty(M) if want_stats else None
    with torch.cuda.device(q.device):
        check(_lib.lib().iff_attn_logits(dptr(q), dptr(k), M, N, D, float(math.sqrt(D)), dptr(logits), dptr(rmax), dptr(rsum),
                                         min(int(gemm_mode), 1), stream_ptr(q.device)), "iff_attn_logits")
    return logits, rmax, rsum


def attn_colsum(logits: torch.Tensor, row_max: torch.Tensor, row_sumexp: torch.Tensor, write_attention: bool = True):
    """In place: logits -> attention (when write_attention); returns score[N] = column sums of the attention."""
    M, N = logits.shape
    score = logits.new_empty(N)
    with torch.cuda.device(logits.device):
        check(_lib.lib().iff_attn_colsum(dptr(logits), M, N, dptr(row_max), dptr(row_sumexp), int(write_attention),
                                         dptr(score), stream_ptr(logits.device)), "iff_attn_colsum")
    return score


def topk(score: torch.Tensor, k: int):
    s = _gpu(score, "scores").reshape(-1)
    N = s.shape[0]
    if k > N:
        raise RuntimeError(f"selected index k out of range (k={k}, N={N})")   # torch.topk raises RuntimeError too
    if N >= TOPK_TWO_LEVEL_N and _shards_for(N, k) > 1:
        idx, val = topk_batched(s.view(1, N), k)
        return idx[0], val[0]
    idx = torch.empty(k, dtype=torch.int64, device=s.device)
    val = s.new_empty(k)
    with torch.cuda.device(s.device):
        check(_lib.lib().iff_topk(dptr(s), N, k, dptr(idx, torch.int64), dptr(val), None, 0, stream_ptr(s.device)), "iff_topk")
    return idx, val


def pose_from_topk(idx, val, rays_o, rays_d, model_up, want_parts: bool = False):
    """Per-image pose solve of pose_estimation/test.py:133-174 on the GPU -> c2w [4,4] (device tensor)."""
    idx = idx.detach().to(torch.int64).contiguous()
    val = _gpu(val, "weights").reshape(-1)
    o, d = _gpu(rays_o, "rays_ori", 3), _gpu(rays_d, "rays_dirs", 3)
    if not idx.is_cuda:
        raise RuntimeError("idx must live on the GPU; libiffnerf_hip has no CPU path")
    k = idx.shape[0]
    c2w = o.new_empty(4, 4)
    parts = o.new_empty(8 + k) if want_parts else None
    up = fvec(torch.as_tensor(model_up).detach().cpu().reshape(-1).tolist())
    with torch.cuda.device(o.device):
        check(_lib.lib().iff_pose_from_topk(dptr(idx, torch.int64), dptr(val), k, dptr(o), dptr(d), o.shape[0], up, dptr(c2w),
                                            dptr(parts), stream_ptr(o.device)), "iff_pose_from_topk")
    return (c2w, parts) if want_parts else c2w


# ---------------------------------------------------------------------------------------------- batches of queries
def attn_colsum_batched(logits: torch.Tensor, row_max: torch.Tensor, row_sumexp: torch.Tensor, Q: int,
                        write_attention: bool = True, rows: Optional[torch.Tensor] = None):
    """logits [Q*M, N] (Q queries of M token rows each), statistics [Q*M] -> score [Q, N]; attention in place if asked.
    ``rows`` (int32 [Q], device): only the first rows[q] rows of query q count (``iff_attn_colsum_rows``)."""
    QM, N = logits.shape
    if Q < 1 or QM % Q:
        raise RuntimeError(f"logits rows ({QM}) must be a multiple of the number of queries ({Q})")
    score = logits.new_empty(Q, N)
    with torch.cuda.device(logits.device):
        if rows is None:
            check(_lib.lib().iff_attn_colsum_batched(dptr(logits), Q, QM // Q, N, dptr(row_max), dptr(row_sumexp),
                                                     int(write_attention), dptr(score), stream_ptr(logits.device)),
                  "iff_attn_colsum_batched")
        else:
            if rows.dtype != torch.int32 or rows.numel() != Q or rows.device != logits.device or not rows.is_contiguous():
                raise RuntimeError(f"rows must be a contiguous int32 tensor of {Q} counts on {logits.device}")
            check(_lib.lib().iff_attn_colsum_rows(dptr(logits), Q, QM // Q, N, dptr(row_max), dptr(row_sumexp), rows.data_ptr(),
                                                  int(write_attention), dptr(score), stream_ptr(logits.device)),
                  "iff_attn_colsum_rows")
    return score


TOPK_TWO_LEVEL_N = 131072          # from here on a row is selected in column shards first (iff_topk_batched serves a row with ONE workgroup)


def _shards_for(N: int, k: int) -> int:
    for G in (32, 27, 16, 9, 8, 4, 3, 2):
        if N % G == 0 and N // G >= 4 * k:
            return G
    return 1


def topk_batched(score: torch.Tensor, k: int):
    """score [Q, N] -> (idx [Q, k] int64, val [Q, k]), each row as ``topk``.

    Rows of 131 072 scores and more (the reference's default 540 000 rays: one workgroup spends 0.6 ms on such a row) are selected
    in two levels of the same kernel: the top k of each of G contiguous column shards ([Q*G, N/G] is a view of the same memory),
    then the top k of the Q x (G k) candidates.  Exactly the one-level result, ties included: the candidates of a row are ordered
    (shard, value descending, index ascending), so "lowest position first" among equal values is "lowest ray index first"."""
    s = _gpu(score, "scores")
    if s.dim() != 2:
        raise RuntimeError("topk_batched expects scores [Q, N]")
    Q, N = s.shape
    if k > N:
        raise RuntimeError(f"selected index k out of range (k={k}, N={N})")
    G = _shards_for(N, k) if N >= TOPK_TWO_LEVEL_N else 1
    if G > 1:
        idx1, val1 = topk_batched(s.view(Q * G, N // G), k)
        pos, val = topk_batched(val1.view(Q, G * k), k)
        base = (torch.arange(G, device=s.device, dtype=torch.int64) * (N // G)).repeat_interleave(k)
        return torch.gather(idx1.view(Q, G * k) + base, 1, pos), val
    idx = torch.empty(Q, k, dtype=torch.int64, device=s.device)
    val = s.new_empty(Q, k)
    with torch.cuda.device(s.device):
        check(_lib.lib().iff_topk_batched(dptr(s), Q, N, k, dptr(idx, torch.int64), dptr(val), stream_ptr(s.device)),
              "iff_topk_batched")
    return idx, val


# ---------------------------------------------------------------------------------------------- merges of the ray-sharded path
def merge_row_stats(stats_all: torch.Tensor):
    """stats_all [G, R, 2] (every rank's per-row (max, sum-exp), rank order; what an all_gather delivers) -> (gmax [R], gsum [R])
    over all columns -- ``iff_merge_row_stats``, one launch (distributed.merge_row_stats_gathered states the arithmetic)."""
    st = _gpu(stats_all, "row statistics")
    if st.dim() != 3 or st.shape[-1] != 2:
        raise RuntimeError("stats_all must be [G, R, 2]")
    G, R = st.shape[:2]
    gmax, gsum = st.new_empty(R), st.new_empty(R)
    with torch.cuda.device(st.device):
        check(_lib.lib().iff_merge_row_stats(dptr(st), G, R, dptr(gmax), dptr(gsum), stream_ptr(st.device)), "iff_merge_row_stats")
    return gmax, gsum


def pack_candidates(idx: torch.Tensor, val: torch.Tensor, rays_o: torch.Tensor, rays_d: torch.Tensor, k: int, first_ray: int):
    """A rank's local top-kl (idx, val [Q, kl]) + its rays ([n, 3] shared or [Q, n, 3] per query) -> the message [Q, k, 8]
    (score, global ray index bits, origin, direction; slots kl.. pad with -inf / 2^31 - 1) -- ``iff_pack_candidates``."""
    idx = idx.detach().to(torch.int64).contiguous()
    val = _gpu(val, "scores")
    o, d = _gpu(rays_o, "rays"), _gpu(rays_d, "rays")
    Q, kl = idx.shape
    if o.shape != d.shape or o.shape[-1] != 3 or o.dim() not in (2, 3) or (o.dim() == 3 and o.shape[0] != Q) or kl > k:
        raise RuntimeError("rays must both be [n,3] or [Q,n,3], and kl <= k")
    stride = 0 if o.dim() == 2 else o.shape[1] * 3
    msg = val.new_empty(Q, k, 8)
    with torch.cuda.device(val.device):
        check(_lib.lib().iff_pack_candidates(dptr(idx, torch.int64), dptr(val), dptr(o), dptr(d), stride, Q, kl, k, int(first_ray), dptr(msg),
                                             stream_ptr(val.device)), "iff_pack_candidates")
    return msg


def merge_candidates(cand_all: torch.Tensor, k: int, q0: int = 0, n_queries=None):
    """cand_all [G, Qt, k, 8] (every rank's message) -> the global top-k of queries q0 .. q0 + Q - 1 in torch.topk's order:
    (val [Q, k], idx [Q, k] int64 global, origins [Q, k, 3], directions [Q, k, 3]) -- ``iff_merge_candidates``."""
    c = _gpu(cand_all, "candidates")
    if c.dim() != 4 or c.shape[2] != k or c.shape[3] != 8:
        raise RuntimeError(f"cand_all must be [G, Qt, {k}, 8]")
    G, Qt = c.shape[:2]
    Q = Qt - q0 if n_queries is None else int(n_queries)
    val = c.new_empty(Q, k)
    idx = torch.empty(Q, k, dtype=torch.int64, device=c.device)
    ori, dirs = c.new_empty(Q, k, 3), c.new_empty(Q, k, 3)
    with torch.cuda.device(c.device):
        check(_lib.lib().iff_merge_candidates(dptr(c), G, Qt, int(q0), Q, k, dptr(val), dptr(idx, torch.int64), dptr(ori), dptr(dirs),
                                              stream_ptr(c.device)), "iff_merge_candidates")
    return val, idx, ori, dirs


def pose_from_topk_batched(idx, val, rays_o, rays_d, model_up, want_parts: bool = False):
    """idx, val [Q, k]; rays_o / rays_d either one shared ray set [N, 3] or per-query candidates [Q, n, 3] -> c2w [Q, 4, 4]
    (+ the solver's internals [Q, 8 + k] per query as ``pose_from_topk`` when ``want_parts``)."""
    idx = idx.detach().to(torch.int64).contiguous()
    val = _gpu(val, "weights")
    if not idx.is_cuda:
        raise RuntimeError("idx must live on the GPU; libiffnerf_hip has no CPU path")
    Q, k = idx.shape
    o = rays_o.detach().to(torch.float32).contiguous()
    d = rays_d.detach().to(torch.float32).contiguous()
    if not (o.is_cuda and d.is_cuda):
        raise RuntimeError("rays must live on the GPU; libiffnerf_hip has no CPU path")
    if o.shape != d.shape or o.shape[-1] != 3 or (o.dim() == 3 and o.shape[0] != Q) or o.dim() not in (2, 3):
        raise RuntimeError("rays_o / rays_d must both be [N,3] or [Q,n,3]")
    n = o.shape[-2]
    stride = n * 3 if o.dim() == 3 else 0
    c2w = o.new_empty(Q, 4, 4)
    parts = o.new_empty(Q, 8 + k) if want_parts else None
    up = fvec(torch.as_tensor(model_up).detach().cpu().reshape(-1).tolist())
    with torch.cuda.device(o.device):
        check(_lib.lib().iff_pose_from_topk_batched(dptr(idx, torch.int64), dptr(val), Q, k, dptr(o), dptr(d), n, stride, up,
                                                    dptr(c2w), dptr(parts), stream_ptr(o.device)), "iff_pose_from_topk_batched")
    return (c2w, parts) if want_parts else c2w


def pose_errors(c2w: torch.Tensor, gt_c2w: torch.Tensor, parts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The per-image error metrics of pose_estimation/test.py:213-241 for Q poses in one launch: c2w, gt_c2w [Q,4,4] (or [4,4]),
    ``parts`` [Q, 8 + k] from the pose solve -> summary [Q,4] = (loss = mean kept weight, translation error, angular error in
    degrees, rays kept).  Nothing is read back: the caller decides when (``iff_pose_errors``)."""
    c = _gpu(c2w, "c2w").reshape(-1, 4, 4)
    g = gt_c2w.detach().to(device=c.device, dtype=torch.float32).reshape(-1, 4, 4).contiguous()
    Q = c.shape[0]
    if g.shape[0] != Q:
        raise RuntimeError("one ground-truth pose per estimated pose expected")
    k = 0
    if parts is not None:
        parts = _gpu(parts, "parts").reshape(Q, -1)
        k = parts.shape[1] - 8
    out = c.new_empty(Q, 4)
    with torch.cuda.device(c.device):
        check(_lib.lib().iff_pose_errors(dptr(c), dptr(g), dptr(parts), Q, k, dptr(out), stream_ptr(c.device)), "iff_pose_errors")
    return out
