// vit_kernels.hip -- the image backbone of stage C: DINOv2's ViT-S/14 forward (pose_estimation/backbone.py:12-14, consumed by
// pose_estimation/identification_module.py:137-146: forward_features(img)["x_norm_patchtokens"]) on the matrix cores.
//
//   patch embedding   14 x 14 / stride-14 convolution as a GEMM over an im2col image (bf16), + bias + position embedding
//   12 blocks         LayerNorm -> QKV (384 -> 1152) -> 6-head attention over the 257 tokens -> projection, LayerScale, residual;
//                     LayerNorm -> MLP 384 -> 1536 (exact GELU) -> 384, LayerScale, residual
//   final LayerNorm   -> patch tokens [Q, 256, 384] fp32 (and the class token)
//
// Arithmetic, two precisions behind one template parameter PREC:
//   PREC 1 (the default of the Python side): fp32-accurate products on the fp16 matrix cores.  Every operand is the exact split
//     a = hi + lo (hi = fp16(a), lo = fp16(a - hi): 22 significant bits) kept as two fp16 planes, and a product block is
//     hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16 (the dropped lo lo term is <= 2^-22 of the product) -- the k loop simply
//     runs three times over the planes.  Weights are stored times a power of two chosen at create so that fp16's range is used
//     (the accumulator is multiplied back by the exact inverse).  The reference runs DINOv2 in fp32
//     (pose_estimation/identification_module.py:137-142, backbone.py:12-14): this is that accuracy class.
//   PREC 0: bf16 operands on v_mfma_f32_32x32x16_bf16 (one product per block: 8 significant bits) -- a throughput option.
// fp32 accumulation, residual stream, LayerNorms, softmax statistics and epilogues in both.  One GEMM kernel (128 x 128 x 64 tiles, weights as the MFMA's A operand so that a lane
// ends up with 4 consecutive output features of one token: 8- / 16-byte stores) with four epilogues; one attention kernel per
// (image, head, 128-query block) that keeps K and V^T of the head in LDS, forms S^T = K Q^T so that a lane owns one query's whole
// row of scores (row maximum and sum without cross-lane traffic) and feeds the probabilities to the P V product straight from the
// accumulator registers (the k order of that product is permuted to the accumulator's row order, MI355X_MICROARCH / cdna guide 3).
#include "iff_device.h"
#include "iff_launch.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// operand element of a precision, its 16- and 8-byte vectors, the MFMA that multiplies it, and the conversion from fp32:
// PREC 0 rounds to bf16; PREC 1 returns the (hi, lo) pair of the exact split (the value clamped into fp16's range first)
template <int PREC> struct VT;
template <> struct VT<0> {
    typedef __bf16 e; typedef bf16x8 v8; typedef bf16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct VT<1> {
    typedef _Float16 e; typedef f16x8 v8; typedef f16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// Accuracy of the split: hi carries 11 significant bits; lo = fp16(v - hi) carries 11 more WHILE IT IS A NORMAL fp16, i.e. for |v| >= 2^-3
// (lo ~ 2^-11 |v| >= 2^-14).  Below that lo is an fp16 subnormal with an absolute resolution of 2^-24: the pair then represents v to
// 2^-25 ABSOLUTE instead of 2^-22 relative -- activations are not rescaled the way the weights are (their hi planes sit in [2^13, 2^14)).
// That is below fp32's own resolution of the O(1) values these operands are mixed with (LayerNorm outputs, softmax probabilities
// that sum to 1, GELU outputs), and it relies on v_mfma_f32_32x32x16_f16 multiplying fp16 subnormals exactly (it does not flush them:
// tests/test_hip_image_side.py::test_native_vit_small_activations holds the 1e-4 bound with every LayerNorm output scaled by 2^-7).
__device__ __forceinline__ void split_h(float v, _Float16& hi, _Float16& lo) {
    v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
// element i (scalar) / elements i .. i + 3 of a row: PREC 0 one bf16 plane, PREC 1 the hi plane at `p` and the lo plane `lo_off` elements on
template <int PREC>
__device__ __forceinline__ void put1(void* p, int64_t i, int64_t lo_off, float v) {
    if (PREC == 0) { reinterpret_cast<__bf16*>(p)[i] = (__bf16)v; return; }
    _Float16 h, l;
    split_h(v, h, l);
    reinterpret_cast<_Float16*>(p)[i] = h;
    reinterpret_cast<_Float16*>(p)[i + lo_off] = l;
}
template <int PREC>
__device__ __forceinline__ void put4(void* p, int64_t i, int64_t lo_off, const float v[4]) {
    if (PREC == 0) {
        bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p) + i) = o;
        return;
    }
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) { _Float16 a, b; split_h(v[j], a, b); h[j] = a; l[j] = b; }
    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(p) + i) = h;
    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(p) + i + lo_off) = l;
}

// ------------------------------------------------------------------------------------------------ im2col, class token, LayerNorm
// images [Q,3,H,W] fp32 (already resized / cropped / normalised) -> patches [Q * gh * gw][KP] bf16, k = c * P * P + dy * P + dx
// (the flattening of patch_embed.proj.weight [D,3,P,P]); columns >= 3 P P are zero
template <int PREC>
__global__ void k_vit_im2col(const float* __restrict__ img, int Q, int H, int W, int P, int gh, int gw, int KP, void* __restrict__ out) {
    const int64_t n = (int64_t)Q * gh * gw * KP;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(t % KP);
        const int64_t row = t / KP;
        const int px = (int)(row % gw), py = (int)((row / gw) % gh), q = (int)(row / ((int64_t)gw * gh));
        float v = 0.0f;
        if (k < 3 * P * P) {
            const int c = k / (P * P), r = k - c * P * P, dy = r / P, dx = r - dy * P;
            v = img[(((int64_t)q * 3 + c) * H + (py * P + dy)) * W + (px * P + dx)];
        }
        put1<PREC>(out, t, n, v);
    }
}

// x[q * T + 0][:] = cls + pos[0]
__global__ void k_vit_cls(const float* __restrict__ cls, const float* __restrict__ pos, int Q, int T, int D, float* __restrict__ x) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Q * D; t += gridDim.x * blockDim.x) {
        const int q = t / D, c = t - q * D;
        x[(int64_t)q * T * D + c] = cls[c] + pos[c];
    }
}

// LayerNorm over rows of D = 384 (eps inside the sqrt, affine), one wave per row.  OUT = 1 / 2: operand rows for the next GEMM
// (bf16 / fp16 hi + lo planes); OUT = 0: fp32 rows with the class-token row of every image dropped (the x_norm_patchtokens
// output) and, when cls_out is given, the class-token rows there.
template <int OUT>
__global__ void __launch_bounds__(256) k_vit_layernorm(const float* __restrict__ x, int64_t M, int T, const float* __restrict__ g,
                                                       const float* __restrict__ b, float eps, void* __restrict__ out_op,
                                                       float* __restrict__ out_f, float* __restrict__ cls_out) {
    constexpr int D = 384;
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * D;
    float v[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float2 t = *reinterpret_cast<const float2*>(xr + 2 * lane + 128 * j);
        v[2 * j] = t.x; v[2 * j + 1] = t.y;
    }
    float s = ((v[0] + v[1]) + (v[2] + v[3])) + (v[4] + v[5]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)D;
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; ++j) { const float d = v[j] - mean; ss = fmaf(d, d, ss); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float rstd = 1.0f / sqrtf(ss / (float)D + eps);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = 2 * lane + 128 * j;
        const float2 gg = *reinterpret_cast<const float2*>(g + c), bb = *reinterpret_cast<const float2*>(b + c);
        const float y0 = (v[2 * j] - mean) * rstd * gg.x + bb.x, y1 = (v[2 * j + 1] - mean) * rstd * gg.y + bb.y;
        if (OUT == 1) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 o = {(__bf16)y0, (__bf16)y1};
            *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(out_op) + row * D + c) = o;
        } else if (OUT == 2) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            _Float16 h0, l0, h1, l1;
            split_h(y0, h0, l0); split_h(y1, h1, l1);
            f16x2 oh = {h0, h1}, ol = {l0, l1};
            _Float16* op = reinterpret_cast<_Float16*>(out_op) + row * D + c;
            *reinterpret_cast<f16x2*>(op) = oh;
            *reinterpret_cast<f16x2*>(op + M * D) = ol;
        } else {
            const int64_t q = row / T;
            const int t = (int)(row - q * T);
            float* dst = t == 0 ? (cls_out ? cls_out + q * D : nullptr) : out_f + (q * (T - 1) + (t - 1)) * D;
            if (dst) *reinterpret_cast<float2*>(dst + c) = make_float2(y0, y1);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GEMM
// C[m][n] = sum_k X[m][k] W[n][k]   (X [M][K] bf16 activations, W [N][K] bf16 = nn.Linear weight), fp32 accumulation.
enum { EPI_QKV = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_EMBED = 3 };
constexpr int AT_TP = 288;          // attention: keys padded to 9 blocks of 32 (T = 257 tokens); also the row length of V^T
struct GemmEpi {
    const float* bias;        // [N]
    float wscale;             // PREC 1: the weights were stored times 1 / wscale (a power of two): acc * wscale is the product
    // EPI_QKV: q (times `qscale`) and k as [img][head][T][64] operand rows, v TRANSPOSED as [img][head][64][AT_TP] (the attention
    // kernel's P V product needs its keys contiguous per feature; columns T .. AT_TP - 1 are never written and are masked by the
    // reader).  PREC 1: the lo plane of each sits `qk_lo` / `v_lo` elements behind its hi plane
    void* q; void* k; void* v; int T; int heads; float qscale; int64_t qk_lo, v_lo;
    // EPI_GELU: out [M][N] operand rows (PREC 1: lo plane M * N elements on)
    void* out;
    // EPI_RESID: x[m][n] += ls[n] * (acc + bias[n])
    float* x; const float* ls;
    // EPI_EMBED: x[(m / G) * T + 1 + m % G][n] = acc + bias[n] + pos[1 + m % G][n]   (G patches per image)
    const float* pos; int G;
};

constexpr int GBN = 128, GBK = 64, GLD = GBK + 8;      // LDS rows padded to 144 B: conflict-free ds_read_b128

// GBM = 128 (a wave: 64 features x 64 tokens) or 64 (64 x 32: twice the workgroups for the N = 384 products, which would
// otherwise occupy 99 of the 256 CUs)
// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26; erff itself is good to ~1e-7): GELU's 0.5 x (1 + erf) keeps that absolute
// error times |x| / 2, far inside the 1e-4 the fp32 class is held to and invisible after the bf16 rounding of the other class; erff's
// ~40 instructions per value were half of the MLP's first GEMM (32 values per thread against 48 MFMAs per wave), in the fp32 class
// 7 of that launch's 37 us (16 images)
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}

// PREC 1: X and Wt are the hi planes, the lo planes sit x_lo / w_lo elements behind them; a k tile stages BOTH planes of both
// operands in LDS (four tiles) and every 16-deep k-step issues the three products lo hi, hi lo, hi hi (smallest first) from them --
// the operand bytes of a tile are fetched once for its three products.
template <int EPI, int GBM, int PREC>
__global__ void __launch_bounds__(256, 2) k_vit_gemm(const typename VT<PREC>::e* __restrict__ X, int64_t x_lo,
                                                     const typename VT<PREC>::e* __restrict__ Wt, int64_t w_lo, int64_t M, int N, int K,
                                                     GemmEpi e) {
    typedef typename VT<PREC>::e ET;
    typedef typename VT<PREC>::v8 bf16x8;                      // (the 16-byte operand vector of this precision)
    constexpr int MB = GBM / 64;                                // 32-token blocks per wave
    constexpr int NP = PREC ? 2 : 1;                            // operand planes
    // one pool: the operand tiles, and after the k loop the waves' output tiles (PREC 1 epilogue below)
    __shared__ __attribute__((aligned(16))) ET s_all[NP * (GBN + GBM) * GLD];
    ET (*sW)[GBN][GLD] = reinterpret_cast<ET (*)[GBN][GLD]>(s_all);
    ET (*sX)[GBM][GLD] = reinterpret_cast<ET (*)[GBM][GLD]>(s_all + NP * GBN * GLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = (wave & 1) * 64, wm = (wave >> 1) * (32 * MB);      // the wave's corner of the tile
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * GBM;
    const int n0 = blockIdx.x * GBN;
    f32x16 acc[2][MB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // a 128-row tile is 128 x 64 halves = 1024 chunks of 16 B: 4 per thread; the token tile 4 or 2.  PREC 0: TWO k tiles are in
    // flight in registers (sets A and B, the k loop unrolled by two): a k step is 8-16 MFMAs per wave, a third of the latency of the
    // tile that travels, and one tile ahead left every step waiting for it.  PREC 1: a k step is three times the MFMAs and the tile
    // is two planes: one tile ahead (set A only) covers the latency and keeps the registers
    struct Tile { bf16x8 w[NP][4], x[NP][2 * MB]; };
    auto fetch = [&](Tile& t, int k0) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const ET* Wp = Wt + (pl ? w_lo : 0);
            const ET* Xp = X + (pl ? x_lo : 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int chunk = tid + 256 * r, row = chunk >> 3, kc = (chunk & 7) * 8;
                t.w[pl][r] = *reinterpret_cast<const bf16x8*>(Wp + (int64_t)(n0 + row) * K + k0 + kc);
            }
#pragma unroll
            for (int r = 0; r < 2 * MB; ++r) {
                const int chunk = tid + 256 * r, row = chunk >> 3, kc = (chunk & 7) * 8;
                const int64_t m = min(m0 + row, M - 1);
                t.x[pl][r] = *reinterpret_cast<const bf16x8*>(Xp + m * K + k0 + kc);
            }
        }
    };
    auto step = [&](Tile& t, int k_next) {
        __syncthreads();                       // the previous step's fragments have been read
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int chunk = tid + 256 * r, row = chunk >> 3, kc = (chunk & 7) * 8;
                *reinterpret_cast<bf16x8*>(&sW[pl][row][kc]) = t.w[pl][r];
            }
#pragma unroll
            for (int r = 0; r < 2 * MB; ++r) {
                const int chunk = tid + 256 * r, row = chunk >> 3, kc = (chunk & 7) * 8;
                *reinterpret_cast<bf16x8*>(&sX[pl][row][kc]) = t.x[pl][r];
            }
        }
        __syncthreads();
        if (k_next < K) fetch(t, k_next);       // this set's next tile travels while the staged one is multiplied
#pragma unroll
        for (int ks = 0; ks < GBK / 16; ++ks) {
            bf16x8 fa[NP][2], fb[NP][MB];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[pl][a] = *reinterpret_cast<const bf16x8*>(&sW[pl][wn + 32 * a + lr][16 * ks + 8 * lh]);
#pragma unroll
                for (int b = 0; b < MB; ++b) fb[pl][b] = *reinterpret_cast<const bf16x8*>(&sX[pl][wm + 32 * b + lr][16 * ks + 8 * lh]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < MB; ++b) {
                    if (PREC) {
                        acc[a][b] = VT<PREC>::mfma(fa[NP - 1][a], fb[0][b], acc[a][b]);      // W lo x X hi
                        acc[a][b] = VT<PREC>::mfma(fa[0][a], fb[NP - 1][b], acc[a][b]);      // W hi x X lo
                    }
                    acc[a][b] = VT<PREC>::mfma(fa[0][a], fb[0][b], acc[a][b]);
                }
        }
    };
    if (PREC) {
        Tile tA;
        fetch(tA, 0);
        for (int k0 = 0; k0 < K; k0 += GBK) step(tA, k0 + GBK);
    } else {
        Tile tA, tB;
        fetch(tA, 0);
        if (GBK < K) fetch(tB, GBK);
        for (int k0 = 0; k0 < K; k0 += 2 * GBK) {
            step(tA, k0 + 2 * GBK);
            if (k0 + GBK < K) step(tB, k0 + 3 * GBK);
        }
    }
    // D[i = n][j = m]: the lane holds column m = lr and rows n = (reg & 3) + 8 (reg >> 2) + 4 lh: four consecutive n per reg group.
    // Stored from there, an instruction touches 32 token rows with 8 or 16 bytes each.  PREC 1 (two planes to write, 4-16 us of such
    // stores per launch): every wave turns its 64-feature x 32 MB-token tile over in LDS (the operand tiles are done) and stores whole
    // rows -- 128 B of a plane per 8 lanes.  Same values, same rounding: only the path differs.  (The V third of the QKV product is
    // written transposed -- tokens contiguous -- which the register layout already is; the residual update's 16-byte read-modify-writes
    // measured the same either way, 23.3 against 23.9 us, and stay direct.)  Per launch of 16 images: QKV 28.1 -> 25.7, MLP-in 33.3 -> 30.7 us.
    if constexpr (PREC == 1 && (EPI == EPI_QKV || EPI == EPI_GELU)) {
        const bool by_rows = !(EPI == EPI_QKV && n0 / (N / 3) == 2);
        if (by_rows) {
            constexpr int TLD = 68;                        // floats per tile row: 64 + 4 (conflict-free 16-byte writes down a column of tokens)
            static_assert(4 * 32 * MB * TLD * 4 <= (int)sizeof(s_all), "the four output tiles fit the operand pool");
            float* const tile = reinterpret_cast<float*>(s_all) + wave * (32 * MB) * TLD;
            __syncthreads();                               // every wave has read its last fragments
#pragma unroll
            for (int b = 0; b < MB; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int nl = 32 * a + 8 * g4 + 4 * lh;
                        const float4 bi = *reinterpret_cast<const float4*>(e.bias + n0 + wn + nl);
                        float v[4] = {acc[a][b][4 * g4] * e.wscale + bi.x, acc[a][b][4 * g4 + 1] * e.wscale + bi.y,
                                      acc[a][b][4 * g4 + 2] * e.wscale + bi.z, acc[a][b][4 * g4 + 3] * e.wscale + bi.w};
                        if (EPI == EPI_QKV) {
                            const float sc = n0 < N / 3 ? e.qscale : 1.0f;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] *= sc;
                        } else if (EPI == EPI_GELU) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = 0.5f * v[i] * (1.0f + erf_as(v[i] * 0.70710678118654752440f));
                        }
                        *reinterpret_cast<float4*>(tile + (32 * b + lr) * TLD + nl) = make_float4(v[0], v[1], v[2], v[3]);
                    }
            __syncthreads();
            {
                const int n8 = (lane & 7) * 8;
#pragma unroll
                for (int it = 0; it < 4 * MB; ++it) {
                    const int row = 8 * it + (lane >> 3);
                    const int64_t m = m0 + wm + row;
                    if (m >= M) continue;
                    const float4 v0 = *reinterpret_cast<const float4*>(tile + row * TLD + n8);
                    const float4 v1 = *reinterpret_cast<const float4*>(tile + row * TLD + n8 + 4);
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    f16x8 h, l;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { _Float16 hh, ll; split_h(v[j], hh, ll); h[j] = hh; l[j] = ll; }
                    _Float16* dst;
                    int64_t at, lo_off;
                    if (EPI == EPI_QKV) {
                        const int D = N / 3, which = n0 / D, head = (n0 - which * D + wn) >> 6;       // a wave's 64 features are one head
                        const int img = (int)(m / e.T), t = (int)(m - (int64_t)img * e.T);
                        dst = reinterpret_cast<_Float16*>(which == 0 ? e.q : e.k);
                        at = (((int64_t)(img * e.heads + head) * e.T + t) << 6) + n8;
                        lo_off = e.qk_lo;
                    } else {
                        dst = reinterpret_cast<_Float16*>(e.out);
                        at = m * N + n0 + wn + n8;
                        lo_off = M * (int64_t)N;
                    }
                    *reinterpret_cast<f16x8*>(dst + at) = h;
                    *reinterpret_cast<f16x8*>(dst + at + lo_off) = l;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) {
        const int64_t m = m0 + wm + 32 * b + lr;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + wn + 32 * a + 8 * g4 + 4 * lh;
                const float4 bi = *reinterpret_cast<const float4*>(e.bias + n);
                float v[4] = {acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
                if (PREC) { v[0] *= e.wscale; v[1] *= e.wscale; v[2] *= e.wscale; v[3] *= e.wscale; }       // exact: a power of two
                v[0] += bi.x; v[1] += bi.y; v[2] += bi.z; v[3] += bi.w;
                if (EPI == EPI_QKV) {
                    // a 128-column tile lies inside one of q / k / v (D = 384 = 3 x 128): `which` is the workgroup's, not the element's
                    const int D = N / 3, which = n0 / D, c = n - which * D, head = c >> 6, d = c & 63;
                    const int img = (int)m / e.T;
                    const int t = (int)m - img * e.T;
                    if (which == 2) {            // V^T: the 32 lanes of a half wave write 32 consecutive tokens of one feature row
                        const int64_t at = ((int64_t)(img * e.heads + head) * 64 + d) * AT_TP + t;
#pragma unroll
                        for (int i = 0; i < 4; ++i) put1<PREC>(e.v, at + (int64_t)i * AT_TP, e.v_lo, v[i]);
                    } else {
                        const int64_t at = (((int64_t)(img * e.heads + head) * e.T + t) << 6) + d;
                        const float sc = which == 0 ? e.qscale : 1.0f;
                        const float o[4] = {v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc};
                        put4<PREC>(which == 0 ? e.q : e.k, at, e.qk_lo, o);
                    }
                } else if (EPI == EPI_GELU) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = 0.5f * v[i] * (1.0f + erf_as(v[i] * 0.70710678118654752440f));
                    put4<PREC>(e.out, m * N + n, M * (int64_t)N, o);
                } else if (EPI == EPI_RESID) {
                    const float4 ls = *reinterpret_cast<const float4*>(e.ls + n);
                    float4* xp = reinterpret_cast<float4*>(e.x + m * N + n);
                    float4 xv = *xp;
                    xv.x = fmaf(ls.x, v[0], xv.x); xv.y = fmaf(ls.y, v[1], xv.y); xv.z = fmaf(ls.z, v[2], xv.z); xv.w = fmaf(ls.w, v[3], xv.w);
                    *xp = xv;
                } else {
                    const int64_t img = m / e.G;
                    const int p = (int)(m - img * e.G);
                    const float4 ps = *reinterpret_cast<const float4*>(e.pos + (int64_t)(1 + p) * N + n);
                    *reinterpret_cast<float4*>(e.x + (img * e.T + 1 + p) * N + n) = make_float4(v[0] + ps.x, v[1] + ps.y, v[2] + ps.z, v[3] + ps.w);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ GEMM, operand tiles by LDS-DMA
// The same product with the operand tiles filled by global_load_lds_dwordx4 (no staging registers, no ds_write pass: those stores
// were a fifth of the register-staged kernel and its tiles one deep) into TWO LDS buffers: the tile of k-step t + 1 travels while
// step t is multiplied, one barrier per step.  An LDS row is 128 B = eight 16-byte chunks of one operand row: PREC 0 64 k of the
// bf16 plane; PREC 1 32 k of the hi plane then the same 32 k of the lo plane (so both precisions stage 128 B per row and step, and
// a PREC 1 step is 32 deep with three products).  A DMA instruction writes 1 KiB lane-linearly -- eight whole rows -- so rows
// cannot be padded; chunk c of row r sits in slot c ^ ((r >> 1) & 7) instead (the swizzle is applied to the per-lane SOURCE
// address), which spreads the 16 lanes of every ds_read_b128 group over the 16 slots of a 256-byte bank row: conflict-free.
// Tile: BN = 32 AR WN features x BM = 32 BR WM tokens over WN x WM waves, a wave 32 AR x 32 BR.  The k order of every
// accumulator (16-deep blocks in increasing k; per block lo hi, hi lo, hi hi) is the register-staged kernel's: the same bits.
#ifndef VIT_XCD_REMAP
#define VIT_XCD_REMAP 1
#endif
#ifndef VIT_ABLATE
#define VIT_ABLATE 0          // timing-only development builds, a bit mask: 1 no epilogue, 2 no LDS reads / MFMAs, 4 no operand DMA (never shipped)
#endif
// Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, workgroup dispatch), each with its own L2: in launch order the
// gx column tiles of one token tile land on gx different XCDs and every one of them fetches the token tile's operand rows again from the
// Infinity Cache.  The workgroups that share an XCD (ids equal mod 8) take a CONTIGUOUS range of tiles instead (bijective for any
// grid: the first total % 8 XCDs serve one tile more), token tile by token tile, so a token tile's rows are fetched into one L2 once.
__device__ __forceinline__ void xcd_tile(int gx, int gy, int& tx, int& ty) {
#if VIT_XCD_REMAP
    const int total = gx * gy, orig = blockIdx.y * gx + blockIdx.x;
    const int xcd = orig & 7, q = total >> 3, r = total & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    ty = id / gx; tx = id - ty * gx;
#else
    tx = blockIdx.x; ty = blockIdx.y;
#endif
}

// NBUF 2: the tile of step t + 1 travels while step t is multiplied (one barrier per step).  NBUF 1: one buffer, filled and then
// multiplied (two barriers per step) -- half the LDS, so twice the workgroups per CU, and the overlap comes from THEM: the product is
// bound by the L2 -> LDS fill and by its epilogue's stores, and workgroups in different phases keep both paths busy.
template <int EPI, int PREC, int WN, int WM, int AR, int BR, int NBUF>
__global__ void __launch_bounds__(64 * WN * WM, (NBUF == 1 ? (WN * WM <= 4 ? 4 : 2) : (NBUF == 2 && WN * WM <= 4 ? 2 : 1)))
k_vit_gemm_dma(const typename VT<PREC>::e* __restrict__ X, int64_t x_lo, const typename VT<PREC>::e* __restrict__ Wt, int64_t w_lo, int64_t M,
               int N, int K, GemmEpi e) {
    typedef typename VT<PREC>::e ET;
    typedef typename VT<PREC>::v8 V8;
    constexpr int NW = WN * WM, BN = 32 * AR * WN, BM = 32 * BR * WM, ROWS = BN + BM;
    constexpr int KT = PREC ? 32 : 64;                          // k per step
    constexpr int KS = PREC ? 2 : 4;                            // 16-deep blocks per step
    constexpr int TILE = ROWS * 128;                            // bytes per buffer
    constexpr int PPW = ROWS / 8 / NW;                          // 1-KiB pieces per wave and step
    static_assert(ROWS % (8 * NW) == 0, "whole pieces per wave");
    __shared__ __attribute__((aligned(1024))) char s_all[NBUF * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = (wave % WN) * (32 * AR), wm = (wave / WN) * (32 * BR);
    const int lr = lane & 31, lh = lane >> 5;
    int tile_x, tile_y;
    xcd_tile((int)gridDim.x, (int)gridDim.y, tile_x, tile_y);
    const int64_t m0 = (int64_t)tile_y * BM;
    const int n0 = tile_x * BN;
    f32x16 acc[AR][BR];
#pragma unroll
    for (int a = 0; a < AR; ++a)
#pragma unroll
        for (int b = 0; b < BR; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // this lane's source of every piece the wave fills (at k = 0): piece p = rows 8 p .. 8 p + 7, lane -> row 8 p + (lane >> 3), slot lane & 7
    const ET* src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int r = 8 * (wave + NW * i) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const int kc = PREC ? (c & 3) * 8 : c * 8;
        const bool lo = PREC && c >= 4;
        if (r < BN) src[i] = Wt + (lo ? w_lo : 0) + (int64_t)(n0 + r) * K + kc;
        else        src[i] = X + (lo ? x_lo : 0) + min(m0 + (r - BN), M - 1) * K + kc;
    }
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            // (the source as `const void*`: with a pointer of template-dependent type here hipcc's host pass silently drops the kernel's launch stub)
            __builtin_amdgcn_global_load_lds((const void*)(src[i] + (int64_t)t * KT),
                                             (__attribute__((address_space(3))) void*)(s_all + buf * TILE + (wave + NW * i) * 1024), 16, 0, 0);
    };
    const int sw = (lr >> 1) & 7;
    const int steps = K / KT;
    // NBUF >= 2: a ring of NBUF buffers with NBUF - 1 steps travelling.  Before step t is multiplied its own pieces must have landed:
    // all but the (up to NBUF - 2) younger steps' -- a COUNTED wait, the raw barrier (a __syncthreads would drain the DMA queue: it
    // fences vmcnt(0)), then step t + NBUF - 1 is requested into the buffer step t - 1 has just been read from.
    constexpr int AHEAD = NBUF - 1;
    if (NBUF >= 2) {
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            if (i < steps) issue(i, i);
    }
    int cur = 0, nxt = AHEAD % (NBUF > 1 ? NBUF : 1);
    for (int t = 0; t < steps; ++t) {
        if (NBUF == 1) {
            if (t) __syncthreads();                            // everybody is done reading the buffer
            issue(t, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            const int younger = min(AHEAD - 1, steps - 1 - t);         // steps behind t that stay in flight
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // everybody's pieces of step t have landed; everybody is done reading step t - 1
            asm volatile("" ::: "memory");                     // (the compiler may not lift this step's LDS reads over the barrier)
#if !(VIT_ABLATE & 4)
            if (t + AHEAD < steps) issue(t + AHEAD, nxt);
#endif
        }
#if VIT_ABLATE & 2
        if (NBUF > 1) { cur = cur + 1 == NBUF ? 0 : cur + 1; nxt = nxt + 1 == NBUF ? 0 : nxt + 1; }
        continue;
#endif
        const char* bw = s_all + cur * TILE + (wn + lr) * 128;
        const char* bx = s_all + cur * TILE + (BN + wm + lr) * 128;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            constexpr int NP = PREC ? 2 : 1;
            V8 fa[NP][AR], fb[NP][BR];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const int slot = ((4 * pl + 2 * ks + lh) ^ sw) << 4;
#pragma unroll
                for (int a = 0; a < AR; ++a) fa[pl][a] = *reinterpret_cast<const V8*>(bw + a * (32 * 128) + slot);
#pragma unroll
                for (int b = 0; b < BR; ++b) fb[pl][b] = *reinterpret_cast<const V8*>(bx + b * (32 * 128) + slot);
            }
#pragma unroll
            for (int a = 0; a < AR; ++a)
#pragma unroll
                for (int b = 0; b < BR; ++b) {
                    if (PREC) {
                        acc[a][b] = VT<PREC>::mfma(fa[NP - 1][a], fb[0][b], acc[a][b]);      // W lo x X hi
                        acc[a][b] = VT<PREC>::mfma(fa[0][a], fb[NP - 1][b], acc[a][b]);      // W hi x X lo
                    }
                    acc[a][b] = VT<PREC>::mfma(fa[0][a], fb[0][b], acc[a][b]);
                }
        }
        if (NBUF > 1) { cur = cur + 1 == NBUF ? 0 : cur + 1; nxt = nxt + 1 == NBUF ? 0 : nxt + 1; }
    }
#if VIT_ABLATE & 1
    if (acc[0][0][0] != 123456.789f) return;
#endif
    // epilogues: the register-staged kernel's, per 64-feature half of the wave's tile (a wave's 64 features are one head) and per
    // 32-token block.  D[i = n][j = m]: the lane holds column m = lr and rows n = (reg & 3) + 8 (reg >> 2) + 4 lh
    if constexpr (PREC == 1 && (EPI == EPI_QKV || EPI == EPI_GELU)) {
        const bool by_rows = !(EPI == EPI_QKV && n0 / (N / 3) == 2);
        if (by_rows) {
            constexpr int TLD = 68;
            constexpr int TT = NW * 32 * TLD * 4 <= NBUF * TILE ? 32 : 16;      // tokens a wave turns over per pass: what the operand pool holds
            static_assert(NW * TT * TLD * 4 <= NBUF * TILE, "the waves' output tiles fit the operand pool");
            float* const tile = reinterpret_cast<float*>(s_all) + wave * TT * TLD;
#pragma unroll
            for (int h = 0; h < AR / 2; ++h)
#pragma unroll
                for (int b = 0; b < BR; ++b)
#pragma unroll
                    for (int ps = 0; ps < 32 / TT; ++ps) {
                        __syncthreads();                       // the last fragments / the previous pass's rows have been read
                        if (TT == 32 || (lr >> 4) == ps) {
#pragma unroll
                            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                                for (int g4 = 0; g4 < 4; ++g4) {
                                    const int a = 2 * h + a2;
                                    const int nl = 32 * a2 + 8 * g4 + 4 * lh;
                                    const float4 bi = *reinterpret_cast<const float4*>(e.bias + n0 + wn + 64 * h + nl);
                                    float v[4] = {acc[a][b][4 * g4] * e.wscale + bi.x, acc[a][b][4 * g4 + 1] * e.wscale + bi.y,
                                                  acc[a][b][4 * g4 + 2] * e.wscale + bi.z, acc[a][b][4 * g4 + 3] * e.wscale + bi.w};
                                    if (EPI == EPI_QKV) {
                                        const float sc = n0 < N / 3 ? e.qscale : 1.0f;
#pragma unroll
                                        for (int i = 0; i < 4; ++i) v[i] *= sc;
                                    } else {
#pragma unroll
                                        for (int i = 0; i < 4; ++i) v[i] = 0.5f * v[i] * (1.0f + erf_as(v[i] * 0.70710678118654752440f));
                                    }
                                    *reinterpret_cast<float4*>(tile + (lr & (TT - 1)) * TLD + nl) = make_float4(v[0], v[1], v[2], v[3]);
                                }
                        }
                        __syncthreads();
                        const int n8 = (lane & 7) * 8;
#pragma unroll
                        for (int it = 0; it < TT / 8; ++it) {
                            const int row = 8 * it + (lane >> 3);
                            const int64_t m = m0 + wm + 32 * b + TT * ps + row;
                            if (m >= M) continue;
                            const float4 v0 = *reinterpret_cast<const float4*>(tile + row * TLD + n8);
                            const float4 v1 = *reinterpret_cast<const float4*>(tile + row * TLD + n8 + 4);
                            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                            f16x8 hh8, ll8;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { _Float16 hh, ll; split_h(v[j], hh, ll); hh8[j] = hh; ll8[j] = ll; }
                            _Float16* dst;
                            int64_t at, lo_off;
                            if (EPI == EPI_QKV) {
                                const int D = N / 3, which = n0 / D, head = (n0 - which * D + wn + 64 * h) >> 6;
                                const int img = (int)(m / e.T), tk = (int)(m - (int64_t)img * e.T);
                                dst = reinterpret_cast<_Float16*>(which == 0 ? e.q : e.k);
                                at = (((int64_t)(img * e.heads + head) * e.T + tk) << 6) + n8;
                                lo_off = e.qk_lo;
                            } else {
                                dst = reinterpret_cast<_Float16*>(e.out);
                                at = m * N + n0 + wn + 64 * h + n8;
                                lo_off = M * (int64_t)N;
                            }
                            *reinterpret_cast<f16x8*>(dst + at) = hh8;
                            *reinterpret_cast<f16x8*>(dst + at + lo_off) = ll8;
                        }
                    }
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < BR; ++b) {
        const int64_t m = m0 + wm + 32 * b + lr;
        if (m >= M) continue;
#pragma unroll
        for (int a = 0; a < AR; ++a)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + wn + 32 * a + 8 * g4 + 4 * lh;
                const float4 bi = *reinterpret_cast<const float4*>(e.bias + n);
                float v[4] = {acc[a][b][4 * g4], acc[a][b][4 * g4 + 1], acc[a][b][4 * g4 + 2], acc[a][b][4 * g4 + 3]};
                if (PREC) { v[0] *= e.wscale; v[1] *= e.wscale; v[2] *= e.wscale; v[3] *= e.wscale; }
                v[0] += bi.x; v[1] += bi.y; v[2] += bi.z; v[3] += bi.w;
                if (EPI == EPI_QKV) {
                    const int D = N / 3, which = n0 / D, c = n - which * D, head = c >> 6, d = c & 63;
                    const int img = (int)m / e.T;
                    const int tk = (int)m - img * e.T;
                    if (which == 2) {
                        const int64_t at = ((int64_t)(img * e.heads + head) * 64 + d) * AT_TP + tk;
#pragma unroll
                        for (int i = 0; i < 4; ++i) put1<PREC>(e.v, at + (int64_t)i * AT_TP, e.v_lo, v[i]);
                    } else {
                        const int64_t at = (((int64_t)(img * e.heads + head) * e.T + tk) << 6) + d;
                        const float sc = which == 0 ? e.qscale : 1.0f;
                        const float o[4] = {v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc};
                        put4<PREC>(which == 0 ? e.q : e.k, at, e.qk_lo, o);
                    }
                } else if (EPI == EPI_GELU) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = 0.5f * v[i] * (1.0f + erf_as(v[i] * 0.70710678118654752440f));
                    put4<PREC>(e.out, m * N + n, M * (int64_t)N, o);
                } else if (EPI == EPI_RESID) {
                    const float4 ls = *reinterpret_cast<const float4*>(e.ls + n);
                    float4* xp = reinterpret_cast<float4*>(e.x + m * N + n);
                    float4 xv = *xp;
                    xv.x = fmaf(ls.x, v[0], xv.x); xv.y = fmaf(ls.y, v[1], xv.y); xv.z = fmaf(ls.z, v[2], xv.z); xv.w = fmaf(ls.w, v[3], xv.w);
                    *xp = xv;
                } else {
                    const int64_t img = m / e.G;
                    const int p = (int)(m - img * e.G);
                    const float4 ps = *reinterpret_cast<const float4*>(e.pos + (int64_t)(1 + p) * N + n);
                    *reinterpret_cast<float4*>(e.x + (img * e.T + 1 + p) * N + n) = make_float4(v[0] + ps.x, v[1] + ps.y, v[2] + ps.z, v[3] + ps.w);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ attention
// One workgroup = (image, head, block of 128 queries); wave w serves queries 32 w .. 32 w + 31 of the block.  (Two workgroups per
// (image, head) with a second pass for the 257th token in one wave -- 384 workgroups, one round of the chip's slots instead of 576 --
// measured the same 20 us: the wave with two passes is the critical path.)
constexpr int AT_KLD = 64 + 8;      // sK rows: 144 B
constexpr int AT_VLD = AT_TP + 12;  // sVt rows: 600 B -> conflict-free ds_read_b64 down a column of d

__global__ void __launch_bounds__(256, 2) k_vit_attention(const __bf16* __restrict__ Qh, const __bf16* __restrict__ Kh,
                                                          const __bf16* __restrict__ Vh, int T, int heads, __bf16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) __bf16 sK[AT_TP][AT_KLD];
    __shared__ __attribute__((aligned(16))) __bf16 sVt[64][AT_VLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int ih = blockIdx.y;                               // image * heads + head
    const int64_t base = (int64_t)ih * T * 64;
    // K rows (zero beyond T) and V^T rows (the QKV epilogue wrote V transposed; columns beyond T hold whatever the workspace held:
    // zeroed here, P is 0 there but 0 x NaN is not): 2304 chunks of 16 B each = 9 per thread, all 18 loads in flight together
    {
        const __bf16* Vt = Vh + (int64_t)ih * 64 * AT_TP;
        bf16x8 kv[9], vv[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int chunk = tid + 256 * r;
            const int t = chunk >> 3, dc = (chunk & 7) * 8;              // K: row t, features dc .. dc + 7
            kv[r] = *reinterpret_cast<const bf16x8*>(Kh + base + min(t, T - 1) * 64 + dc);
            vv[r] = *reinterpret_cast<const bf16x8*>(Vt + chunk * 8);     // V^T: 36 chunks per feature row
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int chunk = tid + 256 * r;
            const int t = chunk >> 3, dc = (chunk & 7) * 8;
            if (t >= T) {
#pragma unroll
                for (int i = 0; i < 8; ++i) kv[r][i] = (__bf16)0.0f;
            }
            *reinterpret_cast<bf16x8*>(&sK[t][dc]) = kv[r];
            const int d = chunk / 36, t8 = (chunk - d * 36) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (t8 + i >= T) vv[r][i] = (__bf16)0.0f;
            bf16x4 lo4 = {vv[r][0], vv[r][1], vv[r][2], vv[r][3]}, hi4 = {vv[r][4], vv[r][5], vv[r][6], vv[r][7]};
            *reinterpret_cast<bf16x4*>(&sVt[d][t8]) = lo4;           // rows of 600 B: 8-byte aligned pieces
            *reinterpret_cast<bf16x4*>(&sVt[d][t8 + 4]) = hi4;
        }
    }
    const int q = blockIdx.x * 128 + wave * 32 + lr;         // this lane's query (column of every MFMA below)
    const int qc = min(q, T - 1);
    bf16x8 fq[4];                                            // Q^T fragments: k = d = 16 ks + 8 lh + j
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fq[ks] = *reinterpret_cast<const bf16x8*>(Qh + base + qc * 64 + 16 * ks + 8 * lh);
    __syncthreads();
    // S^T[t][q] = sum_d K[t][d] Q[q][d]   (Q carries the 1 / sqrt(64))
    f32x16 S[AT_TP / 32];
#pragma unroll
    for (int b = 0; b < AT_TP / 32; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[b][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 fk = *reinterpret_cast<const bf16x8*>(&sK[32 * b + lr][16 * ks + 8 * lh]);
            S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, fq[ks], S[b], 0, 0, 0);
        }
    }
    // softmax over t for this lane's query: rows t = 32 b + (r & 3) + 8 (r >> 2) + 4 lh live in this lane and in lane ^ 32
    float mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < AT_TP / 32; ++b) {
        if (32 * b + 32 > T) {                               // only the last key block(s) hold padding: wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * b + (r & 3) + 8 * (r >> 2) + 4 * lh >= T) S[b][r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, S[b][r]), S[b][r + 1]);      // v_max3_f32
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
    const float mxl = mx * 1.4426950408889634f;              // exp(s - mx) = exp2(s log2 e - mx log2 e): one fma + the hardware exp2;
#pragma unroll                                               // P is rounded to bf16 next
    for (int b = 0; b < AT_TP / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[b][r] = __builtin_amdgcn_exp2f(fmaf(S[b][r], 1.4426950408889634f, -mxl)); sum += S[b][r]; }
    sum += __shfl_xor(sum, 32, 64);
    // O^T[d][q] = sum_t V^T[d][t] P^T[t][q]: the accumulator registers 8 s .. 8 s + 7 of block b are the B fragment of k-step
    // (b, s); its element j is row 16 s + 8 (j >> 2) + 4 lh + (j & 3) of the block, so the V^T fragment is gathered in that order
    f32x16 O[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[a][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < AT_TP / 32; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fp;
#pragma unroll
            for (int j = 0; j < 8; ++j) fp[j] = (__bf16)S[b][8 * s + j];
            const int t0 = 32 * b + 16 * s + 4 * lh;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(&sVt[32 * a + lr][t0]);
                const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(&sVt[32 * a + lr][t0 + 8]);
                bf16x8 fv = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                O[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv, fp, O[a], 0, 0, 0);
            }
        }
    if (q < T) {
        const float inv = 1.0f / sum;
        const int64_t img = ih / heads;
        const int head = ih - (int)img * heads;
        __bf16* dst = out + ((img * T + q) * heads + head) * 64;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o = {(__bf16)(O[a][4 * g4] * inv), (__bf16)(O[a][4 * g4 + 1] * inv), (__bf16)(O[a][4 * g4 + 2] * inv),
                            (__bf16)(O[a][4 * g4 + 3] * inv)};
                *reinterpret_cast<bf16x4*>(dst + 32 * a + 8 * g4 + 4 * lh) = o;
            }
    }
}

// The same attention with fp32-accurate products (PREC 1): Q, K, V^T arrive as fp16 hi / lo planes (the lo plane qk_lo / v_lo
// elements behind the hi plane), S^T = Kh Qh + Kl Qh + Kh Ql and O^T = Vh Ph + Vl Ph + Vh Pl with P split in registers.  The keys
// pass through ONE 46-KB LDS buffer in four pieces -- K of key blocks 0..4, K of blocks 5..8, V^T of blocks 0..4, V^T of blocks 5..8,
// both planes each -- with all nine S blocks kept in registers in between: two workgroups per CU (with K and then V^T whole the
// buffer was 83 KB: one), the same products in the same order.  The next piece travels in registers while the current one is multiplied.
constexpr int AX_B0 = 5, AX_T0 = 32 * AX_B0;                 // key blocks / keys of the first piece (the second: 4 / 128)
constexpr int AX_VLD = AX_T0 + 12;                           // V^T piece rows: 344 B (as AT_VLD: conflict-free ds_read_b64 down a column of d)
constexpr int AX_KPL = AX_T0 * AT_KLD, AX_VPL = 64 * AX_VLD; // halves per plane of a piece
__global__ void __launch_bounds__(256, 2) k_vit_attention_x2(const _Float16* __restrict__ Qh, const _Float16* __restrict__ Kh, int64_t qk_lo,
                                                             const _Float16* __restrict__ Vh, int64_t v_lo, int T, int heads,
                                                             _Float16* __restrict__ out, int64_t out_lo) {
    __shared__ __attribute__((aligned(16))) _Float16 s_buf[2 * (AX_KPL > AX_VPL ? AX_KPL : AX_VPL)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int ih = blockIdx.y;
    const int64_t base = (int64_t)ih * T * 64;
    const _Float16* Vt = Vh + (int64_t)ih * 64 * AT_TP;
    // piece h of K: keys t0 .. t0 + 32 nb - 1, eight 16-byte chunks per key and plane; of V^T: 4 nb chunks per feature row and plane
    f16x8 pre[2][AX_B0];                                      // [plane][round]: 256 chunks per round
    auto fetch_k = [&](int h) {
        const int t0 = h ? AX_T0 : 0, nb = h ? AT_TP / 32 - AX_B0 : AX_B0;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int r = 0; r < AX_B0; ++r)
                if (r < nb) {
                    const int chunk = tid + 256 * r, t = t0 + (chunk >> 3), dc = (chunk & 7) * 8;
                    pre[pl][r] = *reinterpret_cast<const f16x8*>(Kh + pl * qk_lo + base + min(t, T - 1) * 64 + dc);
                }
    };
    auto store_k = [&](int h) {
        const int t0 = h ? AX_T0 : 0, nb = h ? AT_TP / 32 - AX_B0 : AX_B0;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int r = 0; r < AX_B0; ++r)
                if (r < nb) {
                    const int chunk = tid + 256 * r, tl = chunk >> 3, dc = (chunk & 7) * 8;
                    f16x8 v = pre[pl][r];
                    if (t0 + tl >= T) {                       // rows beyond T: zero
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = (_Float16)0.0f;
                    }
                    *reinterpret_cast<f16x8*>(&s_buf[pl * AX_KPL + tl * AT_KLD + dc]) = v;
                }
    };
    auto fetch_v = [&](int h) {
        const int t0 = h ? AX_T0 : 0, nb = h ? AT_TP / 32 - AX_B0 : AX_B0;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int r = 0; r < AX_B0; ++r)
                if (r < nb) {
                    const int chunk = tid + 256 * r, d = chunk / (4 * nb), c8 = chunk - d * (4 * nb);
                    pre[pl][r] = *reinterpret_cast<const f16x8*>(Vt + pl * v_lo + d * AT_TP + t0 + 8 * c8);
                }
    };
    auto store_v = [&](int h) {
        const int t0 = h ? AX_T0 : 0, nb = h ? AT_TP / 32 - AX_B0 : AX_B0;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int r = 0; r < AX_B0; ++r)
                if (r < nb) {
                    const int chunk = tid + 256 * r, d = chunk / (4 * nb), c8 = chunk - d * (4 * nb);
                    f16x8 v = pre[pl][r];
#pragma unroll
                    for (int i = 0; i < 8; ++i)               // columns beyond T hold whatever the workspace held: P is 0 there, 0 x NaN is not
                        if (t0 + 8 * c8 + i >= T) v[i] = (_Float16)0.0f;
                    f16x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f16x4*>(&s_buf[pl * AX_VPL + d * AX_VLD + 8 * c8]) = lo4;      // rows of 344 B: 8-byte aligned pieces
                    *reinterpret_cast<f16x4*>(&s_buf[pl * AX_VPL + d * AX_VLD + 8 * c8 + 4]) = hi4;
                }
    };
    fetch_k(0);
    const int q = blockIdx.x * 128 + wave * 32 + lr;
    const int qc = min(q, T - 1);
    f16x8 fq[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fq[pl][ks] = *reinterpret_cast<const f16x8*>(Qh + pl * qk_lo + base + qc * 64 + 16 * ks + 8 * lh);
    f32x16 S[AT_TP / 32];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        store_k(h);
        __syncthreads();
        if (h == 0) fetch_k(1); else fetch_v(0);             // the next piece, in flight under this piece's products
        const int b0 = h ? AX_B0 : 0, nb = h ? AT_TP / 32 - AX_B0 : AX_B0;
#pragma unroll
        for (int bl = 0; bl < AX_B0; ++bl)
            if (bl < nb) {
                const int b = b0 + bl;
#pragma unroll
                for (int r = 0; r < 16; ++r) S[b][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(&s_buf[(32 * bl + lr) * AT_KLD + 16 * ks + 8 * lh]);
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(&s_buf[AX_KPL + (32 * bl + lr) * AT_KLD + 16 * ks + 8 * lh]);
                    S[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, fq[0][ks], S[b], 0, 0, 0);      // the small terms first
                    S[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, fq[1][ks], S[b], 0, 0, 0);
                    S[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, fq[0][ks], S[b], 0, 0, 0);
                }
            }
        __syncthreads();                                     // every wave has read this piece: the next goes over it
    }
    float mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < AT_TP / 32; ++b) {
        if (32 * b + 32 > T) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (32 * b + (r & 3) + 8 * (r >> 2) + 4 * lh >= T) S[b][r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, S[b][r]), S[b][r + 1]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
    // exp(s - mx) = exp2(s log2 e - mx log2 e): one fma (a single rounding of the argument) + the hardware exp2 (1 ulp).  The argument's
    // rounding is 2^-24 |s log2 e|: for every term that carries weight (s - mx > -20) the probability moves by < 2e-6 relative, far
    // inside the fp32 class; libm's expf -- an extended-precision argument reduction, ~11 instructions per value -- was a quarter of
    // this kernel's vector instructions (144 exponentials per lane)
    const float mxl = mx * 1.4426950408889634f;
#pragma unroll
    for (int b = 0; b < AT_TP / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[b][r] = __builtin_amdgcn_exp2f(fmaf(S[b][r], 1.4426950408889634f, -mxl)); sum += S[b][r]; }
    sum += __shfl_xor(sum, 32, 64);
    f32x16 O[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[a][r] = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        store_v(h);
        __syncthreads();
        if (h == 0) fetch_v(1);
        const int b0 = h ? AX_B0 : 0, nb = h ? AT_TP / 32 - AX_B0 : AX_B0;
#pragma unroll
        for (int bl = 0; bl < AX_B0; ++bl)
            if (bl < nb) {
                const int b = b0 + bl;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    f16x8 ph, pl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { _Float16 hh, ll; split_h(S[b][8 * s2 + j], hh, ll); ph[j] = hh; pl[j] = ll; }
                    const int t0 = 32 * bl + 16 * s2 + 4 * lh;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        f16x8 fv[2];
#pragma unroll
                        for (int p2 = 0; p2 < 2; ++p2) {
                            const f16x4 v0 = *reinterpret_cast<const f16x4*>(&s_buf[p2 * AX_VPL + (32 * a + lr) * AX_VLD + t0]);
                            const f16x4 v1 = *reinterpret_cast<const f16x4*>(&s_buf[p2 * AX_VPL + (32 * a + lr) * AX_VLD + t0 + 8]);
                            fv[p2] = f16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        }
                        O[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv[1], ph, O[a], 0, 0, 0);
                        O[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv[0], pl, O[a], 0, 0, 0);
                        O[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv[0], ph, O[a], 0, 0, 0);
                    }
                }
            }
        if (h == 0) __syncthreads();
    }
    if (q < T) {
        const float inv = 1.0f / sum;
        const int64_t img = ih / heads;
        const int head = ih - (int)img * heads;
        const int64_t at = ((img * T + q) * heads + head) * 64;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float o[4] = {O[a][4 * g4] * inv, O[a][4 * g4 + 1] * inv, O[a][4 * g4 + 2] * inv, O[a][4 * g4 + 3] * inv};
                put4<1>(out, at + 32 * a + 8 * g4 + 4 * lh, out_lo, o);
            }
    }
}

// ------------------------------------------------------------------------------------------------ image preprocessing
// Antialiased resize (the separable triangle / cubic a = -0.5 filters ATen's upsample_*2d_aa kernels use, i.e. what
// F.interpolate(..., antialias=True, align_corners=False) computes) of channels-last images, cropped to a window of the resized
// image and normalised per channel, written channels-first -- Resize(256, BICUBIC) + CenterCrop(224) + Normalize of the reference
// (pose_estimation/identification_module.py:36-61) in one pass over the pixels the crop needs.
constexpr int RS_TAPS = 32;         // filter taps per axis the kernel holds (scale factors up to 7.5 bicubic, 15 bilinear)
__device__ inline float aa_filter(float x, int cubic) {
    x = fabsf(x);
    if (!cubic) return x < 1.0f ? 1.0f - x : 0.0f;
    const float a = -0.5f;
    if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
    return 0.0f;
}
// weights of output index `o` of an axis of `in` samples resized to `out`: first tap index and tap count returned, weights to w[]
__device__ inline void aa_weights(int o, int in, int out, int cubic, float* w, int& first, int& count) {
    const float scale = (float)in / (float)out;
    const float support = (cubic ? 2.0f : 1.0f) * (scale >= 1.0f ? scale : 1.0f);
    const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    const float center = scale * ((float)o + 0.5f);
    first = max((int)(center - support + 0.5f), 0);
    count = min(min((int)(center + support + 0.5f), in) - first, RS_TAPS);
    float total = 0.0f;
    for (int j = 0; j < count; ++j) { w[j] = aa_filter(((float)(j + first) - center + 0.5f) * invscale, cubic); total += w[j]; }
    for (int j = 0; j < count; ++j) w[j] = total != 0.0f ? w[j] / total : w[j];
    for (int j = count; j < RS_TAPS; ++j) w[j] = 0.0f;
}

// `mode` (IFF_RESIZE_*): 0 the C channels of the source as they are (Cin == C); 1 the source is RGBA (Cin = 4) and the three output
// channels are the colour composited on white, rgb * a + (1 - a) -- pose_estimation/test.py:77-81, the three roundings of the three
// torch ops, evaluated per input pixel inside the horizontal pass; 2 the source is RGBA and the one output channel is its alpha.
struct ResizeArgs {
    const float* src; int Q, H, W, C;            // [Q,H,W,Cin]; C output channels (<= 4)
    int Cin, mode;
    int rh, rw;                                  // size of the (virtual) resized image
    int top, left, ch, cw;                       // crop window inside it
    int cubic;
    float mean[4], inv_std[4];
    float* dst;                                  // [Q,C,ch,cw]
};
// One workgroup = a tile of 16 columns x `tr` rows of the output (tr = 16; 1 when 16-row tiles would leave most CUs without a
// workgroup -- the 16 x 16 token-grid mask of a batch is one such tile per image).  Separable inside the tile: every input row the tile's 16 output rows touch is
// filtered horizontally once for the tile's 16 columns (LDS), then the columns are filtered vertically -- the order of ATen's
// kernel (horizontal sum per row, then the weighted sum of rows), ~3x fewer multiply-adds than a 2-D sum per output pixel.
// The input rows pass through LDS in chunks of `rch` rows, each row's segment (the tile's columns with their filter support:
// contiguous in a channels-last image) loaded by consecutive lanes: read straight from global memory the horizontal pass had 16
// lanes walking 16 different pixel runs and was bound by the texture-address path (the kernel took 400 us for 32 images).
__global__ void __launch_bounds__(256) k_resize_crop(ResizeArgs a, int max_rows, int max_cols, int rch, int tr) {
    extern __shared__ float s_dyn[];            // [max_rows][16][C] horizontally filtered rows, then [rch][max_cols * C] input rows
    __shared__ float s_wx[16][RS_TAPS], s_wy[16][RS_TAPS];
    __shared__ int s_fx[16], s_nx[16], s_fy[16], s_ny[16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * tr, q = blockIdx.z;
    if (threadIdx.x < 16) aa_weights(a.left + min(ox0 + tx, a.cw - 1), a.W, a.rw, a.cubic, s_wx[tx], s_fx[tx], s_nx[tx]);
    else if (threadIdx.x >= 64 && threadIdx.x < 64 + tr)        // in another wave than the columns' weights: the two run side by side
        aa_weights(a.top + min(oy0 + tx, a.ch - 1), a.H, a.rh, a.cubic, s_wy[tx], s_fy[tx], s_ny[tx]);
    __syncthreads();
    const int C = a.C, Cin = a.Cin;
    float* const s_tmp = s_dyn;
    float* const s_in = s_dyn + max_rows * 16 * C;
    const int ry0 = s_fy[0], fx0 = s_fx[0];
    int ry1 = ry0, fx1 = fx0;
    for (int i = 0; i < 16; ++i) fx1 = max(fx1, s_fx[i] + s_nx[i]);
    for (int i = 0; i < tr; ++i) ry1 = max(ry1, s_fy[i] + s_ny[i]);
    const int n_rows = min(ry1 - ry0, max_rows);
    const int seg = min(fx1 - fx0, max_cols) * Cin, ld = max_cols * Cin;      // floats of a row this tile reads
    const float* img = a.src + (int64_t)q * a.H * a.W * Cin;
    for (int r0 = 0; r0 < n_rows; r0 += rch) {
        const int nr = min(rch, n_rows - r0);
        if (r0) __syncthreads();                // the previous chunk has been filtered
        // a wave per row, its lanes along the row's segment; twelve loads (4 rows x 3 pieces of 64 floats) in flight per lane
        // before the first is stored -- one at a time, a chunk cost twelve memory latencies in a row.  (Requesting the NEXT chunk
        // before this one is filtered changed nothing: 176 against 180 us for 32 images.)
        for (int rb = threadIdx.x >> 6; rb < nr; rb += 16)
            for (int kb = threadIdx.x & 63; kb < seg; kb += 192) {
                float v[4][3];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = rb + 4 * rr;
                    const float* src = img + ((int64_t)(ry0 + r0 + min(r, nr - 1)) * a.W + fx0) * Cin;
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) v[rr][kk] = src[min(kb + 64 * kk, seg - 1)];
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk)
                        if (rb + 4 * rr < nr && kb + 64 * kk < seg) s_in[(rb + 4 * rr) * ld + kb + 64 * kk] = v[rr][kk];
            }
        __syncthreads();
        for (int item = threadIdx.x; item < nr * 16; item += 256) {
            const int r = item >> 4, ox = item & 15;
            const float* row = s_in + r * ld + (s_fx[ox] - fx0) * Cin;
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (a.mode == 0) {
                for (int i = 0; i < s_nx[ox]; ++i) {
                    const float w = s_wx[ox][i];
                    for (int c = 0; c < C; ++c) acc[c] = fmaf(w, row[i * C + c], acc[c]);
                }
            } else if (a.mode == 1) {
                for (int i = 0; i < s_nx[ox]; ++i) {
                    const float w = s_wx[ox][i], al = row[i * 4 + 3], white = 1.0f - al;
                    for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, row[i * 4 + c] * al + white, acc[c]);
                }
            } else {
                for (int i = 0; i < s_nx[ox]; ++i) acc[0] = fmaf(s_wx[ox][i], row[i * 4 + 3], acc[0]);
            }
            for (int c = 0; c < C; ++c) s_tmp[((r0 + r) * 16 + ox) * C + c] = acc[c];
        }
    }
    __syncthreads();
    const int ox = ox0 + tx, oy = oy0 + ty;
    if (ty >= tr || ox >= a.cw || oy >= a.ch) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rbase = s_fy[ty] - ry0;
    for (int j = 0; j < s_ny[ty]; ++j) {
        const float w = s_wy[ty][j];
        const float* t = s_tmp + ((rbase + j) * 16 + tx) * C;
        for (int c = 0; c < C; ++c) acc[c] = fmaf(w, t[c], acc[c]);
    }
    for (int c = 0; c < C; ++c)
        a.dst[(((int64_t)q * C + c) * a.ch + oy) * a.cw + ox] = (acc[c] - a.mean[c]) * a.inv_std[c];
}

__global__ void k_vit_to_bf16(const float* __restrict__ src, int64_t n, __bf16* __restrict__ dst) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) dst[t] = (__bf16)src[t];
}
// patch_embed.proj.weight [D][3 P P] fp32 -> [D][KP] bf16, zero padded
__global__ void k_vit_pad_rows(const float* __restrict__ src, int rows, int cols, int KP, __bf16* __restrict__ dst) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < rows * KP; t += gridDim.x * blockDim.x) {
        const int r = t / KP, c = t - r * KP;
        dst[t] = (__bf16)(c < cols ? src[r * cols + c] : 0.0f);
    }
}

// fp32 weights times `scale` (a power of two) -> fp16 hi / lo planes; rows of `cols` floats zero-padded to `KP` (patch embedding)
__global__ void k_vit_to_f16_planes(const float* __restrict__ src, int64_t rows, int cols, int KP, float scale, _Float16* __restrict__ hi,
                                    _Float16* __restrict__ lo) {
    const int64_t n = rows * KP;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / KP;
        const int c = (int)(t - r * KP);
        _Float16 h, l;
        split_h(c < cols ? src[r * cols + c] * scale : 0.0f, h, l);
        hi[t] = h; lo[t] = l;
    }
}

inline unsigned grid1(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <int EPI, int PREC, int WN, int WM, int AR, int BR, int NBUF>
void gemm_dma(const typename VT<PREC>::e* X, int64_t x_lo, const typename VT<PREC>::e* W, int64_t w_lo, int64_t M, int N, int K, const GemmEpi& e,
              hipStream_t s) {
    constexpr int BN = 32 * AR * WN, BM = 32 * BR * WM;
    hipLaunchKernelGGL((k_vit_gemm_dma<EPI, PREC, WN, WM, AR, BR, NBUF>), dim3((unsigned)(N / BN), (unsigned)((M + BM - 1) / BM)), dim3(64 * WN * WM), 0, s, X,
                       x_lo, W, w_lo, M, N, K, e);
}

// `form` (iff_vit_desc.gemm_form) names the kernels that multiply a batch of 24 images or more; every form returns the same bits
// (tests/test_hip_image_side.py::test_vit_gemm_forms_return_the_same_bits):
//   0 / 10  the product's choice: 128 x 128 tiles by LDS-DMA into two buffers, 256 x 256 for the MLP's first product
//   1       the register-staged 128 x 128 tiles (round 5's product)
//   2       128 x 128 by DMA for every product          3  128 features x 256 tokens over eight waves
//   4       form 3 + 256 x 256 for the MLP's first product
//   5 / 6   forms 2 / 3 with ONE buffer and twice the workgroups per CU       7 / 8 / 9   rings of 4 / 3 / 3 buffers (128^2, 128^2, 128 x 256)
// Measured, 32 images, image -> pose with 4 graphs in flight / one forward alone (profiles/r06_vit_gemm_forms.txt): 1: 12 490 images/s /
// 2.60 ms; 2: 12 630 / 2.55; 4: 12 810 / 2.88; 10: 12 830 / 2.56 -- none of the shapes moves the forward by more than 4 %, see NOTES.md.
template <int EPI, int PREC>
hipError_t gemm(int form, const void* Xv, int64_t x_lo, const void* Wv, int64_t w_lo, int64_t M, int N, int K, const GemmEpi& e, hipStream_t s) {
    typedef typename VT<PREC>::e ET;
    const ET* X = (const ET*)Xv;
    const ET* W = (const ET*)Wv;
    if (N % GBN != 0 || K % GBK != 0 || M < 1) return hipErrorInvalidValue;
    // 64-token tiles for small batches (with 128 a 4112-token batch -- 16 images -- leaves CUs idle or a single workgroup per CU);
    // from 24 images on 128-token tiles: each launch alone is 5-15 % slower, but the workgroups read 1/3 fewer operand bytes per flop
    // and with several batches in flight (the bench's four graphs) the total is 5 % faster (17 700 -> 18 700 images/s at 32 images)
    const bool wide = M >= 6144;       // (64-token tiles for the N = 384 products only: -1.7 % images/s)
    if (form == 0) form = 10;
    if (!wide)
        hipLaunchKernelGGL((k_vit_gemm<EPI, 64, PREC>), dim3((unsigned)(N / GBN), (unsigned)((M + 63) / 64)), dim3(256), 0, s, X, x_lo, W, w_lo, M, N, K, e);
    else if (form == 1)
        hipLaunchKernelGGL((k_vit_gemm<EPI, 128, PREC>), dim3((unsigned)(N / GBN), (unsigned)((M + 127) / 128)), dim3(256), 0, s, X, x_lo, W, w_lo, M, N, K, e);
    else if (form == 4 && EPI == EPI_GELU && N % 256 == 0)
        gemm_dma<EPI, PREC, 2, 4, 4, 2, 2>(X, x_lo, W, w_lo, M, N, K, e, s);
    else if (form == 3 || form == 4)
        gemm_dma<EPI, PREC, 2, 4, 2, 2, 2>(X, x_lo, W, w_lo, M, N, K, e, s);
    else if (form == 5)
        gemm_dma<EPI, PREC, 2, 2, 2, 2, 1>(X, x_lo, W, w_lo, M, N, K, e, s);
    else if (form == 6)
        gemm_dma<EPI, PREC, 2, 4, 2, 2, 1>(X, x_lo, W, w_lo, M, N, K, e, s);
    else if (form == 7)
        gemm_dma<EPI, PREC, 2, 2, 2, 2, 4>(X, x_lo, W, w_lo, M, N, K, e, s);
    else if (form == 8)
        gemm_dma<EPI, PREC, 2, 2, 2, 2, 3>(X, x_lo, W, w_lo, M, N, K, e, s);
    else if (form == 9)
        gemm_dma<EPI, PREC, 2, 4, 2, 2, 3>(X, x_lo, W, w_lo, M, N, K, e, s);
    else if (form == 10 && EPI == EPI_GELU && N % 256 == 0)
        gemm_dma<EPI, PREC, 2, 4, 4, 2, 2>(X, x_lo, W, w_lo, M, N, K, e, s);
    else
        gemm_dma<EPI, PREC, 2, 2, 2, 2, 2>(X, x_lo, W, w_lo, M, N, K, e, s);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_resize_crop(const float* src, int Q, int H, int W, int C, int mode, int rh, int rw, int top, int left, int ch, int cw, int cubic,
                              const float* mean, const float* std, float* dst, hipStream_t s) {
    if (Q < 1) return hipSuccess;
    ResizeArgs a;
    const int Cin = mode ? 4 : C;
    a.Cin = Cin; a.mode = mode;
    a.src = src; a.Q = Q; a.H = H; a.W = W; a.C = C; a.rh = rh; a.rw = rw; a.top = top; a.left = left; a.ch = ch; a.cw = cw; a.cubic = cubic;
    for (int c = 0; c < 4; ++c) { a.mean[c] = (mean && c < C) ? mean[c] : 0.0f; a.inv_std[c] = (std && c < C) ? 1.0f / std[c] : 1.0f; }
    a.dst = dst;
    // input rows one tile can touch: tr output rows apart by the scale factor, plus the filter support on both sides
    const float sy = (float)H / (float)rh, sup = (cubic ? 2.0f : 1.0f) * (sy >= 1.0f ? sy : 1.0f);
    const int tiles16 = ((cw + 15) / 16) * ((ch + 15) / 16);
    const int tr = (int64_t)tiles16 * Q < 256 ? 1 : 16;
    const int max_rows = (int)((float)(tr - 1) * sy + 2.0f * sup) + 4;
    // ... and input columns likewise; the rows are staged `rch` at a time (about 12 KB of LDS)
    const float sx = (float)W / (float)rw, supx = (cubic ? 2.0f : 1.0f) * (sx >= 1.0f ? sx : 1.0f);
    const int max_cols = (int)(15.0f * sx + 2.0f * supx) + 4;
    int rch = 3072 / (max_cols * Cin);
    rch = rch < 1 ? 1 : (rch > max_rows ? max_rows : rch);
    const size_t lds = ((size_t)max_rows * 16 * C + (size_t)rch * max_cols * Cin) * sizeof(float);
    if (lds > 60 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_resize_crop, dim3((unsigned)((cw + 15) / 16), (unsigned)((ch + tr - 1) / tr), (unsigned)Q), dim3(256), lds, s, a, max_rows,
                       max_cols, rch, tr);
    return hipGetLastError();
}

hipError_t launch_vit_to_bf16(const float* src, int64_t n, void* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_vit_to_bf16, dim3(grid1(n)), dim3(256), 0, s, src, n, (__bf16*)dst);
    return hipGetLastError();
}
hipError_t launch_vit_pad_rows(const float* src, int rows, int cols, int KP, void* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_vit_pad_rows, dim3(grid1((int64_t)rows * KP)), dim3(256), 0, s, src, rows, cols, KP, (__bf16*)dst);
    return hipGetLastError();
}

hipError_t launch_vit_to_f16_planes(const float* src, int64_t rows, int cols, int KP, float scale, void* hi, void* lo, hipStream_t s) {
    hipLaunchKernelGGL(k_vit_to_f16_planes, dim3(grid1(rows * KP)), dim3(256), 0, s, src, rows, cols, KP, scale, (_Float16*)hi, (_Float16*)lo);
    return hipGetLastError();
}

size_t vit_workspace_bytes(const VitDev& v, int Q) {
    const size_t M = (size_t)Q * v.T, G = (size_t)Q * (v.T - 1);
    const size_t eb = v.prec ? 4 : 2;           // bytes per operand element: bf16, or fp16 hi + lo planes
    size_t b = 0;
    auto take = [&](size_t bytes) { b += (bytes + 255) / 256 * 256; };
    take(M * v.dim * 4);            // x      residual stream, fp32
    take(M * v.dim * eb);           // xn     LayerNorm output / attention output
    take(M * v.dim * 2 * eb);       // q, k per head
    take((size_t)Q * v.heads * 64 * AT_TP * eb);     // v per head, transposed, rows of AT_TP keys
    take(M * v.mlp * eb);           // MLP hidden
    take(G * v.kp * eb);            // im2col
    return b;
}

template <int PREC>
static hipError_t vit_forward(const VitDev& v, const float* images, int Q, int H, int W, float* patch_tokens, float* cls_opt, void* ws,
                              hipStream_t s) {
    typedef typename VT<PREC>::e ET;
    const int64_t M = (int64_t)Q * v.T, G = (int64_t)Q * (v.T - 1);
    const int D = v.dim, L = v.depth;
    const size_t eb = PREC ? 4 : 2;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) / 256 * 256; return r; };
    float* x = (float*)take((size_t)M * D * 4);
    ET* xn = (ET*)take((size_t)M * D * eb);                    // PREC 1: [2][M][D], the lo plane behind the hi plane (likewise below)
    ET* q = (ET*)take((size_t)M * D * 2 * eb);                 // q | k (| q lo | k lo)
    ET* k = q + M * D;
    ET* vv = (ET*)take((size_t)Q * v.heads * 64 * AT_TP * eb);
    ET* hid = (ET*)take((size_t)M * v.mlp * eb);
    ET* col = (ET*)take((size_t)G * v.kp * eb);
    const int64_t xn_lo = M * D, qk_lo = 2 * M * D, v_lo = (int64_t)Q * v.heads * 64 * AT_TP, hid_lo = M * v.mlp, col_lo = G * v.kp;
    // weight planes: the lo plane of a stacked weight sits the whole stack behind its hi plane
    const int64_t w_patch_lo = (int64_t)D * v.kp, w_qkv_lo = (int64_t)L * 3 * D * D, w_proj_lo = (int64_t)L * D * D, w_fc_lo = (int64_t)L * v.mlp * D;
    hipError_t e;
    hipLaunchKernelGGL((k_vit_im2col<PREC>), dim3(grid1(G * v.kp)), dim3(256), 0, s, images, Q, H, W, v.patch, v.gh, v.gw, v.kp, (void*)col);
    hipLaunchKernelGGL(k_vit_cls, dim3(grid1((int64_t)Q * D)), dim3(256), 0, s, v.cls, v.pos, Q, v.T, D, x);
    GemmEpi ep = {};
    ep.bias = v.patch_b; ep.x = x; ep.pos = v.pos; ep.G = v.T - 1; ep.T = v.T; ep.wscale = v.s_patch;
    if ((e = gemm<EPI_EMBED, PREC>(v.gemm_form, col, col_lo, v.patch_w, w_patch_lo, G, D, v.kp, ep, s)) != hipSuccess) return e;
    const unsigned ln_grid = (unsigned)((M + 3) / 4);
    constexpr int LN_OUT = PREC ? 2 : 1;
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL((k_vit_layernorm<LN_OUT>), dim3(ln_grid), dim3(256), 0, s, x, M, v.T, v.ln1_w + (size_t)l * D, v.ln1_b + (size_t)l * D,
                           v.eps, (void*)xn, nullptr, nullptr);
        GemmEpi e0 = {};
        e0.bias = v.qkv_b + (size_t)l * 3 * D; e0.q = q; e0.k = k; e0.v = vv; e0.T = v.T; e0.heads = v.heads; e0.qscale = 0.125f;
        e0.qk_lo = qk_lo; e0.v_lo = v_lo; e0.wscale = v.s_qkv[l];
        if ((e = gemm<EPI_QKV, PREC>(v.gemm_form, xn, xn_lo, (const ET*)v.qkv_w + (size_t)l * 3 * D * D, w_qkv_lo, M, 3 * D, D, e0, s)) != hipSuccess) return e;
        if (PREC)
            hipLaunchKernelGGL(k_vit_attention_x2, dim3((unsigned)((v.T + 127) / 128), (unsigned)(Q * v.heads)), dim3(256), 0, s, (const _Float16*)q,
                               (const _Float16*)k, qk_lo, (const _Float16*)vv, v_lo, v.T, v.heads, (_Float16*)xn, xn_lo);
        else
            hipLaunchKernelGGL(k_vit_attention, dim3((unsigned)((v.T + 127) / 128), (unsigned)(Q * v.heads)), dim3(256), 0, s, (const __bf16*)q,
                               (const __bf16*)k, (const __bf16*)vv, v.T, v.heads, (__bf16*)xn);
        GemmEpi e1 = {};
        e1.bias = v.proj_b + (size_t)l * D; e1.x = x; e1.ls = v.ls1 + (size_t)l * D; e1.wscale = v.s_proj[l];
        if ((e = gemm<EPI_RESID, PREC>(v.gemm_form, xn, xn_lo, (const ET*)v.proj_w + (size_t)l * D * D, w_proj_lo, M, D, D, e1, s)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_vit_layernorm<LN_OUT>), dim3(ln_grid), dim3(256), 0, s, x, M, v.T, v.ln2_w + (size_t)l * D, v.ln2_b + (size_t)l * D,
                           v.eps, (void*)xn, nullptr, nullptr);
        GemmEpi e2 = {};
        e2.bias = v.fc1_b + (size_t)l * v.mlp; e2.out = hid; e2.wscale = v.s_fc1[l];
        if ((e = gemm<EPI_GELU, PREC>(v.gemm_form, xn, xn_lo, (const ET*)v.fc1_w + (size_t)l * v.mlp * D, w_fc_lo, M, v.mlp, D, e2, s)) != hipSuccess) return e;
        GemmEpi e3 = {};
        e3.bias = v.fc2_b + (size_t)l * D; e3.x = x; e3.ls = v.ls2 + (size_t)l * D; e3.wscale = v.s_fc2[l];
        if ((e = gemm<EPI_RESID, PREC>(v.gemm_form, hid, hid_lo, (const ET*)v.fc2_w + (size_t)l * D * v.mlp, w_fc_lo, M, D, v.mlp, e3, s)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_vit_layernorm<0>), dim3(ln_grid), dim3(256), 0, s, x, M, v.T, v.norm_w, v.norm_b, v.eps, nullptr, patch_tokens, cls_opt);
    return hipGetLastError();
}

hipError_t launch_vit_forward(const VitDev& v, const float* images, int Q, int H, int W, float* patch_tokens, float* cls_opt, void* ws,
                              size_t ws_bytes, hipStream_t s) {
    if (Q < 1) return hipSuccess;
    if (ws_bytes < vit_workspace_bytes(v, Q) || v.dim != 384 || v.heads * 64 != v.dim || v.T > AT_TP || H != v.gh * v.patch ||
        W != v.gw * v.patch || v.depth > VIT_MAX_DEPTH)
        return hipErrorInvalidValue;
    return v.prec ? vit_forward<1>(v, images, Q, H, W, patch_tokens, cls_opt, ws, s) : vit_forward<0>(v, images, Q, H, W, patch_tokens, cls_opt, ws, s);
}
