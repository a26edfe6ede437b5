// trunk_f16_kernels.hip -- the fused ray encoder + attention logits (k5_trunk_h) on the fp16 matrix cores with a TWO-term
// split of every fp32 operand: IFF_GEMM_F16X2.
//
// Reference: pose_estimation/ray_preprocessor.py:29-39 (three ReLU layers), multihead_attention.py:4-12,60-61 (folded
// into one token-side Linear, api.hip fold_heads).  Same work split as k5_trunk (identify_kernels.hip): a workgroup owns
// TR = 32 RG rays, a wave 32 FG output features of every layer; activations stay in LDS between the layers and are the
// MFMA's B operand, weights stream from L2 in fragment order as the A operand; the logits are "layer 4" with swapped
// operands so that the softmax statistics run down a lane's registers.
//
// Arithmetic.  fp16 carries 11 significant bits, so a = hi + lo with hi = fp16(a), lo = fp16(a - hi) represents a to
// 2^-22 relative (as long as lo stays a normal or subnormal fp16: the MFMA keeps fp16 subnormals in the default float mode),
// and a*b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b drops only lo_a lo_b <= 2^-22 |a b|: three v_mfma_f32_32x32x16_f16 per
// product block where the bf16 split needs six for the same fp32-class accuracy (measured against an fp64 evaluation of the
// chain: rms logit error 4.0e-6 vs 4.0e-6 for a plain fp32 matmul chain, max 3.9e-5 vs 5.1e-5 for the reference's own fp32
// CPU run -- tests/test_hip_identify.py).  fp16 has a 5-bit exponent, so every operand is first multiplied by a power of
// two chosen at iff_idnet_create (api.hip plan_f16_scales) from worst-case bounds on the activations (row L1 norms of the
// weights x the previous layer's bound), so that nothing can overflow 65504 and the lo terms of typical values stay normal;
// the fp32 accumulator is multiplied back by the exact inverse power of two.  The token side (folded queries) is scaled
// per token row at run time (k_qf_frag_h).  If the bounds leave fewer than ~3 bits of headroom the handle falls back to
// the 3xBF16 kernel (iff_idnet_gemm_mode reports what runs).
#include "iff_device.h"
#include "iff_launch.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int HSLD = 264;          // halves per LDS row: 256 + 8 -> 528 B, an odd multiple of 16 B (conflict-free 16-B reads)
constexpr int HC = 256;            // feature_c this kernel is built for
constexpr float H_MAX = 65504.0f;

__device__ inline void split_h(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// Four accumulator values -> the hi/lo halves of min(relu(acc inv + bias) s, H_MAX), with `is` = inv s and `bs` = bias s (s is a
// power of two: relu(fma(acc, inv, bias)) s == relu(fma(acc, inv s, bias s)) bit for bit in the normal range).  Written on
// register pairs: v_med3_f32 is the relu and the clamp in one, v_cvt_pk_f16_f32 converts two values per instruction; the pair-wise
// fma / subtract compile to two scalar instructions each, not to v_pk_fma_f32 / v_pk_add_f32 -- the library is built without packed
// fp32 arithmetic (iffnerf_amd/build.py, fan_common.h; tests/test_isa_rules.py holds the line).
__device__ __forceinline__ void relu_split_quad(float a0, float a1, float a2, float a3, f32x2 is, f32x2 bs01, f32x2 bs23, f16x4& hi, f16x4& lo) {
    f32x2 t01 = __builtin_elementwise_fma(f32x2{a0, a1}, is, bs01);
    f32x2 t23 = __builtin_elementwise_fma(f32x2{a2, a3}, is, bs23);
    t01.x = __builtin_amdgcn_fmed3f(t01.x, 0.0f, H_MAX); t01.y = __builtin_amdgcn_fmed3f(t01.y, 0.0f, H_MAX);
    t23.x = __builtin_amdgcn_fmed3f(t23.x, 0.0f, H_MAX); t23.y = __builtin_amdgcn_fmed3f(t23.y, 0.0f, H_MAX);
    const f16x2 h01 = __builtin_convertvector(t01, f16x2), h23 = __builtin_convertvector(t23, f16x2);
    const f32x2 r01 = t01 - __builtin_convertvector(h01, f32x2), r23 = t23 - __builtin_convertvector(h23, f32x2);
    const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
    hi = f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = f16x4{l01.x, l01.y, l23.x, l23.y};
}

// ------------------------------------------------------------------------------------------------ weight prep
// nn.Linear weight W [256][ld] (columns col0 .. col0+ncols-1), times `scale` (a power of two) -> fragment order
// Wf[ks][plane][nb][lane][8] halves, plane 0 = hi, 1 = lo: lane l of the (nb, ks) fragment holds
// W[32 nb + (l & 31)][col0 + 16 ks + 8 (l >> 5) + 0..7], the 32x32x16 A-operand map; columns beyond ncols are zero
__global__ void k_frag_order_h(const float* __restrict__ W, int ld, int col0, int ncols, int nks, float scale,
                               _Float16* __restrict__ Wf) {
    const int64_t n = (int64_t)nks * 8 * 64 * 8;            // elements per plane... per (ks): 2 planes x 8 nb x 64 lanes x 8
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t & 7), lane = (int)((t >> 3) & 63), nb = (int)((t >> 9) & 7), ks = (int)(t >> 12);
        const int row = nb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + i;
        const float v = (k < ncols) ? W[(size_t)row * ld + col0 + k] * scale : 0.0f;
        _Float16 hi, lo;
        split_h(v, hi, lo);
        const size_t base = (((size_t)ks * 2) * 8 + nb) * 512 + lane * 8 + i;
        Wf[base] = hi;
        Wf[base + 8 * 512] = lo;
    }
}
hipError_t launch_frag_order_h(const float* W, int ld, int col0, int ncols, int nks, float scale, void* Wf, hipStream_t s) {
    const int64_t n = (int64_t)nks * 8 * 64 * 8;
    hipLaunchKernelGGL(k_frag_order_h, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, ld, col0, ncols, nks, scale, (_Float16*)Wf);
    return hipGetLastError();
}

// folded query rows qf [M][ld] (iff_q_fold; columns 0..255) -> per-token power-of-two scale, hi/lo planes in fragment
// order, one 256-token block after the other: Qf[tb][ks][plane][nb][lane][8], token = 256 tb + 32 nb + (lane & 31),
// feature = 16 ks + 8 (lane >> 5) + i; tokens beyond M are zero.  qscale[token] = 2^-(e + e3): what the logits epilogue
// multiplies the accumulator with (e: this token's exponent, e3: the exponent the h3 planes were scaled by).
// One workgroup per 32 tokens: thread = (token, 8-feature chunk group), 8 threads per token.
__global__ void __launch_bounds__(256) k_qf_frag_h(const float* __restrict__ qf, int ld, int M, int n_tb, int e3,
                                                   _Float16* __restrict__ Qf, float* __restrict__ qscale) {
    const int tid = threadIdx.x, tl = tid >> 3, sub = tid & 7;
    const int tok = blockIdx.x * 32 + tl;                      // token of this query
    const int tb = tok >> 8, nb = (tok >> 5) & 7;
    const size_t plane_elems = (size_t)n_tb * (HC / 16) * 2 * 8 * 64 * 8;   // elements of one query's Qf (both planes)
    qf += (size_t)blockIdx.y * M * ld;
    Qf += (size_t)blockIdx.y * plane_elems;
    qscale += (size_t)blockIdx.y * n_tb * 256;
    const bool ok = tok < M;
    float v[4][8];
    float amax = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = sub + 8 * j;                              // 8-feature chunk: features 8c .. 8c+7
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[j][i] = ok ? qf[(size_t)tok * ld + 8 * c + i] : 0.0f;
            amax = fmaxf(amax, fabsf(v[j][i]));
        }
    }
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    // amax = m 2^ex with m in [0.5, 1): scaling by 2^(15 - ex) puts the row's largest magnitude in [16384, 32768)
    int ex = 0;
    if (amax > 0.0f && amax < INFINITY) (void)frexpf(amax, &ex);
    int e = 15 - ex;
    e = e > 40 ? 40 : (e < -40 ? -40 : e);
    const float sc = ldexpf(1.0f, e);
    if (sub == 0) qscale[tok] = ldexpf(1.0f, -(e + e3));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = sub + 8 * j, ks = c >> 1, h = c & 1;
        f16x8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            _Float16 a, b;
            split_h(fminf(fmaxf(v[j][i] * sc, -H_MAX), H_MAX), a, b);
            hi[i] = a; lo[i] = b;
        }
        const size_t base = ((((size_t)tb * (HC / 16) + ks) * 2) * 8 + nb) * 512 + ((tok & 31) + 32 * h) * 8;
        *reinterpret_cast<f16x8*>(Qf + base) = hi;
        *reinterpret_cast<f16x8*>(Qf + base + 8 * 512) = lo;
    }
}

// The encoder input of one 64-ray tile, by eight waves (ray_preprocessor.py:30-37, tensorBase.py:14-20):
// x = [o, d, rgb, PE(o,8), PE(d,8), PE(rgb,6)] (141 columns, zero-padded to 160; rows beyond N are zero) as hi/lo planes in LDS.
// lane = ray; wave w < 8 = source component w of [o.xyz, d.xyz, rgb.xy]: its raw column and its F frequencies, unrolled -- no
// per-item decoding, the column arithmetic is scalar; rgb.z (the 66 = 8 x 8 + 2 frequency pairs do not divide by eight) goes
// three frequencies each to waves 6 and 7, whose own components have F = 6: every wave evaluates 8 or 9 sincosf (one argument
// reduction serves the sin and the cos column).  Values are those of the per-item loop this replaces (same arguments, same sincosf).
// Measured with the stage stamps (-DTRUNK_STAMPS): the encoder input was 20 % of a tile's time in the item loop.
__device__ __forceinline__ void encode_tile64(_Float16 (*S)[64][HSLD], const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                              const float* __restrict__ ray_c, int64_t row0, int64_t N, float sx, int wave, int lane) {
    auto put = [&](int col, float v) {
        _Float16 hi, lo;
        split_h(__builtin_amdgcn_fmed3f(v * sx, -H_MAX, H_MAX), hi, lo);
        S[0][lane][col] = hi; S[1][lane][col] = lo;
    };
    const int64_t gr = row0 + lane;
    const bool ok = gr < N;
    const int blk = wave < 3 ? 0 : (wave < 6 ? 1 : 2), j = wave - 3 * blk;          // scalar
    const float* __restrict__ base = blk == 0 ? ray_o : (blk == 1 ? ray_d : ray_c);
    const float v = ok ? base[3 * gr + j] : 0.0f;
    const float vz = (ok && wave >= 6) ? ray_c[3 * gr + 2] : 0.0f;
    put(wave, v);
    if (wave == 7) put(8, vz);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
        if (wave + 8 * pc < 19) {
            S[0][lane][141 + wave + 8 * pc] = (_Float16)0.0f; S[1][lane][141 + wave + 8 * pc] = (_Float16)0.0f;
        }
    if (wave < 6) {
        const int col0 = 9 + 48 * blk + 8 * j;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float sv, cv;
            sincosf(v * (float)(1 << k), &sv, &cv);
            put(col0 + k, sv);
            put(col0 + 24 + k, cv);
        }
    } else {
        const int col0 = 105 + 6 * j;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float sv, cv;
            sincosf(v * (float)(1 << k), &sv, &cv);
            put(col0 + k, sv);
            put(col0 + 18 + k, cv);
        }
        const int k0 = 3 * (wave - 6);
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            float sv, cv;
            sincosf(vz * (float)(1 << (k0 + kk)), &sv, &cv);
            put(105 + 12 + k0 + kk, sv);
            put(105 + 18 + 12 + k0 + kk, cv);
        }
    }
}

// ------------------------------------------------------------------------------------------------ the fused kernel
struct TrunkHArgs {
    const float* ray_o; const float* ray_d; const float* ray_c; int64_t N;
    const uint4* W1; const uint4* W2; const uint4* W3h; const uint4* W3x;      // fragment-ordered fp16 hi/lo planes
    const float* b1; const float* b2; const float* b3;
    float sx;                 // x is multiplied by sx before the split
    float inv1, s1;           // h1 = relu(acc inv1 + b1); its planes hold h1 s1
    float inv2, s2;
    float inv3, s3;           // layer 3: the h-part and the x-part accumulate at one common scale
    float* h3;                // MODE 0: [N][256] fp32 out
    _Float16* planes;         // MODE 2 (out) / MODE 3 (in): the h3 hi/lo planes [2][N][256], times s3 -- the per-model cache
    const uint4* Qf; const float* qscale; const float* rowc; int rowc_ld; int M; float divisor;
    float* logits; float2* part; int Mpad;
    const int* rows;          // MODE 3, optional: kept token rows per 256-token block (kept rows first); the rows behind them are skipped
};

template <int FG> struct WFragH { f16x8 p[FG][2]; };      // [feature group][hi, lo]

// NP = products per block: 3 (hi hi + hi lo + lo hi: IFF_GEMM_F16X2, the fp32 class) or 1 (hi hi only: IFF_GEMM_F16X1, the lo planes are
// neither loaded nor multiplied)
template <int FG, int NP = 3>
__device__ inline void trunk_load_w_h(WFragH<FG>& w, const uint4* __restrict__ Wf, int ks, int wave, int lane) {
#pragma unroll
    for (int pl = 0; pl < (NP == 1 ? 1 : 2); ++pl)
#pragma unroll
        for (int fg = 0; fg < FG; ++fg) {
            // uniform base (scalar registers, advanced by scalar adds) + one 32-bit lane offset: no vector address arithmetic
            typedef const __attribute__((address_space(1))) char* gchar_p;          // stays a GLOBAL pointer through the asm
            typedef unsigned u32q __attribute__((ext_vector_type(4)));              // (HIP's uint4 class cannot be read through it)
            typedef const __attribute__((address_space(1))) u32q* guint4_p;
            gchar_p base = (gchar_p)(Wf + ((size_t)ks * 2 + pl) * 8 * 64);
            asm("" : "+s"(base));          // pinned in scalar registers: LLVM would re-associate the constant onto the lane part
            const unsigned voff = (unsigned)(((FG * wave + fg) * 64 + lane) * 16);
            u32q v = *(guint4_p)(base + voff);
            w.p[fg][pl] = *reinterpret_cast<f16x8*>(&v);
        }
}

// acc[fg][rg] += W(fg) * act(rg) over one 16-wide k-step: lo*hi, hi*lo, hi*hi (smallest contributions first)
template <int FG, int RG, int NP = 3>
__device__ inline void trunk_mfma_h(f32x16 (&acc)[FG][RG], const WFragH<FG>& w, const f16x8 (&a)[RG][2]) {
#pragma unroll
    for (int fg = 0; fg < FG; ++fg)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            if (NP == 3) {
                acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.p[fg][1], a[rg][0], acc[fg][rg], 0, 0, 0);
                acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.p[fg][0], a[rg][1], acc[fg][rg], 0, 0, 0);
            }
            acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.p[fg][0], a[rg][0], acc[fg][rg], 0, 0, 0);
        }
}
// the logits tile: rows = rays, columns = tokens (operands swapped)
template <int FG, int RG, int NP = 3>
__device__ inline void trunk_mfma_ht(f32x16 (&acc)[FG][RG], const WFragH<FG>& w, const f16x8 (&a)[RG][2]) {
#pragma unroll
    for (int tg = 0; tg < FG; ++tg)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            if (NP == 3) {
                acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rg][0], w.p[tg][1], acc[tg][rg], 0, 0, 0);
                acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rg][1], w.p[tg][0], acc[tg][rg], 0, 0, 0);
            }
            acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rg][0], w.p[tg][0], acc[tg][rg], 0, 0, 0);
        }
}

struct __attribute__((packed, aligned(4))) f4uh { float x, y, z, w; };     // 16-byte store at 4-byte alignment

// The epilogue of one logits tile of one wave: acc[rg][r] (rows = rays row0 + 32 rg + (r & 3) + 8 (r >> 2) + 4 lh, this lane's
// column = one token) -> logits (acc qs + rc) / divisor into `dst_row` (= logits + token N), and the tile's softmax partial
// (max, sum exp) of that token over the rays < N.  (acc qs + rc) / divisor is correctly rounded (qs is a power of two, so acc qs is
// exact): quotient estimate by the reciprocal, exact remainder, one correction.  Tiles that lie wholly inside N (all but the last)
// take the pair-wise form (register pairs, no per-ray bound checks; scalar fma / mul under the library's no-packed-fp32 build); the
// values, the maximum and the ORDER of the sum are those of the masked form, so a row's statistics do not depend on which form
// served a tile.
template <int RG>
__device__ __forceinline__ float2 logits_tile_epilogue(f32x16 (&acc)[RG], float qs, float rc, float divisor, float inv_div, bool tok_ok,
                                              float* __restrict__ dst_row, int64_t row0, int64_t N, int lh) {
    float vmax = -INFINITY, ssum = 0.0f;
    if (row0 + 32 * RG <= N) {                                  // wave-uniform
        const f32x2 qs2 = {qs, qs}, rc2 = {rc, rc}, dv2 = {divisor, divisor}, id2 = {inv_div, inv_div};
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 num = __builtin_elementwise_fma(f32x2{acc[rg][r], acc[rg][r + 1]}, qs2, rc2);
                const f32x2 q1 = num * id2;
                const f32x2 v = __builtin_elementwise_fma(__builtin_elementwise_fma(-q1, dv2, num), id2, q1);
                acc[rg][r] = v.x; acc[rg][r + 1] = v.y;
                vmax = fmaxf(fmaxf(vmax, v.x), v.y);             // v_max3_f32
            }
        vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
        const f32x2 vm2 = {vmax, vmax};
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 d01 = f32x2{acc[rg][4 * q], acc[rg][4 * q + 1]} - vm2, d23 = f32x2{acc[rg][4 * q + 2], acc[rg][4 * q + 3]} - vm2;
                ssum += __expf(d01.x); ssum += __expf(d01.y); ssum += __expf(d23.x); ssum += __expf(d23.y);
                if (tok_ok) {
                    f4uh o4 = {acc[rg][4 * q], acc[rg][4 * q + 1], acc[rg][4 * q + 2], acc[rg][4 * q + 3]};
                    *reinterpret_cast<f4uh*>(dst_row + row0 + 32 * rg + 8 * q + 4 * lh) = o4;
                }
            }
    } else {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t ray = row0 + 32 * rg + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float num = fmaf(acc[rg][r], qs, rc);
                const float q1 = num * inv_div;
                const float v = fmaf(fmaf(-q1, divisor, num), inv_div, q1);
                acc[rg][r] = v;
                vmax = fmaxf(vmax, ray < N ? v : -INFINITY);
            }
        vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t ray0 = row0 + 32 * rg + 8 * q + 4 * lh;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ray0 + i < N) ssum += __expf(acc[rg][4 * q + i] - vmax);   // denominator only: <= 4e-6 relative, common to the whole token row
                if (tok_ok) {
                    float* dst = dst_row + ray0;
                    if (ray0 + 3 < N) {
                        f4uh o4 = {acc[rg][4 * q], acc[rg][4 * q + 1], acc[rg][4 * q + 2], acc[rg][4 * q + 3]};
                        *reinterpret_cast<f4uh*>(dst) = o4;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (ray0 + i < N) dst[i] = acc[rg][4 * q + i];
                    }
                }
            }
    }
    ssum += __shfl_xor(ssum, 32, 64);
    return make_float2(vmax, ssum);
}

// FG = 32-feature groups per wave (8 / FG waves per workgroup), RG = 32-ray groups per workgroup (TR = 32 RG rays).
// MODE 0: rays -> h3 [N][256] fp32 (iff_ray_trunk).  MODE 1: rays -> logits + softmax partials (the fused per-query launch).
// MODE 2: rays -> the h3 hi/lo planes in HBM (iff_ray_cache_build: the encoder once per resident ray set).
// MODE 3: cached planes -> logits + softmax partials (iff_logits_from_cache: every later query batch skips the encoder).
// MODE_ = MODE + 4: the same launch with ONE fp16 product per block (IFF_GEMM_F16X1: 11 significant bits per operand, a third of the
// matrix work, half the weight stream -- a throughput class that is NOT the reference's accuracy class; never the default).
template <int MODE_, int FG, int RG>
__global__ void __launch_bounds__(64 * 8 / FG, (FG == 1 && RG == 2) ? 4 : 2) k5_trunk_h(TrunkHArgs a) {
    constexpr int MODE = MODE_ & 3, NP = (MODE_ & 4) ? 1 : 3;
    constexpr bool LOGITS = MODE == 1 || MODE == 3;
    constexpr int TR = 32 * RG, NT = 64 * 8 / FG, NWAVE = 8 / FG;
    __shared__ __attribute__((aligned(16))) _Float16 S[2][TR][HSLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * TR;
    const int64_t N = a.N;
#ifdef TRUNK_STAMPS
    // diagnostic build only: wave w stores the low word of s_memtime at stage boundary k over logits[32 w + k][row0] (of query 0's
    // first token block) once the tile is done -- scripts/trunk_stamps.py reads the per-stage timeline of every tile out of them
    uint32_t stamp[16];
#define TSTAMP(k) stamp[k] = (uint32_t)__builtin_amdgcn_s_memtime()
#else
#define TSTAMP(k) do { } while (0)
#endif
    TSTAMP(0);
    {   // blockIdx.y = query of a batch: its own rays [N,3], folded query planes, logits [M,N] and partials
        const size_t qb = blockIdx.y;
        a.ray_o += qb * N * 3; a.ray_d += qb * N * 3; a.ray_c += qb * N * 3;
        if (MODE == 2 || MODE == 3) a.planes += qb * 2 * (size_t)N * HC;
        if (LOGITS) {
            a.Qf += qb * (size_t)(a.Mpad / 256) * (HC / 16) * 2 * 8 * 64;
            a.qscale += qb * (size_t)a.Mpad;
            a.rowc += qb * (size_t)a.M * a.rowc_ld;
            a.logits += qb * (size_t)a.M * N;
            a.part += qb * (size_t)((N + 63) / 64) * a.Mpad;       // partials are per 64-RAY BLOCK whatever the tile (below)
        } else if (MODE == 0) {
            a.h3 += qb * N * HC;
        }
    }
    if (MODE == 3) {
        // the cached planes of this workgroup's rays straight into LDS (rows of 512 B, 16 B per lane; rows beyond N are zero)
        for (int t = tid; t < 2 * TR * (HC / 8); t += NT) {
            const int pl = t / (TR * (HC / 8)), rem = t - pl * (TR * (HC / 8)), ray = rem / (HC / 8), ch = rem - ray * (HC / 8);
            const int64_t gr = row0 + ray;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (gr < N) v = *reinterpret_cast<const uint4*>(a.planes + ((size_t)pl * N + gr) * HC + 8 * ch);
            *reinterpret_cast<uint4*>(&S[pl][ray][8 * ch]) = v;
        }
        __syncthreads();
    }

    if (MODE != 3) {
        // encoder input x (ray_preprocessor.py:30-37, tensorBase.py:14-20) straight into LDS as hi/lo planes:
        // x = [o, d, rgb, PE(o,8), PE(d,8), PE(rgb,6)] (141 columns, zero-padded to 160; rows beyond N are zero).
        // Work items per ray: 66 (source component, frequency) pairs -- one sincosf serves the sin and the cos column --
        // plus the 9 raw values and the 19 pad columns: 94 items x TR rays over NT threads.
        if constexpr (NT == 512 && TR == 64) {
            encode_tile64(S, a.ray_o, a.ray_d, a.ray_c, row0, N, a.sx, __builtin_amdgcn_readfirstlane(wave), lane);
        } else {
            const float sx = a.sx;
            auto put = [&](int ray, int col, float v) {
                _Float16 hi, lo;
                split_h(fminf(fmaxf(v * sx, -H_MAX), H_MAX), hi, lo);
                S[0][ray][col] = hi; S[1][ray][col] = lo;
            };
            const int ray = tid % TR;
            const int64_t gr = row0 + ray;
            const bool ok = gr < N;
            float src[9];
    #pragma unroll
            for (int c = 0; c < 3; ++c) {
                src[c] = ok ? a.ray_o[3 * gr + c] : 0.0f;
                src[3 + c] = ok ? a.ray_d[3 * gr + c] : 0.0f;
                src[6 + c] = ok ? a.ray_c[3 * gr + c] : 0.0f;
            }
            for (int item = tid / TR; item < 94; item += NT / TR) {       // wave-uniform item -> no divergence
                if (item < 66) {
                    // blocks: PE(o) at column 9, PE(d) at 57, PE(rgb) at 105; each [sin (F*3) | cos (F*3)], component-major
                    int blk = item < 24 ? 0 : (item < 48 ? 1 : 2);
                    int b = item - 24 * blk;
                    int F = blk == 2 ? 6 : 8;
                    int j = b / F, k = b - j * F;
                    // select instead of src[3 * blk + j]: a dynamically indexed local array would live in scratch memory
                    const float c0 = blk == 0 ? src[0] : (blk == 1 ? src[3] : src[6]);
                    const float c1 = blk == 0 ? src[1] : (blk == 1 ? src[4] : src[7]);
                    const float c2 = blk == 0 ? src[2] : (blk == 1 ? src[5] : src[8]);
                    float arg = (j == 0 ? c0 : (j == 1 ? c1 : c2)) * (float)(1 << k);
                    float sv, cv;
                    sincosf(arg, &sv, &cv);               // one argument reduction for both columns
                    int col = 9 + 48 * blk + b;
                    put(ray, col, sv);
                    put(ray, col + 3 * F, cv);
                } else if (item < 75) {
                    const int ci = item - 66;
                    float rv = src[0];
    #pragma unroll
                    for (int u = 1; u < 9; ++u) rv = (ci == u) ? src[u] : rv;
                    put(ray, ci, rv);
                } else {
                    put(ray, 141 + (item - 75), 0.0f);
                }
            }
        }
        __syncthreads();
        TSTAMP(1);
    }

    auto load_act = [&](f16x8 (&v)[RG][2], int ks) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int pl = 0; pl < (NP == 1 ? 1 : 2); ++pl) v[rg][pl] = *reinterpret_cast<const f16x8*>(&S[pl][32 * rg + lr][16 * ks + 8 * lh]);
    };
    // relu(acc inv + bias) s -> hi/lo planes of this wave's 32 FG features for all TR rays
    auto write_planes = [&](const f32x16 (&acc)[FG][RG], const float* __restrict__ bias, float inv, float s) {
#pragma unroll
        for (int fg = 0; fg < FG; ++fg)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 32 * FG * wave + 32 * fg + 8 * q + 4 * lh;     // features f0..f0+3 <- registers 4q..4q+3
                const float4 bv = *reinterpret_cast<const float4*>(bias + f0);
                const f32x2 is = {inv * s, inv * s}, bs01 = {bv.x * s, bv.y * s}, bs23 = {bv.z * s, bv.w * s};
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) {
                    f16x4 p0, p1;
                    relu_split_quad(acc[fg][rg][4 * q], acc[fg][rg][4 * q + 1], acc[fg][rg][4 * q + 2], acc[fg][rg][4 * q + 3], is, bs01, bs23, p0, p1);
                    *reinterpret_cast<f16x4*>(&S[0][32 * rg + lr][f0]) = p0;
                    *reinterpret_cast<f16x4*>(&S[1][32 * rg + lr][f0]) = p1;
                }
            }
    };
    auto zero = [](f32x16 (&acc)[FG][RG]) {
#pragma unroll
        for (int x = 0; x < FG; ++x)
#pragma unroll
            for (int y = 0; y < RG; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
    };

    f32x16 acc[FG][RG], acc3[FG][RG];
    zero(acc); zero(acc3);
    int wsel = wave;          // whose 32 FG weight rows (encoder: features; logits: tokens) this wave streams: the wave's own, except below
    constexpr int KX = (141 + 15) / 16, KH = HC / 16;

    // One k loop, software-pipelined in registers: the weight fragments of k-step ks + DEPTH - 1 are requested before
    // k-step ks is multiplied.  DUAL: two weight streams over the same activations (layer 1 and the x-part of layer 3).
    constexpr int AB = (FG == 1 && RG == 2) ? 1 : 2;
    auto phase = [&](auto nk_c, auto depth_c, auto dual_c, auto swap_c, f32x16 (&accA)[FG][RG], const uint4* __restrict__ WA,
                     f32x16 (&accB)[FG][RG], const uint4* __restrict__ WB) {
        constexpr int NK = decltype(nk_c)::value, DEPTH = decltype(depth_c)::value;
        constexpr bool DUAL = decltype(dual_c)::value, SWAP = decltype(swap_c)::value;
        WFragH<FG> wa[DEPTH], wb[DUAL ? DEPTH : 1];
        f16x8 act[AB][RG][2];                  // AB = 2: activations one k-step ahead as well (LDS latency)
        load_act(act[0], 0);
#pragma unroll
        for (int i = 0; i < DEPTH - 1; ++i) {
            if (i < NK) {
                trunk_load_w_h<FG, NP>(wa[i], WA, i, wsel, lane);
                if (DUAL) trunk_load_w_h<FG, NP>(wb[i], WB, i, wsel, lane);
            }
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks + DEPTH - 1 < NK) {
                trunk_load_w_h<FG, NP>(wa[(ks + DEPTH - 1) % DEPTH], WA, ks + DEPTH - 1, wsel, lane);
                if (DUAL) trunk_load_w_h<FG, NP>(wb[(ks + DEPTH - 1) % DEPTH], WB, ks + DEPTH - 1, wsel, lane);
            }
            if (AB == 2 && ks + 1 < NK) load_act(act[(ks + 1) & 1], ks + 1);
            __builtin_amdgcn_sched_barrier(0);       // keep the requests ahead of this k-step's MFMAs (the scheduler sinks them)
            if (SWAP) trunk_mfma_ht<FG, RG, NP>(accA, wa[ks % DEPTH], act[ks & (AB - 1)]);
            else trunk_mfma_h<FG, RG, NP>(accA, wa[ks % DEPTH], act[ks & (AB - 1)]);
            if (DUAL) trunk_mfma_h<FG, RG, NP>(accB, wb[ks % DEPTH], act[ks & (AB - 1)]);
            __builtin_amdgcn_sched_barrier(0);
            if (AB == 1 && ks + 1 < NK) load_act(act[0], ks + 1);
        }
    };
    using std::integral_constant;
    using no_t = integral_constant<bool, false>;
    using yes_t = integral_constant<bool, true>;
    // the 8-wave x 64-ray form is built for two workgroups per CU (4 waves per SIMD, 128 registers): shallower register
    // prefetch, the other waves of the SIMD cover the latency instead
    constexpr bool LEAN = (FG == 1 && RG == 2);
    constexpr int DEPTH_DUAL = LEAN ? 2 : 3, DEPTH_ONE = LEAN ? 3 : 4;
    constexpr int DEPTH_LOGITS = LEAN ? 6 : 4;       // acc3 is dead by then: its 32 registers hold query fragments further ahead

    if (MODE != 3) {
        // layer 1 and the x-part of layer 3, one pass over x
        phase(integral_constant<int, KX>{}, integral_constant<int, DEPTH_DUAL>{}, yes_t{}, no_t{}, acc, a.W1, acc3, a.W3x);
        TSTAMP(2);
        __syncthreads();                      // every wave has finished reading x
        TSTAMP(3);
        write_planes(acc, a.b1, a.inv1, a.s1);
        TSTAMP(4);
        __syncthreads();
        TSTAMP(5);

        // layer 2
        zero(acc);
        phase(integral_constant<int, KH>{}, integral_constant<int, DEPTH_ONE>{}, no_t{}, no_t{}, acc, a.W2, acc, a.W2);
        TSTAMP(6);
        __syncthreads();
        TSTAMP(7);
        write_planes(acc, a.b2, a.inv2, a.s2);
        TSTAMP(8);
        __syncthreads();
        TSTAMP(9);

        // layer 3, h-part, on top of the x-part
        phase(integral_constant<int, KH>{}, integral_constant<int, DEPTH_ONE>{}, no_t{}, no_t{}, acc3, a.W3h, acc3, a.W3h);
        TSTAMP(10);
        __syncthreads();                      // every wave has finished reading h2
        TSTAMP(11);
    }
    if (MODE == 2) {
        // the cache: h3 planes from LDS to HBM, whole 512-B rows per 32 lanes
        write_planes(acc3, a.b3, a.inv3, a.s3);
        __syncthreads();
        for (int t = tid; t < 2 * TR * (HC / 8); t += NT) {
            const int pl = t / (TR * (HC / 8)), rem = t - pl * (TR * (HC / 8)), ray = rem / (HC / 8), ch = rem - ray * (HC / 8);
            const int64_t gr = row0 + ray;
            if (gr < N) *reinterpret_cast<uint4*>(a.planes + ((size_t)pl * N + gr) * HC + 8 * ch) = *reinterpret_cast<const uint4*>(&S[pl][ray][8 * ch]);
        }
        return;
    }
    if (LOGITS) {
        if (MODE == 1) {
            write_planes(acc3, a.b3, a.inv3, a.s3);           // h3 planes
        TSTAMP(12);
            __syncthreads();
        TSTAMP(13);
        }
        // gridDim.z workgroups share a ray tile's token blocks (many query images against one ray set: without the split a
        // 16 011-ray set is 251 workgroups -- half the chip's workgroup slots -- however many images it serves)
        const int n_tb = a.Mpad / 256;
        const int tb_per = (n_tb + (int)gridDim.z - 1) / (int)gridDim.z;
        const int tb0 = (int)blockIdx.z * tb_per, tb1 = min(n_tb, tb0 + tb_per);
        const float divisor = a.divisor, inv_div = 1.0f / divisor;
        for (int tb = tb0; tb < tb1; ++tb) {
            // a token block whose kept rows end before this wave's 32 tokens: nothing to multiply, nothing to store (the statistics
            // merge gives those rows the pair of a dropped row, the column pass stops at the count).  No barrier inside this loop.
            // With row counts the waves take the block's 32 FG-token groups in an order that ROTATES with the block: the kept groups
            // are the first ones, and a workgroup's waves sit on the four SIMDs cyclically -- unrotated, 5 kept groups of 8 would
            // leave SIMD 0 with two waves' products in every block and the others with one (no barrier separates the blocks, so
            // rotated the SIMDs even out over the blocks a workgroup serves).  Which wave computes a group does not touch its bits.
            if (MODE == 3 && a.rows) {
                wsel = __builtin_amdgcn_readfirstlane((wave + tb) & (NWAVE - 1));
                if (32 * FG * wsel >= a.rows[tb]) continue;
            }
            zero(acc);
            phase(integral_constant<int, KH>{}, integral_constant<int, DEPTH_LOGITS>{}, no_t{}, yes_t{}, acc,
                  a.Qf + (size_t)tb * (KH * 2 * 8 * 64), acc, a.Qf);
            TSTAMP(14);
            // tile: rows = rays 32 rg + (reg & 3) + 8 (reg >> 2) + 4 lh, columns = tokens 256 tb + 32 FG wave + 32 tg + lr
#pragma unroll
            for (int tg = 0; tg < FG; ++tg) {
                const int tok = tb * 256 + 32 * FG * wsel + 32 * tg + lr;
                const bool tok_ok = tok < a.M;
                const float rc = tok_ok ? a.rowc[(size_t)tok * a.rowc_ld] : 0.0f;
                const float qs = a.qscale[tok];
                // one softmax partial per 64-RAY BLOCK, computed as the 64-ray tile computes it: a 128-ray tile (RG = 4) leaves two,
                // so the merged row statistics -- and with them every score -- do not depend on the tile a launch was given
                static_assert(RG % 2 == 0, "tiles are whole 64-ray blocks");
#pragma unroll
                for (int hb = 0; hb < RG / 2; ++hb) {
                    const int64_t r0 = row0 + 64 * hb;
                    if (r0 < N) {                                     // (wave-uniform; only the last tile of a ray set can lack its second block)
                        const float2 st = logits_tile_epilogue<2>(*reinterpret_cast<f32x16 (*)[2]>(&acc[tg][2 * hb]), qs, rc, divisor, inv_div, tok_ok,
                                                                  a.logits + (size_t)tok * N, r0, N, lh);
                        if (lh == 0) a.part[((size_t)blockIdx.x * (RG / 2) + hb) * a.Mpad + tok] = st;
                    }
                }
            }
        }
#ifdef TRUNK_STAMPS
        TSTAMP(15);
        if (MODE == 1 && lane == 0 && row0 + TR <= N)
            for (int k = 0; k < 16; ++k) reinterpret_cast<uint32_t*>(a.logits)[(size_t)(32 * wave + k) * N + row0] = stamp[k];
#endif
        return;
    }
    // h3 = relu(acc3 inv3 + b3).  A lane holds 4 consecutive features of one ray per register quad; the tile is transposed
    // through LDS (fp32 [ray][260]) so that every global store instruction writes one whole 1-KiB row of h3.
    constexpr int OLD = HC + 4;           // 1040-B rows: an odd multiple of 16 B
    float* O = reinterpret_cast<float*>(&S[0][0][0]);
    static_assert(TR * OLD * 4 <= 2 * TR * HSLD * 2, "output tile must fit in the activation planes");
#pragma unroll
    for (int fg = 0; fg < FG; ++fg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * FG * wave + 32 * fg + 8 * q + 4 * lh;
            const float4 bv = *reinterpret_cast<const float4*>(a.b3 + f0);
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                float4 o4;
                o4.x = fmaxf(fmaf(acc3[fg][rg][4 * q + 0], a.inv3, bv.x), 0.0f);
                o4.y = fmaxf(fmaf(acc3[fg][rg][4 * q + 1], a.inv3, bv.y), 0.0f);
                o4.z = fmaxf(fmaf(acc3[fg][rg][4 * q + 2], a.inv3, bv.z), 0.0f);
                o4.w = fmaxf(fmaf(acc3[fg][rg][4 * q + 3], a.inv3, bv.w), 0.0f);
                *reinterpret_cast<float4*>(&O[(32 * rg + lr) * OLD + f0]) = o4;
            }
        }
    __syncthreads();
    for (int ray = wave; ray < TR; ray += NWAVE) {
        const int64_t gr = row0 + ray;
        if (gr < N) *reinterpret_cast<float4*>(a.h3 + gr * HC + 4 * lane) = *reinterpret_cast<const float4*>(&O[ray * OLD + 4 * lane]);
    }
}

// ------------------------------------------------------------------------------------------------ two tiles, one stage apart
// k5_trunk_h2: a workgroup of SIXTEEN waves serves two 64-ray tiles.  Waves 0-7 ("half A") run on tile 2 b, waves 8-15 ("half B")
// on tile 2 b + 1, each exactly the program of k5_trunk_h<MODE, 1, 2> on its tile -- same work split (a wave = 32 features of all
// 64 rays), same arithmetic, same results bit for bit, softmax partials included -- but as a list of stages in which matrix-core
// stages and vector-ALU stages ALTERNATE
//     [encoder input | layer 1 + x-part of layer 3 | split | layer 2 | split | layer 3 | split | (logits product | epilogue) per block]
// and half B runs ONE STAGE BEHIND half A.  A workgroup's waves go to the four SIMDs in cyclic order, so every SIMD hosts two
// waves of each half: at every moment two of its waves issue MFMAs and two do vector work (hi/lo split, sincos, exact division,
// exp, stores), and the two pipes run side by side.  In k5_trunk_h all waves of a workgroup are in the same phase between two
// barriers, and the two workgroups of a CU start together, take equally long and so stay in lockstep: the matrix pipe idles
// through every vector phase.  The weight stream per ray is that of k5_trunk_h (one pass over the weights per 64 rays) -- a
// split of ONE tile's rays over the halves doubles it (measured: 0.75-0.89 ms against 0.48-0.50 ms for k5_trunk_h, same box).
// Stages are separated by workgroup barriers; both halves pass the same number of them.
// MEASURED (same box, 16 x 16 011 rays, M = 256): 0.548 ms against 0.501 ms for k5_trunk_h<1, 1, 2> -- the enforced overlap does
// NOT pay.  What it gives up is the reason the lockstep of two workgroups is not the loss it looks like: the two workgroups of a
// CU stream the SAME weight fragments at the same time, so the second one hits the first one's lines in the vector L1 (hit rate
// 0.71 in the counters); one stage apart the halves stream different layers and every fragment comes from L2.  The weight stream
// (852 KB per 64-ray tile through a 64-B/clk path) is what the matrix pipe waits for.  Kept as iff_idnet_desc.trunk_variant = 4
// for the A/B; not the default.
template <int MODE>
__global__ void __launch_bounds__(1024, 4) k5_trunk_h2(TrunkHArgs a, int64_t n_tiles) {
    static_assert(MODE == 1 || MODE == 3, "the two-tile form serves the two logits launches");
    constexpr int FG = 1, RG = 2, TR = 64;
    __shared__ __attribute__((aligned(16))) _Float16 SS[2][2][TR][HSLD];
    const int tid = threadIdx.x, lane = tid & 63, wave16 = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar registers
    const int half = wave16 >> 3, wave = wave16 & 7, th = tid & 511;
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t tile = 2 * (int64_t)blockIdx.x + half;
    const bool tile_on = tile < n_tiles;                      // an odd tile count leaves the last workgroup's half B idle
    const int64_t row0 = tile * TR;
    const int64_t N = a.N;
    _Float16 (*S)[TR][HSLD] = SS[half];
    {
        const size_t qb = blockIdx.y;
        if (MODE == 1) { a.ray_o += qb * N * 3; a.ray_d += qb * N * 3; a.ray_c += qb * N * 3; }
        if (MODE == 3) a.planes += qb * 2 * (size_t)N * HC;
        a.Qf += qb * (size_t)(a.Mpad / 256) * (HC / 16) * 2 * 8 * 64;
        a.qscale += qb * (size_t)a.Mpad;
        a.rowc += qb * (size_t)a.M * a.rowc_ld;
        a.logits += qb * (size_t)a.M * N;
        a.part += qb * (size_t)n_tiles * a.Mpad;
    }
    f32x16 acc[FG][RG], acc3[FG][RG];
    auto zero = [](f32x16 (&ac)[FG][RG]) {
#pragma unroll
        for (int y = 0; y < RG; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) ac[0][y][r] = 0.0f;
    };
    constexpr int KX = (141 + 15) / 16, KH = HC / 16;

    auto load_act = [&](f16x8 (&v)[RG][2], int ks) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) v[rg][pl] = *reinterpret_cast<const f16x8*>(&S[pl][32 * rg + lr][16 * ks + 8 * lh]);
    };
    auto write_planes = [&](const f32x16 (&ac)[FG][RG], const float* __restrict__ bias, float inv, float s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * wave + 8 * q + 4 * lh;
            const float4 bv = *reinterpret_cast<const float4*>(bias + f0);
            const f32x2 is = {inv * s, inv * s}, bs01 = {bv.x * s, bv.y * s}, bs23 = {bv.z * s, bv.w * s};
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                f16x4 p0, p1;
                relu_split_quad(ac[0][rg][4 * q], ac[0][rg][4 * q + 1], ac[0][rg][4 * q + 2], ac[0][rg][4 * q + 3], is, bs01, bs23, p0, p1);
                *reinterpret_cast<f16x4*>(&S[0][32 * rg + lr][f0]) = p0;
                *reinterpret_cast<f16x4*>(&S[1][32 * rg + lr][f0]) = p1;
            }
        }
    };
    // the k loop of k5_trunk_h: weight fragments DEPTH - 1 k-steps ahead in registers; DUAL: two weight streams over the same activations
    auto phase = [&](auto nk_c, auto depth_c, auto dual_c, auto swap_c, f32x16 (&accA)[FG][RG], const uint4* __restrict__ WA,
                     f32x16 (&accB)[FG][RG], const uint4* __restrict__ WB) {
        constexpr int NK = decltype(nk_c)::value, DEPTH = decltype(depth_c)::value;
        constexpr bool DUAL = decltype(dual_c)::value, SWAP = decltype(swap_c)::value;
        WFragH<FG> wa[DEPTH], wb[DUAL ? DEPTH : 1];
        f16x8 act[RG][2];
        load_act(act, 0);
#pragma unroll
        for (int i = 0; i < DEPTH - 1; ++i) {
            if (i < NK) {
                trunk_load_w_h(wa[i], WA, i, wave, lane);
                if (DUAL) trunk_load_w_h(wb[i], WB, i, wave, lane);
            }
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks + DEPTH - 1 < NK) {
                trunk_load_w_h(wa[(ks + DEPTH - 1) % DEPTH], WA, ks + DEPTH - 1, wave, lane);
                if (DUAL) trunk_load_w_h(wb[(ks + DEPTH - 1) % DEPTH], WB, ks + DEPTH - 1, wave, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (SWAP) trunk_mfma_ht<FG, RG>(accA, wa[ks % DEPTH], act);
            else trunk_mfma_h<FG, RG>(accA, wa[ks % DEPTH], act);
            if (DUAL) trunk_mfma_h<FG, RG>(accB, wb[ks % DEPTH], act);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < NK) load_act(act, ks + 1);
        }
    };
    using std::integral_constant;
    using no_t = integral_constant<bool, false>;
    using yes_t = integral_constant<bool, true>;

    const int n_tb = a.Mpad / 256;
    const int tb_per = (n_tb + (int)gridDim.z - 1) / (int)gridDim.z;
    const int tb0 = (int)blockIdx.z * tb_per, tb1 = min(n_tb, tb0 + tb_per);
    const float divisor = a.divisor, inv_div = 1.0f / divisor;

    // The program of one half: every stage ends at a workgroup barrier.  `on` = this half has a tile (an odd tile count leaves the
    // last workgroup's half B without one: it only keeps the barrier count).  Straight-line code per half (the two halves are
    // two inlined copies): a loop over a runtime stage index keeps both accumulator sets live through every stage and spills.
    auto program = [&](bool on) {
        if (MODE == 3) {
            if (on) {
            for (int u = th; u < 2 * TR * (HC / 8); u += 512) {
                const int pl = u / (TR * (HC / 8)), rem = u - pl * (TR * (HC / 8)), ray = rem / (HC / 8), ch = rem - ray * (HC / 8);
                const int64_t gr = row0 + ray;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (gr < N) v = *reinterpret_cast<const uint4*>(a.planes + ((size_t)pl * N + gr) * HC + 8 * ch);
                *reinterpret_cast<uint4*>(&S[pl][ray][8 * ch]) = v;
            }
            }
            __syncthreads();
        } else {
            if (on) encode_tile64(S, a.ray_o, a.ray_d, a.ray_c, row0, N, a.sx, wave, lane);     // the encoder input (see k5_trunk_h)
            __syncthreads();
            zero(acc); zero(acc3);
            if (on) phase(integral_constant<int, KX>{}, integral_constant<int, 2>{}, yes_t{}, no_t{}, acc, a.W1, acc3, a.W3x);
            __syncthreads();
            if (on) write_planes(acc, a.b1, a.inv1, a.s1);
            __syncthreads();
            zero(acc);
            if (on) phase(integral_constant<int, KH>{}, integral_constant<int, 3>{}, no_t{}, no_t{}, acc, a.W2, acc, a.W2);
            __syncthreads();
            if (on) write_planes(acc, a.b2, a.inv2, a.s2);
            __syncthreads();
            if (on) phase(integral_constant<int, KH>{}, integral_constant<int, 3>{}, no_t{}, no_t{}, acc3, a.W3h, acc3, a.W3h);
            __syncthreads();
            if (on) write_planes(acc3, a.b3, a.inv3, a.s3);
            __syncthreads();
        }
        for (int tb = tb0; tb < tb1; ++tb) {
            zero(acc);
            if (on) phase(integral_constant<int, KH>{}, integral_constant<int, 4>{}, no_t{}, yes_t{}, acc,
                          a.Qf + (size_t)tb * (KH * 2 * 8 * 64), acc, a.Qf);
            __syncthreads();
            if (on) {
                // logits epilogue of k5_trunk_h: rows = rays 32 rg + (reg & 3) + 8 (reg >> 2) + 4 lh, columns = tokens 256 tb + 32 wave + lr
                const int tok = tb * 256 + 32 * wave + lr;
                const bool tok_ok = tok < a.M;
                const float rc = tok_ok ? a.rowc[(size_t)tok * a.rowc_ld] : 0.0f;
                const float qs = a.qscale[tok];
                const float2 st = logits_tile_epilogue<RG>(acc[0], qs, rc, divisor, inv_div, tok_ok, a.logits + (size_t)tok * N, row0, N, lh);
                if (lh == 0) a.part[(size_t)tile * a.Mpad + tok] = st;
            }
            __syncthreads();
        }
    };
    // half B runs one stage behind half A: one barrier ahead of its program, half A one barrier after its own
    if (half == 0) { program(tile_on); __syncthreads(); }
    else { __syncthreads(); program(tile_on); }
}

// ------------------------------------------------------------------------------------------------ launchers
// variant: 0 -> 8 waves x 32 features, 64 rays;  1 -> 4 waves x 64 features, 64 rays (two workgroups per CU);
//          2 -> 8 waves x 32 features, 128 rays (half the weight stream per ray);
//          3 -> the logits launches as k5_trunk_h2: sixteen waves, two 64-ray tiles one stage apart (the work split of 0 per tile)
int trunk_h_rays_per_wg(int variant) { return variant == 2 ? 128 : 64; }
// The logits launch against CACHED encoder planes (MODE 3) has no encoder to feed: what a workgroup streams is the query planes,
// 256 KB per 256-token block whatever its rays, so twice the rays per workgroup halve that stream per logit -- variant 2 unless the
// handle names a form of its own (1: four waves, 3: two tiles a stage apart).  Measured (scripts/time_warm.py, same box):
// 32 queries x 16 011 rays 67 000 -> 72 600 poses/s, 8 x 540 000 rays 2 460 -> 2 680; the fused launch (MODE 1) keeps 64 rays,
// where the second accumulator set of the encoder makes 128 rays the slower form.  The results are the same bits either way.
int trunk_h_cached_variant(int variant) { return variant == 0 ? 2 : variant; }

// workgroups per ray tile for the token blocks of a logits launch: enough that tiles x ray sets x split fills the chip's
// workgroup slots about twice (2 per CU), never more than the blocks there are; `recompute`: every split workgroup of the
// fused launch runs the encoder again (300 of its 300 + 96 n MFMAs per wave), so it is split only while that stays < 25 %
static unsigned token_split(int64_t tiles, int B, int n_tb, bool recompute) {
    int64_t want = (1024 + tiles * B - 1) / (tiles * B);
    if (want > n_tb) want = n_tb;
    if (recompute) {
        const int64_t cap = n_tb / 8;            // >= 8 blocks per workgroup: encoder share 300 / (300 + 96 * 8) = 28 %
        if (want > cap) want = cap;
    }
    if (want < 1) want = 1;
    const int64_t per = (n_tb + want - 1) / want;          // blocks per workgroup; no workgroup without a block
    return (unsigned)((n_tb + per - 1) / per);
}

static TrunkHArgs base_args(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N) {
    TrunkHArgs a;
    a.ray_o = o; a.ray_d = d; a.ray_c = rgb; a.N = N;
    a.W1 = (const uint4*)n.h1; a.W2 = (const uint4*)n.h2; a.W3h = (const uint4*)n.h3h; a.W3x = (const uint4*)n.h3x;
    a.b1 = n.b1; a.b2 = n.b2; a.b3 = n.b3;
    a.sx = ldexpf(1.0f, n.e_x);
    a.inv1 = ldexpf(1.0f, -(n.e_w1 + n.e_x)); a.s1 = ldexpf(1.0f, n.e_h1);
    a.inv2 = ldexpf(1.0f, -(n.e_w2 + n.e_h1)); a.s2 = ldexpf(1.0f, n.e_h2);
    a.inv3 = ldexpf(1.0f, -(n.e_w3h + n.e_h2)); a.s3 = ldexpf(1.0f, n.e_h3);
    a.h3 = nullptr; a.planes = nullptr; a.Qf = nullptr; a.qscale = nullptr; a.rowc = nullptr; a.rowc_ld = 0; a.M = 0; a.divisor = 1.0f;
    a.logits = nullptr; a.part = nullptr; a.Mpad = 0; a.rows = nullptr;
    return a;
}

template <int MODE>
static hipError_t launch_variant(int variant, dim3 grid, const TrunkHArgs& a, hipStream_t s, bool one_product = false) {
    if (one_product) {                   // IFF_GEMM_F16X1: the eight-wave forms only (64 rays, or 128 for the cached logits launch)
        if (variant == 2) hipLaunchKernelGGL((k5_trunk_h<MODE + 4, 1, 4>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k5_trunk_h<MODE + 4, 1, 2>), grid, dim3(512), 0, s, a);
        return hipGetLastError();
    }
    if constexpr (MODE == 1 || MODE == 3) {
        if (variant == 3) {          // two tiles per workgroup, one stage apart (the cache build and the feature output keep the 8-wave form)
            const int64_t n_tiles = grid.x;
            hipLaunchKernelGGL((k5_trunk_h2<MODE>), dim3((unsigned)((n_tiles + 1) / 2), grid.y, grid.z), dim3(1024), 0, s, a, n_tiles);
            return hipGetLastError();
        }
    }
    if (variant == 1) hipLaunchKernelGGL((k5_trunk_h<MODE, 2, 2>), grid, dim3(256), 0, s, a);
    else if (variant == 2) hipLaunchKernelGGL((k5_trunk_h<MODE, 1, 4>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((k5_trunk_h<MODE, 1, 2>), grid, dim3(512), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_trunk_h_features(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, int B, float* h3,
                                   hipStream_t s) {
    TrunkHArgs a = base_args(n, o, d, rgb, N);
    a.h3 = h3;
    const int TR = trunk_h_rays_per_wg(n.trunk_variant);
    return launch_variant<0>(n.trunk_variant, dim3((unsigned)((N + TR - 1) / TR), (unsigned)B), a, s, n.trunk_f16 == 2);
}

// qf [B*M][qf_ld] -> Qf planes + qscale in `ws` (layout: Qf | qscale | part), then the fused launch; `part_out` / `n_blk_out`
// tell the caller where the per-workgroup softmax partials are for k6_merge_stats
hipError_t launch_trunk_h_logits(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, const float* qf, int M,
                                 int B, float divisor, float* logits, void* Qf, float* qscale, float2* part, hipStream_t s) {
    const int n_tb = (M + 255) / 256, Mpad = n_tb * 256;
    hipLaunchKernelGGL(k_qf_frag_h, dim3((unsigned)(Mpad / 32), (unsigned)B), dim3(256), 0, s, qf, n.qf_ld, M, n_tb, n.e_h3,
                       (_Float16*)Qf, qscale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    TrunkHArgs a = base_args(n, o, d, rgb, N);
    a.Qf = (const uint4*)Qf; a.qscale = qscale; a.rowc = qf + HC; a.rowc_ld = n.qf_ld; a.M = M; a.divisor = divisor;
    a.logits = logits; a.part = part; a.Mpad = Mpad;
    const int TR = trunk_h_rays_per_wg(n.trunk_variant);
    const int64_t tiles = (N + TR - 1) / TR;
    return launch_variant<1>(n.trunk_variant, dim3((unsigned)tiles, (unsigned)B, token_split(tiles, B, n_tb, true)), a, s, n.trunk_f16 == 2);
}

// The per-model cache: the encoder's last hidden activation as fp16 hi/lo planes [2][N][256] (1 KB per ray),
// and the logits of a batch of token rows against it.
hipError_t launch_trunk_h_cache(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, void* planes,
                                hipStream_t s) {
    TrunkHArgs a = base_args(n, o, d, rgb, N);
    a.planes = (_Float16*)planes;
    const int TR = trunk_h_rays_per_wg(n.trunk_variant);
    return launch_variant<2>(n.trunk_variant, dim3((unsigned)((N + TR - 1) / TR), 1u), a, s, n.trunk_f16 == 2);
}

hipError_t launch_trunk_h_logits_cached(const IdNetDev& n, const void* planes, int64_t N, const float* qf, int M, float divisor,
                                        float* logits, void* Qf, float* qscale, float2* part, const int* rows, hipStream_t s) {
    const int n_tb = (M + 255) / 256, Mpad = n_tb * 256;
    hipLaunchKernelGGL(k_qf_frag_h, dim3((unsigned)(Mpad / 32), 1u), dim3(256), 0, s, qf, n.qf_ld, M, n_tb, n.e_h3, (_Float16*)Qf, qscale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    TrunkHArgs a = base_args(n, nullptr, nullptr, nullptr, N);
    a.planes = (_Float16*)const_cast<void*>(planes);
    a.Qf = (const uint4*)Qf; a.qscale = qscale; a.rowc = qf + HC; a.rowc_ld = n.qf_ld; a.M = M; a.divisor = divisor;
    a.logits = logits; a.part = part; a.Mpad = Mpad;
    int variant = trunk_h_cached_variant(n.trunk_variant);
    if (rows && variant == 3) variant = 2;            // the two-tile form keeps barriers inside its token loop: row counts go to the 128-ray form
    a.rows = rows;
    const int TR = trunk_h_rays_per_wg(variant);
    const int64_t tiles = (N + TR - 1) / TR;
    return launch_variant<3>(variant, dim3((unsigned)tiles, 1u, token_split(tiles, 1, n_tb, false)), a, s, n.trunk_f16 == 2);
}
