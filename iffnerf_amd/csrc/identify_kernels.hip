// identify_kernels.hip -- K5 ray encoder (+k_proj), q_proj, K6 attention logits / row statistics / column-sum score,
// K7 top-k.  gfx950, fp32 throughout: the matrix products run on the fp32-input MFMA (v_mfma_f32_32x32x2_f32), which is
// bit-for-bit a k-ordered fmaf chain (exact f32, no reduced precision), so attention logits stay within fp32 rounding
// of the reference and the top-k set is the reference's.
#include "iff_device.h"
#include "iff_launch.h"
#include "iff_select.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ weight prep
// nn.Linear weight [out][in] -> k-major [in_pad][out] at row offset row_off (rows beyond in stay as the caller zeroed them)
__global__ void k_transpose_pad(const float* __restrict__ w, float* __restrict__ dst, int out_f, int in_f, int row_off) {
    int n = out_f * in_f;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        int i = t / out_f, o = t - i * out_f;
        dst[(size_t)(row_off + i) * out_f + o] = w[(size_t)o * in_f + i];
    }
}
hipError_t launch_transpose_pad(const float* w, float* dst, int out_f, int in_f, int in_pad, int row_off, hipStream_t s) {
    (void)in_pad;
    int grid = (out_f * in_f + 255) / 256;
    hipLaunchKernelGGL(k_transpose_pad, dim3(grid), dim3(256), 0, s, w, dst, out_f, in_f, row_off);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ encoder input
// ray_preprocessor.py:30-37 + tensorBase.py:14-20: x = [o, d, rgb, PE(o,8), PE(d,8), PE(rgb,6)] (141), zero-padded to XW = 160 (a multiple of both GEMM k-tiles, 16 and 32)
constexpr int XW = 160;
__global__ void k5_ray_input(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ c,
                             int64_t N, float* __restrict__ x) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < N * XW; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t / XW;
        int col = (int)(t - r * XW);
        float v = 0.0f;
        if (col < 3) v = o[3 * r + col];
        else if (col < 6) v = d[3 * r + col - 3];
        else if (col < 9) v = c[3 * r + col - 6];
        else if (col < 141) {
            // blocks: PE(o,8) = 48 at 9, PE(d,8) = 48 at 57, PE(rgb,6) = 36 at 105; each = [sin(F*3) | cos(F*3)], j-major k-minor
            int b = col - 9;
            const float* src;
            int F;
            if (b < 48) { src = o; F = 8; }
            else if (b < 96) { src = d; F = 8; b -= 48; }
            else { src = c; F = 6; b -= 96; }
            int half = F * 3;
            bool is_cos = b >= half;
            if (is_cos) b -= half;
            int j = b / F, k = b - j * F;
            float arg = src[3 * r + j] * (float)(1 << k);
            v = is_cos ? cosf(arg) : sinf(arg);
        }
        x[t] = v;
    }
}

// ------------------------------------------------------------------------------------------------ fp32 MFMA GEMM
// Y[M][ldy] (cols 0..Nout) = act( [A1 | A2] * Wt + bias ), A1 [M][lda1] uses K1 columns, A2 [M][lda2] uses K2 columns,
// Wt [K1+K2][Nout] k-major.  K1, K2 multiples of BK.  Tile 128x128x16, 4 waves each 64x64 (2x2 MFMA 32x32 blocks).
// NT variant (B_IS_ROWS): B operand given as rows Bm [Nout][ldb] (k contiguous), i.e. Y = A * Bm^T -- the attention
// logits; there `divisor` divides the product (multihead_attention.py:6-7) and bias is not applied.
constexpr int BN = 128, BK = 16;

// MT = 32-row MFMA blocks per wave in M: MT = 2 -> 128-row tile, MT = 1 -> 64-row tile (twice the workgroups: 2-3 are
// co-resident per CU and cover each other's barrier / LDS-latency bubbles, and 384-wide outputs balance over 256 CUs).
template <bool RELU, bool B_IS_ROWS, int MT>
__global__ void __launch_bounds__(256) k_gemm_f32(const float* __restrict__ A1, int lda1, int K1,
                                                  const float* __restrict__ A2, int lda2, int K2,
                                                  const float* __restrict__ B, int ldb, const float* __restrict__ bias,
                                                  float* __restrict__ Y, int64_t ldy, int64_t M, int64_t Nout, float divisor) {
    constexpr int BM = 64 * MT;
    constexpr int A_LD = BM + 4, B_LD = BN + 4;
    __shared__ float As[2][BK][A_LD];
    __shared__ float Bs[2][BK][B_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int64_t col0 = (int64_t)blockIdx.y * BN;
    const int wr = (wave >> 1) * 32 * MT, wc = (wave & 1) * 64;
    const int K = K1 + K2;
    const int nk = K / BK;

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging registers: A tile BM rows x 16 k = BM*4 float4 -> MT per thread (row = f>>2, kq = (f&3)*4); B 512 float4 -> 2.
    // Two register sets: global loads run TWO k-tiles ahead of the MFMAs (one tile of MFMAs is shorter than an L2/HBM
    // round trip), LDS is double-buffered one tile ahead.
    float4 raA[MT], rbA[2], raB[MT], rbB[2];
    auto load_tile = [&](int kt, float4* ra, float4* rb) {
        const int k0 = kt * BK;
        const float* Asrc; int lda, kk;
        if (k0 < K1) { Asrc = A1; lda = lda1; kk = k0; } else { Asrc = A2; lda = lda2; kk = k0 - K1; }
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            int f = tid + u * 256;
            int r = f >> 2, kq = (f & 3) * 4;
            int64_t gr = row0 + r;
            ra[u] = (gr < M) ? ld4(Asrc + gr * lda + kk + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int f = tid + u * 256;
            if (B_IS_ROWS) {
                int r = f >> 2, kq = (f & 3) * 4;
                int64_t gc = col0 + r;
                rb[u] = (gc < Nout) ? ld4(B + gc * ldb + k0 + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                int kr = f >> 5, cq = (f & 31) * 4;          // 16 k-rows x 32 float4
                int64_t gc = col0 + cq;
                rb[u] = (gc < Nout) ? ld4(B + (int64_t)(k0 + kr) * ldb + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tile = [&](int buf, const float4* ra, const float4* rb) {
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            int f = tid + u * 256;
            int r = f >> 2, kq = (f & 3) * 4;
            As[buf][kq + 0][r] = ra[u].x; As[buf][kq + 1][r] = ra[u].y; As[buf][kq + 2][r] = ra[u].z; As[buf][kq + 3][r] = ra[u].w;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int f = tid + u * 256;
            if (B_IS_ROWS) {
                int r = f >> 2, kq = (f & 3) * 4;
                Bs[buf][kq + 0][r] = rb[u].x; Bs[buf][kq + 1][r] = rb[u].y; Bs[buf][kq + 2][r] = rb[u].z; Bs[buf][kq + 3][r] = rb[u].w;
            } else {
                int kr = f >> 5, cq = (f & 31) * 4;
                *reinterpret_cast<float4*>(&Bs[buf][kr][cq]) = rb[u];
            }
        }
    };
    const int li = lane & 31, lk = lane >> 5;
    auto compute_tile = [&](int buf) {
        // operand reads run one k-step ahead of the MFMAs that consume them
        float a_cur[MT], b_cur[2], a_nxt[MT], b_nxt[2];
#pragma unroll
        for (int i = 0; i < MT; ++i) a_cur[i] = As[buf][lk][wr + 32 * i + li];
        b_cur[0] = Bs[buf][lk][wc + li]; b_cur[1] = Bs[buf][lk][wc + 32 + li];
#pragma unroll
        for (int k2 = 0; k2 < BK; k2 += 2) {
            if (k2 + 2 < BK) {
#pragma unroll
                for (int i = 0; i < MT; ++i) a_nxt[i] = As[buf][k2 + 2 + lk][wr + 32 * i + li];
                b_nxt[0] = Bs[buf][k2 + 2 + lk][wc + li]; b_nxt[1] = Bs[buf][k2 + 2 + lk][wc + 32 + li];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[1], acc[i][1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) a_cur[i] = a_nxt[i];
            b_cur[0] = b_nxt[0]; b_cur[1] = b_nxt[1];
        }
    };

    load_tile(0, raA, rbA);
    store_tile(0, raA, rbA);
    if (nk > 1) load_tile(1, raA, rbA);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even tile: LDS buffer 0; set A holds tile kt+1 (in flight since the previous phase); fetch tile kt+2 into set B
        if (kt + 2 < nk) load_tile(kt + 2, raB, rbB);
        compute_tile(0);
        if (kt + 1 < nk) store_tile(1, raA, rbA);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd tile: LDS buffer 1; set B holds tile kt+2; fetch tile kt+3 into set A
        if (kt + 3 < nk) load_tile(kt + 3, raA, rbA);
        compute_tile(1);
        if (kt + 2 < nk) store_tile(0, raB, rbB);
        __syncthreads();
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int64_t gc = col0 + wc + j * 32 + (lane & 31);
            float bv = 0.0f;
            if (!B_IS_ROWS && bias && gc < Nout) bv = bias[gc];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int64_t gr = row0 + wr + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gr < M && gc < Nout) {
                    float v = acc[i][j][r];
                    if (B_IS_ROWS) v = v / divisor;
                    else v = v + bv;
                    if (RELU) v = fmaxf(v, 0.0f);
                    Y[gr * ldy + gc] = v;
                }
            }
        }
}

constexpr int GEMM_MT = 1;

template <bool RELU>
static hipError_t gemm_nn(const float* A1, int lda1, int K1, const float* A2, int lda2, int K2, const float* Wt, int Nout,
                          const float* bias, float* Y, int64_t ldy, int64_t M, hipStream_t s) {
    constexpr int BMv = 64 * GEMM_MT;
    dim3 grid((unsigned)((M + BMv - 1) / BMv), (unsigned)((Nout + BN - 1) / BN));
    hipLaunchKernelGGL((k_gemm_f32<RELU, false, GEMM_MT>), grid, dim3(256), 0, s, A1, lda1, K1, A2, lda2, K2, Wt, Nout, bias, Y,
                       ldy, M, (int64_t)Nout, 1.0f);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 3xBF16 split GEMM
// fp32-accurate products on the bf16 matrix cores.  Every fp32 operand a is split exactly into three bf16 pieces
// a = a0 + a1 + a2 (round-to-nearest each time, residuals are exact), and a*b is accumulated in fp32 as
// a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0: the dropped cross terms are below 2^-25 |a b|, i.e. under half an ulp of
// the product -- the same class of error as any re-association of an fp32 sum.  6 x v_mfma_f32_32x32x16_bf16 cover the
// work of 8 x v_mfma_f32_32x32x2_f32 in 192 instead of 512 cycles per SIMD (bf16 MFMA runs 16x the fp32-MFMA rate).
// Weights are pre-split at load time ([3][out][K_pad] bf16, the nn.Linear row layout); activations are split while
// they are staged into LDS.  Tile 64 x 128 x 32, 4 waves of 32 x 64, single LDS buffer + register prefetch.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BK3 = 32, LD3 = BK3 + 8;     // LDS row: 32 bf16 + 8 pad = 80 B, conflict-free for 16-B fragment reads

__device__ inline void split3(float4 v, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __bf16 h0 = (__bf16)x[i];
        float r1 = x[i] - (float)h0;
        __bf16 h1 = (__bf16)r1;
        float r2 = r1 - (float)h1;
        p0[i] = h0; p1[i] = h1; p2[i] = (__bf16)r2;
    }
}

__global__ void k_split_rows(const float* __restrict__ w, __bf16* __restrict__ planes, int out_f, int in_f, int in_pad) {
    // w [out][in] fp32 -> planes [3][out][in_pad] bf16 (zero beyond in)
    int64_t n = (int64_t)out_f * in_pad;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int o = (int)(t / in_pad), i = (int)(t - (int64_t)o * in_pad);
        float x = (i < in_f) ? w[(size_t)o * in_f + i] : 0.0f;
        __bf16 h0 = (__bf16)x;
        float r1 = x - (float)h0;
        __bf16 h1 = (__bf16)r1;
        float r2 = r1 - (float)h1;
        planes[t] = h0; planes[n + t] = h1; planes[2 * n + t] = (__bf16)r2;
    }
}
hipError_t launch_split_rows(const float* w, void* planes, int out_f, int in_f, int in_pad, hipStream_t s) {
    int64_t n = (int64_t)out_f * in_pad;
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, (__bf16*)planes, out_f, in_f, in_pad);
    return hipGetLastError();
}

// Y = act([A1 | A2] * W^T + bias) with W given as pre-split planes Wp [3][Nout][Kp] (B_FP32 = false), or
// Y = (A1 * Bf^T) / divisor with Bf fp32 rows [Nout][ldb] split on the fly (B_FP32 = true; the attention logits).
template <bool RELU, bool B_FP32>
__global__ void __launch_bounds__(256) k_gemm_bf3(const float* __restrict__ A1, int lda1, int K1,
                                                  const float* __restrict__ A2, int lda2, int K2,
                                                  const __bf16* __restrict__ Wp, int Kp, const float* __restrict__ Bf, int ldb,
                                                  const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy,
                                                  int64_t M, int64_t Nout, float divisor, const float* __restrict__ rowc,
                                                  int rowc_ld) {
    constexpr int BM = 64;
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM][LD3];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[3][BN][LD3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int64_t col0 = (int64_t)blockIdx.y * BN;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 64;
    const int K = K1 + K2;
    const int nk = K / BK3;
    const size_t plane = (size_t)Nout * Kp;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // staging registers.  A: 64 rows x 32 k fp32 = 512 float4 -> 2 per thread (row = f >> 3, k = (f & 7) * 4).
    // B pre-split: 3 planes x 128 cols x 32 k bf16 = 1536 16-B chunks -> 6 per thread; B fp32: 1024 float4 -> 4 per thread.
    float4 ra[2];
    uint4 rbw[6];
    float4 rbf[4];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK3;
        const float* Asrc; int lda, kk;
        if (k0 < K1) { Asrc = A1; lda = lda1; kk = k0; } else { Asrc = A2; lda = lda2; kk = k0 - K1; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int f = tid + u * 256;
            int r = f >> 3, kq = (f & 7) * 4;
            int64_t gr = row0 + r;
            ra[u] = (gr < M) ? ld4(Asrc + gr * lda + kk + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (B_FP32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int f = tid + u * 256;
                int r = f >> 3, kq = (f & 7) * 4;
                int64_t gc = col0 + r;
                rbf[u] = (gc < Nout) ? ld4(Bf + gc * ldb + k0 + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                int f = tid + u * 256;                  // chunk id: plane p = f / 512, col = (f % 512) >> 2, k8 = (f & 3) * 8
                int p = f >> 9, c = (f & 511) >> 2, k8 = (f & 3) * 8;
                int64_t gc = col0 + c;
                rbw[u] = (gc < Nout) ? *reinterpret_cast<const uint4*>(Wp + p * plane + gc * Kp + k0 + k8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int f = tid + u * 256;
            int r = f >> 3, kq = (f & 7) * 4;
            bf16x4 p0, p1, p2;
            split3(ra[u], p0, p1, p2);
            *reinterpret_cast<bf16x4*>(&As[0][r][kq]) = p0;
            *reinterpret_cast<bf16x4*>(&As[1][r][kq]) = p1;
            *reinterpret_cast<bf16x4*>(&As[2][r][kq]) = p2;
        }
        if (B_FP32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int f = tid + u * 256;
                int r = f >> 3, kq = (f & 7) * 4;
                bf16x4 p0, p1, p2;
                split3(rbf[u], p0, p1, p2);
                *reinterpret_cast<bf16x4*>(&Bs[0][r][kq]) = p0;
                *reinterpret_cast<bf16x4*>(&Bs[1][r][kq]) = p1;
                *reinterpret_cast<bf16x4*>(&Bs[2][r][kq]) = p2;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                int f = tid + u * 256;
                int p = f >> 9, c = (f & 511) >> 2, k8 = (f & 3) * 8;
                *reinterpret_cast<uint4*>(&Bs[p][c][k8]) = rbw[u];
            }
        }
    };

    const int lr = lane & 31, lh = lane >> 5;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                 // the previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);          // in flight while this tile is multiplied
#pragma unroll
        for (int ks = 0; ks < BK3 / 16; ++ks) {
            const int ko = 16 * ks + 8 * lh;
            bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[0][wr + lr][ko]);
            bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[1][wr + lr][ko]);
            bf16x8 a2 = *reinterpret_cast<const bf16x8*>(&As[2][wr + lr][ko]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[0][wc + 32 * j + lr][ko]);
                bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[1][wc + 32 * j + lr][ko]);
                bf16x8 b2 = *reinterpret_cast<const bf16x8*>(&Bs[2][wc + 32 * j + lr][ko]);
                // smallest contributions first
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
            }
        }
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int64_t gc = col0 + wc + j * 32 + (lane & 31);
        float bv = 0.0f;
        if (!B_FP32 && bias && gc < Nout) bv = bias[gc];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int64_t gr = row0 + wr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gr < M && gc < Nout) {
                float v = acc[j][r];
                if (B_FP32) v = (rowc ? v + rowc[gr * rowc_ld] : v) / divisor;
                else v = v + bv;
                if (RELU) v = fmaxf(v, 0.0f);
                Y[gr * ldy + gc] = v;
            }
        }
    }
}

template <bool RELU>
static hipError_t gemm_bf3(const float* A1, int lda1, int K1, const float* A2, int lda2, int K2, const void* Wp, int Kp,
                           int Nout, const float* bias, float* Y, int64_t ldy, int64_t M, hipStream_t s) {
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((Nout + BN - 1) / BN));
    hipLaunchKernelGGL((k_gemm_bf3<RELU, false>), grid, dim3(256), 0, s, A1, lda1, K1, A2, lda2, K2, (const __bf16*)Wp, Kp,
                       (const float*)nullptr, 0, bias, Y, ldy, M, (int64_t)Nout, 1.0f, (const float*)nullptr, 0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ fused encoder trunk
// The three ReLU layers of the ray encoder (ray_preprocessor.py:9-22,30-38) in ONE launch for feature_c = 256:
//   h1 = relu(W1 x + b1), h2 = relu(W2 h1 + b2), h3 = relu(W3 [h2 | x] + b3)
// A workgroup owns 64 rays; wave w owns output features 64w..64w+63 of every layer.  Activations never leave the CU:
// they sit in LDS as three bf16 planes [plane][ray][feature] (the 3xBF16 split above) and are the MFMA's B operand
// (k = feature, column = ray); the weights are the A operand, read straight from L2 in "fragment order" (each
// wave-instruction one contiguous KiB, prepared once by k_frag_order) -- every wave needs different output rows, so
// staging them through LDS would buy no reuse.  The products come out transposed (rows = features, columns = rays):
// a lane holds 4 consecutive features of one ray per register quad, so the next layer's planes are written with 8-byte
// LDS stores and h3 with 16-byte global stores.  The x-part of layer 3 is accumulated together with layer 1 (both read
// x), which lets h1/h2 reuse x's LDS: 99 KiB, one workgroup (4 waves) per CU, two barriers per layer boundary and none
// inside the k loops.  Per launch at 16 011 rays: 251 workgroups, 52 k-steps x 24 MFMA per wave.
#ifndef TRUNK_FG
#define TRUNK_FG 1
#endif
constexpr int TR = 64;            // rays per workgroup
constexpr int SLD = 264;          // bf16 per LDS row: 256 + 8 -> 528 B, an odd multiple of 16 B (conflict-free 16-B reads)
constexpr int TC = 256;           // feature_c this kernel is built for

// nn.Linear planes Wp [3][256][Kp] -> fragment order Wf[ks][p][nb][lane][8]: lane l of the (nb, ks) fragment holds
// W[32 nb + (l & 31)][16 ks + 8 (l >> 5) + 0..7], the 32x32x16 A-operand map
__global__ void k_frag_order(const __bf16* __restrict__ Wp, __bf16* __restrict__ Wf, int Kp) {
    const int nks = Kp / 16;
    const int64_t n = (int64_t)nks * 3 * 8 * 64 * 8;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(t & 7), lane = (int)((t >> 3) & 63), nb = (int)((t >> 9) & 7);
        int64_t rest = t >> 12;
        int pl = (int)(rest % 3), ks = (int)(rest / 3);
        int row = nb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + i;
        Wf[t] = Wp[((size_t)pl * TC + row) * Kp + k];
    }
}
hipError_t launch_frag_order(const void* Wp, void* Wf, int Kp, hipStream_t s) {
    int64_t n = (int64_t)(Kp / 16) * 3 * 8 * 64 * 8;
    hipLaunchKernelGGL(k_frag_order, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const __bf16*)Wp, (__bf16*)Wf, Kp);
    return hipGetLastError();
}

// FG = 32-feature groups per wave: 2 -> four waves of 64 features (256 threads), 1 -> eight waves of 32 features (512
// threads, two waves per SIMD: the non-MFMA phases -- encoder input, plane writes, logits epilogue -- are spread over
// twice the waves and one wave's waits hide behind the other's MFMAs).  Same arithmetic per output element either way.
template <int FG> struct WFrag { bf16x8 p[FG][3]; };      // [feature group][plane]

template <int FG>
__device__ inline void trunk_load_w(WFrag<FG>& w, const uint4* __restrict__ Wf, int ks, int wave, int lane) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int fg = 0; fg < FG; ++fg) {
            uint4 v = Wf[(((size_t)ks * 3 + pl) * 8 + FG * wave + fg) * 64 + lane];
            w.p[fg][pl] = *reinterpret_cast<bf16x8*>(&v);
        }
}

// acc[fg][rg] += W(fg) * act(rg) over one 16-wide k-step, six bf16 products per tile, smallest contributions first
template <int FG>
__device__ inline void trunk_mfma(f32x16 (&acc)[FG][2], const WFrag<FG>& w, const bf16x8 (&a)[2][3]) {
#pragma unroll
    for (int fg = 0; fg < FG; ++fg)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[fg][2], a[rg][0], acc[fg][rg], 0, 0, 0);
            acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[fg][0], a[rg][2], acc[fg][rg], 0, 0, 0);
            acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[fg][1], a[rg][1], acc[fg][rg], 0, 0, 0);
            acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[fg][1], a[rg][0], acc[fg][rg], 0, 0, 0);
            acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[fg][0], a[rg][1], acc[fg][rg], 0, 0, 0);
            acc[fg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.p[fg][0], a[rg][0], acc[fg][rg], 0, 0, 0);
        }
}

// acc[tg][rg] += act(rg) * Q(tg): rows = rays, columns = tokens (the logits tile; operands swapped so that the softmax
// reduction over rays runs down a lane's registers instead of across lanes)
template <int FG>
__device__ inline void trunk_mfma_t(f32x16 (&acc)[FG][2], const WFrag<FG>& w, const bf16x8 (&a)[2][3]) {
#pragma unroll
    for (int tg = 0; tg < FG; ++tg)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rg][0], w.p[tg][2], acc[tg][rg], 0, 0, 0);
            acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rg][2], w.p[tg][0], acc[tg][rg], 0, 0, 0);
            acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rg][1], w.p[tg][1], acc[tg][rg], 0, 0, 0);
            acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rg][0], w.p[tg][1], acc[tg][rg], 0, 0, 0);
            acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rg][1], w.p[tg][0], acc[tg][rg], 0, 0, 0);
            acc[tg][rg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rg][0], w.p[tg][0], acc[tg][rg], 0, 0, 0);
        }
}

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };     // 16-byte store at 4-byte alignment

// folded query rows qf [M][ld] (iff_q_fold) -> three bf16 planes in the fragment order of k5_trunk, one 256-token block
// after the other: Qf[tb][ks][plane][nb][lane][8], token = 256 tb + 32 nb + (lane & 31), feature = 16 ks + 8 (lane >> 5) + i;
// tokens beyond M are zero
__global__ void k_qf_frag(const float* __restrict__ qf, int ld, int M, __bf16* __restrict__ Qf, int n_tb) {
    const int64_t n = (int64_t)n_tb * (TC / 16) * 8 * 64 * 8;           // elements per plane
    qf += (size_t)blockIdx.y * M * ld;                                   // blockIdx.y = query of a batch: qf [B*M][ld]
    Qf += (size_t)blockIdx.y * 3 * n;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(t & 7), lane = (int)((t >> 3) & 63), nb = (int)((t >> 9) & 7);
        int rest = (int)(t >> 12);
        int ks = rest % (TC / 16), tb = rest / (TC / 16);
        int tok = tb * 256 + nb * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + i;
        float v = tok < M ? qf[(size_t)tok * ld + k] : 0.0f;
        __bf16 h0 = (__bf16)v;
        float r1 = v - (float)h0;
        __bf16 h1 = (__bf16)r1;
        float r2 = r1 - (float)h1;
        // destination index of (tb, ks, plane, nb, lane, i)
        size_t base = ((((size_t)tb * (TC / 16) + ks) * 3) * 8 + nb) * 512 + lane * 8 + i;
        Qf[base] = h0; Qf[base + 8 * 512] = h1; Qf[base + 16 * 512] = (__bf16)r2;
    }
}

// per-token softmax statistics from the per-workgroup partials of k5_trunk<true>: part [n_blk][Mpad][2] = (max, sum exp)
// over each workgroup's 64 rays -> row_max [M], row_sumexp [M]; one wave per token, fixed merge order
// `rows` (optional): kept rows per 256-token block (iff_logits_from_cache_rows): the logits launch skipped the rows behind them and left
// no partials; their statistics become (+inf, 1), the pair k_mask_token_rows gives a dropped row.
__global__ void __launch_bounds__(256) k6_merge_stats(const float2* __restrict__ part, int n_blk, int Mpad, int M,
                                                      float* __restrict__ row_max, float* __restrict__ row_sumexp, const int* __restrict__ rows) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= M) return;
    part += (size_t)blockIdx.y * n_blk * Mpad; row_max += (size_t)blockIdx.y * M; row_sumexp += (size_t)blockIdx.y * M;
    if (rows && (t & 255) >= rows[t >> 8]) {
        if (lane == 0) { row_max[t] = INFINITY; row_sumexp[t] = 1.0f; }
        return;
    }
    float m = -INFINITY, sacc = 0.0f;
    for (int b = lane; b < n_blk; b += 64) {
        float2 p = part[(size_t)b * Mpad + t];
        if (p.x > m) { sacc = sacc * expf(m - p.x); m = p.x; }
        if (p.x != -INFINITY) sacc += p.y * expf(p.x - m);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(sacc, off, 64);
        float mm = fmaxf(m, m2);
        sacc = ((m == -INFINITY) ? 0.0f : sacc * expf(m - mm)) + ((m2 == -INFINITY) ? 0.0f : s2 * expf(m2 - mm));
        m = mm;
    }
    if (lane == 0) { row_max[t] = m; row_sumexp[t] = sacc; }
}

// LOGITS = false: h3 [N][256] out.  LOGITS = true: h3 stays in LDS and is multiplied with the folded query planes Qf
// ("layer 4", multihead_attention.py:6-7 folded): logits [M][N] = (qf[:, :256] h3^T + qf[:, 256]) / divisor out, plus
// this workgroup's softmax partials (max, sum exp over its 64 rays) per token.
template <bool LOGITS, int FG>
__global__ void __launch_bounds__(FG == 2 ? 256 : 512) k5_trunk(const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                                const float* __restrict__ ray_c, int64_t N, const uint4* __restrict__ Wf1,
                                                const uint4* __restrict__ Wf2, const uint4* __restrict__ Wf3,
                                                const float* __restrict__ b1, const float* __restrict__ b2,
                                                const float* __restrict__ b3, float* __restrict__ h3,
                                                const uint4* __restrict__ Qf, const float* __restrict__ rowc, int rowc_ld,
                                                int M, float divisor, float* __restrict__ logits,
                                                float2* __restrict__ part, int Mpad) {
    __shared__ __attribute__((aligned(16))) __bf16 S[3][TR][SLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * TR;
    {   // blockIdx.y = query of a batch: its own rays [N,3], folded query planes, logits [M,N] and partials
        const size_t qb = blockIdx.y;
        ray_o += qb * N * 3; ray_d += qb * N * 3; ray_c += qb * N * 3;
        if (LOGITS) {
            Qf += qb * (size_t)(Mpad / 256) * (TC / 16) * 3 * 8 * 64;
            rowc += qb * (size_t)M * rowc_ld;
            logits += qb * (size_t)M * N;
            part += qb * (size_t)gridDim.x * Mpad;
        } else {
            h3 += qb * N * TC;
        }
    }

    // encoder input x (ray_preprocessor.py:30-37, tensorBase.py:14-20) straight into LDS as three bf16 planes:
    // x = [o, d, rgb, PE(o,8), PE(d,8), PE(rgb,6)] (141 columns, zero-padded to 160; rows beyond N are zero).
    // Work items per ray: 66 (source component, frequency) pairs -- one sincosf each serves the sin and the cos column --
    // plus the 9 raw values and the 19 pad columns: 94 items x 64 rays over 256 threads.
    {
        auto put = [&](int ray, int col, float v) {
            __bf16 h0 = (__bf16)v;
            float r1 = v - (float)h0;
            __bf16 h1 = (__bf16)r1;
            float r2 = r1 - (float)h1;
            S[0][ray][col] = h0; S[1][ray][col] = h1; S[2][ray][col] = (__bf16)r2;
        };
        const int ray = tid & 63;
        const int64_t gr = row0 + ray;
        const bool ok = gr < N;
        float src[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            src[c] = ok ? ray_o[3 * gr + c] : 0.0f;
            src[3 + c] = ok ? ray_d[3 * gr + c] : 0.0f;
            src[6 + c] = ok ? ray_c[3 * gr + c] : 0.0f;
        }
        for (int item = tid >> 6; item < 94; item += 8 / FG) {       // wave-uniform item -> no divergence
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

    auto load_act = [&](bf16x8 (&a)[2][3], int ks) {
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[rg][pl] = *reinterpret_cast<const bf16x8*>(&S[pl][32 * rg + lr][16 * ks + 8 * lh]);
    };
    // relu(acc + bias) -> three bf16 planes of this wave's 64 features for all 64 rays
    auto write_planes = [&](const f32x16 (&acc)[FG][2], const float* __restrict__ bias) {
#pragma unroll
        for (int fg = 0; fg < FG; ++fg)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 32 * FG * wave + 32 * fg + 8 * q + 4 * lh;     // features f0..f0+3 <- registers 4q..4q+3
                const float4 bv = *reinterpret_cast<const float4*>(bias + f0);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    bf16x4 p0, p1, p2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = fmaxf(acc[fg][rg][4 * q + i] + bb[i], 0.0f);
                        __bf16 h0 = (__bf16)v;
                        float r1 = v - (float)h0;
                        __bf16 h1 = (__bf16)r1;
                        float r2 = r1 - (float)h1;
                        p0[i] = h0; p1[i] = h1; p2[i] = (__bf16)r2;
                    }
                    *reinterpret_cast<bf16x4*>(&S[0][32 * rg + lr][f0]) = p0;
                    *reinterpret_cast<bf16x4*>(&S[1][32 * rg + lr][f0]) = p1;
                    *reinterpret_cast<bf16x4*>(&S[2][32 * rg + lr][f0]) = p2;
                }
            }
    };
    auto zero = [](f32x16 (&acc)[FG][2]) {
#pragma unroll
        for (int a = 0; a < FG; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    };

    f32x16 acc[FG][2], acc3[FG][2];
    zero(acc); zero(acc3);
    // x has 141 real columns: 9 k-steps cover them (columns 144..159 of the padded layout are zero in x and are skipped)
    constexpr int KX = (141 + 15) / 16, KH = TC / 16;
    static_assert(KX * 16 <= XW, "x planes are laid out for XW columns");

    // One k loop, software-pipelined in registers: the weight fragments of k-step ks + DEPTH - 1 are requested before
    // k-step ks is multiplied (an L2 hit under load takes longer than one k-step's 24 MFMAs).  DUAL: two weight streams
    // over the same activations (layer 1 and the x-part of layer 3).
    auto phase = [&](auto nk_c, auto depth_c, auto dual_c, auto swap_c, f32x16 (&accA)[FG][2], const uint4* __restrict__ WA,
                     int ksA, f32x16 (&accB)[FG][2], const uint4* __restrict__ WB, int ksB) {
        constexpr int NK = decltype(nk_c)::value, DEPTH = decltype(depth_c)::value;
        constexpr bool DUAL = decltype(dual_c)::value, SWAP = decltype(swap_c)::value;
        WFrag<FG> wa[DEPTH], wb[DUAL ? DEPTH : 1];
        bf16x8 act[2][2][3];                   // activations one k-step ahead as well (LDS latency)
        load_act(act[0], 0);
#pragma unroll
        for (int i = 0; i < DEPTH - 1; ++i) {
            if (i < NK) {
                trunk_load_w(wa[i], WA, ksA + i, wave, lane);
                if (DUAL) trunk_load_w(wb[i], WB, ksB + i, wave, lane);
            }
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks + DEPTH - 1 < NK) {
                trunk_load_w(wa[(ks + DEPTH - 1) % DEPTH], WA, ksA + ks + DEPTH - 1, wave, lane);
                if (DUAL) trunk_load_w(wb[(ks + DEPTH - 1) % DEPTH], WB, ksB + ks + DEPTH - 1, wave, lane);
            }
            if (ks + 1 < NK) load_act(act[(ks + 1) & 1], ks + 1);
            __builtin_amdgcn_sched_barrier(0);       // keep the requests ahead of this k-step's MFMAs (the scheduler sinks them)
            if (SWAP) trunk_mfma_t(accA, wa[ks % DEPTH], act[ks & 1]);
            else trunk_mfma(accA, wa[ks % DEPTH], act[ks & 1]);
            if (DUAL) trunk_mfma(accB, wb[ks % DEPTH], act[ks & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;
    using no_t = integral_constant<bool, false>;
    using yes_t = integral_constant<bool, true>;

    // layer 1 and the x-part of layer 3 (weight columns 256..415 = k-steps 16..25 of Wf3), one pass over x
    phase(integral_constant<int, KX>{}, integral_constant<int, 3>{}, yes_t{}, no_t{}, acc, Wf1, 0, acc3, Wf3, KH);
    __syncthreads();                      // every wave has finished reading x
    write_planes(acc, b1);
    __syncthreads();

    // layer 2
    zero(acc);
    phase(integral_constant<int, KH>{}, integral_constant<int, 4>{}, no_t{}, no_t{}, acc, Wf2, 0, acc, Wf2, 0);
    __syncthreads();
    write_planes(acc, b2);
    __syncthreads();

    // layer 3, h-part (k-steps 0..15 of Wf3), on top of the x-part
    phase(integral_constant<int, KH>{}, integral_constant<int, 4>{}, no_t{}, no_t{}, acc3, Wf3, 0, acc3, Wf3, 0);
    __syncthreads();                      // every wave has finished reading h2
    if (LOGITS) {
        write_planes(acc3, b3);           // h3 planes
        __syncthreads();
        const int n_tb = Mpad / 256;
        const float inv_div = 1.0f / divisor;
        for (int tb = 0; tb < n_tb; ++tb) {
            zero(acc);
            phase(integral_constant<int, KH>{}, integral_constant<int, 4>{}, no_t{}, yes_t{}, acc, Qf + (size_t)tb * (KH * 3 * 8 * 64), 0,
                  acc, Qf, 0);
            // tile: rows = rays 32 rg + (reg & 3) + 8 (reg >> 2) + 4 lh, columns = tokens 256 tb + 32 FG wave + 32 tg + lr
#pragma unroll
            for (int tg = 0; tg < FG; ++tg) {
                const int tok = tb * 256 + 32 * FG * wave + 32 * tg + lr;
                const bool tok_ok = tok < M;
                const float rc = tok_ok ? rowc[(size_t)tok * rowc_ld] : 0.0f;
                float vmax = -INFINITY;
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t ray = row0 + 32 * rg + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        // (acc + rc) / divisor, correctly rounded, without the ten-instruction division sequence: quotient
                        // estimate by the (correctly rounded) reciprocal, exact remainder, one correction
                        const float num = acc[tg][rg][r] + rc;
                        const float q1 = num * inv_div;
                        float v = fmaf(fmaf(-q1, divisor, num), inv_div, q1);
                        acc[tg][rg][r] = v;
                        vmax = fmaxf(vmax, ray < N ? v : -INFINITY);
                    }
                vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
                float ssum = 0.0f;
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int64_t ray0 = row0 + 32 * rg + 8 * q + 4 * lh;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (ray0 + i < N) ssum += __expf(acc[tg][rg][4 * q + i] - vmax);
                        if (tok_ok) {
                            float* dst = logits + (size_t)tok * N + ray0;
                            if (ray0 + 3 < N) {
                                f4u o4 = {acc[tg][rg][4 * q], acc[tg][rg][4 * q + 1], acc[tg][rg][4 * q + 2], acc[tg][rg][4 * q + 3]};
                                *reinterpret_cast<f4u*>(dst) = o4;
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (ray0 + i < N) dst[i] = acc[tg][rg][4 * q + i];
                            }
                        }
                    }
                ssum += __shfl_xor(ssum, 32, 64);
                if (lh == 0) part[(size_t)blockIdx.x * Mpad + tok] = make_float2(vmax, ssum);
            }
        }
        return;
    }
    // h3 = relu(. + b3).  A lane holds 4 consecutive features of one ray per register quad; the tile is transposed
    // through LDS (fp32 [ray][260]) so that every global store instruction writes one whole 1-KiB row of h3.
    constexpr int OLD = TC + 4;           // 1040-B rows: an odd multiple of 16 B
    float* O = reinterpret_cast<float*>(&S[0][0][0]);
    static_assert(TR * OLD * 4 <= 3 * TR * SLD * 2, "output tile must fit in the activation planes");
#pragma unroll
    for (int fg = 0; fg < FG; ++fg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * FG * wave + 32 * fg + 8 * q + 4 * lh;
            const float4 bv = *reinterpret_cast<const float4*>(b3 + f0);
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                float4 o4;
                o4.x = fmaxf(acc3[fg][rg][4 * q + 0] + bv.x, 0.0f);
                o4.y = fmaxf(acc3[fg][rg][4 * q + 1] + bv.y, 0.0f);
                o4.z = fmaxf(acc3[fg][rg][4 * q + 2] + bv.z, 0.0f);
                o4.w = fmaxf(acc3[fg][rg][4 * q + 3] + bv.w, 0.0f);
                *reinterpret_cast<float4*>(&O[(32 * rg + lr) * OLD + f0]) = o4;
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TR * FG / 8; ++i) {
        const int ray = wave * (TR * FG / 8) + i;
        const int64_t gr = row0 + ray;
        if (gr < N) *reinterpret_cast<float4*>(h3 + gr * TC + 4 * lane) = *reinterpret_cast<const float4*>(&O[ray * OLD + 4 * lane]);
    }
}

hipError_t launch_merge_stats(const float2* part, int n_blk, int Mpad, int M, int B, float* row_max, float* row_sumexp, const int* rows,
                              hipStream_t s) {
    hipLaunchKernelGGL(k6_merge_stats, dim3((unsigned)((M + 3) / 4), (unsigned)B), dim3(256), 0, s, part, n_blk, Mpad, M, row_max, row_sumexp, rows);
    return hipGetLastError();
}

size_t ray_encode_workspace_bytes(const IdNetDev& n, int64_t N) {
    // x [N,XW] + two ping-pong activations [N, max(feature_c, fea)]
    int wide = n.feature_c > n.fea ? n.feature_c : n.fea;
    return (size_t)N * (XW + 2 * (size_t)wide) * sizeof(float) + 256;
}
size_t ray_trunk_workspace_bytes(const IdNetDev& n, int64_t N) {
    // x [N,XW] + two activations [N, feature_c]
    return (size_t)N * (XW + 2 * (size_t)n.feature_c) * sizeof(float) + 256;
}

// the three ReLU layers: mlp.0, mlp.2, mlp2.0 (ray_preprocessor.py:9-22,30-38) -> h3 [N, feature_c]
static hipError_t trunk(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, float* x, float* h1,
                        float* h2, float* h3, hipStream_t s) {
    const int C = n.feature_c;
    int64_t tot = N * XW;
    int grid = (int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256);
    if (n.trunk_f16) return launch_trunk_h_features(n, o, d, rgb, N, 1, h3, s);
    if (n.gemm_mode == 1 && C == TC && n.f1 && n.fused_trunk) {
        hipLaunchKernelGGL((k5_trunk<false, TRUNK_FG>), dim3((unsigned)((N + TR - 1) / TR)), dim3(64 * 8 / TRUNK_FG), 0, s, o, d, rgb, N, (const uint4*)n.f1,
                           (const uint4*)n.f2, (const uint4*)n.f3, n.b1, n.b2, n.b3, h3, (const uint4*)nullptr,
                           (const float*)nullptr, 0, 0, 1.0f, (float*)nullptr, (float2*)nullptr, 0);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k5_ray_input, dim3(grid), dim3(256), 0, s, o, d, rgb, N, x);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (n.gemm_mode == 1) {
        if ((e = gemm_bf3<true>(x, XW, XW, nullptr, 0, 0, n.p1, XW, C, n.b1, h1, C, N, s)) != hipSuccess) return e;
        if ((e = gemm_bf3<true>(h1, C, C, nullptr, 0, 0, n.p2, C, C, n.b2, h2, C, N, s)) != hipSuccess) return e;
        return gemm_bf3<true>(h2, C, C, x, XW, XW, n.p3, C + XW, C, n.b3, h3, C, N, s);
    }
    if ((e = gemm_nn<true>(x, XW, XW, nullptr, 0, 0, n.w1, C, n.b1, h1, C, N, s)) != hipSuccess) return e;
    if ((e = gemm_nn<true>(h1, C, C, nullptr, 0, 0, n.w2, C, n.b2, h2, C, N, s)) != hipSuccess) return e;
    return gemm_nn<true>(h2, C, C, x, XW, XW, n.w3, C, n.b3, h3, C, N, s);
}

hipError_t launch_ray_trunk(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, float* h3, void* ws,
                            size_t ws_bytes, hipStream_t s) {
    if (N == 0) return hipSuccess;
    if (ws_bytes < ray_trunk_workspace_bytes(n, N)) return hipErrorInvalidValue;
    float* x = (float*)ws;
    float* h1 = x + (size_t)N * XW;
    float* h2 = h1 + (size_t)N * n.feature_c;
    return trunk(n, o, d, rgb, N, x, h1, h2, h3, s);
}

hipError_t launch_ray_encode(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, float* feat,
                             float* kout, void* ws, size_t ws_bytes, hipStream_t s) {
    if (N == 0) return hipSuccess;
    if (ws_bytes < ray_encode_workspace_bytes(n, N)) return hipErrorInvalidValue;
    int wide = n.feature_c > n.fea ? n.feature_c : n.fea;
    float* x = (float*)ws;
    float* h1 = x + (size_t)N * XW;
    float* h2 = h1 + (size_t)N * wide;
    const int C = n.feature_c;
    // mlp: Linear(141,C) ReLU Linear(C,C) ReLU ; mlp2: Linear(C+141,C) ReLU Linear(C,fea)   (ray_preprocessor.py:9-25)
    hipError_t e = trunk(n, o, d, rgb, N, x, h1, h2, h1, s);
    if (e != hipSuccess) return e;
    float* f_out = feat ? feat : h2;
    if (n.gemm_mode == 1) {
        if ((e = gemm_bf3<false>(h1, C, C, nullptr, 0, 0, n.p4, C, n.fea, n.b4, f_out, n.fea, N, s)) != hipSuccess) return e;
        if (kout) e = gemm_bf3<false>(f_out, n.fea, n.fea, nullptr, 0, 0, n.pk, n.fea, n.fea, n.bk, kout, n.fea, N, s);
        return e;
    }
    if ((e = gemm_nn<false>(h1, C, C, nullptr, 0, 0, n.w4, n.fea, n.b4, f_out, n.fea, N, s)) != hipSuccess) return e;
    if (kout) e = gemm_nn<false>(f_out, n.fea, n.fea, nullptr, 0, 0, n.wk, n.fea, n.bk, kout, n.fea, N, s);
    return e;
}

hipError_t launch_k_proj(const IdNetDev& n, const float* feat, int64_t N, float* kout, hipStream_t s) {
    if (N == 0) return hipSuccess;
    if (n.gemm_mode == 1) return gemm_bf3<false>(feat, n.fea, n.fea, nullptr, 0, 0, n.pk, n.fea, n.fea, n.bk, kout, n.fea, N, s);
    return gemm_nn<false>(feat, n.fea, n.fea, nullptr, 0, 0, n.wk, n.fea, n.bk, kout, n.fea, N, s);
}

// q_proj: M <= 256 token rows -- far too few rows for the 128-row tile.  One wave per 16x16 output tile on
// v_mfma_f32_16x16x4_f32, operands straight from L2 (A [M][K] row-major, unpadded; Wt [K_pad][N] k-major with zero rows
// beyond K), 4 waves = a 16 x 64 strip per workgroup: 96 workgroups for 256 x 384.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_gemm_small(const float* __restrict__ A, int K, int K_pad, const float* __restrict__ Wt,
                                                    int Nout, const float* __restrict__ bias, float* __restrict__ Y, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 16, n0 = (blockIdx.y * 4 + wave) * 16;
    if (n0 >= Nout) return;
    const int r = lane & 15, kk = lane >> 4;
    const bool row_ok = (m0 + r) < M;
    const float* arow = A + (size_t)(m0 + r) * K;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    // 16 k-steps (64 k) per trip: all 32 operand loads are issued before the first MFMA consumes one, so the L2
    // latency is paid once per trip instead of once per step
    for (; k < K_pad; k += 64) {
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int ka = k + 4 * u + kk;
            av[u] = (row_ok && ka < K) ? arow[ka] : 0.0f;
            bv[u] = (ka < K_pad) ? Wt[(size_t)ka * Nout + n0 + r] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
        }
    }
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
    const float bv = bias ? bias[n0 + r] : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int row = m0 + 4 * (lane >> 4) + g;
        if (row < M) Y[(size_t)row * Nout + n0 + r] = (acc0[g] + acc1[g]) + bv;
    }
}

// Token-side Linears for MANY token rows (a batch of query images: 16 x 256 rows): Y [M][Nout] = A [M][K] Wt + bias with
// Wt [K_pad][Nout] k-major (zero rows beyond K), on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate).  A
// workgroup owns 32 token rows and ALL output columns: the 16-deep k-tile of Wt (17 KB for 272 columns) is staged in LDS
// once and read by the four waves (row half x column half), instead of every 16 x 16 output tile streaming its own 51 KB
// of operands from L2 as k_gemm_small does -- that kernel stays for a single image's <= 256 rows, where it has 4x the
// workgroups.  NTW = 16-column tiles per wave (ceil(Nout / 32)).
// RW = token rows per workgroup: 32 (waves = row half x column half) or 16 (waves = column quarters): a batch of 16-32 images is
// 128-256 workgroups of 32 rows -- one wave per SIMD with nothing to run while it waits for LDS -- so up to 16 384 rows take 16.
template <int NTW, int RW>
__global__ void __launch_bounds__(256) k_gemm_tokens(const float* __restrict__ A, int K, int K_pad, const float* __restrict__ Wt,
                                                     int Nout, const float* __restrict__ bias, float* __restrict__ Y, int M) {
    constexpr int BK = 16, A_LD = RW + 1, NCH = 64 / RW;        // NCH = column shares (2 or 4)
    extern __shared__ float smem_tok[];
    float* As = smem_tok;                        // [BK][A_LD]   (k-major: As[k][row])
    float* Bs = smem_tok + BK * A_LD;            // [BK][Nout]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rh = RW == 32 ? (wave & 1) : 0, ch = RW == 32 ? (wave >> 1) : wave;
    const int r = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.x * RW;
    const int NT = Nout / 16, per = (NT + NCH - 1) / NCH, t0 = ch * per, nt = max(0, min(per, NT - t0));
    // two accumulators per tile, by parity of the 4-deep k-step, added at the end: k_gemm_small's association, so a token row
    // gets the same bits whichever of the two kernels serves it (batched and per-image calls stay bit-identical)
    f32x4 acc[NTW][2];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n4 = Nout / 4, nb4 = BK * n4;      // float4s of one Wt k-tile
    // staging registers: A RW rows x 16 k = 512 / 256 floats -> AE = 2 / 1 per thread; Wt k-tile: ceil(nb4 / 256) float4 per thread (<= 7)
    constexpr int AE = RW / 16;
    const int arow = RW == 32 ? tid >> 3 : tid >> 4, ak = RW == 32 ? (tid & 7) * 2 : (tid & 15);
    float a_reg[AE];
    float4 b_reg[7];
    auto load_tile = [&](int k0) {
        const int gr = m0 + arow;
#pragma unroll
        for (int e = 0; e < AE; ++e) a_reg[e] = (gr < M && k0 + ak + e < K) ? A[(size_t)gr * K + k0 + ak + e] : 0.0f;
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int f = tid + u * 256;
            b_reg[u] = f < nb4 ? *reinterpret_cast<const float4*>(Wt + (size_t)(k0 + f / n4) * Nout + 4 * (f % n4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < AE; ++e) As[(ak + e) * A_LD + arow] = a_reg[e];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int f = tid + u * 256;
            if (f < nb4) *reinterpret_cast<float4*>(Bs + (size_t)(f / n4) * Nout + 4 * (f % n4)) = b_reg[u];
        }
    };
    load_tile(0);
    for (int k0 = 0; k0 < K_pad; k0 += BK) {
        __syncthreads();                 // the previous tile's operand reads are done
        store_tile();
        __syncthreads();
        if (k0 + BK < K_pad) load_tile(k0 + BK);      // in flight while this tile is multiplied
#pragma unroll
        for (int k4 = 0; k4 < BK; k4 += 4) {
            const float av = As[(k4 + kk) * A_LD + rh * 16 + r];
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                if (j < nt) acc[j][(k4 >> 2) & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bs[(size_t)(k4 + kk) * Nout + 16 * (t0 + j) + r],
                                                                                          acc[j][(k4 >> 2) & 1], 0, 0, 0);
        }
    }
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        if (j >= nt) continue;
        const int col = 16 * (t0 + j) + r;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = m0 + rh * 16 + 4 * kk + g;
            if (row < M) Y[(size_t)row * Nout + col] = (acc[j][0][g] + acc[j][1][g]) + bv;
        }
    }
}

static hipError_t gemm_tokens(const float* A, int K, int K_pad, const float* Wt, int Nout, const float* bias, float* Y, int M,
                              hipStream_t s) {
    // many rows: the LDS-tiled form; few rows (one image): one wave per 16 x 16 tile fills the chip better
    const int NT = Nout / 16;
    if (M > 512 && Nout % 16 == 0 && Nout / 4 * 16 <= 7 * 256 && (NT + 1) / 2 <= 12) {
        const size_t lds = (size_t)(16 * 33 + 16 * Nout) * sizeof(float);
        if (M <= 16384) {                        // 16-row workgroups: twice as many, two or more per CU
            dim3 grid((unsigned)((M + 15) / 16));
            if ((NT + 3) / 4 <= 5) hipLaunchKernelGGL((k_gemm_tokens<5, 16>), grid, dim3(256), lds, s, A, K, K_pad, Wt, Nout, bias, Y, M);
            else hipLaunchKernelGGL((k_gemm_tokens<6, 16>), grid, dim3(256), lds, s, A, K, K_pad, Wt, Nout, bias, Y, M);
            return hipGetLastError();
        }
        dim3 grid((unsigned)((M + 31) / 32));
        if ((NT + 1) / 2 <= 9) hipLaunchKernelGGL((k_gemm_tokens<9, 32>), grid, dim3(256), lds, s, A, K, K_pad, Wt, Nout, bias, Y, M);
        else hipLaunchKernelGGL((k_gemm_tokens<12, 32>), grid, dim3(256), lds, s, A, K, K_pad, Wt, Nout, bias, Y, M);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((M + 15) / 16), (unsigned)((Nout + 63) / 64));
    hipLaunchKernelGGL(k_gemm_small, grid, dim3(256), 0, s, A, K, K_pad, Wt, Nout, bias, Y, M);
    return hipGetLastError();
}

hipError_t launch_q_proj(const IdNetDev& n, const float* img, int M, float* q, void* scratch, hipStream_t s) {
    (void)scratch;
    if (M == 0) return hipSuccess;
    int kp = (n.img_fea + 15) / 16 * 16;
    return gemm_tokens(img, n.img_fea, kp, n.wq, n.fea, n.bq, q, M, s);
}

// ------------------------------------------------------------------------------------------------ K6
// one workgroup per image token row: row_max, row_sumexp = sum_j exp(l_ij - row_max), in ONE pass over the row
// (running maximum with rescale per thread, then a fixed-order merge of the 256 (max, sum) pairs)
__global__ void __launch_bounds__(256) k6_row_stats(const float* __restrict__ logits, int64_t N, float* __restrict__ row_max,
                                                    float* __restrict__ row_sumexp) {
    __shared__ float red_m[4], red_s[4];
    const float* row = logits + (int64_t)blockIdx.x * N;
    const int tid = threadIdx.x;
    float m = -INFINITY, s = 0.0f;
    for (int64_t j = tid; j < N; j += 256) {
        float x = row[j];
        if (x > m) { s = s * expf(m - x); m = x; }       // exp(-inf) = 0 on the first element
        s += expf(x - m);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
        float mm = fmaxf(m, m2);
        s = ((m == -INFINITY) ? 0.0f : s * expf(m - mm)) + ((m2 == -INFINITY) ? 0.0f : s2 * expf(m2 - mm));
        m = mm;
    }
    if ((tid & 63) == 0) { red_m[tid >> 6] = m; red_s[tid >> 6] = s; }
    __syncthreads();
    if (tid == 0) {
        float mm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
        float ss = 0.0f;
        for (int w = 0; w < 4; ++w) ss += (red_m[w] == -INFINITY) ? 0.0f : red_s[w] * expf(red_m[w] - mm);
        row_max[blockIdx.x] = mm;
        row_sumexp[blockIdx.x] = ss;
    }
}

hipError_t launch_attn_logits(const float* q, const float* k, int M, int64_t N, int D, float divisor, float* logits,
                              float* row_max, float* row_sumexp, int gemm_mode, hipStream_t s) {
    if (M == 0 || N == 0) return hipSuccess;
    if (D % BK3 != 0) return hipErrorInvalidValue;
    if (gemm_mode == 1) {
        dim3 grid3((unsigned)((M + 63) / 64), (unsigned)((N + BN - 1) / BN));
        hipLaunchKernelGGL((k_gemm_bf3<false, true>), grid3, dim3(256), 0, s, q, D, D, (const float*)nullptr, 0, 0,
                           (const __bf16*)nullptr, 0, k, D, (const float*)nullptr, logits, N, (int64_t)M, N, divisor,
                           (const float*)nullptr, 0);
        hipError_t e3 = hipGetLastError();
        if (e3 != hipSuccess) return e3;
        if (row_max && row_sumexp) {
            hipLaunchKernelGGL(k6_row_stats, dim3(M), dim3(256), 0, s, logits, N, row_max, row_sumexp);
            e3 = hipGetLastError();
        }
        return e3;
    }
    constexpr int MTL = 1;
    dim3 grid((unsigned)((M + 64 * MTL - 1) / (64 * MTL)), (unsigned)((N + BN - 1) / BN));
    hipLaunchKernelGGL((k_gemm_f32<false, true, MTL>), grid, dim3(256), 0, s, q, D, D, (const float*)nullptr, 0, 0, k, D,
                       (const float*)nullptr, logits, N, (int64_t)M, N, divisor);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (row_max && row_sumexp) {
        hipLaunchKernelGGL(k6_row_stats, dim3(M), dim3(256), 0, s, logits, N, row_max, row_sumexp);
        e = hipGetLastError();
    }
    return e;
}

// Folded token side (api.hip: fold_heads): qf [M][qf_ld] = img * wqf + bqf.  Columns 0..C-1 dotted with a ray's h3 plus
// column C give q . k of the unfolded chain (multihead_attention.py:60-63 after ray_preprocessor.py:38).
hipError_t launch_q_fold(const IdNetDev& n, const float* img, int M, float* qf, hipStream_t s) {
    if (M == 0) return hipSuccess;
    int kp = (n.img_fea + 15) / 16 * 16;
    return gemm_tokens(img, n.img_fea, kp, n.wqf, n.qf_ld, n.bqf, qf, M, s);
}

hipError_t launch_attn_logits_folded(const float* qf, int ldq, const float* h3, int M, int64_t N, int C, float divisor,
                                     float* logits, float* row_max, float* row_sumexp, hipStream_t s) {
    if (M == 0 || N == 0) return hipSuccess;
    if (C % BK3 != 0 || ldq <= C) return hipErrorInvalidValue;
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + BN - 1) / BN));
    hipLaunchKernelGGL((k_gemm_bf3<false, true>), grid, dim3(256), 0, s, qf, ldq, C, (const float*)nullptr, 0, 0,
                       (const __bf16*)nullptr, 0, h3, C, (const float*)nullptr, logits, N, (int64_t)M, N, divisor, qf + C, ldq);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (row_max && row_sumexp) {
        hipLaunchKernelGGL(k6_row_stats, dim3(M), dim3(256), 0, s, logits, N, row_max, row_sumexp);
        e = hipGetLastError();
    }
    return e;
}

// One call from rays to logits (+ row statistics): ray_input -> trunk -> folded logits.  With feature_c = 256 and the
// 3xBF16 mode this is k5_ray_input_planes + k5_trunk<true> (h3 never leaves the CU) + k6_merge_stats; otherwise the
// layered trunk followed by launch_attn_logits_folded.
static inline size_t up256z(size_t v) { return (v + 255) & ~(size_t)255; }
size_t ray_logits_workspace_bytes(const IdNetDev& n, int64_t N, int M, int B) {
    const size_t n_tb = (size_t)(M + 255) / 256, n_blk = (size_t)(N + TR - 1) / TR;
    size_t fused = up256z((size_t)B * n_tb * (TC / 16) * 3 * 8 * 64 * 16) + up256z((size_t)B * n_blk * n_tb * 256 * 8) +
                   up256z((size_t)B * n_tb * 256 * 4);          // query planes | softmax partials | per-token scales (F16X2)
    size_t layered = up256z(ray_trunk_workspace_bytes(n, N)) + (size_t)N * n.feature_c * sizeof(float);
    return (fused > layered ? fused : layered) + 256;
}

// B queries, each with its own ray set and its own M tokens: o, d, rgb [B][N][3], qf [B*M][qf_ld] -> logits [B][M][N],
// row_max / row_sumexp [B][M].
hipError_t launch_ray_logits_folded(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, const float* qf,
                                    int M, int B, float divisor, float* logits, float* row_max, float* row_sumexp, void* ws,
                                    size_t ws_bytes, float* trunk_ms_host, hipStream_t s) {
    if (N == 0 || M == 0 || B == 0) return hipSuccess;
    if (ws_bytes < ray_logits_workspace_bytes(n, N, M, B)) return hipErrorInvalidValue;
    const int C = n.feature_c;
    if (!n.trunk_f16 && !(n.gemm_mode == 1 && C == TC && n.f1 && n.fused_trunk)) {
        if (trunk_ms_host) *trunk_ms_host = -1.0f;        // no fused launch to time in this configuration
        float* h3 = (float*)((char*)ws + up256z(ray_trunk_workspace_bytes(n, N)));
        for (int q = 0; q < B; ++q) {
            hipError_t e = launch_ray_trunk(n, o + (size_t)q * N * 3, d + (size_t)q * N * 3, rgb + (size_t)q * N * 3, N, h3, ws,
                                            ray_trunk_workspace_bytes(n, N), s);
            if (e != hipSuccess) return e;
            e = launch_attn_logits_folded(qf + (size_t)q * M * n.qf_ld, n.qf_ld, h3, M, N, C, divisor, logits + (size_t)q * M * N,
                                          row_max ? row_max + (size_t)q * M : nullptr, row_sumexp ? row_sumexp + (size_t)q * M : nullptr, s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    const int n_tb = (M + 255) / 256, Mpad = n_tb * 256;
    const int64_t n_blk = (N + TR - 1) / TR;             // softmax partials per token: one per 64-ray block in every form of the launch
    if (B > 65535) return hipErrorInvalidValue;
    char* base = (char*)ws;
    __bf16* Qf = (__bf16*)base;
    float2* part = (float2*)((char*)Qf + up256z((size_t)B * n_tb * (TC / 16) * 3 * 8 * 64 * 16));
    float* qscale = (float*)((char*)part + up256z((size_t)B * ((N + TR - 1) / TR) * n_tb * 256 * 8));
    hipError_t e;
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (n.trunk_f16) {
        // the token-side split (k_qf_frag_h) is part of launch_trunk_h_logits: the timed span covers it too (a few us)
        if (trunk_ms_host) {
            for (auto& x : ev) if ((e = hipEventCreate(&x)) != hipSuccess) return e;
            (void)hipEventRecord(ev[0], s);
        }
        e = launch_trunk_h_logits(n, o, d, rgb, N, qf, M, B, divisor, logits, (void*)Qf, qscale, part, s);
    } else {
        const int64_t nq = (int64_t)n_tb * (TC / 16) * 8 * 64 * 8;
        hipLaunchKernelGGL(k_qf_frag, dim3((unsigned)((nq + 255) / 256), (unsigned)B), dim3(256), 0, s, qf, n.qf_ld, M, Qf, n_tb);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (trunk_ms_host) {
            for (auto& x : ev) if ((e = hipEventCreate(&x)) != hipSuccess) return e;
            (void)hipEventRecord(ev[0], s);
        }
        hipLaunchKernelGGL((k5_trunk<true, TRUNK_FG>), dim3((unsigned)n_blk, (unsigned)B), dim3(64 * 8 / TRUNK_FG), 0, s, o, d, rgb, N, (const uint4*)n.f1,
                           (const uint4*)n.f2, (const uint4*)n.f3, n.b1, n.b2, n.b3, (float*)nullptr, (const uint4*)Qf, qf + C, n.qf_ld, M,
                           divisor, logits, part, Mpad);
        e = hipGetLastError();
    }
    if (trunk_ms_host) {
        (void)hipEventRecord(ev[1], s);
        hipError_t es = hipEventSynchronize(ev[1]);
        (void)hipEventElapsedTime(trunk_ms_host, ev[0], ev[1]);
        for (auto& x : ev) (void)hipEventDestroy(x);
        if (e == hipSuccess) e = es;
    }
    if (e != hipSuccess) return e;
    if (row_max && row_sumexp) {
        hipLaunchKernelGGL(k6_merge_stats, dim3((unsigned)((M + 3) / 4), (unsigned)B), dim3(256), 0, s, part, (int)n_blk, Mpad, M, row_max,
                           row_sumexp, (const int*)nullptr);
        e = hipGetLastError();
    }
    return e;
}

// attention_ij = exp(l_ij - max_i) / sumexp_i ; score_j = sum_i attention_ij.  A 256-thread workgroup owns 64 ray columns;
// wave g sums the rows of its quarter of the token rows in order, the 4 partial sums are added in fixed order
// (deterministic).  Each wave-instruction reads one 256-B row segment.
// `rows` (optional): rows of query q that count (kept rows first: iff_token_assemble_compact); the others are neither read nor written.
__global__ void __launch_bounds__(256) k6_colsum(float* __restrict__ logits, int M_all, int64_t N, const float* __restrict__ row_max,
                                                 const float* __restrict__ row_sumexp, int write_attention,
                                                 float* __restrict__ score, const int* __restrict__ rows) {
    // 256 ray columns per workgroup, 64 per wave, every wave over ALL token rows: the four waves read four adjacent 256-B
    // pieces of the same rows (1 KiB of a row per workgroup at a time -- the logits stream from HBM, and a row is N floats
    // long); a lane keeps four partial sums (rows i mod 4) and adds them pairwise at the end.
    extern __shared__ float s_stats[];   // [2][M]
    const int tid = threadIdx.x;
    {   // blockIdx.y = query of a batch: logits [Q][M][N], statistics [Q][M], score [Q][N]
        // LAST query first, and below the last ray columns first: the launch follows the logits launch, which wrote query 0 .. Q - 1
        // in dispatch order, so what it wrote last is what the Infinity Cache still holds (256 MB of a 585-MB map at 32 x 16 011
        // rays); read in the writer's own order every line has been evicted by the time it is asked for.  Measured, same box, the
        // two launches back to back: 111 -> 93 us (score + top-k + pose stage 0.175 -> 0.157 ms).  Which workgroup sums which
        // columns does not touch the sums.
        const int64_t qb = gridDim.y - 1 - blockIdx.y;
        logits += qb * M_all * N; row_max += qb * M_all; row_sumexp += qb * M_all; score += qb * N;
    }
    const int M = rows ? min(M_all, max(rows[gridDim.y - 1 - blockIdx.y], 0)) : M_all;
    // scores only (no attention map asked for): exp through the hardware exponential and the row's reciprocal sum instead of
    // expf and a division per element -- 2e-7 relative per term, far inside the 5e-5 the logits themselves carry; the kernel
    // was bound by those ~25 extra vector instructions per logit, not by the 262 MB it streams
    const bool fast = !write_attention;
    for (int i = tid; i < M; i += 256) { s_stats[i] = row_max[i]; s_stats[M + i] = fast ? 1.0f / row_sumexp[i] : row_sumexp[i]; }
    __syncthreads();
    const int64_t j = (int64_t)(gridDim.x - 1 - blockIdx.x) * 256 + tid;
    if (j >= N) return;
    if (fast) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int i = 0;
        for (; i + 16 <= M; i += 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = logits[(int64_t)(i + u) * N + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] = fmaf(__expf(x[u] - s_stats[i + u]), s_stats[M + i + u], acc[u & 3]);
        }
        for (; i < M; ++i) acc[i & 3] = fmaf(__expf(logits[(int64_t)i * N + j] - s_stats[i]), s_stats[M + i], acc[i & 3]);
        score[j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        return;
    }
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int i = 0;
    for (; i + 16 <= M; i += 16) {           // 16 rows requested before the first is used
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = logits[(int64_t)(i + u) * N + j];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float a = expf(x[u] - s_stats[i + u]) / s_stats[M + i + u];
            if (write_attention) logits[(int64_t)(i + u) * N + j] = a;
            acc[u & 3] += a;
        }
    }
    for (; i < M; ++i) {
        const float a = expf(logits[(int64_t)i * N + j] - s_stats[i]) / s_stats[M + i];
        if (write_attention) logits[(int64_t)i * N + j] = a;
        acc[i & 3] += a;
    }
    score[j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

hipError_t launch_attn_colsum(float* logits, int Q, int M, int64_t N, const float* row_max, const float* row_sumexp,
                              int write_attention, float* score, const int* rows, hipStream_t s) {
    if (N == 0 || Q == 0) return hipSuccess;
    hipLaunchKernelGGL(k6_colsum, dim3((unsigned)((N + 255) / 256), (unsigned)Q), dim3(256), 2 * (size_t)M * sizeof(float), s, logits,
                       M, N, row_max, row_sumexp, write_attention, score, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K7
// torch.topk(score, k): one workgroup.  (1) radix-select the key T of the k-th largest score (iff_select.h);
// (2) one sweep gathers everything above T (any order) and the indices of the scores equal to T; (3) ties at T are taken
// lowest index first (sorted only when there are more ties than free slots; more than 1024 ties fall back to an ordered
// sweep); (4) bitonic sort of the next power of two >= k slots by (value desc, index asc).
constexpr int TK_THREADS = 1024;

__device__ inline bool tk_before(float va, int ia, float vb, int ib) {
    uint32_t ka = iff_order_key(va), kb = iff_order_key(vb);
    return (ka > kb) || (ka == kb && ia < ib);
}

__global__ void __launch_bounds__(TK_THREADS) k7_topk(const float* __restrict__ score, int64_t N, int k, int64_t* __restrict__ idx,
                                                      float* __restrict__ val) {
    __shared__ int hist[264];
    __shared__ float s_val[1024];
    __shared__ int s_idx[1024];
    __shared__ int s_eq[1024];
    __shared__ int n_gt, n_eq, eq_base;
    __shared__ int scan[TK_THREADS];
    const int tid = threadIdx.x;
    score += (int64_t)blockIdx.x * N; idx += (int64_t)blockIdx.x * k; val += (int64_t)blockIdx.x * k;   // one workgroup per query
    const uint32_t T = iff_wg_select_key<true>(score, N, k, hist);
    if (tid == 0) { n_gt = 0; n_eq = 0; eq_base = 0; }
    s_val[tid] = -INFINITY; s_idx[tid] = 0x7fffffff; s_eq[tid] = 0x7fffffff;
    __syncthreads();
    for (int64_t i = tid; i < N; i += TK_THREADS) {
        float v = score[i];
        uint32_t key = iff_order_key(v);
        if (key > T) {
            int p = atomicAdd(&n_gt, 1);          // fewer than k elements are above the k-th largest
            s_val[p] = v; s_idx[p] = (int)i;
        } else if (key == T) {
            int p = atomicAdd(&n_eq, 1);
            if (p < 1024) s_eq[p] = (int)i;
        }
    }
    __syncthreads();
    const int gt = n_gt, eq = n_eq, need = k - gt;
    const float vT = iff_order_key_inv(T);
    if (eq <= 1024) {
        if (eq > need) {
            // more ties than slots: ascending bitonic sort of the tie indices, keep the lowest `need`
            for (int size = 2; size <= 1024; size <<= 1)
                for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                    int j = tid ^ stride;
                    if (j > tid) {
                        bool up = ((tid & size) == 0);
                        int a = s_eq[tid], b = s_eq[j];
                        if ((a < b) != up) { s_eq[tid] = b; s_eq[j] = a; }
                    }
                    __syncthreads();
                }
        }
        if (tid < need) { s_val[gt + tid] = vT; s_idx[gt + tid] = s_eq[tid]; }
        __syncthreads();
    } else {
        // pathological tie count: ordered sweep, chunk by chunk, taking the first `need` equal elements
        for (int64_t c0 = 0; c0 < N && eq_base < need; c0 += TK_THREADS) {
            int64_t i = c0 + tid;
            bool e = (i < N) && iff_order_key(score[i]) == T;
            scan[tid] = e ? 1 : 0;
            __syncthreads();
            for (int off = 1; off < TK_THREADS; off <<= 1) {
                int v = (tid >= off) ? scan[tid - off] : 0;
                __syncthreads();
                scan[tid] += v;
                __syncthreads();
            }
            int pos = eq_base + scan[tid] - (e ? 1 : 0);
            if (e && pos < need) { s_val[gt + pos] = vT; s_idx[gt + pos] = (int)i; }
            __syncthreads();
            if (tid == TK_THREADS - 1) eq_base += scan[tid];
            __syncthreads();
        }
    }
    int slots = 1;
    while (slots < k) slots <<= 1;
    for (int size = 2; size <= slots; size <<= 1)
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            int j = tid ^ stride;
            if (tid < slots && j > tid) {
                bool up = ((tid & size) == 0);
                float vi = s_val[tid], vj = s_val[j];
                int ii = s_idx[tid], ij = s_idx[j];
                if (tk_before(vi, ii, vj, ij) != up) { s_val[tid] = vj; s_val[j] = vi; s_idx[tid] = ij; s_idx[j] = ii; }
            }
            __syncthreads();
        }
    if (tid < k) { idx[tid] = (int64_t)s_idx[tid]; val[tid] = s_val[tid]; }
}

size_t topk_workspace_bytes(int64_t N, int k) { (void)N; (void)k; return 256; }

hipError_t launch_topk(const float* score, int Q, int64_t N, int k, int64_t* idx, float* val, void* ws, size_t ws_bytes, hipStream_t s) {
    (void)ws; (void)ws_bytes;
    if (k < 1 || k > 1024 || k > N || N >= 0x7fffffff || Q < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k7_topk, dim3((unsigned)Q), dim3(TK_THREADS), 0, s, score, N, k, idx, val);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ image-token assembly
// IdentificationModule.image_processing after the backbone (identification_module.py:149-160): every patch token gets the
// 14-channel position code of get_img_position_encoding (:76-99) appended -- [pos_i, pos_j, sin(pos_i {1,2,4}), sin(pos_j
// {1,2,4}), cos(...)], pos = torch.linspace(-1, 1, g) per axis, 'ij' indexing (the two linspace tables come from the host
// so that their bits are torch's) -- and the boolean row selection `[mask > 0.1]` (:157-160) becomes a keep flag per token:
// the rows stay in place (static shapes, no host sync) and k_mask_token_rows later removes the dropped rows' contribution
// from the column sums exactly (exp(l - inf) = 0), which is what deleting the rows does.
struct LinTab { float h[32], w[32]; };
__global__ void k_token_assemble(const float* __restrict__ tok, int Q, int gh, int gw, int C, const float* __restrict__ mask,
                                 float thres, LinTab lt, float* __restrict__ out, uint8_t* __restrict__ keep) {
    const int G = gh * gw, CO = C + 14;
    const int64_t n = (int64_t)Q * G * CO;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(t % CO);
        const int64_t row = t / CO;                  // q * G + cell
        const int cell = (int)(row % G), i = cell / gw, j = cell - i * gw;
        float v;
        if (col < C) v = tok[row * C + col];
        else {
            const int c = col - C;
            const float pi = lt.h[i], pj = lt.w[j];
            if (c < 2) v = c == 0 ? pi : pj;
            else {
                const int u = (c - 2) % 6;                              // (axis, octave) = (u / 3, u % 3)
                const float ang = (u < 3 ? pi : pj) * (float)(1 << (u % 3));
                v = (c - 2) < 6 ? sinf(ang) : cosf(ang);
            }
        }
        out[t] = v;
        if (col == 0) keep[row] = (!mask || mask[row] > thres) ? 1 : 0;
    }
}
hipError_t launch_token_assemble(const float* tok, int Q, int gh, int gw, int C, const float* mask, float thres, const float* lin_h,
                                 const float* lin_w, float* out, uint8_t* keep, hipStream_t s) {
    LinTab lt;
    for (int i = 0; i < 32; ++i) { lt.h[i] = i < gh ? lin_h[i] : 0.0f; lt.w[i] = i < gw ? lin_w[i] : 0.0f; }
    const int64_t n = (int64_t)Q * gh * gw * (C + 14);
    int64_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_token_assemble, dim3((unsigned)grid), dim3(256), 0, s, tok, Q, gh, gw, C, mask, thres, lt, out, keep);
    return hipGetLastError();
}

// The same rows, KEPT ROWS FIRST: one workgroup per image partitions its G rows stably (kept rows in grid order, then the dropped
// ones in grid order) and reports how many it kept.  identification_module.py:157-160 deletes the dropped rows before the attention,
// so its work per image follows the kept count; with the kept rows in front the logits launch and the column pass can stop at
// rows_out[q] (iff_logits_from_cache_rows, iff_attn_colsum_rows) -- the same saving without a host sync or a changing shape.
// The values are k_token_assemble's, element for element; keep_out becomes 1 ... 1 0 ... 0.
__global__ void __launch_bounds__(256) k_token_assemble_compact(const float* __restrict__ tok, int gh, int gw, int C, const float* __restrict__ mask,
                                                                float thres, LinTab lt, float* __restrict__ out, uint8_t* __restrict__ keep,
                                                                int* __restrict__ rows_out) {
    __shared__ int s_dest[1024];                    // destination row of grid cell g
    __shared__ int s_count[256 + 1];
    const int G = gh * gw, CO = C + 14, tid = threadIdx.x, q = blockIdx.x;
    const int per = (G + 255) / 256;                // cells per thread, consecutive: a thread's cells keep their order
    const int g0 = tid * per, g1 = min(G, g0 + per);
    int mine = 0;
    for (int g = g0; g < g1; ++g) mine += (!mask || mask[(int64_t)q * G + g] > thres) ? 1 : 0;
    s_count[tid + 1] = mine;
    if (tid == 0) s_count[0] = 0;
    __syncthreads();
    if (tid == 0) for (int i = 1; i <= 256; ++i) s_count[i] += s_count[i - 1];          // 256 adds: not worth a scan
    __syncthreads();
    const int n_keep = s_count[256];
    int k = s_count[tid], d = n_keep + (g0 - s_count[tid]);                           // next kept / dropped destination of this thread
    for (int g = g0; g < g1; ++g) {
        const bool kp = !mask || mask[(int64_t)q * G + g] > thres;
        s_dest[g] = kp ? k++ : d++;
    }
    __syncthreads();
    if (tid == 0) rows_out[q] = n_keep;
    for (int g = tid; g < G; g += 256) keep[(int64_t)q * G + s_dest[g]] = s_dest[g] < n_keep ? 1 : 0;
    const int64_t n = (int64_t)G * CO;
    for (int64_t t = tid; t < n; t += 256) {
        const int col = (int)(t % CO), cell = (int)(t / CO), i = cell / gw, j = cell - i * gw;
        float v;
        if (col < C) v = tok[((int64_t)q * G + cell) * C + col];
        else {
            const int c = col - C;
            const float pi = lt.h[i], pj = lt.w[j];
            if (c < 2) v = c == 0 ? pi : pj;
            else {
                const int u = (c - 2) % 6;                              // (axis, octave) = (u / 3, u % 3)
                const float ang = (u < 3 ? pi : pj) * (float)(1 << (u % 3));
                v = (c - 2) < 6 ? sinf(ang) : cosf(ang);
            }
        }
        out[((int64_t)q * G + s_dest[cell]) * CO + col] = v;
    }
}
hipError_t launch_token_assemble_compact(const float* tok, int Q, int gh, int gw, int C, const float* mask, float thres, const float* lin_h,
                                         const float* lin_w, float* out, uint8_t* keep, int* rows_out, hipStream_t s) {
    LinTab lt;
    for (int i = 0; i < 32; ++i) { lt.h[i] = i < gh ? lin_h[i] : 0.0f; lt.w[i] = i < gw ? lin_w[i] : 0.0f; }
    hipLaunchKernelGGL(k_token_assemble_compact, dim3((unsigned)Q), dim3(256), 0, s, tok, gh, gw, C, mask, thres, lt, out, keep, rows_out);
    return hipGetLastError();
}

__global__ void k_mask_token_rows(const uint8_t* __restrict__ keep, int64_t rows, float* __restrict__ row_max, float* __restrict__ row_sumexp) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < rows && !keep[i]) { row_max[i] = INFINITY; row_sumexp[i] = 1.0f; }
}
hipError_t launch_mask_token_rows(const uint8_t* keep, int64_t rows, float* row_max, float* row_sumexp, hipStream_t s) {
    hipLaunchKernelGGL(k_mask_token_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, keep, rows, row_max, row_sumexp);
    return hipGetLastError();
}
