// identify_kernels.hip -- K5 ray encoder (+k_proj), q_proj, K6 attention logits / row statistics / column-sum score,
// K7 top-k.  gfx950, fp32 throughout: the matrix products run on the fp32-input MFMA (v_mfma_f32_32x32x2_f32), which is
// bit-for-bit a k-ordered fmaf chain (exact f32, no reduced precision), so attention logits stay within fp32 rounding
// of the reference and the top-k set is the reference's.
#include "iff_device.h"
#include "iff_launch.h"

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
// ray_preprocessor.py:30-37 + tensorBase.py:14-20: x = [o, d, rgb, PE(o,8), PE(d,8), PE(rgb,6)] (141), zero-padded to 144
__global__ void k5_ray_input(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ c,
                             int64_t N, float* __restrict__ x) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < N * 144; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t / 144;
        int col = (int)(t - r * 144);
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
constexpr int BM = 128, BN = 128, BK = 16;

template <bool RELU, bool B_IS_ROWS>
__global__ void __launch_bounds__(256) k_gemm_f32(const float* __restrict__ A1, int lda1, int K1,
                                                  const float* __restrict__ A2, int lda2, int K2,
                                                  const float* __restrict__ B, int ldb, const float* __restrict__ bias,
                                                  float* __restrict__ Y, int64_t ldy, int64_t M, int64_t Nout, float divisor) {
    __shared__ float As[2][BK][BM + 4];
    __shared__ float Bs[2][BK][BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int64_t col0 = (int64_t)blockIdx.y * BN;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    const int K = K1 + K2;
    const int nk = K / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging registers: A tile 128 rows x 16 k = 512 float4 -> 2 per thread (row = f>>2, kq = f&3)
    float4 ra[2], rb[2];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const float* Asrc; int lda, kk;
        if (k0 < K1) { Asrc = A1; lda = lda1; kk = k0; } else { Asrc = A2; lda = lda2; kk = k0 - K1; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int f = tid + u * 256;
            int r = f >> 2, kq = (f & 3) * 4;
            int64_t gr = row0 + r;
            ra[u] = (gr < M) ? ld4(Asrc + gr * lda + kk + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (B_IS_ROWS) {
                int64_t gc = col0 + r;
                rb[u] = (gc < Nout) ? ld4(B + gc * ldb + k0 + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                int kr = f >> 5, cq = (f & 31) * 4;          // 16 k-rows x 32 float4
                int64_t gc = col0 + cq;
                rb[u] = (gc < Nout) ? ld4(B + (int64_t)(k0 + kr) * ldb + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int f = tid + u * 256;
            int r = f >> 2, kq = (f & 3) * 4;
            As[buf][kq + 0][r] = ra[u].x; As[buf][kq + 1][r] = ra[u].y; As[buf][kq + 2][r] = ra[u].z; As[buf][kq + 3][r] = ra[u].w;
            if (B_IS_ROWS) {
                Bs[buf][kq + 0][r] = rb[u].x; Bs[buf][kq + 1][r] = rb[u].y; Bs[buf][kq + 2][r] = rb[u].z; Bs[buf][kq + 3][r] = rb[u].w;
            } else {
                int kr = f >> 5, cq = (f & 31) * 4;
                *reinterpret_cast<float4*>(&Bs[buf][kr][cq]) = rb[u];
            }
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int k2 = 0; k2 < BK; k2 += 2) {
            float a0 = As[buf][k2 + lk][wr + li], a1 = As[buf][k2 + lk][wr + 32 + li];
            float b0 = Bs[buf][k2 + lk][wc + li], b1 = Bs[buf][k2 + lk][wc + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
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

template <bool RELU>
static hipError_t gemm_nn(const float* A1, int lda1, int K1, const float* A2, int lda2, int K2, const float* Wt, int Nout,
                          const float* bias, float* Y, int64_t ldy, int64_t M, hipStream_t s) {
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((Nout + BN - 1) / BN));
    hipLaunchKernelGGL((k_gemm_f32<RELU, false>), grid, dim3(256), 0, s, A1, lda1, K1, A2, lda2, K2, Wt, Nout, bias, Y, ldy, M,
                       (int64_t)Nout, 1.0f);
    return hipGetLastError();
}

size_t ray_encode_workspace_bytes(const IdNetDev& n, int64_t N) {
    // x [N,144] + two ping-pong activations [N, max(feature_c, fea)]
    int wide = n.feature_c > n.fea ? n.feature_c : n.fea;
    return (size_t)N * (144 + 2 * (size_t)wide) * sizeof(float) + 256;
}

hipError_t launch_ray_encode(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, float* feat,
                             float* kout, void* ws, size_t ws_bytes, hipStream_t s) {
    if (N == 0) return hipSuccess;
    if (ws_bytes < ray_encode_workspace_bytes(n, N)) return hipErrorInvalidValue;
    int wide = n.feature_c > n.fea ? n.feature_c : n.fea;
    float* x = (float*)ws;
    float* h1 = x + (size_t)N * 144;
    float* h2 = h1 + (size_t)N * wide;
    const int C = n.feature_c;
    int64_t tot = N * 144;
    int grid = (int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256);
    hipLaunchKernelGGL(k5_ray_input, dim3(grid), dim3(256), 0, s, o, d, rgb, N, x);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // mlp: Linear(141,C) ReLU Linear(C,C) ReLU ; mlp2: Linear(C+141,C) ReLU Linear(C,fea)   (ray_preprocessor.py:9-25)
    if ((e = gemm_nn<true>(x, 144, 144, nullptr, 0, 0, n.w1, C, n.b1, h1, C, N, s)) != hipSuccess) return e;
    if ((e = gemm_nn<true>(h1, C, C, nullptr, 0, 0, n.w2, C, n.b2, h2, C, N, s)) != hipSuccess) return e;
    if ((e = gemm_nn<true>(h2, C, C, x, 144, 144, n.w3, C, n.b3, h1, C, N, s)) != hipSuccess) return e;
    float* f_out = feat ? feat : h2;
    if ((e = gemm_nn<false>(h1, C, C, nullptr, 0, 0, n.w4, n.fea, n.b4, f_out, n.fea, N, s)) != hipSuccess) return e;
    if (kout) {
        if ((e = gemm_nn<false>(f_out, n.fea, n.fea, nullptr, 0, 0, n.wk, n.fea, n.bk, kout, n.fea, N, s)) != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_k_proj(const IdNetDev& n, const float* feat, int64_t N, float* kout, hipStream_t s) {
    if (N == 0) return hipSuccess;
    return gemm_nn<false>(feat, n.fea, n.fea, nullptr, 0, 0, n.wk, n.fea, n.bk, kout, n.fea, N, s);
}

// q_proj: img [M][img_fea=398] -> pad K to 400 through a scratch copy [M][400]
__global__ void k_pad_rows(const float* __restrict__ src, int cols, float* __restrict__ dst, int cols_pad, int64_t rows) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < rows * cols_pad; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = t / cols_pad;
        int c = (int)(t - r * cols_pad);
        dst[t] = (c < cols) ? src[r * cols + c] : 0.0f;
    }
}
hipError_t launch_q_proj(const IdNetDev& n, const float* img, int M, float* q, void* scratch, hipStream_t s) {
    if (M == 0) return hipSuccess;
    int kp = (n.img_fea + 15) / 16 * 16;
    float* xp = (float*)scratch;
    int grid = (int)(((int64_t)M * kp + 255) / 256);
    hipLaunchKernelGGL(k_pad_rows, dim3(grid), dim3(256), 0, s, img, n.img_fea, xp, kp, (int64_t)M);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return gemm_nn<false>(xp, kp, kp, nullptr, 0, 0, n.wq, n.fea, n.bq, q, n.fea, M, s);
}

// ------------------------------------------------------------------------------------------------ K6
// one workgroup per image token row: row_max, row_sumexp = sum_j exp(l_ij - row_max)   (softmax denominators)
__global__ void __launch_bounds__(256) k6_row_stats(const float* __restrict__ logits, int64_t N, float* __restrict__ row_max,
                                                    float* __restrict__ row_sumexp) {
    __shared__ float red[4];
    const float* row = logits + (int64_t)blockIdx.x * N;
    const int tid = threadIdx.x;
    float m = -INFINITY;
    for (int64_t j = tid; j < N; j += 256) m = fmaxf(m, row[j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.0f;
    for (int64_t j = tid; j < N; j += 256) sum += expf(row[j] - m);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        row_max[blockIdx.x] = m;
        row_sumexp[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

hipError_t launch_attn_logits(const float* q, const float* k, int M, int64_t N, int D, float divisor, float* logits,
                              float* row_max, float* row_sumexp, hipStream_t s) {
    if (M == 0 || N == 0) return hipSuccess;
    if (D % BK != 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    hipLaunchKernelGGL((k_gemm_f32<false, true>), grid, dim3(256), 0, s, q, D, D, (const float*)nullptr, 0, 0, k, D,
                       (const float*)nullptr, logits, N, (int64_t)M, N, divisor);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (row_max && row_sumexp) {
        hipLaunchKernelGGL(k6_row_stats, dim3(M), dim3(256), 0, s, logits, N, row_max, row_sumexp);
        e = hipGetLastError();
    }
    return e;
}

// one lane per ray column: attention_ij = exp(l_ij - max_i) / sumexp_i ; score_j = sum_i attention_ij (row order)
__global__ void __launch_bounds__(64) k6_colsum(float* __restrict__ logits, int M, int64_t N, const float* __restrict__ row_max,
                                                const float* __restrict__ row_sumexp, int write_attention,
                                                float* __restrict__ score) {
    extern __shared__ float s_stats[];   // [2][M]
    for (int i = threadIdx.x; i < M; i += 64) { s_stats[i] = row_max[i]; s_stats[M + i] = row_sumexp[i]; }
    __syncthreads();
    int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= N) return;
    float acc = 0.0f;
#pragma unroll 8
    for (int i = 0; i < M; ++i) {
        float a = expf(logits[(int64_t)i * N + j] - s_stats[i]) / s_stats[M + i];
        if (write_attention) logits[(int64_t)i * N + j] = a;
        acc += a;
    }
    score[j] = acc;
}

hipError_t launch_attn_colsum(float* logits, int M, int64_t N, const float* row_max, const float* row_sumexp,
                              int write_attention, float* score, hipStream_t s) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k6_colsum, dim3((unsigned)((N + 63) / 64)), dim3(64), 2 * (size_t)M * sizeof(float), s, logits, M, N,
                       row_max, row_sumexp, write_attention, score);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K7
// torch.topk(score, k): single-workgroup radix select on order-preserving keys, ordered gather, bitonic sort.
__device__ inline uint32_t order_key(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // larger float <-> larger key ; NaN sorts above +inf like torch
}

constexpr int TK_THREADS = 1024;

// k-th largest key (k >= 1) of n values; all threads of the workgroup call it; result broadcast.  hist: 256 + 2 ints of LDS
__device__ uint32_t wg_kth_largest_key(const float* __restrict__ v, int64_t n, int64_t k, int* hist) {
    uint32_t prefix = 0, mask = 0;
    int64_t remaining = k;
    for (int pass = 3; pass >= 0; --pass) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            uint32_t key = order_key(v[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int64_t rem = remaining;
            int b = 255;
            for (; b > 0; --b) {
                if (hist[b] >= rem) break;
                rem -= hist[b];
            }
            hist[256] = b;
            hist[257] = (int)rem;
        }
        __syncthreads();
        int b = hist[256];
        remaining = hist[257];
        prefix |= ((uint32_t)b) << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    return prefix;
}

__global__ void __launch_bounds__(TK_THREADS) k7_topk(const float* __restrict__ score, int64_t N, int k, int64_t* __restrict__ idx,
                                                      float* __restrict__ val) {
    __shared__ int hist[258];
    __shared__ int scan[TK_THREADS];
    __shared__ float s_val[1024];
    __shared__ int s_idx[1024];
    __shared__ int base_gt, base_eq;
    const int tid = threadIdx.x;
    const uint32_t T = wg_kth_largest_key(score, N, k, hist);
    // count strictly greater to know how many ties at T we may take (lowest indices first)
    if (tid == 0) { base_gt = 0; base_eq = 0; }
    for (int i = tid; i < 1024; i += TK_THREADS) { s_val[i] = -INFINITY; s_idx[i] = 0x7fffffff; }
    __syncthreads();
    // ordered sweep: chunk by chunk so equal keys are taken in index order
    int n_gt_total = 0;
    {
        int c = 0;
        for (int64_t i = tid; i < N; i += TK_THREADS) c += (order_key(score[i]) > T) ? 1 : 0;
        scan[tid] = c;
        __syncthreads();
        for (int off = TK_THREADS / 2; off >= 1; off >>= 1) {
            if (tid < off) scan[tid] += scan[tid + off];
            __syncthreads();
        }
        n_gt_total = scan[0];
        __syncthreads();
    }
    const int need_eq = k - n_gt_total;
    for (int64_t c0 = 0; c0 < N; c0 += TK_THREADS) {
        int64_t i = c0 + tid;
        uint32_t key = (i < N) ? order_key(score[i]) : 0u;
        bool gt = (i < N) && key > T, eq = (i < N) && key == T;
        // exclusive scans of both flags (packed: eq in the high half)
        int packed = (gt ? 1 : 0) | ((eq ? 1 : 0) << 16);
        scan[tid] = packed;
        __syncthreads();
        for (int off = 1; off < TK_THREADS; off <<= 1) {
            int v = (tid >= off) ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        int incl = scan[tid];
        int pos_gt = base_gt + (incl & 0xffff) - (gt ? 1 : 0);
        int pos_eq = base_eq + (incl >> 16) - (eq ? 1 : 0);
        if (gt) { s_val[pos_gt] = score[i]; s_idx[pos_gt] = (int)i; }
        if (eq && pos_eq < need_eq) { s_val[n_gt_total + pos_eq] = score[i]; s_idx[n_gt_total + pos_eq] = (int)i; }
        __syncthreads();
        if (tid == TK_THREADS - 1) { base_gt += incl & 0xffff; base_eq += incl >> 16; }
        __syncthreads();
    }
    // bitonic sort of 1024 slots: descending value, ascending index on ties (padding = -inf, idx max)
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            int i = tid;
            int j = i ^ stride;
            if (j > i) {
                bool up = ((i & size) == 0);
                float vi = s_val[i], vj = s_val[j];
                int ii = s_idx[i], ij = s_idx[j];
                uint32_t ki = order_key(vi), kj = order_key(vj);
                bool i_first = (ki > kj) || (ki == kj && ii < ij);   // i should precede j in the final order
                if (i_first != up) { s_val[i] = vj; s_val[j] = vi; s_idx[i] = ij; s_idx[j] = ii; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += TK_THREADS) { idx[i] = (int64_t)s_idx[i]; val[i] = s_val[i]; }
}

size_t topk_workspace_bytes(int64_t N, int k) { (void)N; (void)k; return 256; }

hipError_t launch_topk(const float* score, int64_t N, int k, int64_t* idx, float* val, void* ws, size_t ws_bytes, hipStream_t s) {
    (void)ws; (void)ws_bytes;
    if (k < 1 || k > 1024 || k > N || N >= 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k7_topk, dim3(1), dim3(TK_THREADS), 0, s, score, N, k, idx, val);
    return hipGetLastError();
}
