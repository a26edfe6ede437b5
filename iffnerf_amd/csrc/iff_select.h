// iff_select.h -- single-workgroup radix selection on order-preserving float keys (shared by k7_topk and the sampler's
// torch.quantile).  4 passes of 8 bits; the 256-bin scan of every pass is done in parallel by the first 256 threads
// (wave shuffles + 4 wave totals), not by one lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ inline uint32_t iff_order_key(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // larger float <-> larger key; NaN (positive) sorts above +inf
}
__device__ inline float iff_order_key_inv(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Key of the element of rank `rank1` (1-based) among n values, counted from the largest (LARGEST) or the smallest.
// All threads of the workgroup (blockDim.x >= 256, multiple of 64) must call it; `hist` needs 264 ints of LDS.
template <bool LARGEST>
__device__ inline uint32_t iff_wg_select_key(const float* __restrict__ v, int64_t n, int64_t rank1, int* hist) {
    uint32_t prefix = 0, mask = 0;
    int64_t remaining = rank1;
    const int tid = threadIdx.x;
    for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (int64_t i = tid; i < n; i += blockDim.x) {
            uint32_t key = iff_order_key(v[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        // bins in selection order: position p <-> bin (LARGEST ? 255 - p : p); inclusive scan over p
        int cnt = 0, incl = 0;
        if (tid < 256) {
            cnt = hist[LARGEST ? 255 - tid : tid];
            incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                int up = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += up;
            }
            if ((tid & 63) == 63) hist[256 + (tid >> 6)] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            int base = 0;
            for (int w = 0; w < (tid >> 6); ++w) base += hist[256 + w];
            incl += base;
            int excl = incl - cnt;
            if ((int64_t)incl >= remaining && (int64_t)excl < remaining) {
                hist[260] = LARGEST ? 255 - tid : tid;
                hist[261] = (int)(remaining - excl);
            }
        }
        __syncthreads();
        prefix |= ((uint32_t)hist[260]) << shift;
        remaining = hist[261];
        mask |= 255u << shift;
        __syncthreads();
    }
    return prefix;
}
