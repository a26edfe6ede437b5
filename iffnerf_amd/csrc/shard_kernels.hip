// shard_kernels.hip -- the merge steps of the ray-sharded path (iffnerf_amd/distributed.py states the
// exchange): what every rank does with the messages an RCCL all_gather has just delivered, one launch each, so that a captured
// segment holds them and nothing runs on the host between two collectives.
//
// The reference is single-process: these kernels reproduce, over G column shards, exactly what its one softmax over the ray axis
// (pose_estimation/multihead_attention.py:11, dim=-1) and its one torch.topk (identification_module.py:207) compute:
//   k_merge_row_stats    per-row (max, sum exp) of every rank -> the statistics over all columns, ranks added in rank order
//   k_pack_candidates    a rank's local top-k -> its message [Q, k, 8] = (score, global ray index bits, origin, direction)
//   k_merge_candidates   every rank's message -> the global top-k with torch.topk's order (value descending, lower ray index
//                        first): the lists arrive sorted, so an element's place is its own position plus, for every other
//                        rank's list, the number of entries that precede it (binary search) -- no sort
#include "iff_device.h"
#include "iff_launch.h"

namespace {

__global__ void k_merge_row_stats(const float* __restrict__ stats, int G, int64_t R, float* __restrict__ gmax, float* __restrict__ gsum) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        float m = -INFINITY;
        for (int g = 0; g < G; ++g) m = fmaxf(m, stats[((int64_t)g * R + r) * 2]);
        float s = 0.0f;
        for (int g = 0; g < G; ++g) {                      // fixed rank order: reproducible
            const float mg = stats[((int64_t)g * R + r) * 2], sg = stats[((int64_t)g * R + r) * 2 + 1];
            s = s + sg * expf(mg - m);
        }
        gmax[r] = m; gsum[r] = s;
    }
}

// slot (q, j): j < kl -> the rank's j-th best ray of query q; the others are padding that sorts last (-inf, index 2^31 - 1)
__global__ void k_pack_candidates(const int64_t* __restrict__ idx, const float* __restrict__ val, const float* __restrict__ ori,
                                  const float* __restrict__ dirs, int64_t ray_stride, int Q, int kl, int k, int64_t first_ray,
                                  float* __restrict__ msg) {
    const int n = Q * k;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int q = t / k, j = t - q * k;
        float* m = msg + (int64_t)t * 8;
        if (j < kl) {
            const int64_t i = idx[(int64_t)q * kl + j];
            const float* o = ori + q * ray_stride + 3 * i;
            const float* d = dirs + q * ray_stride + 3 * i;
            m[0] = val[(int64_t)q * kl + j];
            m[1] = __int_as_float((int)(i + first_ray));
            m[2] = o[0]; m[3] = o[1]; m[4] = o[2]; m[5] = d[0]; m[6] = d[1]; m[7] = d[2];
        } else {
            m[0] = -INFINITY;
            m[1] = __int_as_float(0x7fffffff);
            m[2] = m[3] = m[4] = m[5] = m[6] = m[7] = 0.0f;
        }
    }
}

// does (va, ia) come before (vb, ib) in torch.topk's order?
__device__ __forceinline__ bool before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// one workgroup per query: cand [G][Qt][k][8], this launch's queries q0 .. q0 + Q - 1 of the Qt in the message
__global__ void __launch_bounds__(256) k_merge_candidates(const float* __restrict__ cand, int G, int Qt, int q0, int k, float* __restrict__ val,
                                                          int64_t* __restrict__ idx, float* __restrict__ ori, float* __restrict__ dir) {
    extern __shared__ float s_key[];                       // [G][k] values, then [G][k] index bits
    float* const s_val = s_key;
    int* const s_idx = reinterpret_cast<int*>(s_key + G * k);
    const int q = blockIdx.x;
    const int n = G * k;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int g = t / k, j = t - g * k;
        const float* m = cand + (((int64_t)g * Qt + q0 + q) * k + j) * 8;
        s_val[t] = m[0];
        s_idx[t] = __float_as_int(m[1]);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int g = t / k, j = t - g * k;
        const float v = s_val[t];
        const int i = s_idx[t];
        int rank = j;                                      // its own list is sorted: j entries precede it there
        for (int h = 0; h < G; ++h) {
            if (h == g) continue;
            // entries of list h that come before (v, i); equal keys (padding only: ray indices are unique) go by rank order
            int lo = 0, hi = k;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float vm = s_val[h * k + mid];
                const int im = s_idx[h * k + mid];
                const bool pre = before(vm, im, v, i) || (h < g && vm == v && im == i);
                if (pre) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            const float* m = cand + (((int64_t)g * Qt + q0 + q) * k + j) * 8;
            const int64_t out = (int64_t)q * k + rank;
            val[out] = v;
            idx[out] = (int64_t)i;
            ori[3 * out] = m[2]; ori[3 * out + 1] = m[3]; ori[3 * out + 2] = m[4];
            dir[3 * out] = m[5]; dir[3 * out + 1] = m[6]; dir[3 * out + 2] = m[7];
        }
    }
}

}  // namespace

hipError_t launch_merge_row_stats(const float* stats_all, int G, int64_t R, float* gmax, float* gsum, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    const int64_t blocks = (R + 255) / 256;
    hipLaunchKernelGGL(k_merge_row_stats, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, stats_all, G, R, gmax, gsum);
    return hipGetLastError();
}

hipError_t launch_pack_candidates(const int64_t* idx, const float* val, const float* ori, const float* dirs, int64_t ray_stride, int Q, int kl,
                                  int k, int64_t first_ray, float* msg, hipStream_t s) {
    if (Q <= 0 || k <= 0) return hipSuccess;
    const int n = Q * k;
    hipLaunchKernelGGL(k_pack_candidates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx, val, ori, dirs, ray_stride, Q, kl, k, first_ray, msg);
    return hipGetLastError();
}

hipError_t launch_merge_candidates(const float* cand_all, int G, int Qt, int q0, int Q, int k, float* val, int64_t* idx, float* ori, float* dir,
                                   hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    const size_t lds = (size_t)G * k * 8;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_merge_candidates, dim3((unsigned)Q), dim3(256), lds, s, cand_all, G, Qt, q0, k, val, idx, ori, dir);
    return hipGetLastError();
}
