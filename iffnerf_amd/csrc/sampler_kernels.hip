// sampler_kernels.hip -- stage A surface sampler as ONE persistent launch (no host synchronisation).
//
// Reference: pose_estimation/sampling.py:509-532 iterative_surface_sampling_process
//   seeds    :78-116  uniform points inside occupied mask voxels, alpha = compute_alpha(seed)
//   epoch    :143-213 thresh = quantile(alpha, 0.6); while some samples are still "invalid": every invalid sample gets
//            m = (5 P) // n_invalid jittered candidates (:35-67: uniform direction, |N(0,rho)| radius); those with
//            alpha > thresh pass; one passing candidate per sample is picked uniformly and replaces the sample.
// The reference does this with argsort/argwhere/scatter_reduce and two host syncs per iteration.  Here every candidate is
// an independent (sample, j) work item with its own counter-based random stream (Philox4x32-10 keyed by
// seed / epoch / iteration / sample / j), a passing candidate competes with a random 32-bit priority through a 64-bit
// atomicMax (uniform pick, order independent => bitwise reproducible), and iterations are separated by an in-kernel
// grid barrier (agent-scope release/acquire, bounded spins).  The random streams necessarily differ from torch's CPU
// generator, so parity for this stage is distributional (tests/test_sampler.py), as SURVEY.md section 7.4 #2 records.
#include "iff_device.h"
#include "iff_launch.h"

// ------------------------------------------------------------------------------------------------ Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
__device__ inline U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        U4 n;
        n.x = hi1 ^ ctr.y ^ k0; n.y = lo1; n.z = hi0 ^ ctr.w ^ k1; n.w = lo0;
        ctr = n;
        k0 += W0; k1 += W1;
    }
    return ctr;
}
__device__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1), 24 bits like torch.rand

// ------------------------------------------------------------------------------------------------ workspace layout
constexpr int SAMPLER_MAX_EPOCHS = 64;
constexpr int SAMPLER_MAX_ITERS = 4096;
struct SamplerWs {
    unsigned barrier_count;     // monotonic arrivals
    unsigned abort_flag;
    float thresh[SAMPLER_MAX_EPOCHS];   // one slot per epoch: written once, never reused inside a launch
    // followed by: left[n_epochs*max_iterations] (int; samples still invalid after each iteration, one slot each so
    // no counter is ever reset while another workgroup may still read it), winners[P] (u64), list_a[P], list_b[P] (int),
    // cand_pos[5P][3], cand_alpha[5P]
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct SamplerLayout { size_t left, winners, list_a, list_b, cand_pos, cand_alpha, total; };
__host__ __device__ inline SamplerLayout sampler_layout(int64_t P) {
    SamplerLayout L;
    size_t p = align_up(sizeof(SamplerWs), 256);
    L.left = p; p = align_up(p + (size_t)SAMPLER_MAX_EPOCHS * SAMPLER_MAX_ITERS / 16 * 4, 256);   // 16 Ki counters
    L.winners = p; p = align_up(p + (size_t)P * 8, 256);
    L.list_a = p; p = align_up(p + (size_t)P * 4, 256);
    L.list_b = p; p = align_up(p + (size_t)P * 4, 256);
    L.cand_pos = p; p = align_up(p + (size_t)5 * P * 3 * 4, 256);
    L.cand_alpha = p; p = align_up(p + (size_t)5 * P * 4, 256);
    L.total = p;
    return L;
}
size_t sampler_workspace_bytes(int64_t P) { return sampler_layout(P).total; }

// ------------------------------------------------------------------------------------------------ grid barrier
// Placement-independent (cdna guide, Guideline 16): every wave drains its stores, workgroup barrier, lane 0 releases at
// agent scope and arrives on a monotonic counter, polls it relaxed, acquires at agent scope, workgroup barrier.  Spins are
// bounded: on timeout the abort flag is raised and every workgroup leaves at its next check.
__device__ inline bool grid_sync(SamplerWs* ws, unsigned& generation) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        generation += 1;
        const unsigned target = generation * gridDim.x;
        __hip_atomic_fetch_add(&ws->barrier_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&ws->barrier_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(&ws->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if (wall_clock64() - t0 > 200000000LL) {   // 2 s at 100 MHz
                __hip_atomic_store(&ws->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return __hip_atomic_load(&ws->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// alpha = compute_alpha(p, length=1) by the 4 lanes that share the point (all 4 return it)
__device__ inline float alpha4(const FieldDev& f, const float p[3], int sub, bool live) {
    bool valid = live;
    if (valid && f.mask) valid = mask_value(f, p) > 0.0f;
    float part = 0.0f;
    if (valid) {
        float xn[3];
        field_normalize(f, p, xn);
        part = density_partial(f, xn, sub);
    }
    float feat = sum4(part);
    float sigma = valid ? feature2density(f, feat) : 0.0f;
    return 1.0f - expf(-sigma * 1.0f);
}

// order-preserving key and single-workgroup k-th smallest (radix select), used for torch.quantile
__device__ inline uint32_t okey(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float okey_inv(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
__device__ float wg_kth_smallest(const float* v, int n, int k /*0-based*/, int* hist) {
    uint32_t prefix = 0, mask = 0;
    int remaining = k + 1;
    for (int pass = 3; pass >= 0; --pass) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            uint32_t key = okey(v[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int rem = remaining, b = 0;
            for (; b < 255; ++b) {
                if (hist[b] >= rem) break;
                rem -= hist[b];
            }
            hist[256] = b; hist[257] = rem;
        }
        __syncthreads();
        prefix |= ((uint32_t)hist[256]) << shift;
        remaining = hist[257];
        mask |= 255u << shift;
        __syncthreads();
    }
    return okey_inv(prefix);
}

struct SamplerArgs {
    int64_t P;
    int n_epochs, max_iterations;
    uint32_t seed_lo, seed_hi;
    float rho;
    float* samples;   // [P,3]
    float* alpha;     // [P]
    int* stats;       // [n_epochs,4]
    unsigned char* ws;
    const int* occ_list;   // occupied mask voxel ids (z*H*W + y*W + x), ascending
    int n_occ;
};

__global__ void __launch_bounds__(256) k_surface_sample(FieldDev f, SamplerArgs a) {
    __shared__ int hist[258];
    SamplerWs* ws = (SamplerWs*)a.ws;
    const SamplerLayout L = sampler_layout(a.P);
    int* left = (int*)(a.ws + L.left);
    unsigned long long* winners = (unsigned long long*)(a.ws + L.winners);
    int* list_cur = (int*)(a.ws + L.list_a);
    int* list_nxt = (int*)(a.ws + L.list_b);
    float* cand_pos = (float*)(a.ws + L.cand_pos);
    float* cand_alpha = (float*)(a.ws + L.cand_alpha);
    const int P = (int)a.P;
    const int64_t gtid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t gthreads = (int64_t)gridDim.x * blockDim.x;
    unsigned generation = 0;

    // ---------------- seeds (sampling.py:78-116,131-140)
    {
        const int W = f.mask_dims[2], H = f.mask_dims[1], D = f.mask_dims[0];
        const int64_t nt = (int64_t)P * 4;
        for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
            bool live = t < nt;
            int i = live ? (int)(t >> 2) : 0;
            int sub = (int)(t & 3);
            U4 r = philox4x32_10(U4{(uint32_t)i, 0u, 0u, 0x5eedu}, a.seed_lo, a.seed_hi);
            float p[3];
            if (f.mask && a.n_occ > 0) {
                int pick = (int)(((unsigned long long)r.x * (unsigned long long)a.n_occ) >> 32);
                int v = a.occ_list[pick];
                int x = v % W, y = (v / W) % H, z = v / (W * H);
                float s[3] = {(float)x + u01(r.y), (float)y + u01(r.z), (float)z + u01(r.w)};
                float dims[3] = {(float)W - 1.0f, (float)H - 1.0f, (float)D - 1.0f};
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = ((f.mask_hi[c] - f.mask_lo[c]) * s[c]) / dims[c] + f.mask_lo[c];
            } else {
                float u[3] = {u01(r.y), u01(r.z), u01(r.w)};   // sampling.py:119-128
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = u[c] * (f.aabb_hi[c] - f.aabb_lo[c]) + f.aabb_lo[c];
            }
            float al = alpha4(f, p, sub, live);
            if (live && sub == 0) {
                a.samples[3 * i] = p[0]; a.samples[3 * i + 1] = p[1]; a.samples[3 * i + 2] = p[2];
                a.alpha[i] = al;
                winners[i] = 0ull;
                list_cur[i] = i;
            }
        }
    }
#define IFF_SYNC_OR_ABORT()                                   \
    if (!grid_sync(ws, generation)) {                        \
        if (gtid == 0) a.stats[3] = -1; /* timed out */      \
        return;                                              \
    }
    IFF_SYNC_OR_ABORT();

    for (int epoch = 0; epoch < a.n_epochs; ++epoch) {
        // ---------------- threshold = torch.quantile(alpha, 0.6) (linear interpolation), workgroup 0
        if (blockIdx.x == 0) {
            float pos = 0.6f * (float)(P - 1);
            int lo = (int)floorf(pos);
            int hi = min(lo + 1, P - 1);
            float frac = pos - (float)lo;
            float vlo = wg_kth_smallest(a.alpha, P, lo, hist);
            float vhi = wg_kth_smallest(a.alpha, P, hi, hist);
            if (threadIdx.x == 0) {
                // torch.lerp: lo + w (hi - lo) for w < 0.5, hi - (hi - lo)(1 - w) otherwise
                ws->thresh[epoch] = (frac < 0.5f) ? (vlo + (vhi - vlo) * frac) : (vhi - (vhi - vlo) * (1.0f - frac));
            }
        }
        for (int64_t t = gtid; t < P; t += gthreads) list_cur[t] = (int)t;
        IFF_SYNC_OR_ABORT();
        const float thresh = ws->thresh[epoch];
        int K = P, it = 0, m_last = 0;
        while (K != 0 && it < a.max_iterations) {
            // ---------------- candidates: m per invalid sample, 4 lanes each
            const int m = (5 * P) / K;
            m_last = m;
            const int64_t n_slots = (int64_t)K * m;
            const int64_t nt = n_slots * 4;
            for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
                bool live = t < nt;
                int64_t slot = live ? (t >> 2) : 0;
                int sub = (int)(t & 3);
                int li = (int)(slot / m), j = (int)(slot - (int64_t)li * m);
                int i = list_cur[li];
                U4 c0 = U4{(uint32_t)i, (uint32_t)j, (uint32_t)(epoch * 4096 + it), 0xA5u};
                U4 r0 = philox4x32_10(c0, a.seed_lo, a.seed_hi);
                c0.w = 0xA6u;
                U4 r1 = philox4x32_10(c0, a.seed_lo, a.seed_hi);
                // sampling.py:38-66: theta = 2 pi u, phi = arccos(1 - 2u), radius |N(0, rho)|
                float theta = 6.283185307179586f * u01(r0.x);
                float phi = acosf(1.0f - 2.0f * u01(r0.y));
                float sp = sinf(phi);
                float dir[3] = {sp * cosf(theta), sp * sinf(theta), cosf(phi)};
                float u1 = 1.0f - u01(r0.z);                       // (0,1]
                float g = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u01(r0.w));
                float dist = fabsf(g * a.rho);
                float p[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = a.samples[3 * i + c] + dir[c] * dist;
                float al = alpha4(f, p, sub, live);
                if (live && sub == 0) {
                    cand_pos[3 * slot] = p[0]; cand_pos[3 * slot + 1] = p[1]; cand_pos[3 * slot + 2] = p[2];
                    cand_alpha[slot] = al;
                    if (al > thresh) {
                        unsigned long long key = ((unsigned long long)r1.x << 32) | (unsigned long long)(unsigned)(j + 1);
                        atomicMax(&winners[i], key);
                    }
                }
            }
            IFF_SYNC_OR_ABORT();
            // ---------------- resolve: accepted samples move, the rest queue for the next iteration
            for (int64_t li = gtid; li < K; li += gthreads) {
                int i = list_cur[li];
                unsigned long long wv = atomicExch(&winners[i], 0ull);   // memory-side read-and-clear
                if (wv != 0ull) {
                    int j = (int)(wv & 0xffffffffull) - 1;
                    int64_t slot = (int64_t)li * m + j;
                    a.samples[3 * i] = cand_pos[3 * slot]; a.samples[3 * i + 1] = cand_pos[3 * slot + 1];
                    a.samples[3 * i + 2] = cand_pos[3 * slot + 2];
                    a.alpha[i] = cand_alpha[slot];
                } else {
                    int pos = atomicAdd(&left[epoch * a.max_iterations + it], 1);
                    list_nxt[pos] = i;
                }
            }
            IFF_SYNC_OR_ABORT();
            K = __hip_atomic_load(&left[epoch * a.max_iterations + it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            it += 1;
            int* tmp = list_cur; list_cur = list_nxt; list_nxt = tmp;
        }
        if (gtid == 0) {
            a.stats[epoch * 4 + 0] = it;
            a.stats[epoch * 4 + 1] = K;
            a.stats[epoch * 4 + 2] = __float_as_int(thresh);
            a.stats[epoch * 4 + 3] = m_last;
        }
    }
}

hipError_t launch_surface_sample_occ(const FieldDev& f, const int* occ_list, int n_occ, int64_t P, int n_epochs,
                                     int max_iterations, uint64_t seed, float rho, float* samples, float* alpha, int* stats,
                                     void* ws, size_t ws_bytes, int n_cus, hipStream_t s) {
    if (P < 1 || P > (1 << 24) || n_epochs < 0 || n_epochs > SAMPLER_MAX_EPOCHS || max_iterations < 0 ||
        (int64_t)n_epochs * max_iterations > (int64_t)SAMPLER_MAX_EPOCHS * SAMPLER_MAX_ITERS / 16)
        return hipErrorInvalidValue;
    if (ws_bytes < sampler_workspace_bytes(P)) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ws, 0, sampler_layout(P).winners, s);   // header + per-iteration counters
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(stats, 0, sizeof(int) * 4 * (size_t)n_epochs, s);
    if (e != hipSuccess) return e;
    SamplerArgs a;
    a.P = P; a.n_epochs = n_epochs; a.max_iterations = max_iterations;
    a.seed_lo = (uint32_t)(seed & 0xffffffffu); a.seed_hi = (uint32_t)(seed >> 32);
    a.rho = rho; a.samples = samples; a.alpha = alpha; a.stats = stats; a.ws = (unsigned char*)ws;
    a.occ_list = occ_list; a.n_occ = n_occ;
    // one workgroup per CU at most, so the grid is co-resident and the in-kernel barrier cannot deadlock
    int64_t want = (5 * P * 4 + 255) / 256;
    int grid = (int)(want < 1 ? 1 : (want > n_cus ? n_cus : want));
    hipLaunchKernelGGL(k_surface_sample, dim3(grid), dim3(256), 0, s, f, a);
    return hipGetLastError();
}
