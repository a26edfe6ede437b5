// sampler_kernels.hip -- stage A surface sampler as ONE persistent launch (no host synchronisation).
//
// Reference: pose_estimation/sampling.py:509-532 iterative_surface_sampling_process
//   seeds    :78-116  uniform points inside occupied mask voxels, alpha = compute_alpha(seed)
//   epoch    :143-213 thresh = quantile(alpha, 0.6); while some samples are still "invalid": every invalid sample gets
//            m = (5 P) // n_invalid jittered candidates (:35-67: uniform direction, |N(0,rho)| radius); those with
//            alpha > thresh pass; one passing candidate per sample is picked uniformly and replaces the sample.
// The reference does this with argsort/argwhere/scatter_reduce and two host syncs per iteration.  Here
//   * every candidate is an independent (sample, j) work item with its own counter-based random stream (Philox4x32-10
//     keyed by seed / epoch / iteration / sample / j) served by 4 lanes (one texel quarter each);
//   * a passing candidate competes with a random 32-bit priority through a 64-bit atomicMax on its sample's slot: a
//     uniform pick that is independent of execution order, so runs are bitwise reproducible;
//   * the slot keeps (iteration, j) of the winner, so nothing but that one word crosses workgroups during an epoch:
//     every workgroup rebuilds the (sorted) list of still-invalid samples in its own LDS from the slots, and the
//     accepted positions are re-derived from the random stream when the epoch ends;
//   * iterations are separated by a grid barrier that needs NO cache maintenance (only memory-side atomics were
//     exchanged); one release/acquire barrier per epoch publishes the moved samples.  All spins are bounded.
// torch's CPU generator cannot be reproduced on the device, so parity for this stage is distributional
// (tests/test_hip_sampler.py), as SURVEY.md section 7.4 #2 records.
#include "iff_device.h"
#include "iff_launch.h"
#include "iff_select.h"

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
constexpr int SAMPLER_MAX_ITERS = 4095;
constexpr int SAMPLER_MAX_POINTS = 32768;          // the invalid list lives in LDS (4 B per sample)
constexpr int SAMPLER_CACHE_POINTS = 4096;        // up to here alpha + positions are cached in LDS too (20 B per sample)
struct SamplerWs {
    unsigned barrier_count;     // monotonic arrivals (persistent form)
    unsigned abort_flag;
    float thresh;               // stepped form: this epoch's threshold, written by the epoch's first iteration launch
    int done_epoch;             // stepped form: epoch + 1 once an iteration launch found no invalid sample left in `epoch`
    int K_list;                 // stepped form, large P: entries of list[] (k_ss_list -> k_ss_cand)
    unsigned pad[59];
    // followed by: winners[2][P] (u64, double-buffered by epoch parity)
};
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// header | winners[2][P] u64 | list[P] i32 (the stepped form's invalid list of a run when P is large: k_ss_list)
size_t sampler_workspace_bytes(int64_t P) { return align_up(sizeof(SamplerWs), 256) + align_up((size_t)P * 16, 256) + align_up((size_t)P * 4, 256); }

// ------------------------------------------------------------------------------------------------ grid barriers
// Placement-independent (cdna guide, Guideline 16).  FENCED: every wave drains its stores, workgroup barrier, lane 0
// releases at agent scope, arrives on a monotonic counter, polls it relaxed, acquires at agent scope, workgroup barrier
// -- plain loads after it see every plain store before it.  Unfenced: same arrival protocol without the cache
// maintenance; correct when the only cross-workgroup traffic since the last fenced barrier went through atomics.
// Spins are bounded: on timeout the abort flag is raised and every workgroup leaves at its next check.
template <bool FENCED>
__device__ inline bool grid_sync(SamplerWs* ws, unsigned& generation, unsigned n_wg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores and no-return atomics of this wave are complete
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCED) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        generation += 1;
        const unsigned target = generation * n_wg;      // workgroups of THIS query's group
        __hip_atomic_fetch_add(&ws->barrier_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // One agent-scope load per poll and ~0.2 us between polls: these loads are served at the memory side, all from one
        // address, so a tight spin from every workgroup saturates that memory channel and slows every other kernel on
        // the chip (measured: queries on other streams ran 1.6x slower next to one spinning sampler).  A workgroup that
        // times out raises the abort flag AND the counter's top bit, which releases every poller at once.
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&ws->barrier_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 200000000LL) {   // 2 s at 100 MHz
                __hip_atomic_store(&ws->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_or(&ws->barrier_count, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (FENCED) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    return __hip_atomic_load(&ws->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// alpha = compute_alpha(p, length=1) by the 4 lanes that share the point (all 4 return it).  The occupancy test and the
// density taps are issued together (both are safe for any coordinate); the result is selected afterwards.
// alpha by ONE lane (density_full: taps once for all 16 channels, the 4-lane form's summation order -- same bits).
// Used when n_density == 16; LPC = lanes per candidate is then 1 and a run needs a quarter of the workgroups.
__device__ inline float alpha1(const FieldDev& f, const float p[3], bool live) {
    float xn[3];
    field_normalize(f, p, xn);
    const bool occ = f.mask ? mask_occupied(f, p, xn) : true;
    float part = density_full(f, xn);
    bool valid = live && occ;
    float sigma = valid ? feature2density(f, valid ? part : 0.0f) : 0.0f;
    return 1.0f - expf(-sigma * 1.0f);
}

__device__ inline float alpha4(const FieldDev& f, const float p[3], int sub, bool live) {
    float xn[3];
    field_normalize(f, p, xn);
    const bool occ = f.mask ? mask_occupied(f, p, xn) : true;
    float part = density_partial(f, xn, sub);
    bool valid = live && occ;
    float feat = sum4(valid ? part : 0.0f);
    float sigma = valid ? feature2density(f, feat) : 0.0f;
    return 1.0f - expf(-sigma * 1.0f);
}

// One candidate per lane for everything that is per candidate (random stream, position, occupancy test), and the density taps of
// the four candidates of a quad gathered by the quad TOGETHER, candidate after candidate, each lane one 16-B texel quarter (the
// positions go round the quad as DPP moves): the coalesced loads of the four-lane form (a quarter of the texture-path cycles of
// the one-lane form) with a quarter of its waves and half its vector instructions.  Same bits as alpha4 / alpha1.
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {          // lane (l & ~3) + Q's value: quad_perm [Q,Q,Q,Q]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), Q * 0x55, 0xF, 0xF, true));
}
template <int Q>
__device__ __forceinline__ void alpha_quad_step(const FieldDev& f, const float xn[3], bool valid, int sub, float& feat) {
    const float xq[3] = {quad_bcast<Q>(xn[0]), quad_bcast<Q>(xn[1]), quad_bcast<Q>(xn[2])};
    const bool vq = quad_bcast<Q>(valid ? 1.0f : 0.0f) != 0.0f;
    const float fq = sum4(vq ? density_partial(f, xq, sub) : 0.0f);
    if (sub == Q) feat = fq;
}
__device__ inline float alpha_quad(const FieldDev& f, const float p[3], bool live) {
    float xn[3];
    field_normalize(f, p, xn);
    const bool valid = live && (f.mask ? mask_occupied(f, p, xn) : true);
    const int sub = threadIdx.x & 3;
    float feat = 0.0f;
    alpha_quad_step<0>(f, xn, valid, sub, feat);
    alpha_quad_step<1>(f, xn, valid, sub, feat);
    alpha_quad_step<2>(f, xn, valid, sub, feat);
    alpha_quad_step<3>(f, xn, valid, sub, feat);
    const float sigma = valid ? feature2density(f, feat) : 0.0f;
    return 1.0f - expf(-sigma * 1.0f);
}

struct SamplerArgs {
    int64_t P;
    int n_epochs, max_iterations;
    uint32_t seed_lo, seed_hi;
    const unsigned long long* seed_dev;   // nullable: added to the by-value seed at kernel start (graph replays vary it)
    float rho;
    float* samples;   // [P,3]
    float* alpha;     // [P]
    int* stats;       // [n_epochs,4]
    unsigned char* ws;
    const int* occ_list;   // occupied mask voxel ids (z*H*W + y*W + x), ascending
    int n_occ;
    int wgs_per_query;     // the grid is B consecutive groups of this many workgroups, one independent sampler each
    size_t ws_stride;      // bytes between the queries' workspaces
    int lpc;               // lanes per candidate / point: 4 (one texel quarter each) or 1 (n_density == 16)
    int cache_lds;         // every workgroup keeps this epoch's alpha [P] and positions [P,3] in LDS (P <= SAMPLER_CACHE_POINTS)
    int quad;              // lpc == 1 with the density taps gathered per quad (alpha_quad): the stepped form's default
};

// jittered candidate j of sample i (sampling.py:38-66): theta = 2 pi u, phi = arccos(1 - 2u), radius |N(0, rho)|
__device__ inline void candidate_position(const SamplerArgs& a, const float base[3], int i, int j, int epoch, int it,
                                          float p[3], uint32_t& prio) {
    U4 c0 = U4{(uint32_t)i, (uint32_t)j, (uint32_t)(epoch * 4096 + it), 0xA5u};
    U4 r0 = philox4x32_10(c0, a.seed_lo, a.seed_hi);
    // the same distribution with cheaper arithmetic: cos(phi) = 1 - 2u is the z component itself, sin(phi) its
    // complement; sincospi/cospi need no range reduction; the four uniforms use the top 24 bits of each word, the
    // selection priority the four low bytes (independent bits of the same draw)
    float z = 1.0f - 2.0f * u01(r0.y);
    float sp = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float st, ct;
    sincospif(2.0f * u01(r0.x), &st, &ct);
    float dir[3] = {sp * ct, sp * st, z};
    float u1 = 1.0f - u01(r0.z);                       // (0,1]
    float g = sqrtf(-2.0f * __logf(u1)) * cospif(2.0f * u01(r0.w));
    float dist = fabsf(g * a.rho);
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = base[c] + dir[c] * dist;
    prio = (r0.x & 0xffu) | ((r0.y & 0xffu) << 8) | ((r0.z & 0xffu) << 16) | ((r0.w & 0xffu) << 24);
}

__global__ void __launch_bounds__(256) k_surface_sample(FieldDev f, SamplerArgs a) {
    extern __shared__ int s_list[];               // [P] still-invalid sample ids, ascending; then (cache_lds) alpha [P], pos [3P]
    // batch: query q = blockIdx.x / wgs_per_query owns its outputs, its workspace (barrier words, winners) and its seed
    const int q_id = (int)(blockIdx.x / (unsigned)a.wgs_per_query);
    const int wg_id = (int)(blockIdx.x - (unsigned)q_id * (unsigned)a.wgs_per_query);
    {
        unsigned long long sd = (((unsigned long long)a.seed_hi << 32) | a.seed_lo) + (unsigned long long)q_id * 0x9E3779B97F4A7C15ull;
        // agent-scope load: a scalar/L1-cached read can be stale when a graph node just before this one rewrote the word
        if (a.seed_dev) sd += __hip_atomic_load(a.seed_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.seed_lo = (uint32_t)(sd & 0xffffffffull); a.seed_hi = (uint32_t)(sd >> 32);
    }
    a.ws += (size_t)q_id * a.ws_stride;
    a.samples += (size_t)q_id * a.P * 3;
    a.alpha += (size_t)q_id * a.P;
    a.stats += (size_t)q_id * 4 * (a.n_epochs > 0 ? a.n_epochs : 1);
    __shared__ int hist[264];
    __shared__ int s_tot[4];
    SamplerWs* ws = (SamplerWs*)a.ws;
    unsigned long long* winners_base = (unsigned long long*)(a.ws + align_up(sizeof(SamplerWs), 256));
    const int P = (int)a.P;
    float* s_alpha = reinterpret_cast<float*>(s_list + P);
    float* s_pos = s_alpha + P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t gtid = wg_id * (int64_t)blockDim.x + tid;
    const int64_t gthreads = (int64_t)a.wgs_per_query * blockDim.x;
    unsigned generation = 0;
    const int lpc = a.lpc, lsh = (a.lpc == 4) ? 2 : 0;
#define IFF_SYNC_OR_ABORT(FENCED)                            \
    if (!grid_sync<FENCED>(ws, generation, (unsigned)a.wgs_per_query)) { \
        if (gtid == 0) a.stats[3] = -1; /* timed out */      \
        return;                                              \
    }

    for (int64_t t = gtid; t < 4 * (int64_t)a.n_epochs; t += gthreads) a.stats[t] = 0;
    // ---------------- seeds (sampling.py:78-116,131-140)
    {
        const int W = f.mask_dims[2], H = f.mask_dims[1], D = f.mask_dims[0];
        const int64_t nt = (int64_t)P * lpc;
        for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
            bool live = t < nt;
            int i = live ? (int)(t >> lsh) : 0;
            int sub = (int)(t & (lpc - 1));
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
            float al = (lpc == 1) ? alpha1(f, p, live) : alpha4(f, p, sub, live);
            if (live && sub == 0) {
                a.samples[3 * i] = p[0]; a.samples[3 * i + 1] = p[1]; a.samples[3 * i + 2] = p[2];
                a.alpha[i] = al;
            }
        }
    }
    IFF_SYNC_OR_ABORT(true);

    for (int epoch = 0; epoch < a.n_epochs; ++epoch) {
        unsigned long long* winners = winners_base + (size_t)(epoch & 1) * P;
        unsigned long long* winners_next = winners_base + (size_t)((epoch + 1) & 1) * P;
        // ---------------- threshold = torch.quantile(alpha, 0.6), linear interpolation; every workgroup computes it
        float thresh;
        const float* alpha_src = a.alpha;
        if (a.cache_lds) {
            // this epoch's alpha and positions (fixed until the apply step) once from global into LDS: the two radix
            // selects (8 passes over alpha) and every candidate's base position then stay on the CU
            for (int t = tid; t < P; t += 256) {
                s_alpha[t] = a.alpha[t];
                s_pos[3 * t] = a.samples[3 * t]; s_pos[3 * t + 1] = a.samples[3 * t + 1]; s_pos[3 * t + 2] = a.samples[3 * t + 2];
            }
            __syncthreads();
            alpha_src = s_alpha;
        }
        {
            float pos = 0.6f * (float)(P - 1);
            int lo = (int)floorf(pos);
            int hi = min(lo + 1, P - 1);
            float frac = pos - (float)lo;
            // one radix select for rank lo; rank lo + 1 is either the same value (duplicates) or the smallest larger one:
            // a single counting pass instead of a second four-pass select
            const uint32_t klo = iff_wg_select_key<false>(alpha_src, P, lo + 1, hist);
            int cnt_le = 0;
            uint32_t kmin = 0xffffffffu;
            for (int t = tid; t < P; t += 256) {
                uint32_t key = iff_order_key(alpha_src[t]);
                cnt_le += (key <= klo) ? 1 : 0;
                kmin = (key > klo && key < kmin) ? key : kmin;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                cnt_le += __shfl_xor(cnt_le, off, 64);
                uint32_t o = (uint32_t)__shfl_xor((int)kmin, off, 64);
                kmin = o < kmin ? o : kmin;
            }
            if (lane == 0) { hist[wave] = cnt_le; hist[4 + wave] = (int)kmin; }
            __syncthreads();
            cnt_le = hist[0] + hist[1] + hist[2] + hist[3];
            kmin = (uint32_t)hist[4];
#pragma unroll
            for (int w = 1; w < 4; ++w) kmin = (uint32_t)hist[4 + w] < kmin ? (uint32_t)hist[4 + w] : kmin;
            __syncthreads();
            const uint32_t khi = (hi == lo || cnt_le >= hi + 1) ? klo : kmin;
            float vlo = iff_order_key_inv(klo);
            float vhi = iff_order_key_inv(khi);
            // torch.lerp: lo + w (hi - lo) for w < 0.5, hi - (hi - lo)(1 - w) otherwise
            thresh = (frac < 0.5f) ? (vlo + (vhi - vlo) * frac) : (vhi - (vhi - vlo) * (1.0f - frac));
        }
        for (int t = tid; t < P; t += 256) s_list[t] = t;
        __syncthreads();
        int K = P, it = 0, m_last = 0;
        while (K != 0 && it < a.max_iterations) {
            // ---------------- candidates: m per invalid sample, 4 lanes each
            const int m = (5 * P) / K;
            m_last = m;
            const int64_t nt = (int64_t)K * m * lpc;
            for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
                bool live = t < nt;
                int64_t slot = live ? (t >> lsh) : 0;
                int sub = (int)(t & (lpc - 1));
                int li = (int)(slot / m), j = (int)(slot - (int64_t)li * m);
                int i = s_list[li];
                const float* ps = a.cache_lds ? s_pos : a.samples;
                float base[3] = {ps[3 * i], ps[3 * i + 1], ps[3 * i + 2]}, p[3];
                uint32_t prio;
                candidate_position(a, base, i, j, epoch, it, p, prio);
                float al = (lpc == 1) ? alpha1(f, p, live) : alpha4(f, p, sub, live);
                if (live && sub == 0 && al > thresh) {
                    unsigned long long key = ((unsigned long long)prio << 32) | ((unsigned long long)it << 20) |
                                             (unsigned long long)(unsigned)(j + 1);
                    atomicMax(&winners[i], key);
                }
            }
            IFF_SYNC_OR_ABORT(false);
            // ---------------- every workgroup drops the accepted samples from its own copy of the list (in place,
            // order kept): chunk of 256 entries at a time, ballot prefix inside a wave, 4 wave totals through LDS
            int kept = 0;
            for (int c0 = 0; c0 < K; c0 += 256) {
                int li = c0 + tid;
                int i = (li < K) ? s_list[li] : -1;
                bool stay = false;
                if (i >= 0) {
                    // a workgroup that is already one iteration ahead may have posted a key for iteration it+1 on a
                    // slot that was empty at the barrier: such keys do not count yet (skew is bounded by one barrier)
                    unsigned long long wv = __hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stay = (wv == 0ull) || ((int)((wv >> 20) & 0xfffull) > it);
                }
                unsigned long long bal = __ballot(stay);
                int before = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) s_tot[wave] = __popcll(bal);
                __syncthreads();                    // all reads of this chunk are done; totals visible
                int base = kept;
                for (int w = 0; w < wave; ++w) base += s_tot[w];
                if (stay) s_list[base + before] = i;
                kept += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
                __syncthreads();
            }
            K = kept;
            it += 1;
        }
        // ---------------- apply: the accepted samples move to their winning candidate (re-derived from the stream)
        {
            const int64_t nt = (int64_t)P * lpc;
            for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
                bool live = t < nt;
                int i = live ? (int)(t >> lsh) : 0;
                int sub = (int)(t & (lpc - 1));
                unsigned long long wv = live ? __hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                bool moved = wv != 0ull;
                int j = (int)(wv & 0xfffffull) - 1, wit = (int)((wv >> 20) & 0xfffull);
                const float* ps = a.cache_lds ? s_pos : a.samples;
                float base[3] = {ps[3 * i], ps[3 * i + 1], ps[3 * i + 2]}, p[3];
                uint32_t prio;
                candidate_position(a, base, i, moved ? j : 0, epoch, wit, p, prio);
                float al = (lpc == 1) ? alpha1(f, p, live && moved) : alpha4(f, p, sub, live && moved);
                if (live && sub == 0) {
                    if (moved) {
                        a.samples[3 * i] = p[0]; a.samples[3 * i + 1] = p[1]; a.samples[3 * i + 2] = p[2];
                        a.alpha[i] = al;
                    }
                    // the other parity's slots were last read one epoch ago: clear them for the next epoch
                    __hip_atomic_store(&winners_next[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (gtid == 0) {
            a.stats[epoch * 4 + 0] = it;
            a.stats[epoch * 4 + 1] = K;
            a.stats[epoch * 4 + 2] = __float_as_int(thresh);
            a.stats[epoch * 4 + 3] = m_last;
        }
        IFF_SYNC_OR_ABORT(true);
    }
}

// ------------------------------------------------------------------------------------------------ the stepped form
// The same sampler as a chain of SHORT launches -- seeds | per epoch: S iteration launches, a finisher, the apply step -- with the
// kernel boundary as the grid barrier.  The persistent launch above is parked at its barriers for 80 % of its life, and next to
// the throughput kernels of other steps it stays resident 0.8 ms per step (2-3 launches at any time under four steps in flight):
// each of its waves holds 128 VGPR, so a SIMD that hosts one cannot take the third wave of the fan march (3 x 168) or the
// fourth of the trunk (4 x 126) -- measured with the sampler free-running on side streams: march 0.53 -> 0.77 ms, trunk 0.39 ->
// 0.66 ms per step (scripts/sampler_interference.py).  A launch of this form holds its slots for the few microseconds it works.
// Same arithmetic, same random streams, same order-independent picks: the samples are those of the persistent form bit for bit
// (tests/test_hip_sampler.py).  No workgroup waits for another one: no co-residency requirement, no spins, no timeout.
//   k_ss_seed        seeds and their alpha (sampling.py:78-116,131-140); clears the statistics
//   k_ss_iter        iteration `it` of `epoch` for every run that still has invalid samples: every workgroup rebuilds the run's
//                    ascending list of invalid samples from the winner slots (it == 0: also the epoch's threshold, which workgroup
//                    0 of the run leaves in the workspace for the later launches), then its share of the 5 P candidates
//   k_ss_iter<true>  (the finisher) iterations S, S + 1, ... up to max_iterations for a run that has not converged after the S static launches
//                    (one workgroup per run looping with workgroup barriers; returns at once otherwise -- the reference's bound
//                    of 200 iterations is kept, 3-6 are observed on every bench model)
//   k_ss_apply       the accepted samples move to their winning candidate (sampling.py:205-213); per-epoch statistics
// static iteration launches per epoch: the first epoch moves every sample (5-6 iterations on the bench models), the later ones only
// the 40 % below the new quantile (3); a run that needs more continues in the finisher.  An idle launch costs 4.6 us of the chain
__host__ __device__ inline int ss_slots(int epoch) { return epoch == 0 ? 8 : 5; }
constexpr int SS_SMALL_P = 2560;         // up to here an iteration is ONE launch whose workgroups each rebuild the run's list (k_ss_iter)

struct SsRun {                           // one run's views, resolved from blockIdx
    int q_id, wg_id;
    SamplerWs* ws;
    unsigned long long* winners_base;
};
__device__ __forceinline__ SsRun ss_resolve(SamplerArgs& a, int wgs_per_run) {
    SsRun r;
    // Workgroups are dealt round-robin over the 8 XCDs (observed; speed only): give the workgroups that share an XCD consecutive
    // run-major ids, so that a run's candidates -- which touch the same few hundred KB of density texels in every iteration -- stay
    // with ONE XCD's 4-MB L2 instead of being spread over all eight (16 runs x 1.5 MB each then thrash every L2)
    unsigned vb = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    r.q_id = (int)(vb / (unsigned)wgs_per_run);
    r.wg_id = (int)(vb - (unsigned)r.q_id * (unsigned)wgs_per_run);
    unsigned long long sd = (((unsigned long long)a.seed_hi << 32) | a.seed_lo) + (unsigned long long)r.q_id * 0x9E3779B97F4A7C15ull;
    if (a.seed_dev) sd += __hip_atomic_load(a.seed_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a.seed_lo = (uint32_t)(sd & 0xffffffffull); a.seed_hi = (uint32_t)(sd >> 32);
    a.ws += (size_t)r.q_id * a.ws_stride;
    a.samples += (size_t)r.q_id * a.P * 3;
    a.alpha += (size_t)r.q_id * a.P;
    a.stats += (size_t)r.q_id * 4 * (a.n_epochs > 0 ? a.n_epochs : 1);
    r.ws = (SamplerWs*)a.ws;
    r.winners_base = (unsigned long long*)(a.ws + align_up(sizeof(SamplerWs), 256));
    return r;
}

__global__ void __launch_bounds__(256) k_ss_seed(FieldDev f, SamplerArgs a, int wgs_per_run) {
    const SsRun r = ss_resolve(a, wgs_per_run);
    const int P = (int)a.P, lpc = a.lpc, lsh = (a.lpc == 4) ? 2 : 0;
    const int64_t gtid = r.wg_id * (int64_t)blockDim.x + threadIdx.x, gthreads = (int64_t)wgs_per_run * blockDim.x;
    for (int64_t t = gtid; t < 4 * (int64_t)a.n_epochs; t += gthreads) a.stats[t] = 0;
    const int W = f.mask_dims[2], H = f.mask_dims[1], D = f.mask_dims[0];
    const int64_t nt = (int64_t)P * lpc;
    for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
        bool live = t < nt;
        int i = live ? (int)(t >> lsh) : 0;
        int sub = (int)(t & (lpc - 1));
        U4 rr = philox4x32_10(U4{(uint32_t)i, 0u, 0u, 0x5eedu}, a.seed_lo, a.seed_hi);
        float p[3];
        if (f.mask && a.n_occ > 0) {
            int pick = (int)(((unsigned long long)rr.x * (unsigned long long)a.n_occ) >> 32);
            int v = a.occ_list[pick];
            int x = v % W, y = (v / W) % H, z = v / (W * H);
            float sv[3] = {(float)x + u01(rr.y), (float)y + u01(rr.z), (float)z + u01(rr.w)};
            float dims[3] = {(float)W - 1.0f, (float)H - 1.0f, (float)D - 1.0f};
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = ((f.mask_hi[c] - f.mask_lo[c]) * sv[c]) / dims[c] + f.mask_lo[c];
        } else {
            float u[3] = {u01(rr.y), u01(rr.z), u01(rr.w)};
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = u[c] * (f.aabb_hi[c] - f.aabb_lo[c]) + f.aabb_lo[c];
        }
        float al = a.quad ? alpha_quad(f, p, live) : ((lpc == 1) ? alpha1(f, p, live) : alpha4(f, p, sub, live));
        if (live && sub == 0) {
            a.samples[3 * i] = p[0]; a.samples[3 * i + 1] = p[1]; a.samples[3 * i + 2] = p[2];
            a.alpha[i] = al;
        }
    }
}

// threshold = torch.quantile(alpha, 0.6), linear interpolation (the persistent form's code): every thread of the workgroup returns it
__device__ inline float ss_threshold(const float* alpha_src, int P, int* hist) {
    __shared__ int s_red[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    float pos = 0.6f * (float)(P - 1);
    int lo = (int)floorf(pos);
    int hi = min(lo + 1, P - 1);
    float frac = pos - (float)lo;
    const uint32_t klo = iff_wg_select_key<false>(alpha_src, P, lo + 1, hist);
    int cnt_le = 0;
    uint32_t kmin = 0xffffffffu;
    for (int t = tid; t < P; t += (int)blockDim.x) {
        uint32_t key = iff_order_key(alpha_src[t]);
        cnt_le += (key <= klo) ? 1 : 0;
        kmin = (key > klo && key < kmin) ? key : kmin;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cnt_le += __shfl_xor(cnt_le, off, 64);
        uint32_t o = (uint32_t)__shfl_xor((int)kmin, off, 64);
        kmin = o < kmin ? o : kmin;
    }
    if (lane == 0) { s_red[wave] = cnt_le; s_red[16 + wave] = (int)kmin; }
    __syncthreads();
    cnt_le = 0;
    kmin = 0xffffffffu;
    for (int w = 0; w < nwave; ++w) {
        cnt_le += s_red[w];
        kmin = (uint32_t)s_red[16 + w] < kmin ? (uint32_t)s_red[16 + w] : kmin;
    }
    __syncthreads();
    const uint32_t khi = (hi == lo || cnt_le >= hi + 1) ? klo : kmin;
    float vlo = iff_order_key_inv(klo);
    float vhi = iff_order_key_inv(khi);
    return (frac < 0.5f) ? (vlo + (vhi - vlo) * frac) : (vhi - (vhi - vlo) * (1.0f - frac));
}

// A sample is still invalid at the start of iteration `it` if its slot is empty or holds a key of iteration >= it: the other
// workgroups of the same launch post their keys of iteration `it` while this one may still be reading the slots
__device__ __forceinline__ bool ss_still_invalid(unsigned long long wv, int it) { return wv == 0ull || (int)((wv >> 20) & 0xfffull) >= it; }

// the ascending list of samples that are still invalid at the start of iteration `it`, into s_list; returns their number.  Each wave owns a contiguous
// quarter of the samples: one counting pass, the four wave totals through LDS, one writing pass (two workgroup barriers whatever P)
__device__ inline int ss_build_list(const unsigned long long* winners, int P, int it, int* s_list, int* s_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_wave = ((P + 255) / 256) * 64;                  // a multiple of 64
    const int w0 = wave * per_wave, w1 = min(P, w0 + per_wave);
    int cnt = 0;
    for (int c0 = w0; c0 < w1; c0 += 64) {
        const int i = c0 + lane;
        const bool stay = i < w1 && ss_still_invalid(__hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), it);
        cnt += __popcll(__ballot(stay));
    }
    if (lane == 0) s_tot[wave] = cnt;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_tot[w];
    const int K = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    for (int c0 = w0; c0 < w1; c0 += 64) {
        const int i = c0 + lane;
        const bool stay = i < w1 && ss_still_invalid(__hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), it);
        const unsigned long long bal = __ballot(stay);
        if (stay) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        base += __popcll(bal);
    }
    __syncthreads();
    return K;
}

// the candidates of iteration `it`: K invalid samples (s_list), m = 5 P / K candidates each, this workgroup's share
__device__ inline void ss_candidates(const FieldDev& f, const SamplerArgs& a, unsigned long long* winners, int epoch, int it, int wg_id, int wgs,
                                     float thresh, int K, const int* s_list, const float* s_pos, int& m_out) {
    const int P = (int)a.P, lpc = a.lpc, lsh = (a.lpc == 4) ? 2 : 0;
    const int64_t gtid = wg_id * (int64_t)blockDim.x + threadIdx.x, gthreads = (int64_t)wgs * blockDim.x;
    const int m = (5 * P) / K;
    m_out = m;
    const int64_t nt = (int64_t)K * m * lpc;
    for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
        bool live = t < nt;
        int64_t slot = live ? (t >> lsh) : 0;
        int sub = (int)(t & (lpc - 1));
        int li = (int)(slot / m), j = (int)(slot - (int64_t)li * m);
        int i = s_list[li];
        const float* ps = a.cache_lds ? s_pos : a.samples;
        float base[3] = {ps[3 * i], ps[3 * i + 1], ps[3 * i + 2]}, p[3];
        uint32_t prio;
        candidate_position(a, base, i, j, epoch, it, p, prio);
        float al = a.quad ? alpha_quad(f, p, live) : ((lpc == 1) ? alpha1(f, p, live) : alpha4(f, p, sub, live));
        if (live && sub == 0 && al > thresh) {
            unsigned long long key = ((unsigned long long)prio << 32) | ((unsigned long long)it << 20) | (unsigned long long)(unsigned)(j + 1);
            atomicMax(&winners[i], key);
        }
    }
}

// The list from winner slots that are already in registers (chunk c of this wave = samples w0 + 64 c + lane): the iteration launch
// requests them together with the positions and the converged flag, so that ONE memory round trip precedes the candidates
template <int NW>
__device__ inline int ss_build_list_regs(const unsigned long long (&wv)[NW], int nch, int w0, int w1, int it, int* s_list, int* s_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < NW; ++c)
        if (c < nch) cnt += __popcll(__ballot(w0 + 64 * c + lane < w1 && ss_still_invalid(wv[c], it)));
    if (lane == 0) s_tot[wave] = cnt;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_tot[w];
    const int K = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
#pragma unroll
    for (int c = 0; c < NW; ++c)
        if (c < nch) {
            const int i = w0 + 64 * c + lane;
            const bool stay = i < w1 && ss_still_invalid(wv[c], it);
            const unsigned long long bal = __ballot(stay);
            if (stay) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            base += __popcll(bal);
        }
    __syncthreads();
    return K;
}

// one iteration of one run by `wgs` workgroups (this one is number wg_id): list, candidate budget, candidates.  Returns the number
// of invalid samples it found (0: nothing was done).  `thresh`, s_alpha / s_pos are the caller's
__device__ inline int ss_iteration(const FieldDev& f, const SamplerArgs& a, unsigned long long* winners, int epoch, int it, int wg_id, int wgs,
                                   float thresh, int* s_list, const float* s_pos, int* s_tot, int& m_out) {
    const int K = ss_build_list(winners, (int)a.P, it, s_list, s_tot);
    if (K == 0) return 0;
    ss_candidates(f, a, winners, epoch, it, wg_id, wgs, thresh, K, s_list, s_pos, m_out);
    return K;
}

// FINISH = false: iteration `it0` by the run's wgs_per_run workgroups.  FINISH = true: one workgroup per run, iterations it0 .. max
template <bool FINISH>
__global__ void __launch_bounds__(256) k_ss_iter(FieldDev f, SamplerArgs a, int wgs_per_run, int epoch, int it0) {
    extern __shared__ int s_list[];               // [P] invalid sample ids; then (cache_lds) alpha [P], pos [3P]
    __shared__ int hist[264];
    __shared__ int s_tot[4];
    __shared__ int s_done;
    const SsRun r = ss_resolve(a, wgs_per_run);
    const int P = (int)a.P, tid = threadIdx.x;
    unsigned long long* winners = r.winners_base + (size_t)(epoch & 1) * P;
    float* s_alpha = reinterpret_cast<float*>(s_list + P);
    float* s_pos = s_alpha + P;
    // this run's epoch has converged already?  ONE read per workgroup: workgroup 0 of this very launch may be setting the flag, and
    // threads of one workgroup that read it at different times would part ways before a workgroup barrier
    constexpr int NW = SS_SMALL_P / 256;
    if (!FINISH && a.cache_lds && P <= SS_SMALL_P) {
        // the short chain: winner slots, converged flag, positions (and alphas) requested together -- one memory round trip -- then the
        // list from registers, then the candidates.  An iteration launch holds its slots for ~10 us next to the other steps' kernels
        // (752 workgroups whose waves each block a fan-march or trunk wave meanwhile): every round trip less is throughput
        const int lane = tid & 63, wave = tid >> 6, nch = (P + 255) / 256;
        const int w0 = wave * nch * 64, w1 = min(P, w0 + nch * 64);
        unsigned long long wv[NW];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            const int i = w0 + 64 * c + lane;
            wv[c] = (c < nch && i < w1) ? __hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        if (tid == 0) s_done = __hip_atomic_load(&r.ws->done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int t = tid; t < P; t += 256) {
            if (it0 == 0) s_alpha[t] = a.alpha[t];
            s_pos[3 * t] = a.samples[3 * t]; s_pos[3 * t + 1] = a.samples[3 * t + 1]; s_pos[3 * t + 2] = a.samples[3 * t + 2];
        }
        __syncthreads();
        if (s_done == epoch + 1) return;
        float thresh;
        if (it0 == 0) {
            thresh = ss_threshold(s_alpha, P, hist);
            if (r.wg_id == 0 && tid == 0) { r.ws->thresh = thresh; a.stats[epoch * 4 + 2] = __float_as_int(thresh); }
        } else {
            thresh = r.ws->thresh;
        }
        if (it0 >= a.max_iterations) return;
        const int K = ss_build_list_regs<NW>(wv, nch, w0, w1, it0, s_list, s_tot);
        if (K == 0) {
            if (r.wg_id == 0 && tid == 0) __hip_atomic_store(&r.ws->done_epoch, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        int m = 0;
        ss_candidates(f, a, winners, epoch, it0, r.wg_id, wgs_per_run, thresh, K, s_list, s_pos, m);
        if (r.wg_id == 0 && tid == 0) { a.stats[epoch * 4 + 0] = it0 + 1; a.stats[epoch * 4 + 3] = m; }
        return;
    }
    if (tid == 0) s_done = __hip_atomic_load(&r.ws->done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_done == epoch + 1) return;
    if (a.cache_lds) {
        for (int t = tid; t < P; t += 256) {
            if (it0 == 0) s_alpha[t] = a.alpha[t];
            s_pos[3 * t] = a.samples[3 * t]; s_pos[3 * t + 1] = a.samples[3 * t + 1]; s_pos[3 * t + 2] = a.samples[3 * t + 2];
        }
        __syncthreads();
    }
    float thresh;
    if (it0 == 0) {
        thresh = ss_threshold(a.cache_lds ? s_alpha : a.alpha, P, hist);
        if (r.wg_id == 0 && tid == 0) { r.ws->thresh = thresh; a.stats[epoch * 4 + 2] = __float_as_int(thresh); }
    } else {
        thresh = r.ws->thresh;                    // written by an earlier launch of this epoch
    }
    for (int it = it0; it < a.max_iterations; ++it) {
        int m = 0;
        const int K = ss_iteration(f, a, winners, epoch, it, FINISH ? 0 : r.wg_id, FINISH ? 1 : wgs_per_run, thresh, s_list, s_pos, s_tot, m);
        if (K == 0) {
            if (r.wg_id == 0 && tid == 0) __hip_atomic_store(&r.ws->done_epoch, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (r.wg_id == 0 && tid == 0) { a.stats[epoch * 4 + 0] = it + 1; a.stats[epoch * 4 + 3] = m; }
        if (!FINISH) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's atomics have reached the memory side
        __syncthreads();
    }
}

// ---- large P (> 2560): an iteration is TWO launches.  k_ss_list: one 1024-thread workgroup per run writes the run's list of invalid
// samples (and, in the epoch's first iteration, the threshold) to the workspace -- with thousands of samples, every one of a run's
// ~400 candidate workgroups rebuilding the list for itself reads the winner slots 400 times over (measured on the reference's
// default P = 20 000: 316 poses/s against 369 with the persistent form).  k_ss_cand: the candidates, list entries read from there.
constexpr int SS_LIST_THREADS = 1024;
__global__ void __launch_bounds__(SS_LIST_THREADS) k_ss_list(FieldDev f, SamplerArgs a, int epoch, int it) {
    __shared__ int hist[264];
    __shared__ int s_tot16[16];
    __shared__ int s_done;
    const SsRun r = ss_resolve(a, 1);
    const int P = (int)a.P, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_done = __hip_atomic_load(&r.ws->done_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_done == epoch + 1) return;
    unsigned long long* winners = r.winners_base + (size_t)(epoch & 1) * P;
    int* list = reinterpret_cast<int*>(a.ws + align_up(sizeof(SamplerWs), 256) + align_up((size_t)P * 16, 256));
    if (it == 0) {
        const float thresh = ss_threshold(a.alpha, P, hist);
        if (tid == 0) { r.ws->thresh = thresh; a.stats[epoch * 4 + 2] = __float_as_int(thresh); }
    }
    // each of the 16 waves owns a contiguous sixteenth of the samples: count, wave totals through LDS, write
    const int per_wave = ((P + SS_LIST_THREADS - 1) / SS_LIST_THREADS) * 64;
    const int w0 = wave * per_wave, w1 = min(P, w0 + per_wave);
    int cnt = 0;
    for (int c0 = w0; c0 < w1; c0 += 64) {
        const int i = c0 + lane;
        const bool stay = i < w1 && ss_still_invalid(__hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), it);
        cnt += __popcll(__ballot(stay));
    }
    if (lane == 0) s_tot16[wave] = cnt;
    __syncthreads();
    int base = 0, K = 0;
    for (int w = 0; w < SS_LIST_THREADS / 64; ++w) { base += w < wave ? s_tot16[w] : 0; K += s_tot16[w]; }
    for (int c0 = w0; c0 < w1; c0 += 64) {
        const int i = c0 + lane;
        const bool stay = i < w1 && ss_still_invalid(__hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), it);
        const unsigned long long bal = __ballot(stay);
        if (stay) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        base += __popcll(bal);
    }
    if (tid == 0) {
        r.ws->K_list = K;
        if (K == 0) __hip_atomic_store(&r.ws->done_epoch, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else { a.stats[epoch * 4 + 0] = it + 1; a.stats[epoch * 4 + 3] = (5 * P) / K; }
    }
}

__global__ void __launch_bounds__(256) k_ss_cand(FieldDev f, SamplerArgs a, int wgs_per_run, int epoch, int it) {
    const SsRun r = ss_resolve(a, wgs_per_run);
    const int P = (int)a.P;
    // written by k_ss_list of this iteration (an earlier launch): every thread reads the same values
    const int K = r.ws->K_list;
    if (r.ws->done_epoch == epoch + 1 || K == 0) return;
    const float thresh = r.ws->thresh;
    unsigned long long* winners = r.winners_base + (size_t)(epoch & 1) * P;
    const int* list = reinterpret_cast<const int*>(a.ws + align_up(sizeof(SamplerWs), 256) + align_up((size_t)P * 16, 256));
    SamplerArgs b = a;
    b.cache_lds = 0;                               // base positions straight from a.samples
    int m = 0;
    ss_candidates(f, b, winners, epoch, it, r.wg_id, wgs_per_run, thresh, K, list, nullptr, m);
}

__global__ void __launch_bounds__(256) k_ss_apply(FieldDev f, SamplerArgs a, int wgs_per_run, int epoch) {
    __shared__ int s_cnt[4];
    const SsRun r = ss_resolve(a, wgs_per_run);
    const int P = (int)a.P, lpc = a.lpc, lsh = (a.lpc == 4) ? 2 : 0;
    unsigned long long* winners = r.winners_base + (size_t)(epoch & 1) * P;
    unsigned long long* winners_next = r.winners_base + (size_t)((epoch + 1) & 1) * P;
    const int64_t gtid = r.wg_id * (int64_t)blockDim.x + threadIdx.x, gthreads = (int64_t)wgs_per_run * blockDim.x;
    const int64_t nt = (int64_t)P * lpc;
    int left = 0;
    for (int64_t t = gtid; t < ((nt + 63) & ~(int64_t)63); t += gthreads) {
        bool live = t < nt;
        int i = live ? (int)(t >> lsh) : 0;
        int sub = (int)(t & (lpc - 1));
        unsigned long long wv = live ? __hip_atomic_load(&winners[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        bool moved = wv != 0ull;
        int j = (int)(wv & 0xfffffull) - 1, wit = (int)((wv >> 20) & 0xfffull);
        float base[3] = {a.samples[3 * i], a.samples[3 * i + 1], a.samples[3 * i + 2]}, p[3];
        uint32_t prio;
        candidate_position(a, base, i, moved ? j : 0, epoch, wit, p, prio);
        float al = a.quad ? alpha_quad(f, p, live && moved) : ((lpc == 1) ? alpha1(f, p, live && moved) : alpha4(f, p, sub, live && moved));
        if (live && sub == 0) {
            if (moved) {
                a.samples[3 * i] = p[0]; a.samples[3 * i + 1] = p[1]; a.samples[3 * i + 2] = p[2];
                a.alpha[i] = al;
            } else {
                left += 1;
            }
            __hip_atomic_store(&winners_next[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // samples still invalid when the epoch's loop ended (stats[1]; 0 unless max_iterations cut it short)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) left += __shfl_xor(left, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = left;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (tot) atomicAdd(&a.stats[epoch * 4 + 1], tot);
    }
}

__global__ void k_zero_u64(unsigned long long* p, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

// workgroups one sampler run uses at P points, and how many sampler workgroups the device can hold at once (from the
// kernel's own register / LDS footprint): callers that keep several launches in flight must stay below the second number
// in total, or the in-kernel barriers of different launches could wait on each other until the spin timeout.
// lanes per candidate.  Both forms draw the same samples bit for bit; they trade latency for footprint.  Four lanes (one
// texel quarter each): 47 workgroups per run at P = 593, 245 us -- best for single queries and small batches.  One lane
// (needs a density texel of one 64-B line, i.e. n_density = 16): 12 workgroups per run, 340 us -- four times as many runs
// fit on the device at once, which is what large batches in flight need (measured: 16 queries per launch x 4 launches in
// flight 6790 poses/s against 6190 for the best four-lane configuration).  iff_field_desc.density_lanes forces a form.
int sampler_lpc(const FieldDev& f, int B) {
    if (f.n_density != 16 || f.density_lanes == 4) return 4;
    if (f.density_lanes == 1) return 1;
    if (sampler_stepped(f)) return 4;          // (the stepped form's default is the quad form: launch_surface_sample_occ; 4 = its footprint class)
    return B >= 8 ? 1 : 4;
}

// B = runs per launch.  A launch takes at most a quarter of the device's sampler slots when it can: a caller that keeps steps in
// flight on several streams (bench.py: four) then has room for all of them -- the sampler is latency-bound (parked at grid
// barriers 80 % of its life), so fewer, longer-looping workgroups per run cost little, while a launch that fills the device alone
// (640^3 model: 47 KB of LDS per workgroup, 3 per CU) clamps the whole pipeline to two steps in flight.  Results do not depend on
// the workgroup count (every candidate has its own counter-based stream; picks go through order-independent atomics).
hipError_t sampler_residency(int64_t P, int n_cus, int lpc, int B, int* wgs_per_query, int* capacity) {
    if (P < 1 || P > SAMPLER_MAX_POINTS) return hipErrorInvalidValue;
    const size_t lds = (size_t)P * sizeof(int) * (P <= SAMPLER_CACHE_POINTS ? 5 : 1);
    if (lds > 48 * 1024) {
        hipError_t ea = hipFuncSetAttribute((const void*)k_surface_sample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return ea;
    }
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_surface_sample, 256, lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) return hipErrorInvalidValue;
    int64_t want = (5 * P * lpc + 255) / 256;
    want = want < 1 ? 1 : (want > n_cus ? n_cus : want);
    const int64_t cap = (int64_t)per_cu * n_cus;
    if (B >= 1 && want * B * 4 > cap) want = cap / (4 * (int64_t)B) >= 1 ? cap / (4 * (int64_t)B) : (cap / B >= 1 ? (want < cap / B ? want : cap / B) : 1);
    *wgs_per_query = (int)want;
    *capacity = (int)cap;
    return hipSuccess;
}

// iff_field_desc.sampler_persistent = 1 keeps the one-launch form (the parity test builds such a handle)
bool sampler_stepped(const FieldDev& f) { return f.sampler_persistent == 0; }

static hipError_t launch_surface_sample_stepped(const FieldDev& f, SamplerArgs a, int B, hipStream_t s) {
    const int64_t P = a.P;
    const int lpc = a.lpc;
    const int wgs_pts = (int)((P * lpc + 255) / 256);                    // one thread per (point, lane)
    int64_t want = (5 * P * lpc + 255) / 256;                            // one thread per candidate of an iteration
    if (want > 1024) want = 1024;
    const int wgs_it = (int)want;
    const size_t lds = (size_t)P * sizeof(int) * (a.cache_lds ? 5 : 1);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ss_iter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ss_iter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    a.wgs_per_query = wgs_it;
    hipLaunchKernelGGL(k_ss_seed, dim3((unsigned)(wgs_pts * B)), dim3(256), 0, s, f, a, wgs_pts);
    for (int epoch = 0; epoch < a.n_epochs; ++epoch) {
        const int slots = a.max_iterations < ss_slots(epoch) ? a.max_iterations : ss_slots(epoch);
        for (int it = 0; it < slots; ++it) {
            if (P > SS_SMALL_P) {
                hipLaunchKernelGGL(k_ss_list, dim3((unsigned)B), dim3(SS_LIST_THREADS), 0, s, f, a, epoch, it);
                hipLaunchKernelGGL(k_ss_cand, dim3((unsigned)(wgs_it * B)), dim3(256), 0, s, f, a, wgs_it, epoch, it);
            } else {
                hipLaunchKernelGGL((k_ss_iter<false>), dim3((unsigned)(wgs_it * B)), dim3(256), lds, s, f, a, wgs_it, epoch, it);
            }
        }
        if (a.max_iterations > slots)
            hipLaunchKernelGGL((k_ss_iter<true>), dim3((unsigned)B), dim3(256), lds, s, f, a, 1, epoch, slots);
        hipLaunchKernelGGL(k_ss_apply, dim3((unsigned)(wgs_pts * B)), dim3(256), 0, s, f, a, wgs_pts, epoch);
    }
    return hipGetLastError();
}

hipError_t launch_surface_sample_occ(const FieldDev& f, const int* occ_list, int n_occ, int B, int64_t P, int n_epochs,
                                     int max_iterations, uint64_t seed, const uint64_t* seed_dev, float rho, float* samples,
                                     float* alpha, int* stats, void* ws, size_t ws_bytes, int n_cus, hipStream_t s) {
    if (P < 1 || P > SAMPLER_MAX_POINTS || n_epochs < 0 || n_epochs > SAMPLER_MAX_EPOCHS || max_iterations < 0 ||
        max_iterations > SAMPLER_MAX_ITERS || B < 1)
        return hipErrorInvalidValue;
    const size_t per_query = sampler_workspace_bytes(P);
    if (ws_bytes < per_query * (size_t)B) return hipErrorInvalidValue;
    if (sampler_stepped(f)) {
        const int64_t n_words = (int64_t)(per_query * (size_t)B / 8);
        hipLaunchKernelGGL(k_zero_u64, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, (unsigned long long*)ws, n_words);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
        SamplerArgs a;
        a.P = P; a.n_epochs = n_epochs; a.max_iterations = max_iterations;
        a.seed_lo = (uint32_t)(seed & 0xffffffffu); a.seed_hi = (uint32_t)(seed >> 32);
        a.seed_dev = (const unsigned long long*)seed_dev;
        a.rho = rho; a.samples = samples; a.alpha = alpha; a.stats = stats; a.ws = (unsigned char*)ws;
        a.occ_list = occ_list; a.n_occ = n_occ;
        a.wgs_per_query = 0; a.ws_stride = per_query; a.lpc = sampler_lpc(f, B);
        a.quad = f.density_lanes == 0 ? 1 : 0;           // iff_field_desc.density_lanes = 1 / 4 name the plain one- / four-lane forms
        if (a.quad) a.lpc = 1;
        a.cache_lds = P <= SAMPLER_CACHE_POINTS ? 1 : 0;
        return launch_surface_sample_stepped(f, a, B, s);
    }
    // one group of workgroups per query; all groups must be co-resident (in-kernel barriers)
    int wgs = 0, capacity = 0;
    const int lpc = sampler_lpc(f, B);
    hipError_t e = sampler_residency(P, n_cus, lpc, B, &wgs, &capacity);
    if (e != hipSuccess) return e;
    if ((int64_t)wgs * B > capacity) {
        wgs = capacity / B;
        if (wgs < 1) return hipErrorInvalidValue;
    }
    // barrier words + both winner buffers start at zero.  A kernel, not hipMemsetAsync: inside a captured hipGraph the
    // memset node did not reliably precede the sampler on replay (ROCm 7.2) -- the barrier counter then starts from the
    // previous replay's value and the grid barrier lets workgroups through early.
    const int64_t n_words = (int64_t)(per_query * (size_t)B / 8);
    hipLaunchKernelGGL(k_zero_u64, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, (unsigned long long*)ws, n_words);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    SamplerArgs a;
    a.P = P; a.n_epochs = n_epochs; a.max_iterations = max_iterations;
    a.seed_lo = (uint32_t)(seed & 0xffffffffu); a.seed_hi = (uint32_t)(seed >> 32);
    a.seed_dev = (const unsigned long long*)seed_dev;
    a.rho = rho; a.samples = samples; a.alpha = alpha; a.stats = stats; a.ws = (unsigned char*)ws;
    a.occ_list = occ_list; a.n_occ = n_occ;
    a.wgs_per_query = wgs; a.ws_stride = per_query; a.lpc = lpc; a.quad = 0;
    a.cache_lds = P <= SAMPLER_CACHE_POINTS ? 1 : 0;
    const size_t lds = (size_t)P * sizeof(int) * (a.cache_lds ? 5 : 1);
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute((const void*)k_surface_sample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_surface_sample, dim3((unsigned)(wgs * B)), dim3(256), lds, s, f, a);
    return hipGetLastError();
}
