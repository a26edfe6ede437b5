// march_kernels.hip -- K3 isocell_emit and K4 march_composite_shade (TensorBase.forward, models/tensorBase.py:775-917).
//
// K4 is three launches over 16-ray tiles (256-thread workgroups; fused into one kernel the compiler needs ~370 registers
// per lane and the gathers run at one wave per SIMD -- split, they run at 3-4 waves per SIMD):
//   K4a density + compositing
//     - every (ray, sample) point gets 4 lanes (one 16-B quarter of each 64-B texel); sigma lands in LDS
//     - one lane per ray walks its samples: alpha, transmittance product, weights, acc, depth, counters
//       (sequential by definition); the weights [R,S] go to a caller-provided workspace
//   K4b appearance + shading, 16 lanes per ray = 4 sub-groups x 4 channel-quarter lanes
//     - sub-group q takes samples q, q+4, ... whose weight passes rayMarch_weight_thres and accumulates
//       weight * (plane*line) for its 36 channels in registers; a fixed-order xor butterfly merges the sub-groups
//       (deterministic, no atomics)
//     - basis_mat once per ray on the weighted sum (linear, so equal to the reference's per-sample basis_mat up to
//       fp32 rounding); the 27 features + a "has shaded samples" flag go to the workspace
//   K4c Ref head by 16 lanes per ray, background blend, clamp.
// K4a processes samples in chunks of CH (LDS holds one chunk), so the 20-sample point-centred sampler and the
// ~1000-sample slab sampler share the code.
#include "iff_device.h"
#include "iff_launch.h"
#include "march_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------ K3
// 27 iso-cell directions (pose_estimation/isocell.py:6-68, N0=3, isrand=-1) are passed in by the host mirror, which
// evaluates the closed form once; the kernel applies rotate_isocell (isocell.py:144-171) and renormalises
// (sampling.py:455-457).
struct IsoCells { float v[27][3]; };

__global__ void k3_isocell_emit(IsoCells cells, const float* __restrict__ pts, const float* __restrict__ nrm, int64_t P,
                                float* __restrict__ ori, float* __restrict__ dirs, float* __restrict__ rays6) {
    int64_t n = P * 27;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t pi = t / 27;
        int c = (int)(t - pi * 27);
        float nx = -nrm[3 * pi], ny = -nrm[3 * pi + 1], nz = -nrm[3 * pi + 2];
        float nn = sqrtf(nx * nx + ny * ny + nz * nz);
        float bx = nx / nn, by = ny / nn, bz = nz / nn;
        // v = z x b = (-by, bx, 0); c = z.b = bz; s = |v|
        float vx = -by, vy = bx, vz = 0.0f;
        float cs = bz;
        float s = sqrtf(vx * vx + vy * vy + vz * vz);
        float k = (1.0f - cs) / (s * s);
        // K = [v]x ; R = I + K + K^2 k
        float K[3][3] = {{0.f, -vz, vy}, {vz, 0.f, -vx}, {-vy, vx, 0.f}};
        float R[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float k2 = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
                R[i][j] = ((i == j) ? 1.0f : 0.0f) + K[i][j] + k2 * k;
            }
        float a[3] = {cells.v[c][0], cells.v[c][1], cells.v[c][2]};
        float d[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[i] = a[0] * R[i][0] + a[1] * R[i][1] + a[2] * R[i][2];
        float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        dirs[3 * t] = d[0] / dn; dirs[3 * t + 1] = d[1] / dn; dirs[3 * t + 2] = d[2] / dn;
        float ox = pts[3 * pi], oy = pts[3 * pi + 1], oz = pts[3 * pi + 2];
        ori[3 * t] = ox; ori[3 * t + 1] = oy; ori[3 * t + 2] = oz;
        if (rays6) {
            rays6[6 * t] = ox; rays6[6 * t + 1] = oy; rays6[6 * t + 2] = oz;
            rays6[6 * t + 3] = d[0] / dn; rays6[6 * t + 4] = d[1] / dn; rays6[6 * t + 5] = d[2] / dn;
        }
    }
}

hipError_t launch_isocell_emit(const float* cells27x3_host, const float* pts, const float* nrm, int64_t P, float* ori,
                               float* dirs, float* rays6, hipStream_t s) {
    IsoCells c;
    for (int i = 0; i < 27; ++i)
        for (int j = 0; j < 3; ++j) c.v[i][j] = cells27x3_host[3 * i + j];
    int64_t n = P * 27;
    int grid = (int)((n + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k3_isocell_emit, dim3(grid), dim3(256), 0, s, c, pts, nrm, P, ori, dirs, rays6);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K4
constexpr int RPB = 16;   // rays per workgroup tile (256 threads / 16 lanes per ray in phases 3-4)
constexpr int CH = 32;    // samples per chunk held in LDS

// ---- K4a: density gather + alpha compositing.  Writes the per-sample weights for K4b.
// LPS = lanes per sample: 4 (each lane one 16-B quarter of every 64-B texel, 16-ray tiles) or 1 (one lane gathers all 16
// channels: the tap arithmetic, which dominates this kernel's instruction count, is done once per sample instead of
// four times; 64-ray tiles).  Both produce the same bits (density_full).
template <int LPS>
__global__ void __launch_bounds__(256) k4a_density_composite(FieldDev f, MarchArgs a, int64_t n_tiles) {
    constexpr int RPB = LPS == 4 ? 16 : 64;
    __shared__ float s_sigma[RPB * CH];   // validity flag per sample (+1 / -1)
    __shared__ float s_alpha[RPB * CH];
    __shared__ float s_ray[RPB * 8];      // o(3) d(3) t0 last
    const int tid = threadIdx.x;
    const int S = a.S;
    const int n_chunks = (S + CH - 1) / CH;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t ray0 = tile * RPB;
        __syncthreads();   // previous tile's readers are done with s_ray
        if (tid < RPB) {
            int64_t r = ray0 + tid;
            bool live = r < a.R;
            float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f}, last = 0.f;
            if (live) {
                const float* rp = a.rays + r * a.ray_cols;
                o[0] = rp[0]; o[1] = rp[1]; o[2] = rp[2]; d[0] = rp[3]; d[1] = rp[4]; d[2] = rp[5];
                last = rp[a.ray_cols - 1];
            }
            float* sr = s_ray + tid * 8;
            sr[0] = o[0]; sr[1] = o[1]; sr[2] = o[2]; sr[3] = d[0]; sr[4] = d[1]; sr[5] = d[2];
            sr[6] = (a.mode == 1) ? slab_entry(f, o, d) : 0.0f; sr[7] = last;
        }
        // per-ray running state lives in the registers of thread `tid < RPB`
        float run_T = 1.0f, run_acc = 0.0f, run_depth = 0.0f;
        int run_valid = 0, run_app = 0;
        __syncthreads();
        for (int c = 0; c < n_chunks; ++c) {
            const int s_base = c * CH;
            const int ns = min(CH, S - s_base);
            // ---------------- sigma and alpha for RPB x ns points, LPS lanes per point
            const int n_lane_tasks = RPB * ns * LPS;
            for (int t = tid; t < ((n_lane_tasks + 63) & ~63); t += 256) {
                bool live = t < n_lane_tasks;
                int ps = live ? (t / LPS) : 0;
                int lsub = t % LPS;
                int rl = ps / ns, sl = ps - rl * ns;
                const float* sr = s_ray + rl * 8;
                float z = z_of(f, a.mode, S, sr[6], s_base + sl);
                float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                const bool inside = live && (ray0 + rl < a.R) && inside_aabb(f, p);
                // the occupancy bytes and the density taps are requested together (both are safe for any coordinate);
                // the mask decides afterwards which points count -- one memory round trip per pass instead of two
                float part = 0.0f;
                bool occ = true;
                if (inside) {
                    float xn[3];
                    field_normalize(f, p, xn);
                    if (f.mask) occ = mask_occupied(f, p, xn);
                    part = (LPS == 4) ? density_partial(f, xn, lsub) : density_full(f, xn);
                }
                const bool valid = inside && occ;
                float feat = (LPS == 4) ? sum4(valid ? part : 0.0f) : (valid ? part : 0.0f);
                if (live && lsub == 0) {
                    // sigma and, already here, alpha (tensorBase.py:25,849): the sequential pass below keeps only the
                    // transmittance product
                    const int s = s_base + sl;
                    float sigma = valid ? feature2density(f, feat) : 0.0f;
                    float dist = (s + 1 < S) ? (z_of(f, a.mode, S, sr[6], s + 1) - z) : 0.0f;       // tensorBase.py:800-803
                    s_alpha[rl * CH + sl] = 1.0f - expf(-sigma * (dist * f.distance_scale));
                    s_sigma[rl * CH + sl] = valid ? 1.0f : -1.0f;                                     // validity flag
                }
            }
            __syncthreads();
            // ---------------- compositing, one lane per ray
            if (tid < RPB) {
                const float* sr = s_ray + tid * 8;
                int64_t r = ray0 + tid;
                for (int sl = 0; sl < ns; ++sl) {
                    int s = s_base + sl;
                    bool valid = s_sigma[tid * CH + sl] >= 0.0f;
                    float z = z_of(f, a.mode, S, sr[6], s);
                    float alpha = s_alpha[tid * CH + sl];
                    float w = alpha * run_T;
                    run_T = run_T * ((1.0f - alpha) + 1e-10f);                                   // tensorBase.py:27-32
                    run_acc += w;
                    run_depth += w * z;
                    run_valid += valid ? 1 : 0;
                    run_app += (w > f.weight_thres) ? 1 : 0;                                      // tensorBase.py:851
                    if (r < a.R) {
                        a.weights[r * S + s] = w;
                        if (a.alpha) a.alpha[r * S + s] = alpha;
                    }
                }
            }
            __syncthreads();   // s_sigma is rewritten by the next chunk
        }
        if (tid < RPB && ray0 + tid < a.R) {
            const int64_t r = ray0 + tid;
            a.acc[r] = run_acc;
            a.depth[r] = run_depth + (1.0f - run_acc) * s_ray[tid * 8 + 7];
            if (a.counts) { a.counts[r * 2] = run_valid; a.counts[r * 2 + 1] = run_app; }
        }
    }
}

// ---- K4b: appearance gather for the samples that pass the weight threshold, then basis_mat on the weighted sums.
// A ray is served by a group of 16 lanes, 12 of them active; each active lane owns one 16-B quarter of the 192-B
// appearance texels (12 plane*line products).  Regroupings with 6 or 3 lanes per ray (2 / 4 quarters per lane, fewer tap
// computations) were measured slower -- what they save in tap arithmetic they lose in occupancy.
// Output: the per-ray feature vector [R][28] (27 features + a "has shaded samples" flag) for K4c.
#ifndef K4B_WAVES
#define K4B_WAVES 4
#endif
template <int APP, bool SHORT, int RB>
__global__ void __launch_bounds__((RB * 16 + 63) / 64 * 64, K4B_WAVES) k4b_appearance(FieldDev f, MarchArgs a, int64_t n_tiles) {
    extern __shared__ __align__(16) float smem[];
    constexpr int NL = 12;                         // n_app / 4 texel quarters = active lanes per group
    constexpr int G = 16;                          // lanes per ray group
    constexpr int NT = (RB * 16 + 63) / 64 * 64;   // threads: RB rays per workgroup tile, 16 lanes each
    constexpr int NW = (32 + G - 1) / G;           // weight registers per lane for S <= 32
    constexpr int LD = (APP + 3) & ~3;
    float* s_basis = smem;                         // [APP][NL][12]
    const int tid = threadIdx.x;
    for (int i = tid; i < APP * NL * 12; i += NT) s_basis[i] = f.basis_l12[i];
    __syncthreads();
    const int S = a.S;
    const int ray_l = tid / G, lg = tid % G;
    const int c0 = lg < NL ? lg : NL - 1;          // idle lanes shadow the last active lane's addresses (coalesced away)
    const float lane_on = lg < NL ? 1.0f : 0.0f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r = tile * RB + ray_l;
        const bool live = r < a.R && ray_l < RB;
        float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f};
        if (live) {
            const float* rp = a.rays + r * a.ray_cols;
            o[0] = rp[0]; o[1] = rp[1]; o[2] = rp[2]; d[0] = rp[3]; d[1] = rp[4]; d[2] = rp[5];
        }
        const float t0 = (a.mode == 1) ? slab_entry(f, o, d) : 0.0f;
        float accp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) accp[i] = 0.0f;
        bool any = false;
        auto shade_sample = [&](int s, float w) {
            float z = z_of(f, a.mode, S, t0, s);
            float p[3] = {o[0] + d[0] * z, o[1] + d[1] * z, o[2] + d[2] * z}, xn[3];
            field_normalize(f, p, xn);
            float prod[12];
            app_products_lane(f, xn, c0, prod);
#pragma unroll
            for (int i = 0; i < 12; ++i) accp[i] = fmaf(w, prod[i], accp[i]);
        };
        if (SHORT) {
            // short rays (S <= 32: the 20-sample point-centred sampler): the lanes of a ray fetch its weights once (lane l
            // holds samples l, l + G, ...), ballots turn "weight > threshold" (tensorBase.py:851) into a bit mask per ray,
            // and every ray then visits exactly its shaded samples, in order -- no trip spent on a sample no ray of the wave
            // shades, no weight load inside the loop
            float wreg[NW];
            unsigned mask = 0u;
            const int g0 = (tid & 63) / G * G;                     // first lane of this ray's group inside the wave
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int sj = lg + G * j;
                wreg[j] = (live && sj < S) ? a.weights[r * S + sj] : 0.0f;
                const unsigned long long bj = __ballot(wreg[j] > f.weight_thres);
                mask |= (unsigned)((bj >> g0) & ((1ull << G) - 1ull)) << (G * j);
            }
            any = mask != 0u;
            while (mask) {
                const int sidx = __ffs((int)mask) - 1;
                mask &= mask - 1u;
                float w = 0.0f;
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const float vj = __shfl(wreg[j], g0 + (sidx % G), 64);
                    w = (sidx / G == j) ? vj : w;
                }
                shade_sample(sidx, w);
            }
        } else {
            for (int s = 0; s < S; ++s) {
                float w = live ? a.weights[r * S + s] : 0.0f;
                if (w > f.weight_thres) {                                              // tensorBase.py:851
                    any = true;
                    shade_sample(s, w);
                }
            }
        }
        // basis_mat on the weighted sums: quarter c contributes a 12-term fmaf chain to each of the APP outputs; the
        // quarters are then added across the group by xor butterfly (fixed order: deterministic)
        // an opaque zero offset per tile keeps LLVM from hoisting the 324 weight reads out of the tile loop; the pointer stays
        // LDS-typed so that they are ds_read_b128 (an opaque POINTER loses its address space: flat loads, which queue in the
        // vector-memory pipe this kernel is bound by -- they were 23 % of its vector-memory instructions)
        int opaque0 = 0;
        asm volatile("" : "+v"(opaque0));
        const lds_cfloat_p basis_tile = (lds_cfloat_p)s_basis + opaque0;
#pragma unroll 1
        for (int oo = 0; oo < APP; ++oo) {
            const lds_cfloat4_p bq = (lds_cfloat4_p)(basis_tile + (oo * NL + c0) * 12);
            const f32q b0 = bq[0], b1 = bq[1], b2 = bq[2];
            const float bl[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
            float v = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 12; ++kk) v = fmaf(bl[kk], accp[kk], v);
            v = v * lane_on;
#pragma unroll
            for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
            if (lg == 0 && live) a.feat[r * LD + oo] = v;
        }
        if (lg == 0 && live) a.feat[r * LD + APP] = any ? 1.0f : 0.0f;
    }
}

// ---- K4b for short rays (S <= 32: the point-centred sampler), FIVE rays per wave: a ray is served by 12 consecutive lanes (one
// per 16-B quarter of the 192-B texels), so a wave-level gather returns 960 of its 1024 bytes instead of 768 -- the kernel is
// bound by the number of vector-memory instructions (each returns 64 lanes x 16 B at 64 B/clk: 16 cycles of the CU's texture
// path whatever its mask), and five rays per instruction are 20 % fewer of them per ray.  Same arithmetic per ray as
// k4b_appearance; the 12 partial sums of basis_mat are added in a fixed tree (6 + 6, 3 + 3, then the three).
template <int APP>
__global__ void __launch_bounds__(256, K4B_WAVES) k4b_appearance12(FieldDev f, MarchArgs a, int64_t n_tiles) {
    extern __shared__ __align__(16) float smem[];
    constexpr int NL = 12, RW = 5, RB = 4 * RW, NW = 3, LD = (APP + 3) & ~3;
    float* s_basis = smem;                         // [APP][NL][12]
    const int tid = threadIdx.x;
    for (int i = tid; i < APP * NL * 12; i += 256) s_basis[i] = f.basis_l12[i];
    __syncthreads();
    const int S = a.S;
    const int wave = tid >> 6, lane = tid & 63;
    const int g = lane / NL;                       // ray of the wave; 5 = the four spare lanes
    const bool grp_on = g < RW;
    const int c0 = grp_on ? lane - g * NL : 0;     // spare lanes shadow lane 0's addresses
    const int g0 = grp_on ? g * NL : 0;            // first lane of this ray's group
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r = tile * RB + wave * RW + g;
        const bool live = grp_on && r < a.R;
        float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f};
        if (live) {
            const float* rp = a.rays + r * a.ray_cols;
            o[0] = rp[0]; o[1] = rp[1]; o[2] = rp[2]; d[0] = rp[3]; d[1] = rp[4]; d[2] = rp[5];
        }
        float accp[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) accp[i] = 0.0f;
        // lane c of a ray holds the weights of samples c, c + 12, c + 24; ballots give every ray its mask of shaded samples
        float wreg[NW];
        unsigned long long mask = 0ull;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int sj = c0 + NL * j;
            wreg[j] = (live && sj < S) ? a.weights[r * S + sj] : 0.0f;
            const unsigned long long bj = __ballot(wreg[j] > f.weight_thres);              // tensorBase.py:851
            mask |= ((bj >> g0) & 0xfffull) << (NL * j);
        }
        if (!grp_on) mask = 0ull;
        const bool any = mask != 0ull;
        // Under unisphere contraction a normalised coordinate is three powf (utils.py:139-146): ~300 of the ~750 vector instructions
        // a wave spends per shaded sample when all 12 lanes of a ray evaluate it for the ray's current sample.  Lane c evaluates it
        // ONCE for the samples whose weights it holds (c, c + 12, c + 24) and the ray's lanes fetch the current sample's from there
        // -- the same function of the same point: the same bits.
        float xs[NW][3];
        if (f.unisphere) {
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const float zj = z_of(f, 0, S, 0.0f, c0 + NL * j);
                const float pj[3] = {o[0] + d[0] * zj, o[1] + d[1] * zj, o[2] + d[2] * zj};
                field_normalize(f, pj, xs[j]);
            }
        }
        while (mask) {
            const int sidx = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            const int j = sidx >= 2 * NL ? 2 : (sidx >= NL ? 1 : 0);
            const int src = g0 + (sidx - NL * j);
            // the ray's twelve lanes agree on (j, src): every lane picks ITS OWN slot j first and one cross-lane move per value follows
            // (twelve moves -- LDS round trips -- and the picks after them before: 4.2e7 LDS instructions per bicycle64k launch)
            float w = wreg[0];
#pragma unroll
            for (int jj = 1; jj < NW; ++jj) w = (j == jj) ? wreg[jj] : w;
            w = __shfl(w, src, 64);
            float xn[3];
            if (f.unisphere) {
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    float v = xs[0][ax];
#pragma unroll
                    for (int jj = 1; jj < NW; ++jj) v = (j == jj) ? xs[jj][ax] : v;
                    xn[ax] = __shfl(v, src, 64);
                }
            } else {
                const float z = z_of(f, 0, S, 0.0f, sidx);
                const float p[3] = {o[0] + d[0] * z, o[1] + d[1] * z, o[2] + d[2] * z};
                field_normalize(f, p, xn);
            }
            float prod[12];
            app_products_lane(f, xn, c0, prod);
#pragma unroll
            for (int i = 0; i < 12; ++i) accp[i] = fmaf(w, prod[i], accp[i]);
        }
        int opaque0 = 0;
        asm volatile("" : "+v"(opaque0));          // see k4b_appearance
        const lds_cfloat_p basis_tile = (lds_cfloat_p)s_basis + opaque0;
#pragma unroll 1
        for (int oo = 0; oo < APP; ++oo) {
            const lds_cfloat4_p bq = (lds_cfloat4_p)(basis_tile + (oo * NL + c0) * 12);
            const f32q b0 = bq[0], b1 = bq[1], b2 = bq[2];
            const float bl[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
            float v = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 12; ++kk) v = fmaf(bl[kk], accp[kk], v);
            v += __shfl_down(v, 6, 64);             // lanes 0..5 of the group: quarters c and c + 6
            v += __shfl_down(v, 3, 64);             // lanes 0..2
            const float v1 = __shfl_down(v, 1, 64), v2 = __shfl_down(v, 2, 64);
            if (c0 == 0 && live) a.feat[r * LD + oo] = (v + v1) + v2;
        }
        if (c0 == 0 && live) a.feat[r * LD + APP] = any ? 1.0f : 0.0f;
    }
}

// ---- K4c (Ref head on the per-ray features, background blend, clamp) is k_ref_shade<APP, true> in field_kernels.hip.

// workspace = compositing weights [R][S] followed by the per-ray features [R][28]
static size_t march_feat_offset(int64_t R, int S) { return ((size_t)R * (size_t)S * sizeof(float) + 255) / 256 * 256; }
size_t march_workspace_bytes(int64_t R, int S) { return march_feat_offset(R, S) + (size_t)R * 28 * sizeof(float); }

// 0 = the general kernels, 2 = the fused fan kernel (iff_field_desc.density_lanes != 0 names one of the general kernels and
// therefore keeps them: what the parity tests compare the fan kernel with).
int march_plan(const FieldDev& f, int mode, int S) {
    return (f.density_lanes == 0 && fan_kernel_for(f, mode, S) != 0) ? 2 : 0;
}

// Which fused fan kernel serves a march: 0 none, 4 = k4f_fan_march (four waves per fan, 12-texel boxes), 8 = k4g_fan_march (eight
// waves per fan, 12- or 22-texel boxes).  iff_field_desc.fan_waves names one; left at 0 the 22-texel boxes take the only kernel that
// stages them and the 12-texel boxes the one FAN12_DEFAULT_WAVES names (the faster on MI355X: DESIGN.md section 4).
#ifndef FAN12_DEFAULT_WAVES
#define FAN12_DEFAULT_WAVES 4
#endif
int fan_kernel_for(const FieldDev& f, int mode, int S) {
    const int side = fan8_patch_side(f, mode, S);
    const bool four = fan_march_eligible(f, mode, S);
    if (f.fan_waves == 4) return four ? 4 : 0;
    if (f.fan_waves == 8) return side ? 8 : 0;
    if (side == 12 && four) return FAN12_DEFAULT_WAVES;
    if (side) return 8;
    return four ? 4 : 0;
}

// the colours call of the fused plan also runs the Ref head in the fan kernel
bool march_head_fused(const FieldDev& f) { return fan_head_fusable(f) && f.head_lanes != 16; }

hipError_t launch_march(const FieldDev& f, const float* rays, int ray_cols, int64_t R, int mode, int S, const float* bg,
                        float* rgb, float* depth, float* acc, float* alpha, int* counts, float* feat_out, void* ws,
                        size_t ws_bytes, float* stage_ms_host, hipStream_t s) {
    if (ws_bytes < march_workspace_bytes(R, S)) return hipErrorInvalidValue;
    // optional per-launch timing (bench.py's roofline): events on the launch stream, one synchronise at the end
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (stage_ms_host) {
        for (auto& x : ev) { hipError_t ee = hipEventCreate(&x); if (ee != hipSuccess) return ee; }
        (void)hipEventRecord(ev[0], s);
    }
    MarchArgs a;
    a.rays = rays; a.ray_cols = ray_cols; a.R = R; a.mode = mode; a.S = S;
    a.bg[0] = bg[0]; a.bg[1] = bg[1]; a.bg[2] = bg[2];
    a.rgb = rgb; a.depth = depth; a.acc = acc; a.alpha = alpha; a.counts = counts; a.weights = (float*)ws;
    // feat_out: stop before the Ref head and hand the per-ray features [R][28] to the caller (iff_march_features)
    a.feat = feat_out ? feat_out : (float*)((char*)ws + march_feat_offset(R, S));
    a.n_tiles = (R + RPB - 1) / RPB;
    if (a.n_tiles == 0) return hipSuccess;
    const int fan = march_plan(f, mode, S);
    hipError_t e = hipSuccess;
    int64_t grid;
    if (fan != 2) {
        // one lane per sample when the density texel is one 64-B line (n_density = 16, every reference config)
        const bool one_lane = f.n_density == 16 && f.density_lanes != 4;
        const int64_t tiles_a = one_lane ? (R + 63) / 64 : a.n_tiles;
        grid = tiles_a < 256 * 8 ? tiles_a : 256 * 8;
        if (one_lane) hipLaunchKernelGGL((k4a_density_composite<1>), dim3((unsigned)grid), dim3(256), 0, s, f, a, tiles_a);
        else hipLaunchKernelGGL((k4a_density_composite<4>), dim3((unsigned)grid), dim3(256), 0, s, f, a, tiles_a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (stage_ms_host) (void)hipEventRecord(ev[1], s);
    size_t lds = (size_t)f.app_dim * 3 * f.n_app * sizeof(float);
    const bool fuse_head = fan == 2 && !feat_out && march_head_fused(f);
    if (fan) {
        e = fan_kernel_for(f, mode, S) == 8 ? launch_fan8_march(f, a, fuse_head ? 3 : 2, s) : launch_fan_march(f, a, fuse_head ? 3 : 2, s);
        if (e != hipSuccess) return e;
    } else if (mode == 0 && S <= 32) {
        const int64_t tiles12 = (R + 19) / 20;
        grid = tiles12 < 256 * 8 ? tiles12 : 256 * 8;
        hipLaunchKernelGGL((k4b_appearance12<27>), dim3((unsigned)grid), dim3(256), lds, s, f, a, tiles12);
    } else {
        grid = a.n_tiles < 256 * 8 ? a.n_tiles : 256 * 8;
        if (S <= 32) hipLaunchKernelGGL((k4b_appearance<27, true, 16>), dim3((unsigned)grid), dim3(256), lds, s, f, a, a.n_tiles);
        else hipLaunchKernelGGL((k4b_appearance<27, false, 16>), dim3((unsigned)grid), dim3(256), lds, s, f, a, a.n_tiles);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (stage_ms_host) (void)hipEventRecord(ev[2], s);
    if (!feat_out && !fuse_head) e = launch_shade_blend(f, rays, ray_cols, a.feat, acc, a.bg, R, rgb, s);
    if (stage_ms_host) {
        (void)hipEventRecord(ev[3], s);
        hipError_t es = hipEventSynchronize(ev[3]);
        for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&stage_ms_host[i], ev[i], ev[i + 1]);
        for (auto& x : ev) (void)hipEventDestroy(x);
        if (e == hipSuccess) e = es;
    }
    return e;
}
