// fan_march_kernels.hip -- K4f: TensorBase.forward (models/tensorBase.py:775-917) with the point-centred sampler
// (sample_point_color, :623-638) for rays that arrive as iso-cell FANS: 27 consecutive rays leaving one surface point
// (pose_estimation/sampling.py:442-488), 20 samples each, half a voxel apart.  All 540 samples of a fan sit inside a box of
// at most 12 texels per axis, so the VM tables are not gathered per sample from the vector caches (the general kernels of
// march_kernels.hip: 18 x 192-B texels per shaded sample through a 64-B/clk texture path) but staged ONCE per fan as
// coalesced row segments into LDS patches and read from there:
//
//   phase 0   bounding box of the tile's samples in texel indices (ray end points; positions are monotone along a ray)
//   phase A   one record per sample: occupancy, clamped tap indices relative to the box, zero-padded tap weights -- the tap
//             arithmetic of a sample is done once, not once per lane and plane
//   phase B   density: the three 12 x 12 x 64-B plane patches + lines in LDS, four lanes per ray (one 16-B quarter of the
//             density texel each), sigma / alpha, then the transmittance product of the ray (tensorBase.py:23-35)
//   phase C   appearance, plane by plane: a 12 x 12 x 192-B patch + its line in LDS, four lanes per ray (three 16-B quarters of
//             the 192-B texel each), samples with weight > rayMarch_weight_thres only (tensorBase.py:851), weight-summed
//             plane*line products in registers
//   phase D   basis_mat (tensoRF.py:158) once per ray on the weighted sums, from an LDS copy
//
// The next patch is fetched into registers while the current phase computes.  A tile is ANY 27 consecutive rays: when its box
// does not fit the patch (arbitrary rays, unisphere contraction) the same workgroup gathers from global memory with the lookup
// functions of iff_device.h -- same arithmetic, same bits, no LDS staging.  Every per-sample operation is the one the general
// kernels perform (shared lerp order, per-ray sequential accumulation), so alpha / acc / depth / counters are bit-identical to
// theirs; the 12 quarter sums of basis_mat are added in this kernel's own fixed tree.
#include "iff_device.h"
#include "iff_launch.h"
#include "march_common.h"

namespace {

constexpr int FR = 27;          // rays per tile = one iso-cell fan (pose_estimation/isocell.py:6-68)
constexpr int FS = 20;          // samples per ray (pose_estimation/sampling.py:247)
constexpr int FP = 12;          // patch side, texels
constexpr int NT = 128;         // threads: 32 groups of 4 lanes; group g serves ray g of the tile
constexpr int REC = 8;          // dwords per sample record
constexpr int PLANE16 = FP * FP * 16, LINE16 = FP * 16;      // density patch (floats)
constexpr int PLANE48 = FP * FP * 48, LINE48 = FP * 48;      // appearance patch (floats)
constexpr int PATCH_FLOATS = PLANE48 + LINE48;               // 7488 floats = 29 952 B = 3 * (PLANE16 + LINE16)
constexpr int BASIS_FLOATS = 27 * 12 * 12;
static_assert(3 * (PLANE16 + LINE16) == PATCH_FLOATS, "the density patches fill the appearance patch exactly");
static_assert(BASIS_FLOATS <= PATCH_FLOATS, "basis_mat is staged in the patch buffer");

typedef uint32_t u32q __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32q splat(float v) { return (f32q)(v); }
// bilinear / linear combination in the operation order of lerp_plane4 / lerp_line4 (iff_device.h)
__device__ __forceinline__ f32q lerp_plane_q(f32q nw, f32q ne, f32q sw, f32q se, const float pw[4]) {
    f32q r = nw * splat(pw[0]);
    r = __builtin_elementwise_fma(ne, splat(pw[1]), r);
    r = __builtin_elementwise_fma(sw, splat(pw[2]), r);
    r = __builtin_elementwise_fma(se, splat(pw[3]), r);
    return r;
}
__device__ __forceinline__ f32q lerp_line_q(f32q lo, f32q hi, const float lw[2]) {
    f32q r = lo * splat(lw[0]);
    return __builtin_elementwise_fma(hi, splat(lw[1]), r);
}

// ---- coalesced patch fetch: chunk = one 16-B piece; a patch row (12 texels) is one contiguous run of the table
// plane patch of a C-channel table into registers: rows lob.., columns loa..; `fast` = the patch lies inside the table
template <int C, int ROUNDS>
__device__ __forceinline__ void fetch_plane(const float* __restrict__ tab, int Ga, int Gb, int loa, int lob, bool fast, int tid,
                                            f32q (&reg)[ROUNDS]) {
    constexpr int CPT = C / 4, CPR = FP * CPT, NCH = FP * CPR;
    static_assert(ROUNDS * NT >= NCH, "rounds");
    if (fast) {
        const char* base = reinterpret_cast<const char*>(tab + ((size_t)lob * Ga + loa) * C);
        const int row_skip = (Ga - FP) * C * 4;                 // bytes between the end of a patch row and the next one
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int chunk = tid + NT * r;
            if (chunk < NCH) {
                const int ry = chunk / CPR;
                reg[r] = *reinterpret_cast<const f32q*>(base + (unsigned)(ry * row_skip + chunk * 16));
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int chunk = tid + NT * r;
            if (chunk < NCH) {
                const int texel = chunk / CPT, q = chunk - texel * CPT;
                const int ry = texel / FP, rx = texel - ry * FP;
                const int row = min(lob + ry, Gb - 1), col = min(loa + rx, Ga - 1);
                reg[r] = *reinterpret_cast<const f32q*>(tab + ((size_t)row * Ga + col) * C + 4 * q);
            }
        }
    }
}
template <int C, int ROUNDS>
__device__ __forceinline__ void fetch_line(const float* __restrict__ tab, int Gv, int lov, bool fast, int tid, f32q (&reg)[ROUNDS]) {
    constexpr int CPT = C / 4, NCH = FP * CPT;
    static_assert(ROUNDS * NT >= NCH, "rounds");
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int chunk = tid + NT * r;
        if (chunk < NCH) {
            const int rz = chunk / CPT, q = chunk - rz * CPT;
            const int row = fast ? lov + rz : min(lov + rz, Gv - 1);
            reg[r] = *reinterpret_cast<const f32q*>(tab + (size_t)row * C + 4 * q);
        }
    }
}
template <int NCH, int ROUNDS>
__device__ __forceinline__ void stash(float* dst, int tid, const f32q (&reg)[ROUNDS]) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int chunk = tid + NT * r;
        if (chunk < NCH) *reinterpret_cast<f32q*>(dst + 4 * chunk) = reg[r];
    }
}

struct RecView {           // one sample record, unpacked (all four lanes of a group read the same record)
    float w;               // compositing weight (after phase B)
    bool valid;
    int r[3], d[3];        // tap index relative to the box, and 1 when the high tap is a different texel
    float wt[3][2];        // zero-padded tap weights per axis
};
__device__ __forceinline__ RecView read_rec(const uint32_t* rec) {
    const u32q a = *reinterpret_cast<const u32q*>(rec), b = *reinterpret_cast<const u32q*>(rec + 4);
    RecView v;
    v.w = __uint_as_float(a.x);
    v.valid = (a.y >> 15) & 1u;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) { v.r[ax] = (a.y >> (5 * ax)) & 15u; v.d[ax] = (a.y >> (5 * ax + 4)) & 1u; }
    v.wt[0][0] = __uint_as_float(a.z); v.wt[0][1] = __uint_as_float(a.w);
    v.wt[1][0] = __uint_as_float(b.x); v.wt[1][1] = __uint_as_float(b.y);
    v.wt[2][0] = __uint_as_float(b.z); v.wt[2][1] = __uint_as_float(b.w);
    return v;
}

// MODE 2: the whole march of a tile (density, compositing, appearance, basis_mat).  MODE 1: appearance + basis_mat only, the
// compositing weights come from K4a's workspace (A/B aid).
template <int MODE>
__global__ void __launch_bounds__(NT, 2) k4f_fan_march(FieldDev f, MarchArgs a, int64_t n_tiles) {
    __shared__ __align__(16) float s_patch[PATCH_FLOATS];
    __shared__ __align__(16) uint32_t s_rec[FR * FS * REC];
    __shared__ __align__(16) float s_feat[FR * 28];
    __shared__ float s_ray[FR * 8];
    __shared__ int s_box[8];
    const int tid = threadIdx.x;
    const int g = tid >> 2, c = tid & 3;
    const bool grp_on = g < FR;
    const int G0 = f.grid[0], G1 = f.grid[1], G2 = f.grid[2];

    // one tile per workgroup (exact grid): inside a persistent tile loop LLVM hoists the per-lane patch offsets and table
    // descriptors out of the loop and spills them
    {
        const int64_t tile = blockIdx.x;
        const int64_t ray0 = tile * FR;
        const int n_live = (int)min((int64_t)FR, a.R - ray0);
        if (tid < FR) {
            float* sr = s_ray + tid * 8;
            if (tid < n_live) {
                const float* rp = a.rays + (ray0 + tid) * a.ray_cols;
                sr[0] = rp[0]; sr[1] = rp[1]; sr[2] = rp[2]; sr[3] = rp[3]; sr[4] = rp[4]; sr[5] = rp[5];
                sr[6] = 0.0f; sr[7] = rp[a.ray_cols - 1];
            } else {
                sr[0] = sr[1] = sr[2] = 0.0f; sr[3] = sr[4] = 0.0f; sr[5] = 1.0f; sr[6] = sr[7] = 0.0f;
            }
        }
        if (tid < 6) s_box[tid] = tid < 3 ? 0x7fffffff : -1;
        __syncthreads();
        // ------------------------------------------------------------------------------------------------ phase 0: the box
        // x(s) is monotone in s along a ray (every operation of the position / normalisation chain is monotone and so is its
        // rounding), hence the taps of a ray's samples lie between the taps of its two end points
        if (tid < 2 * FR) {
            const int rl = tid % FR;
            if (rl < n_live) {
                const float* sr = s_ray + rl * 8;
                const float z = z_of(f, 0, FS, 0.0f, tid < FR ? 0 : FS - 1);
                const float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                float xn[3];
                field_normalize(f, p, xn);
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const int G = f.grid[ax];
                    const float x = unnorm(xn[ax], G);
                    int blo = 0, bhi = G - 1;              // NaN: the whole axis (forces the gather path)
                    if (x == x) {
                        const int fl = (int)floorf(fminf(fmaxf(x, -1.0f), (float)G));
                        blo = min(max(fl, 0), G - 1);
                        bhi = min(max(fl + 1, 0), G - 1);
                    }
                    atomicMin(&s_box[ax], blo);
                    atomicMax(&s_box[3 + ax], bhi);
                }
            }
        }
        __syncthreads();
        int lo[3];
        bool fits = !f.unisphere, inner = true;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            lo[ax] = __builtin_amdgcn_readfirstlane(s_box[ax]);
            const int hi = __builtin_amdgcn_readfirstlane(s_box[3 + ax]);
            fits = fits && (hi - lo[ax] + 1 <= FP);
            inner = inner && (lo[ax] + FP <= f.grid[ax]);
        }
        // ------------------------------------------------------------------------------------------------ prefetch for phase B
        f32q pre[16];
        f32q prel[3];
        if (MODE == 2 && fits) {
            // density planes: 5 rounds each (576 chunks), lines: one round each (48 chunks)
            f32q t0[5], t1[5], t2[5], l0[1], l1[1], l2[1];
            fetch_plane<16, 5>(f.dplane[0], G0, G1, lo[0], lo[1], inner, tid, t0);
            fetch_plane<16, 5>(f.dplane[1], G0, G2, lo[0], lo[2], inner, tid, t1);
            fetch_plane<16, 5>(f.dplane[2], G1, G2, lo[1], lo[2], inner, tid, t2);
            fetch_line<16, 1>(f.dline[0], G2, lo[2], inner, tid, l0);
            fetch_line<16, 1>(f.dline[1], G1, lo[1], inner, tid, l1);
            fetch_line<16, 1>(f.dline[2], G0, lo[0], inner, tid, l2);
#pragma unroll
            for (int r = 0; r < 5; ++r) { pre[r] = t0[r]; pre[5 + r] = t1[r]; pre[10 + r] = t2[r]; }
            prel[0] = l0[0]; prel[1] = l1[0]; prel[2] = l2[0];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ------------------------------------------------------------------------------------------------ phase A: records
        if (grp_on) {
            const float* sr = s_ray + g * 8;
            const bool live = g < n_live;
#pragma unroll 1
            for (int k = 0; k < FS / 4; ++k) {
                const int s = c + 4 * k;
                const float z = z_of(f, 0, FS, 0.0f, s);
                const float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                const bool inside = live && inside_aabb(f, p);
                u32q r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
                if (inside) {
                    float xn[3];
                    field_normalize(f, p, xn);
                    const float mv = f.mask ? mask_value(f, p) : 1.0f;
                    uint32_t packed = (mv > 0.0f) ? (1u << 15) : 0u;
                    float wt[3][2];
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        const AxisTap t = axis_tap(xn[ax], f.grid[ax]);
                        packed |= (uint32_t)(((t.i[0] - lo[ax]) & 15) | ((t.i[1] - t.i[0]) << 4)) << (5 * ax);
                        wt[ax][0] = t.w[0]; wt[ax][1] = t.w[1];
                    }
                    r0.y = packed; r0.z = __float_as_uint(wt[0][0]); r0.w = __float_as_uint(wt[0][1]);
                    r1.x = __float_as_uint(wt[1][0]); r1.y = __float_as_uint(wt[1][1]);
                    r1.z = __float_as_uint(wt[2][0]); r1.w = __float_as_uint(wt[2][1]);
                }
                if (MODE == 1) r0.x = __float_as_uint(live ? a.weights[(ray0 + g) * FS + s] : 0.0f);
                uint32_t* rec = s_rec + (g * FS + s) * REC;
                *reinterpret_cast<u32q*>(rec) = r0;
                *reinterpret_cast<u32q*>(rec + 4) = r1;
            }
        }
        if (MODE == 2 && fits) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32q t[5] = {pre[5 * i], pre[5 * i + 1], pre[5 * i + 2], pre[5 * i + 3], pre[5 * i + 4]};
                stash<FP * FP * 4, 5>(s_patch + i * PLANE16, tid, t);
                f32q l[1] = {prel[i]};
                stash<FP * 4, 1>(s_patch + 3 * PLANE16 + i * LINE16, tid, l);
            }
        }
        __syncthreads();
        // appearance plane 0 on its way while phase B computes
        f32q prea[2];
        if (fits) {
            f32q t[14];
            fetch_plane<48, 14>(f.aplane[0], G0, G1, lo[0], lo[1], inner, tid, t);
            fetch_line<48, 2>(f.aline[0], G2, lo[2], inner, tid, prea);
#pragma unroll
            for (int r = 0; r < 14; ++r) pre[r] = t[r];
        }
        __builtin_amdgcn_sched_barrier(0);
        // ------------------------------------------------------------------------------------------------ phase B: density
        const float* sr = s_ray + (grp_on ? g : 0) * 8;
        const int64_t r_glob = ray0 + g;
        const bool live = grp_on && g < n_live;
        unsigned shmask = 0u;
        if (MODE == 2) {
#pragma unroll 1
            for (int k = 0; k < FS / 4; ++k) {
                float feat_mine = 0.0f;
                bool valid_mine = false;
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    const int s = 4 * k + j;
                    const RecView rv = read_rec(s_rec + ((grp_on ? g : 0) * FS + s) * REC);
                    float part = 0.0f;
                    if (fits) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
                            const float* P = s_patch + i * PLANE16 + ((rv.r[ax_b] * FP + rv.r[ax_a]) * 16 + 4 * c);
                            const int da = rv.d[ax_a] * 16, db = rv.d[ax_b] * (FP * 16);
                            const f32q nw = *reinterpret_cast<const f32q*>(P), ne = *reinterpret_cast<const f32q*>(P + da);
                            const f32q sw = *reinterpret_cast<const f32q*>(P + db), se = *reinterpret_cast<const f32q*>(P + db + da);
                            const float* L = s_patch + 3 * PLANE16 + i * LINE16 + (rv.r[ax_v] * 16 + 4 * c);
                            const f32q ll = *reinterpret_cast<const f32q*>(L), lh = *reinterpret_cast<const f32q*>(L + rv.d[ax_v] * 16);
                            const float pw[4] = {rv.wt[ax_b][0] * rv.wt[ax_a][0], rv.wt[ax_b][0] * rv.wt[ax_a][1],
                                                 rv.wt[ax_b][1] * rv.wt[ax_a][0], rv.wt[ax_b][1] * rv.wt[ax_a][1]};
                            const float lw[2] = {rv.wt[ax_v][0], rv.wt[ax_v][1]};
                            const f32q pq = lerp_plane_q(nw, ne, sw, se, pw), lq = lerp_line_q(ll, lh, lw);
                            part = fmaf(pq.w, lq.w, fmaf(pq.z, lq.z, fmaf(pq.y, lq.y, fmaf(pq.x, lq.x, part))));
                        }
                    } else if (rv.valid) {
                        const float z = z_of(f, 0, FS, 0.0f, s);
                        const float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                        float xn[3];
                        field_normalize(f, p, xn);
                        part = density_partial(f, xn, c);
                    }
                    const float feat = sum4(rv.valid ? part : 0.0f);
                    if (j == c) { feat_mine = feat; valid_mine = rv.valid; }
                }
                // lane c finishes sample 4k + c: sigma and alpha (tensorBase.py:25,849)
                const int s = 4 * k + c;
                const float sigma = valid_mine ? feature2density(f, feat_mine) : 0.0f;
                const float zs = z_of(f, 0, FS, 0.0f, s);
                const float dist = (s + 1 < FS) ? (z_of(f, 0, FS, 0.0f, s + 1) - zs) : 0.0f;       // tensorBase.py:800-803
                const float alpha = 1.0f - expf(-sigma * (dist * f.distance_scale));
                if (grp_on) s_rec[(g * FS + s) * REC] = __float_as_uint(alpha);
            }
            __syncthreads();
            // the transmittance product of the ray (tensorBase.py:27-32), all four lanes alike
            float run_T = 1.0f, run_acc = 0.0f, run_depth = 0.0f;
            int run_valid = 0, run_app = 0;
            uint32_t* rec = s_rec + (grp_on ? g : 0) * FS * REC;
#pragma unroll 1
            for (int s = 0; s < FS; ++s) {
                const float alpha = __uint_as_float(rec[s * REC]);
                const bool valid = (rec[s * REC + 1] >> 15) & 1u;
                const float z = z_of(f, 0, FS, 0.0f, s);
                const float w = alpha * run_T;
                run_T = run_T * ((1.0f - alpha) + 1e-10f);
                run_acc += w;
                run_depth += w * z;
                run_valid += valid ? 1 : 0;
                const bool sh = w > f.weight_thres;                                     // tensorBase.py:851
                run_app += sh ? 1 : 0;
                shmask |= (sh ? 1u : 0u) << s;
                if (c == 0 && live) {
                    rec[s * REC] = __float_as_uint(w);
                    if (a.alpha) a.alpha[r_glob * FS + s] = alpha;
                }
            }
            if (c == 0 && live) {
                a.acc[r_glob] = run_acc;
                a.depth[r_glob] = run_depth + (1.0f - run_acc) * sr[7];
                if (a.counts) { a.counts[r_glob * 2] = run_valid; a.counts[r_glob * 2 + 1] = run_app; }
            }
        } else {
            const uint32_t* rec = s_rec + (grp_on ? g : 0) * FS * REC;
#pragma unroll 1
            for (int s = 0; s < FS; ++s) shmask |= (__uint_as_float(rec[s * REC]) > f.weight_thres ? 1u : 0u) << s;
        }
        if (!live) shmask = 0u;
        const bool any = shmask != 0u;
        // ------------------------------------------------------------------------------------------------ phase C: appearance
        float accp[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) accp[i] = 0.0f;
        const uint32_t* recs = s_rec + (grp_on ? g : 0) * FS * REC;
        if (fits) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                __syncthreads();                              // phase B / the previous plane is done with the patch buffer
                {
                    f32q t[14];
#pragma unroll
                    for (int r = 0; r < 14; ++r) t[r] = pre[r];
                    stash<FP * FP * 12, 14>(s_patch, tid, t);
                    stash<FP * 12, 2>(s_patch + PLANE48, tid, prea);
                }
                __syncthreads();
                if (i < 2) {                                  // the next plane (registers) under this plane's arithmetic
                    const int n = i + 1, na = mat_a(n), nb = mat_b(n), nv = vec_ax(n);
                    f32q t[14];
                    fetch_plane<48, 14>(f.aplane[n], f.grid[na], f.grid[nb], lo[na], lo[nb], inner, tid, t);
                    fetch_line<48, 2>(f.aline[n], f.grid[nv], lo[nv], inner, tid, prea);
#pragma unroll
                    for (int r = 0; r < 14; ++r) pre[r] = t[r];
                } else {                                      // basis_mat for phase D
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int chunk = tid + NT * r;
                        if (chunk < BASIS_FLOATS / 4) pre[r] = *reinterpret_cast<const f32q*>(f.basis_l12 + 4 * chunk);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
                unsigned m = shmask;
                while (m) {
                    const int s = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const RecView rv = read_rec(recs + s * REC);
                    const float* P = s_patch + ((rv.r[ax_b] * FP + rv.r[ax_a]) * 48 + 4 * c);
                    const int da = rv.d[ax_a] * 48, db = rv.d[ax_b] * (FP * 48);
                    const float* L = s_patch + PLANE48 + (rv.r[ax_v] * 48 + 4 * c);
                    const int dv = rv.d[ax_v] * 48;
                    const float pw[4] = {rv.wt[ax_b][0] * rv.wt[ax_a][0], rv.wt[ax_b][0] * rv.wt[ax_a][1],
                                         rv.wt[ax_b][1] * rv.wt[ax_a][0], rv.wt[ax_b][1] * rv.wt[ax_a][1]};
                    const float lw[2] = {rv.wt[ax_v][0], rv.wt[ax_v][1]};
#pragma unroll
                    for (int j = 0; j < 3; ++j) {             // quarter c + 4 j of the 192-B texel
                        const f32q nw = *reinterpret_cast<const f32q*>(P + 16 * j), ne = *reinterpret_cast<const f32q*>(P + 16 * j + da);
                        const f32q sw = *reinterpret_cast<const f32q*>(P + 16 * j + db), se = *reinterpret_cast<const f32q*>(P + 16 * j + db + da);
                        const f32q ll = *reinterpret_cast<const f32q*>(L + 16 * j), lh = *reinterpret_cast<const f32q*>(L + 16 * j + dv);
                        const f32q pr = lerp_plane_q(nw, ne, sw, se, pw) * lerp_line_q(ll, lh, lw);
                        accp[12 * j + 4 * i + 0] = fmaf(rv.w, pr.x, accp[12 * j + 4 * i + 0]);
                        accp[12 * j + 4 * i + 1] = fmaf(rv.w, pr.y, accp[12 * j + 4 * i + 1]);
                        accp[12 * j + 4 * i + 2] = fmaf(rv.w, pr.z, accp[12 * j + 4 * i + 2]);
                        accp[12 * j + 4 * i + 3] = fmaf(rv.w, pr.w, accp[12 * j + 4 * i + 3]);
                    }
                }
            }
        } else {
            // the gather path: a tile whose samples do not fit one patch reads its taps where the general kernels do
            unsigned m = shmask;
            while (m) {
                const int s = __ffs((int)m) - 1;
                m &= m - 1u;
                const float w = __uint_as_float(recs[s * REC]);
                const float z = z_of(f, 0, FS, 0.0f, s);
                const float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                float xn[3];
                field_normalize(f, p, xn);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float prod[12];
                    app_products_lane(f, xn, c + 4 * j, prod);
#pragma unroll
                    for (int q = 0; q < 12; ++q) accp[12 * j + q] = fmaf(w, prod[q], accp[12 * j + q]);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int chunk = tid + NT * r;
                if (chunk < BASIS_FLOATS / 4) pre[r] = *reinterpret_cast<const f32q*>(f.basis_l12 + 4 * chunk);
            }
        }
        // ------------------------------------------------------------------------------------------------ phase D: basis_mat
        __syncthreads();
        {
            f32q t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = pre[r];
            stash<BASIS_FLOATS / 4, 8>(s_patch, tid, t);
        }
        __syncthreads();
#pragma unroll 1
        for (int oo = 0; oo < 27; ++oo) {
            float v[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float* bq = s_patch + (oo * 12 + c + 4 * j) * 12;
                const f32q b0 = *reinterpret_cast<const f32q*>(bq), b1 = *reinterpret_cast<const f32q*>(bq + 4),
                           b2 = *reinterpret_cast<const f32q*>(bq + 8);
                const float bl[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
                float acc = 0.0f;
#pragma unroll
                for (int kk = 0; kk < 12; ++kk) acc = fmaf(bl[kk], accp[12 * j + kk], acc);
                v[j] = acc;
            }
            const float out = sum4((v[0] + v[1]) + v[2]);         // quarters (c, c+4, c+8), then the four lanes by xor butterfly
            if (c == 0 && grp_on) s_feat[g * 28 + oo] = out;
        }
        if (c == 0 && grp_on) s_feat[g * 28 + 27] = any ? 1.0f : 0.0f;
        __syncthreads();
        for (int t = tid; t < n_live * 7; t += NT)
            *reinterpret_cast<f32q*>(a.feat + ray0 * 28 + 4 * t) = *reinterpret_cast<const f32q*>(s_feat + 4 * t);
    }
}

}  // namespace

// The fused fan kernel serves the point-centred sampler with its 20 samples (the emission path) on tables of the reference's
// shapes; a field under unisphere contraction, or one whose ten steps cover many more than 5 texels of an axis (the reference's
// step_ratio 0.5 gives 5), would send most tiles down the gather path, so those keep the general kernels.  A performance
// choice only: the per-tile box test decides what the kernel does.
bool fan_march_eligible(const FieldDev& f, int mode, int S) {
    if (mode != 0 || S != FS || f.unisphere || f.n_density != 16 || f.n_app != 48 || f.app_dim != 27) return false;
    for (int ax = 0; ax < 3; ++ax) {
        const float texels = 10.0f * f.step_size * f.inv_aabb[ax] * 0.5f * (float)(f.grid[ax] - 1);
        if (!(texels <= 6.5f)) return false;
    }
    return true;
}

hipError_t launch_fan_march(const FieldDev& f, const MarchArgs& a, int variant, hipStream_t s) {
    const int64_t n_tiles = (a.R + FR - 1) / FR;
    if (n_tiles == 0) return hipSuccess;
    if (n_tiles > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t grid = n_tiles;
    if (variant == 1) hipLaunchKernelGGL((k4f_fan_march<1>), dim3((unsigned)grid), dim3(NT), 0, s, f, a, n_tiles);
    else hipLaunchKernelGGL((k4f_fan_march<2>), dim3((unsigned)grid), dim3(NT), 0, s, f, a, n_tiles);
    return hipGetLastError();
}
