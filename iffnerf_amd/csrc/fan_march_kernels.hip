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
//   phase B   density: the three 12 x 12 x 64-B plane patches + lines in LDS; a ray is served by EIGHT lanes = two sub-groups of
//             four (one 16-B quarter of the density texel each), sub-group h takes the samples s = h mod 2; sigma / alpha,
//             then the transmittance product of the ray (tensorBase.py:23-35)
//   phase C   appearance, plane by plane: a 12 x 12 x 192-B patch + its line in LDS, each lane three 16-B quarters of the
//             192-B texel, sub-group h the samples s = h mod 2 with weight > rayMarch_weight_thres (tensorBase.py:851),
//             weight-summed plane*line products in registers, the two sub-groups' sums added at the end
//   phase D   basis_mat (tensoRF.py:158) once per ray on the weighted sums, from an LDS copy (sub-group h: outputs h mod 2)
//
// The next patch is fetched into registers while the current phase computes.  A tile is ANY 27 consecutive rays: when its box
// does not fit the patch (arbitrary rays) the same workgroup gathers from global memory with the lookup
// functions of iff_device.h -- same arithmetic, same bits, no LDS staging.  Every per-sample operation is the one the general
// kernels perform (shared lerp order), so alpha / acc / depth / counters are bit-identical to theirs; the weighted feature
// sums are added in this kernel's own fixed order (even samples, odd samples, then the 12 quarter sums of basis_mat).
#include "iff_device.h"
#include "iff_launch.h"
#include "march_common.h"
#include "fan_common.h"
#include "fan_diag.h"

namespace {

constexpr int FR = 27;          // rays per tile = one iso-cell fan (pose_estimation/isocell.py:6-68)
constexpr int FS = 20;          // samples per ray (pose_estimation/sampling.py:247)
constexpr int FP = 12;          // patch side, texels
constexpr int PITCH = 14;       // texels per patch row in LDS: 2 mod 4, so that the 64-B quarter a tap falls into is (i_a + 2 i_b + j) mod 4 and
                                // the taps of neighbouring cells never share a bank quarter at different addresses (see the lane map below)
constexpr int NT = 256;         // threads: 32 groups of 8 lanes (two sub-groups of 4); group g serves ray g of the tile
constexpr int REC = 8;          // dwords per sample record
constexpr int PLANE16 = FP * PITCH * 16, LINE16 = FP * 16;   // density patch (floats)
constexpr int PLANE48 = FP * PITCH * 48, LINE48 = FP * 48;   // appearance patch (floats)
constexpr int PATCH_FLOATS = PLANE48 + LINE48;               // 8640 floats = 34 560 B = 3 * (PLANE16 + LINE16)
constexpr int BASIS_FLOATS = 27 * 12 * 12;
static_assert(3 * (PLANE16 + LINE16) == PATCH_FLOATS, "the density patches fill the appearance patch exactly");
static_assert(BASIS_FLOATS <= PATCH_FLOATS, "basis_mat is staged in the patch buffer");

#ifndef FAN_WAVES
#define FAN_WAVES 3            // waves per SIMD the register budget is set for (three 256-thread workgroups per CU)
#endif
typedef uint32_t u32q __attribute__((ext_vector_type(4)));
// ---- coalesced patch fetch: chunk = one 16-B piece; a patch row (12 texels) is one contiguous run of the table.
// `fast` = the patch lies inside the table (no clamping).  Byte offset of chunk `chunk` of a C-channel plane patch:
template <int C>
__device__ __forceinline__ const f32q* plane_chunk(const float* __restrict__ tab, int Ga, int Gb, int loa, int lob, bool fast, int chunk) {
    constexpr int CPT = C / 4, CPR = FP * CPT;
    if (fast) {
        const char* base = reinterpret_cast<const char*>(tab + ((size_t)lob * Ga + loa) * C);
        const int ry = chunk / CPR;
        return reinterpret_cast<const f32q*>(base + (unsigned)(ry * ((Ga - FP) * C * 4) + chunk * 16));
    }
    const int texel = chunk / CPT, q = chunk - texel * CPT;
    const int ry = texel / FP, rx = texel - ry * FP;
    const int row = min(lob + ry, Gb - 1), col = min(loa + rx, Ga - 1);
    return reinterpret_cast<const f32q*>(tab + ((size_t)row * Ga + col) * C + 4 * q);
}
template <int C>
__device__ __forceinline__ const f32q* line_chunk(const float* __restrict__ tab, int Gv, int lov, bool fast, int chunk) {
    constexpr int CPT = C / 4;
    const int rz = chunk / CPT, q = chunk - rz * CPT;
    const int row = fast ? lov + rz : min(lov + rz, Gv - 1);
    return reinterpret_cast<const f32q*>(tab + (size_t)row * C + 4 * q);
}

// float offset, inside a pitched plane patch, of the fetch order's chunk `chunk` (rows of FP texels, C / 4 chunks per texel)
template <int C>
__device__ __forceinline__ int pitched(int chunk) {
    constexpr int CPR = FP * (C / 4);
    return 4 * chunk + (chunk / CPR) * ((PITCH - FP) * C);
}

struct RecView {           // one sample record, unpacked (all lanes of a sub-group read the same record)
    float w;               // compositing weight (after phase B)
    bool valid;
    int r[3], d[3];        // tap index relative to the box, and 1 when the high tap is a different texel
    float wt[3][2];        // zero-padded tap weights per axis
};
__device__ __forceinline__ RecView unpack_rec(const u32q a, const u32q b) {
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
__device__ __forceinline__ RecView read_rec(const uint32_t* rec) {
    return unpack_rec(*reinterpret_cast<const u32q*>(rec), *reinterpret_cast<const u32q*>(rec + 4));
}

// MODE 2: the whole march of a tile (density, compositing, appearance, basis_mat) -> feature rows.  MODE 3: MODE 2 + the Ref head and
// the background blend (phase E): the tile leaves the kernel as colours, no feature rows and no second launch.
template <int MODE>
__global__ void __launch_bounds__(NT, FAN_WAVES) k4f_fan_march(FieldDev f, MarchArgs a, int64_t n_tiles) {
    // the patch buffer and the sample records are one pool: phase D lays the two operands of its matrix product over both
    __shared__ __align__(16) float s_pool[PATCH_FLOATS + FR * FS * REC];
    float* const s_patch = s_pool;
    float* const s_feat = s_pool + 4 * 32 * 32;        // [FR][28] output rows: written after phase D's partial tiles, behind them
    static_assert(4 * 32 * 32 + FR * 28 <= PATCH_FLOATS + FR * FS * REC, "output rows fit behind the partial tiles");
    uint32_t* const s_rec = reinterpret_cast<uint32_t*>(s_pool + PATCH_FLOATS);
    __shared__ float s_ray[FR * 8];
    __shared__ int s_box[8];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // lane -> (ray of the tile, sub-group, texel quarter).  The two quads of a ray -- sub-group h takes the samples s = h mod 2, so in a
    // trip they sit at consecutive samples, half a voxel apart -- are two quads of ONE ds_read_b128 lane group (MI355X_MICROARCH.md, LDS:
    // a wave's b128 read is served in the groups of quads {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}): with the row pitch above
    // their taps hit the same address or different bank quarters, never the same quarter at different addresses.  (tid >> 3 as the ray put
    // the quads of FOUR unrelated rays into a lane group: a quarter of the LDS cycles were their collisions.)
    const int qx = (tid >> 2) & 7, qpar = __popc(qx) & 1, qpos = qx >> 1;       // quad of the 32-lane half: its lane group, its place in it
    const int g = 8 * wave + 4 * ((tid >> 5) & 1) + 2 * qpar + (qpos >> 1), h = qpos & 1, c = tid & 3;
    const bool grp_on = g < FR;
    const int gg = grp_on ? g : 0;

    // one tile per workgroup (exact grid): inside a persistent tile loop LLVM hoists the per-lane patch offsets and table
    // descriptors out of the loop and spills them
    const int64_t ray0 = (int64_t)blockIdx.x * FR;
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
    STAMP(0);
    __syncthreads();
    // ---------------------------------------------------------------------------------------------------- phase 0: the box
    // x(s) is monotone in s along a ray (every operation of the position / normalisation chain is monotone and so is its
    // rounding), hence the taps of a ray's samples lie between the taps of its two end points.  The 54 end points sit in
    // wave 0: reduced across the wave by xor butterfly (same-address LDS atomics would serialise lane by lane).
    if (wave == 0) {
        int blo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, bhi[3] = {-1, -1, -1};
        const int rl = tid % FR;
        if (tid < 2 * FR && rl < n_live) {
            const float* sr = s_ray + rl * 8;
            const float z = z_of(f, 0, FS, 0.0f, tid < FR ? 0 : FS - 1);
            const float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
            float xn[3];
            field_normalize(f, p, xn);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const int G = f.grid[ax];
                const float x = unnorm(xn[ax], G);
                blo[ax] = 0; bhi[ax] = G - 1;          // NaN: the whole axis (forces the gather path)
                if (x == x) {
                    const int fl = (int)floorf(fminf(fmaxf(x, -1.0f), (float)G));
                    blo[ax] = min(max(fl, 0), G - 1);
                    bhi[ax] = min(max(fl + 1, 0), G - 1);
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                blo[ax] = min(blo[ax], __shfl_xor(blo[ax], off, 64));
                bhi[ax] = max(bhi[ax], __shfl_xor(bhi[ax], off, 64));
            }
        if (lane == 0) {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) { s_box[ax] = blo[ax]; s_box[3 + ax] = bhi[ax]; }
        }
    }
    __syncthreads();
    int lo[3];
    bool fits = true, inner = true;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        lo[ax] = __builtin_amdgcn_readfirstlane(s_box[ax]);
        const int hi = __builtin_amdgcn_readfirstlane(s_box[3 + ax]);
        fits = fits && (hi - lo[ax] + 1 <= FP);
        inner = inner && (lo[ax] + FP <= f.grid[ax]);
    }
    // ---------------------------------------------------------------------------------------------------- prefetch for phase B
    // density patches: 3 x 576 plane chunks = two full rounds per plane + one round in which wave w takes the last 64 chunks of
    // plane w; the three lines (48 chunks each) likewise by waves 0..2 in one round
    f32q pre[8];
    if (fits) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pa = mat_a(i), pb = mat_b(i);
#pragma unroll
            for (int r = 0; r < 2; ++r) pre[2 * i + r] = *plane_chunk<16>(f.dplane[i], f.grid[pa], f.grid[pb], lo[pa], lo[pb], inner, tid + NT * r);
        }
        if (wave < 3) {
            const int pa = mat_a(wave), pb = mat_b(wave), pv = vec_ax(wave);
            pre[6] = *plane_chunk<16>(f.dplane[wave], f.grid[pa], f.grid[pb], lo[pa], lo[pb], inner, 2 * NT + lane);
            if (lane < FP * 4) pre[7] = *line_chunk<16>(f.dline[wave], f.grid[pv], lo[pv], inner, lane);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(1);
    FAN_EXIT(1);
    // ---------------------------------------------------------------------------------------------------- phase A: records
    // (the 540 samples flat over the 256 threads -- two rounds for three of the four waves instead of three -- measured 1.3 % SLOWER
    // than this ray-major split, same box: the integer division and the per-sample ray reads cost more than the idle round)
    if (grp_on) {
        const float* sr = s_ray + g * 8;
        const bool live = g < n_live;
        const int l8 = 4 * h + c;
        // the occupancy bytes of the lane's (up to) three samples first (one byte each from the corner-bit table, iff_device.h
        // mask_occupied): three independent loads in flight together
        bool occ[3] = {true, true, true};
        if (f.mask) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int s = min(l8 + 8 * k, FS - 1);
                const float z = z_of(f, 0, FS, 0.0f, s);
                const float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                occ[k] = mask_occupied(f, p);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int s = l8 + 8 * k;
            if (s >= FS) break;
            const float z = z_of(f, 0, FS, 0.0f, s);
            const float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
            const bool inside = live && inside_aabb(f, p);
            u32q r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
            if (inside) {
                float xn[3];
                field_normalize(f, p, xn);
                uint32_t packed = occ[k] ? (1u << 15) : 0u;
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
            uint32_t* rec = s_rec + (g * FS + s) * REC;
            *reinterpret_cast<u32q*>(rec) = r0;
            *reinterpret_cast<u32q*>(rec + 4) = r1;
        }
    }
    STAMP(2);
    FAN_EXIT(2);
    if (fits) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 2; ++r) *reinterpret_cast<f32q*>(s_patch + i * PLANE16 + pitched<16>(tid + NT * r)) = pre[2 * i + r];
        if (wave < 3) {
            *reinterpret_cast<f32q*>(s_patch + wave * PLANE16 + pitched<16>(2 * NT + lane)) = pre[6];
            if (lane < FP * 4) *reinterpret_cast<f32q*>(s_patch + 3 * PLANE16 + wave * LINE16 + 4 * lane) = pre[7];
        }
    }
    __syncthreads();
    // appearance plane 0 on its way while phase B computes: 1728 plane chunks = 6 full rounds + 192, 144 line chunks
    auto fetch_app = [&](int i) {
        const int pa = mat_a(i), pb = mat_b(i), pv = vec_ax(i);
#pragma unroll
        for (int r = 0; r < 6; ++r) pre[r] = *plane_chunk<48>(f.aplane[i], f.grid[pa], f.grid[pb], lo[pa], lo[pb], inner, tid + NT * r);
        if (tid < FP * FP * 12 - 6 * NT) pre[6] = *plane_chunk<48>(f.aplane[i], f.grid[pa], f.grid[pb], lo[pa], lo[pb], inner, tid + NT * 6);
        if (tid < FP * 12) pre[7] = *line_chunk<48>(f.aline[i], f.grid[pv], lo[pv], inner, tid);
    };
    auto fetch_basis = [&]() {
#pragma unroll
        for (int r = 0; r < 3; ++r) pre[r] = *reinterpret_cast<const f32q*>(f.basis + 4 * (tid + NT * r));
        if (tid < BASIS_FLOATS / 4 - 3 * NT) pre[3] = *reinterpret_cast<const f32q*>(f.basis + 4 * (tid + NT * 3));
    };
    if (fits) fetch_app(0);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    FAN_EXIT(3);
    // ---------------------------------------------------------------------------------------------------- phase B: density
    const float* sr = s_ray + gg * 8;
    const int64_t r_glob = ray0 + g;
    const bool live = grp_on && g < n_live;
    uint32_t* recs = s_rec + gg * FS * REC;
    unsigned shmask = 0u;
    {
        // the 540 samples of the tile over the 64 four-lane sub-groups of the workgroup: sub-group q takes the samples t = q + 64 it
        // (t = 20 ray + s: the record index), its four lanes gather a sample together (one 16-B quarter of the density texel each)
        // and lane c finishes the sample of trip it = 4 k + c; the record of the next sample is read one trip ahead
        const int q4 = 16 * wave + 4 * (2 * ((tid >> 5) & 1) + qpar) + qpos;      // a lane group's four quads: four consecutive samples
        constexpr int NSMP = FR * FS, TRIPS = (NSMP + 63) / 64;
        u32q nra = *reinterpret_cast<const u32q*>(s_rec + q4 * REC), nrb = *reinterpret_cast<const u32q*>(s_rec + q4 * REC + 4);
#pragma unroll 1
        for (int k = 0; k < (TRIPS + 3) / 4; ++k) {
            float feat_mine = 0.0f;
            bool valid_mine = false;
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int it = 4 * k + j;
                if (it >= TRIPS) break;
                const int t = min(q4 + 64 * it, NSMP - 1);
                const RecView rv = unpack_rec(nra, nrb);
                {
                    const uint32_t* rp = s_rec + min(t + 64, NSMP - 1) * REC;
                    nra = *reinterpret_cast<const u32q*>(rp); nrb = *reinterpret_cast<const u32q*>(rp + 4);
                }
                float part = 0.0f;
                if (fits) {
                    // all 18 taps of the three planes in flight before the first one is used (left to itself the compiler takes
                    // them plane by plane: three LDS round trips in a row)
                    f32q tp[3][6];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
                        const float* P = s_patch + i * PLANE16 + ((rv.r[ax_b] * PITCH + rv.r[ax_a]) * 16 + 4 * c);
                        const int da = rv.d[ax_a] * 16, db = rv.d[ax_b] * (PITCH * 16);
                        tp[i][0] = *reinterpret_cast<const f32q*>(P); tp[i][1] = *reinterpret_cast<const f32q*>(P + da);
                        tp[i][2] = *reinterpret_cast<const f32q*>(P + db); tp[i][3] = *reinterpret_cast<const f32q*>(P + db + da);
                        const float* L = s_patch + 3 * PLANE16 + i * LINE16 + (rv.r[ax_v] * 16 + 4 * c);
                        tp[i][4] = *reinterpret_cast<const f32q*>(L); tp[i][5] = *reinterpret_cast<const f32q*>(L + rv.d[ax_v] * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
                        const float pw[4] = {rv.wt[ax_b][0] * rv.wt[ax_a][0], rv.wt[ax_b][0] * rv.wt[ax_a][1],
                                             rv.wt[ax_b][1] * rv.wt[ax_a][0], rv.wt[ax_b][1] * rv.wt[ax_a][1]};
                        const float lw[2] = {rv.wt[ax_v][0], rv.wt[ax_v][1]};
                        const f32q pq = lerp_plane_q(tp[i][0], tp[i][1], tp[i][2], tp[i][3], pw), lq = lerp_line_q(tp[i][4], tp[i][5], lw);
                        part = fmaf(pq.w, lq.w, fmaf(pq.z, lq.z, fmaf(pq.y, lq.y, fmaf(pq.x, lq.x, part))));
                    }
                } else if (rv.valid) {
                    const int ry = t / FS, s = t - ry * FS;
                    const float* srr = s_ray + ry * 8;
                    const float z = z_of(f, 0, FS, 0.0f, s);
                    const float p[3] = {srr[0] + srr[3] * z, srr[1] + srr[4] * z, srr[2] + srr[5] * z};
                    float xn[3];
                    field_normalize(f, p, xn);
                    part = density_partial(f, xn, c);
                }
                const float feat = sum4_dpp(rv.valid ? part : 0.0f);
                if (j == c) { feat_mine = feat; valid_mine = rv.valid; }
            }
            // lane c finishes the sample of trip 4k + c: sigma and alpha (tensorBase.py:25,849)
            const int t = q4 + 64 * (4 * k + c);
            const int s = t % FS;
            const float sigma = valid_mine ? feature2density(f, feat_mine) : 0.0f;
            const float zs = z_of(f, 0, FS, 0.0f, s);
            const float dist = (s + 1 < FS) ? (z_of(f, 0, FS, 0.0f, s + 1) - zs) : 0.0f;       // tensorBase.py:800-803
            const float alpha = 1.0f - expf(-sigma * (dist * f.distance_scale));
            if (4 * k + c < TRIPS && t < NSMP) s_rec[t * REC] = __float_as_uint(alpha);
        }
        STAMP(4);
        __syncthreads();
        // the transmittance product of the ray (tensorBase.py:27-32), all eight lanes alike
        float run_T = 1.0f, run_acc = 0.0f, run_depth = 0.0f;
        int run_valid = 0, run_app = 0;
        const bool writer = live && h == 0 && c == 0;
        typedef uint32_t u32d __attribute__((ext_vector_type(2)));
        u32d av[FS];                                   // (alpha, packed) of every sample: all reads in flight before the chain
#pragma unroll
        for (int s = 0; s < FS; ++s) av[s] = *reinterpret_cast<const u32d*>(recs + s * REC);
#pragma unroll
        for (int s = 0; s < FS; ++s) {
            const float alpha = __uint_as_float(av[s].x);
            const bool valid = (av[s].y >> 15) & 1u;
            const float z = z_of(f, 0, FS, 0.0f, s);
            const float w = alpha * run_T;
            run_T = run_T * ((1.0f - alpha) + 1e-10f);
            run_acc += w;
            run_depth += w * z;
            run_valid += valid ? 1 : 0;
            const bool sh = w > f.weight_thres;                                     // tensorBase.py:851
            run_app += sh ? 1 : 0;
            shmask |= (sh ? 1u : 0u) << s;
            if (writer) {
                recs[s * REC] = __float_as_uint(w);
#ifndef FAN_STAMPS
                if (a.alpha) a.alpha[r_glob * FS + s] = alpha;
#endif
            }
        }
        if (writer) {
            if (MODE == 3) s_ray[g * 8 + 6] = run_acc;          // phase E blends with it
            a.acc[r_glob] = run_acc;
            a.depth[r_glob] = run_depth + (1.0f - run_acc) * sr[7];
            if (a.counts) { a.counts[r_glob * 2] = run_valid; a.counts[r_glob * 2 + 1] = run_app; }
        }
    }
    STAMP(5);
    FAN_EXIT(5);
    const int gr = gg;          // (the ray a group serves in phases C / D: a tile-level order by shaded count was tried, NOTES.md)
    if (!live) shmask = 0u;
    const bool any = shmask != 0u;
    const unsigned mymask = shmask & (h ? 0xAAAAAu : 0x55555u);       // this sub-group's shaded samples
    // ---------------------------------------------------------------------------------------------------- phase C: appearance
    float accp[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) accp[i] = 0.0f;
    if (fits) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __syncthreads();                              // phase B / the previous plane is done with the patch buffer
#pragma unroll
            for (int r = 0; r < 6; ++r) *reinterpret_cast<f32q*>(s_patch + pitched<48>(tid + NT * r)) = pre[r];
            if (tid < FP * FP * 12 - 6 * NT) *reinterpret_cast<f32q*>(s_patch + pitched<48>(tid + NT * 6)) = pre[6];
            if (tid < FP * 12) *reinterpret_cast<f32q*>(s_patch + PLANE48 + 4 * tid) = pre[7];
            __syncthreads();
            if (i < 2) fetch_app(i + 1);                  // the next plane (registers) under this plane's arithmetic
            else fetch_basis();                           // basis_mat for phase D
            __builtin_amdgcn_sched_barrier(0);
            STAMP(6 + 2 * i);
            const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
            {
            float (&acc)[36] = accp;
            unsigned m = mymask;
            // the record of the next sample is read one trip ahead
            u32q na = {0u, 0u, 0u, 0u}, nb = {0u, 0u, 0u, 0u};
            if (m) {
                const uint32_t* rp = recs + (__ffs((int)m) - 1) * REC;
                na = *reinterpret_cast<const u32q*>(rp); nb = *reinterpret_cast<const u32q*>(rp + 4);
            }
            while (m) {
                m &= m - 1u;
                const RecView rv = unpack_rec(na, nb);
                if (m) {
                    const uint32_t* rp = recs + (__ffs((int)m) - 1) * REC;
                    na = *reinterpret_cast<const u32q*>(rp); nb = *reinterpret_cast<const u32q*>(rp + 4);
                }
                const float* P = s_patch + ((rv.r[ax_b] * PITCH + rv.r[ax_a]) * 48 + 4 * c);
                const int da = rv.d[ax_a] * 48, db = rv.d[ax_b] * (PITCH * 48);
                const float* L = s_patch + PLANE48 + (rv.r[ax_v] * 48 + 4 * c);
                const int dv = rv.d[ax_v] * 48;
                const float pw[4] = {rv.wt[ax_b][0] * rv.wt[ax_a][0], rv.wt[ax_b][0] * rv.wt[ax_a][1],
                                     rv.wt[ax_b][1] * rv.wt[ax_a][0], rv.wt[ax_b][1] * rv.wt[ax_a][1]};
                // the compositing weight rides on the line taps' weights (two multiplies per sample): w * (plane * line) becomes plane * (w line),
                // one fma per channel instead of a multiply and an fma -- 10 of a trip's ~122 vector instructions.  The weighted sums move by
                // one rounding per term (~1e-7 relative): they are no longer the bits the general kernels' "w * (plane * line)" returns (alpha,
                // acc, depth and the counters, which do not pass through here, still are)
                const float lw[2] = {rv.w * rv.wt[ax_v][0], rv.w * rv.wt[ax_v][1]};
#pragma unroll
                for (int j = 0; j < 3; ++j) {             // quarter c + 4 j of the 192-B texel
                    const f32q nw = *reinterpret_cast<const f32q*>(P + 16 * j), ne = *reinterpret_cast<const f32q*>(P + 16 * j + da);
                    const f32q sw = *reinterpret_cast<const f32q*>(P + 16 * j + db), se = *reinterpret_cast<const f32q*>(P + 16 * j + db + da);
                    const f32q ll = *reinterpret_cast<const f32q*>(L + 16 * j), lh = *reinterpret_cast<const f32q*>(L + 16 * j + dv);
                    const f32q pl = lerp_plane_q(nw, ne, sw, se, pw), ln = lerp_line_q(ll, lh, lw);
                    acc[12 * j + 4 * i + 0] = fmaf(pl.x, ln.x, acc[12 * j + 4 * i + 0]);
                    acc[12 * j + 4 * i + 1] = fmaf(pl.y, ln.y, acc[12 * j + 4 * i + 1]);
                    acc[12 * j + 4 * i + 2] = fmaf(pl.z, ln.z, acc[12 * j + 4 * i + 2]);
                    acc[12 * j + 4 * i + 3] = fmaf(pl.w, ln.w, acc[12 * j + 4 * i + 3]);
                }
            }
            }
            STAMP(7 + 2 * i);
            FAN_EXIT(7 + 2 * i);
        }
    } else {
        // the gather path: a tile whose samples do not fit one patch reads its taps where the general kernels do
        unsigned m = mymask;
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
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fetch_basis();
    }
    // ---------------------------------------------------------------------------------------------------- phase D: basis_mat
    // F[ray][o] = sum_k basis_mat[o][k] * A[ray][k] over the 144 weighted products (k = 48 plane + channel, the column order of
    // basis_mat): one 32 x 32 x 144 product per tile on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain),
    // k split over the four waves, the four partial tiles added in a fixed order -- the vector ALU only moves operands.
    constexpr int DLD = 33;                            // operand rows padded: conflict-free ds_read_b32 down a column of k
    float* const s_A = s_pool;                         // [144][DLD]  A^T: weighted products, column = ray of the tile
    float* const s_B = s_pool + 144 * DLD;             // [144][DLD]  basis_mat^T, column = output feature
    static_assert(2 * 144 * DLD <= PATCH_FLOATS + FR * FS * REC && 4 * 32 * 32 <= PATCH_FLOATS + FR * FS * REC, "phase D operands fit the pool");
    // even samples + odd samples: both sub-groups of a ray hold the ray's sums afterwards
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        // the ray's other sub-group sits one quad to the left for quads 0, 2, 4, 6 of the half (row_ror 4), one to the right for the others
        const float a4 = dpp_mov<0x124>(accp[i]), a12 = dpp_mov<0x12C>(accp[i]);
        accp[i] = accp[i] + ((qx & 1) == 0 ? a4 : a12);
    }
    __syncthreads();                                   // every wave is done with the patch and the records
    if (grp_on && h == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) s_A[(48 * i + 16 * j + 4 * c + e4) * DLD + gr] = accp[12 * j + 4 * i + e4];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int chunk = tid + NT * r;
        if (chunk < BASIS_FLOATS / 4) {
            const int o = chunk / 36, k4 = (chunk - o * 36) * 4;
            s_B[(k4 + 0) * DLD + o] = pre[r].x; s_B[(k4 + 1) * DLD + o] = pre[r].y;
            s_B[(k4 + 2) * DLD + o] = pre[r].z; s_B[(k4 + 3) * DLD + o] = pre[r].w;
        }
    }
    f32q bias_pre[4], stage_pre = splat(0.0f);
    if (MODE == 3) {
        // the two head slices phase E keeps in LDS (from spec_w on: spec_w, spec_b, ide_mat; up to bott_w: the small heads), 16 B per thread
        const HeadOff hq = head_offsets(f.app_dim, f.feature_c);
        const int n_tail4 = (hq.total - hq.spec_w) / 4, n_small4 = hq.bott_w / 4;
        if (tid < n_tail4) stage_pre = *reinterpret_cast<const f32q*>(f.head + hq.spec_w + 4 * tid);
        else if (tid < n_tail4 + n_small4) stage_pre = *reinterpret_cast<const f32q*>(f.head + 4 * (tid - n_tail4));
    }
    if (MODE == 3 && 32 * wave < f.feature_c) {
        // phase E's matrix operand (this wave's 32 bottleneck rows, one row per lane) and biases: in flight through phase D
        const HeadOff hq = head_offsets(f.app_dim, f.feature_c);
        const float* wr = f.head + hq.bott_w + (32 * wave + (lane & 31)) * 28;
#pragma unroll
        for (int k4 = 0; k4 < 7; ++k4) pre[k4] = *reinterpret_cast<const f32q*>(wr + 4 * k4);
#pragma unroll
        for (int q = 0; q < 4; ++q) bias_pre[q] = *reinterpret_cast<const f32q*>(f.head + hq.bott_b + 32 * wave + 8 * q + 4 * (lane >> 5));
    }
    __syncthreads();
    STAMP(12);
    FAN_EXIT(12);
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 dacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[r] = 0.0f;
    {
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int t = 0; t < 18; ++t) {                 // this wave's 36 values of k, two per instruction
            const int k = 36 * wave + 2 * t + lh;
            dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_A[k * DLD + lr], s_B[k * DLD + lr], dacc, 0, 0, 0);
        }
        __syncthreads();                               // the operands have been read: the partial tiles go over them
#pragma unroll
        for (int r = 0; r < 16; ++r) s_pool[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = dacc[r];
    }
    __syncthreads();
    for (int idx = tid; idx < FR * 27; idx += NT) {
        const int ry = idx / 27, o = idx - ry * 27;
        const float* pp = s_pool + ry * 32 + o;
        s_feat[ry * 28 + o] = (pp[0] + pp[32 * 32]) + (pp[2 * 32 * 32] + pp[3 * 32 * 32]);
    }
    if (h == 0 && c == 0 && grp_on) s_feat[gr * 28 + 27] = any ? 1.0f : 0.0f;
    if (MODE == 3 && tid < 5 * 28) s_feat[FR * 28 + tid] = 0.0f;             // rows 27..31 of the matrix operand
    STAMP(13);
    FAN_EXIT(13);
    __syncthreads();
    if (MODE != 3) {
        if (tid < n_live * 7)
            *reinterpret_cast<f32q*>(a.feat + ray0 * 28 + 4 * tid) = *reinterpret_cast<const f32q*>(s_feat + 4 * tid);
        STAMP(14);
        return;
    }
    // ---------------------------------------------------------------------------------------------------- phase E: the Ref head
    // (models/ref.py:103-152; the separate launch k_ref_shade spends 16 lanes per ray on it.)  The bottleneck (feature_c rows over
    // the ray's 27 features) runs as W[32 rows][28] x F^T[28][32 rays] tiles on the fp32 matrix cores (the same k-ordered fmaf
    // chain as the vector code: identical bits), one 32-row block per wave; four lanes per ray then finish it (ref_head_quad).
    constexpr int BLD = 164;                           // floats per ray in s_b: 160 rows + 4 (16-B aligned rows, spread over the banks)
    float* const s_b = s_pool + 5120;                  // behind the output rows
    static_assert(4 * 32 * 32 + 32 * 28 <= 5120 && 5120 + 32 * BLD + 680 + 296 <= PATCH_FLOATS + FR * FS * REC, "phase E operands fit the pool");
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    const int fc = f.feature_c;
    float* const s_tail = s_b + 32 * BLD;              // the head from spec_w on (spec_w, spec_b, ide_mat) ...
    float* const s_small = s_tail + 680;               // ... and up to bott_w (the small heads): what ref_head_quad reads per lane
    {
        const int n_tail4 = (ho.total - ho.spec_w) / 4, n_small4 = ho.bott_w / 4;         // <= 170 + 74 <= NT (fan_head_fusable)
        if (tid < n_tail4) *reinterpret_cast<f32q*>(s_tail + 4 * tid) = stage_pre;
        else if (tid < n_tail4 + n_small4) *reinterpret_cast<f32q*>(s_small + 4 * (tid - n_tail4)) = stage_pre;
    }
    {
        const int lr = lane & 31, lh = lane >> 5;
        if (32 * wave < fc) {                          // feature_c <= 128: one block per wave
            const int blk = wave;
            const f32q (&w)[8] = pre;
            f32x16 e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 14; ++ks) {
                const float av = lh ? w[ks >> 1][2 * (ks & 1) + 1] : w[ks >> 1][2 * (ks & 1)];
                float bv = s_feat[lr * 28 + 2 * ks + lh];
                if (ks == 13) bv = lh ? 0.0f : bv;                             // column 27 is the shaded flag, not a feature
                e = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, e, 0, 0, 0);
            }
            // rows 32 blk + 8 q + 4 lh + i (register 4 q + i) of ray lr, plus the bias
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r0 = 32 * blk + 8 * q + 4 * lh;
                const f32q bias = bias_pre[q];
                f32q o4 = {e[4 * q] + bias[0], e[4 * q + 1] + bias[1], e[4 * q + 2] + bias[2], e[4 * q + 3] + bias[3]};
                *reinterpret_cast<f32q*>(s_b + lr * BLD + r0) = o4;
            }
        }
    }
    ESTAMP(6);
    __syncthreads();
    ESTAMP(7);
    // four lanes per ray, 32 ray slots = 128 threads = two of the four waves: which two rotates with the tile (a tile's wave w always
    // lands on the CU's w-th SIMD: the head's instructions would otherwise pile up on two SIMDs)
    const int ew = (wave - (int)(blockIdx.x & 3u)) & 3;         // 0, 1: the two waves that run the head
    if (ew < 2) {
        const int eg = 16 * ew + (lane >> 2), sub = lane & 3;  // ray slot of the tile, lane of the quad
        const float* sr = s_ray + (eg < FR ? eg : 0) * 8;
        const float d[3] = {sr[3], sr[4], sr[5]};
        const float cch = ref_head_quad(s_small, ho, fc, s_b + eg * BLD, s_feat + eg * 28, s_tail, d, sub);
        if (sub < 3 && eg < n_live) {
            const bool shaded = s_feat[eg * 28 + 27] != 0.0f;
            const float acc = sr[6];
            float v = shaded ? cch : 0.0f;
            v = v * acc + (sub == 0 ? a.bg[0] : (sub == 1 ? a.bg[1] : a.bg[2])) * (1.0f - acc);
            a.rgb[3 * (ray0 + eg) + sub] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
    }
    STAMP(14);
}

// Ref.forward (models/ref.py:103-152) of n rays with given feature rows and view directions -- the march's shade + blend step on
// tiles the fused kernel does not take, renderer.evaluation's rows, iff_ref_shade -- in the form of the fused kernel's phase E:
// 32 rays per tile, the bottleneck on the fp32 matrix cores (one 32-row block per wave, its weights held in registers over the
// workgroup's tiles), eight lanes per ray for the rest (ref_head_oct).  Same bits as k_ref_shade (field_kernels.hip), which spends
// 16 lanes per ray and the vector ALU on the bottleneck.
template <bool BLEND>
__global__ void __launch_bounds__(NT) k_ref_shade_oct(FieldDev f, const float* __restrict__ dirs, int dir_stride, const float* __restrict__ feat,
                                                      int feat_stride, const float* __restrict__ acc, float bg0, float bg1, float bg2,
                                                      int64_t n, float* __restrict__ rgb, int64_t n_tiles) {
    constexpr int BLD = 164;
    __shared__ __align__(16) float s_F[32 * 28];
    __shared__ __align__(16) float s_b[32 * BLD];
    __shared__ __align__(16) float s_tail[680];
    __shared__ __align__(16) float s_small[296];
    __shared__ float s_flag[32];
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    const int fc = f.feature_c;
    {
        const int n_tail4 = (ho.total - ho.spec_w) / 4, n_small4 = ho.bott_w / 4;         // <= 170 + 74 <= NT (fan_head_fusable)
        if (tid < n_tail4) *reinterpret_cast<f32q*>(s_tail + 4 * tid) = *reinterpret_cast<const f32q*>(f.head + ho.spec_w + 4 * tid);
        else if (tid < n_tail4 + n_small4) *reinterpret_cast<f32q*>(s_small + 4 * (tid - n_tail4)) = *reinterpret_cast<const f32q*>(f.head + 4 * (tid - n_tail4));
    }
    const bool has_block = 32 * wave < fc;
    f32q w[7], bias_q[4];
    if (has_block) {
        const float* wr = f.head + ho.bott_w + (32 * wave + lr) * 28;
#pragma unroll
        for (int k4 = 0; k4 < 7; ++k4) w[k4] = *reinterpret_cast<const f32q*>(wr + 4 * k4);
#pragma unroll
        for (int q = 0; q < 4; ++q) bias_q[q] = *reinterpret_cast<const f32q*>(f.head + ho.bott_b + 32 * wave + 8 * q + 4 * lh);
    }
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t ray0 = tile * 32;
        const int n_live = (int)(n - ray0 < 32 ? n - ray0 : 32);
        __syncthreads();                               // the staged head is in place / the previous tile has been read
        for (int idx = tid; idx < 32 * 28; idx += NT) {
            const int r = idx / 28, k = idx - r * 28;
            float v = 0.0f;
            if (r < n_live) {
                const float* fp = feat + (ray0 + r) * feat_stride;
                if (k < 27) v = fp[k];
                else if (BLEND) s_flag[r] = fp[27];    // the shaded flag of the march's 28-float rows
            }
            s_F[idx] = v;                              // column 27 and the rows past n are zero
        }
        __syncthreads();
        if (has_block) {
            f32x16 e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 14; ++ks) {
                const float av = lh ? w[ks >> 1][2 * (ks & 1) + 1] : w[ks >> 1][2 * (ks & 1)];
                e = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s_F[lr * 28 + 2 * ks + lh], e, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {              // rows 32 wave + 8 q + 4 lh + i (register 4 q + i) of ray lr, plus the bias
                const f32q b4 = bias_q[q];
                f32q o4 = {e[4 * q] + b4[0], e[4 * q + 1] + b4[1], e[4 * q + 2] + b4[2], e[4 * q + 3] + b4[3]};
                *reinterpret_cast<f32q*>(s_b + lr * BLD + 32 * wave + 8 * q + 4 * lh) = o4;
            }
        }
        __syncthreads();
        const int g = tid >> 3, sub = tid & 7;
        const bool live = g < n_live;
        const float* dp = dirs + (live ? ray0 + g : 0) * dir_stride;
        const float d[3] = {dp[0], dp[1], dp[2]};
        const float cch = ref_head_oct(s_small, ho, fc, s_b + g * BLD, s_F + g * 28, s_tail, d, sub);
        if (sub < 3 && live) {
            if (BLEND) {
                const float ac = acc[ray0 + g];
                float v = s_flag[g] != 0.0f ? cch : 0.0f;
                v = v * ac + (sub == 0 ? bg0 : (sub == 1 ? bg1 : bg2)) * (1.0f - ac);
                rgb[3 * (ray0 + g) + sub] = fminf(fmaxf(v, 0.0f), 1.0f);
            } else {
                rgb[3 * (ray0 + g) + sub] = cch;
            }
        }
    }
}

}  // namespace

// The fused fan kernel serves the point-centred sampler with its 20 samples (the emission path) on tables of the reference's
// shapes; a field whose ten steps cover many more than 5 texels of an axis (the reference's step_ratio 0.5 gives 5) would send
// most tiles down the gather path, so it keeps the general kernels.  A performance choice only: the per-tile box test decides what
// the kernel does.  Unisphere contraction (utils.py:139-146, applied per axis) is monotone with slope <= 1, so a fan's box is at
// most that of the uncontracted step and the end-point argument of phase 0 holds unchanged.
bool fan_march_eligible(const FieldDev& f, int mode, int S) {
    if (mode != 0 || S != FS || f.n_density != 16 || f.n_app != 48 || f.app_dim != 27) return false;
    for (int ax = 0; ax < 3; ++ax) {
        const float scale = f.unisphere ? 1.0f : f.inv_aabb[ax];            // d(normalised coordinate) / d(world coordinate), at most
        const float texels = 10.0f * f.step_size * scale * 0.5f * (float)(f.grid[ax] - 1);
        if (!(texels <= 6.5f)) return false;
    }
    return true;
}

// the fused Ref head (phase E) is laid out for the reference's head: 27 features in rows of 28, a bottleneck of at most 128 rows
bool fan_head_fusable(const FieldDev& f) {
    return f.app_dim == 27 && f.feature_c >= 32 && f.feature_c <= 128 && f.feature_c % 32 == 0 && f.head != nullptr;
}

// k_ref_shade's work in the 8-lanes-per-ray form (acc == nullptr: plain Ref.forward; otherwise the march's shade + blend)
hipError_t launch_ref_shade_oct(const FieldDev& f, const float* dirs, int dir_stride, const float* feat, int feat_stride,
                                const float* acc, const float* bg, int64_t n, float* rgb, hipStream_t s) {
    const int64_t n_tiles = (n + 31) / 32;
    if (n_tiles == 0) return hipSuccess;
    const int64_t grid = n_tiles < 256 * 4 ? n_tiles : 256 * 4;
    if (acc) hipLaunchKernelGGL((k_ref_shade_oct<true>), dim3((unsigned)grid), dim3(NT), 0, s, f, dirs, dir_stride, feat, feat_stride, acc,
                                bg[0], bg[1], bg[2], n, rgb, n_tiles);
    else hipLaunchKernelGGL((k_ref_shade_oct<false>), dim3((unsigned)grid), dim3(NT), 0, s, f, dirs, dir_stride, feat, feat_stride, acc,
                            0.0f, 0.0f, 0.0f, n, rgb, n_tiles);
    return hipGetLastError();
}

hipError_t launch_fan_march(const FieldDev& f, const MarchArgs& a, int variant, hipStream_t s) {
    const int64_t n_tiles = (a.R + FR - 1) / FR;
    if (n_tiles == 0) return hipSuccess;
    if (n_tiles > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t grid = n_tiles;
    if (variant == 3) hipLaunchKernelGGL((k4f_fan_march<3>), dim3((unsigned)grid), dim3(NT), 0, s, f, a, n_tiles);
    else hipLaunchKernelGGL((k4f_fan_march<2>), dim3((unsigned)grid), dim3(NT), 0, s, f, a, n_tiles);
    return hipGetLastError();
}
