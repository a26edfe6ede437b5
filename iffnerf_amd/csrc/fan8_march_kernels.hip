// fan8_march_kernels.hip -- K4g: the fused fan march (TensorBase.forward, models/tensorBase.py:775-917, with the point-centred sampler
// sample_point_color, :623-638, on the 27-ray iso-cell fans of pose_estimation/sampling.py:442-488) as ONE 512-thread workgroup per
// fan whose table patches arrive by global -> LDS DMA into two alternating buffers.  One kernel family for both box sizes:
//
//   <FP = 12, NSL = 1>  aabb scenes (lego, truck): a fan's 540 samples span <= 12 texels per axis; the three density patches form one
//                       pass, each appearance plane (48 channels) one pass -- 4 passes per fan
//   <FP = 22, NSL = 3>  unisphere scenes (mip360 bicycle: tensorBase.py:361-365 halves the grid units in the step, a fan spans <= 21
//                       texels): every pass stages ONE 16-channel slice of one plane -- 3 density passes (the per-sample partial sum
//                       is carried in registers in the lookup functions' order, plane 0, 1, 2) and 9 appearance passes
//
// against fan_march_kernels.hip (four waves per fan, patches prefetched in registers, 12-texel boxes only):
//   * a ray is served by SIXTEEN lanes -- four quads, quad p takes the samples s = p mod 4, lane c of a quad one 16-B quarter of a
//     16-channel texel slice -- and the four quads of a ray are the four quads of ONE ds_read_b128 lane group
//     (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): in every trip a lane group reads the taps of four
//     CONSECUTIVE samples of one ray, half a voxel apart.  With a patch row pitch of 2 mod 4 texels the 64-B quarter of a tap is
//     (i_a + 2 i_b + j) mod 4, so neighbouring cells (0 / +-1 in either axis, both diagonals) never share a bank quarter at different
//     addresses and equal cells broadcast: the random quarter collisions between unrelated rays' quads (a quarter of the four-wave
//     kernel's LDS cycles) are gone by construction, what is left are the rare two-texel jumps inside four samples;
//   * the next pass's patch is requested by global_load_lds_dwordx4 (no prefetch registers, no ds_write pass: <= 128 VGPRs, two
//     fans of eight waves per CU) right after the barrier that frees its buffer, and lands under the current pass's arithmetic;
//   * 16-byte sample records (packed tap indices + the three fractional weights; the low weight is 1 - frac, the subtraction
//     axis_tap does; zero-padded taps are flagged and restored under a branch only table-face samples take);
//   * the fan's box is reduced by every wave for itself (DPP + readlane, no LDS, no barrier), so the first DMA is in flight before
//     the records are built.
// Per-sample arithmetic is the lookup functions' (iff_device.h: shared lerp order), so alpha / acc / depth / the sample counters are
// bit-identical to the general kernels and to the four-wave kernel; the weighted feature sums are added per ray as
// (p0 + p1) + (p2 + p3) over the four quads' sample chains, basis_mat and the Ref head (phases D, E) are the four-wave kernel's.
// A fan whose box does not fit (arbitrary rays) gathers its taps from global memory in the same workgroup, same arithmetic.
#include "iff_device.h"
#include "iff_launch.h"
#include "march_common.h"
#include "fan_common.h"
#include "fan_diag.h"

namespace {

constexpr int FR = 27;          // rays per tile = one iso-cell fan (pose_estimation/isocell.py:6-68)
constexpr int FS = 20;          // samples per ray (pose_estimation/sampling.py:247)
constexpr int NT = 512;         // threads: 32 ray slots of 16 lanes
constexpr int NSMP = FR * FS;   // 540 samples per tile
typedef uint32_t u32q __attribute__((ext_vector_type(4)));

// record word 0: per axis 8 bits at 8 ax: [0:4] low tap index relative to the box, [5] high tap is the next texel, [6] / [7] the low /
// high tap lies outside the table (grid_sample's zero padding); then
constexpr uint32_t REC_VALID = 1u << 24;       // the sample is inside the aabb and its mask cell is occupied
constexpr uint32_t REC_PAD = 1u << 25;         // some tap of the sample is zero-padded (table faces only)

template <int FP_, int NSL_>
struct Geo {
    static constexpr int FP = FP_, NSL = NSL_;
    static constexpr int PITCH = (FP % 4 == 2) ? FP : FP + 2;        // texels per patch row: 2 mod 4 (header)
    static constexpr int SLC = 48 / NSL;                              // appearance channels per pass
    static constexpr int QPP = SLC / 16;                              // 16-channel quarters per pass
    static constexpr int NDP = (NSL == 1) ? 1 : 3;                    // density passes
    static constexpr int NP = NDP + 3 * NSL;                          // passes per fan
    static constexpr int PLANE = FP * PITCH * SLC, LINE = FP * SLC;   // floats of an appearance pass
    static constexpr int DPLANE = FP * PITCH * 16, DLINE = FP * 16;   // floats of one density plane / line
    static constexpr int BUF = PLANE + LINE;                          // floats per buffer
    static constexpr int NCH = BUF / 4;                               // 16-B chunks per pass
    static constexpr int NR = (NCH + NT - 1) / NT;                    // DMA rounds per pass (one chunk per thread and round)
    static_assert(NDP == 3 || 3 * (DPLANE + DLINE) == BUF, "the three density patches fill one buffer exactly");
    static_assert(NDP == 1 || DPLANE + DLINE == BUF, "a density plane slice is an appearance slice");
    static constexpr int POOL = 2 * BUF + NSMP * 4 + NSMP;            // two buffers, 16-B records, one weight per sample
};

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
// minimum / maximum over the wave, wave-uniform result: quad, half row, row by DPP, the four rows by readlane
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, dpp_i<0xB1>(v)); v = max(v, dpp_i<0x4E>(v)); v = max(v, dpp_i<0x141>(v)); v = max(v, dpp_i<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// one 16-B piece per lane, global -> LDS: the 64 lanes of a wave write 1 KiB contiguously from `lds_wave_base` (wave-uniform)
__device__ __forceinline__ void dma16(const void* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct RecView {
    bool valid;
    int r[3], d[3];
    float wt[3][2];
};
__device__ __forceinline__ RecView unpack_rec(const u32q a) {
    RecView v;
    v.valid = (a.x & REC_VALID) != 0u;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        v.r[ax] = (a.x >> (8 * ax)) & 31u;
        v.d[ax] = (a.x >> (8 * ax + 5)) & 1u;
    }
    v.wt[0][1] = __uint_as_float(a.y); v.wt[1][1] = __uint_as_float(a.z); v.wt[2][1] = __uint_as_float(a.w);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) v.wt[ax][0] = 1.0f - v.wt[ax][1];          // axis_tap: w0 = 1 - w1
    if (a.x & REC_PAD) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            if ((a.x >> (8 * ax + 6)) & 1u) v.wt[ax][0] = 0.0f;
            if ((a.x >> (8 * ax + 7)) & 1u) v.wt[ax][1] = 0.0f;
        }
    }
    return v;
}

// MODE 2: density, compositing, appearance, basis_mat -> feature rows.  MODE 3: + the Ref head and the background blend.
template <int FP, int NSL, int MODE>
__global__ void __launch_bounds__(NT, 4) k4g_fan_march(FieldDev f, MarchArgs a) {
    typedef Geo<FP, NSL> G;
    constexpr int PITCH = G::PITCH, SLC = G::SLC, QPP = G::QPP, NDP = G::NDP, BUF = G::BUF;
    __shared__ __align__(16) float s_pool[G::POOL];
    __shared__ float s_ray[32 * 8];
    float* const s_buf0 = s_pool;
    float* const s_buf1 = s_pool + BUF;
    uint32_t* const s_rec = reinterpret_cast<uint32_t*>(s_pool + 2 * BUF);        // [NSMP] 16-B records
    float* const s_w = s_pool + 2 * BUF + 4 * NSMP;                                // [NSMP] alpha, then the compositing weight
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // lane -> (ray slot, quad of the ray, texel quarter): the four quads of a ray are one ds_read_b128 lane group
    const int q = lane >> 2, c = lane & 3;
    const int gi = 2 * (q >> 3) + (__popc(q & 7) & 1), p = (q & 7) >> 1;
    const int rs = 4 * wave + gi;                       // ray slot 0..31 (27 in use)
    const bool slot_on = rs < FR;
    const int64_t ray0 = (int64_t)blockIdx.x * FR;
    const int n_live = (int)min((int64_t)FR, a.R - ray0);
    __shared__ int s_box[8];
    __shared__ uint32_t s_sh[32];                       // shaded-sample mask of every ray slot (written by the compositing wave)
    // the rays to LDS for the later phases (compositing, gather path, head): one thread per ray
    if (tid < 32) {
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
    // ---------------------------------------------------------------------------------------------------- phase A, first half
    // the 540 samples flat over the 512 threads: thread t takes sample t (ray t / 20), and the last 28 samples go to lanes 0 .. 27 of
    // wave 7 (wave 0 reduces the box meanwhile).  Everything of a record that does not need the box is computed here, in registers.
    struct Pre { uint32_t flags; int t0[3]; float w1[3]; bool any; };
    auto sample_pre = [&](int t) {
        Pre r;
        r.flags = 0u; r.any = false;
        r.t0[0] = r.t0[1] = r.t0[2] = 0; r.w1[0] = r.w1[1] = r.w1[2] = 0.0f;
        const int ry = t / FS, s = t - ry * FS;
        if (ry < n_live) {
            const float* rp = a.rays + (ray0 + ry) * a.ray_cols;
            const float z = z_of(f, 0, FS, 0.0f, s);
            const float pt[3] = {rp[0] + rp[3] * z, rp[1] + rp[4] * z, rp[2] + rp[5] * z};
            if (inside_aabb(f, pt)) {
                float xn[3];
                field_normalize(f, pt, xn);
                const bool occ = f.mask ? mask_occupied(f, pt, xn) : true;
                r.any = true;
                r.flags = occ ? REC_VALID : 0u;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    // axis_tap (iff_device.h), with the zero padding kept as flags
                    const int size = f.grid[ax];
                    const float x = unnorm(xn[ax], size);
                    const float fx = floorf(x);
                    r.w1[ax] = x - fx;
                    const bool ok = (x > -2.0f) && (x < (float)(size + 1));
                    const int i0 = ok ? (int)fx : -2, i1 = i0 + 1;
                    const int t0 = min(max(i0, 0), size - 1), t1 = min(max(i1, 0), size - 1);
                    const bool z0 = !(i0 >= 0 && i0 < size), z1 = !(i1 >= 0 && i1 < size);
                    r.t0[ax] = t0;
                    r.flags |= (uint32_t)(((t1 - t0) << 5) | (z0 ? 64 : 0) | (z1 ? 128 : 0)) << (8 * ax);
                    if (z0 || z1) r.flags |= REC_PAD;
                }
            }
        }
        return r;
    };
    const Pre pre0 = sample_pre(tid);
    Pre pre1;
    pre1.any = false;
    if (wave == 7 && lane < NSMP - NT) pre1 = sample_pre(NT + lane);
    // ---------------------------------------------------------------------------------------------------- phase 0: the box
    // x(s) is monotone in s along a ray, hence the taps of a ray's samples lie between the taps of its two end points: 54 points
    // in wave 0, reduced by DPP + readlane.
    if (wave == 0) {
        int blo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, bhi[3] = {-1, -1, -1};
        const int rl = lane % FR;
        if (lane < 2 * FR && rl < n_live) {
            const float* rp = a.rays + (ray0 + rl) * a.ray_cols;
            const float z = z_of(f, 0, FS, 0.0f, lane < FR ? 0 : FS - 1);
            const float pt[3] = {rp[0] + rp[3] * z, rp[1] + rp[4] * z, rp[2] + rp[5] * z};
            float xn[3];
            field_normalize(f, pt, xn);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const int Gx = f.grid[ax];
                const float x = unnorm(xn[ax], Gx);
                blo[ax] = 0; bhi[ax] = Gx - 1;          // NaN: the whole axis (forces the gather path)
                if (x == x) {
                    const int fl = (int)floorf(fminf(fmaxf(x, -1.0f), (float)Gx));
                    blo[ax] = min(max(fl, 0), Gx - 1);
                    bhi[ax] = min(max(fl + 1, 0), Gx - 1);
                }
            }
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int l = wave_min(blo[ax]), h = wave_max(bhi[ax]);
            if (lane == 0) { s_box[ax] = l; s_box[3 + ax] = h; }
        }
    }
    __syncthreads();
    int lo[3], ext[3];                                  // the box: its low corner and its side per axis, in texels
    bool fits = n_live > 0, inner = true;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        lo[ax] = __builtin_amdgcn_readfirstlane(s_box[ax]);
        const int hi = __builtin_amdgcn_readfirstlane(s_box[3 + ax]);
        ext[ax] = hi - lo[ax] + 1;
        fits = fits && (ext[ax] <= FP);
        inner = inner && (lo[ax] + FP <= f.grid[ax]);
    }
    // ---------------------------------------------------------------------------------------------------- the DMA of a pass
    // A pass image is [plane rows][PITCH][SL] then [line rows][SL] (NDP = 1, density: three such images of 16 channels, the planes
    // first); chunk L = 512 round + tid is one 16-B piece of it.  The decomposition of an appearance-geometry chunk is the same
    // in every pass (q [0:3], column [4:8], row [9:13], line flag 14, valid 15); it is recomputed per pass from the thread index -- held in
    // registers through the kernel it was spilled, and a reload in front of a DMA waits for every DMA in flight.
    auto chunk_of = [&](int r) -> uint32_t {
        const int L = NT * r + tid;
        constexpr int CPT = SLC / 4, NPL = G::PLANE / 4;
        uint32_t e = 0u;
        if (L < NPL) {
            const int texel = L / CPT, qq = L - texel * CPT;
            const int ry = texel / PITCH, rx = min(texel - ry * PITCH, FP - 1);       // the pad columns repeat the row's last texel
            e = (uint32_t)qq | ((uint32_t)rx << 4) | ((uint32_t)ry << 9) | (1u << 15);
        } else if (L < G::NCH) {
            const int l = L - NPL, rz = l / CPT, qq = l - rz * CPT;
            e = (uint32_t)qq | ((uint32_t)rz << 9) | (1u << 14) | (1u << 15);
        }
        return e;
    };
    // table of `C` channels per texel, channels [ch0, ch0 + SLC) of it, plane axes (pa, pb), line axis pv
    auto dma_slice = [&](const float* ptab, const float* ltab, int C, int ch0, int pa, int pb, int pv, float* buf) {
        const int Ga = f.grid[pa], Gb = f.grid[pb], Gv = f.grid[pv];
        const char* pbase = reinterpret_cast<const char*>(ptab + ((size_t)lo[pb] * Ga + lo[pa]) * C + ch0);
        const char* lbase = reinterpret_cast<const char*>(ltab + (size_t)lo[pv] * C + ch0);
        // only the fan's ACTUAL box travels: every tap of every sample lies inside it (phase 0), the rest of the FP x FP template is never
        // read.  The pass image keeps its layout (the lanes of the texels outside the box are switched off in the DMA instructions: no
        // bytes fetched for them); on the 640^3 unisphere workload a fan's box is 17 x 18.5 x 20 texels on average against the template's
        // 22^3: 0.71 of the staged bytes (scripts/fan_box_stats.py)
        const int na = ext[pa], nb = ext[pb], nv = ext[pv];
        constexpr int NR_ = G::NR;
#pragma unroll
        for (int r = 0; r < NR_; ++r) {
            const uint32_t e = chunk_of(r);
            const int rx_ = (e >> 4) & 31u, ry_ = (e >> 9) & 31u;
            const bool wanted = (e & (1u << 14)) ? ry_ < nv : (rx_ < na && ry_ < nb);
            if ((e & (1u << 15)) && wanted) {
                const int qq = e & 15u, rx = (e >> 4) & 31u, ry = (e >> 9) & 31u;
                const char* src;
                if (inner) {
                    src = (e & (1u << 14)) ? lbase + (unsigned)(ry * (C * 4) + qq * 16)
                                           : pbase + (unsigned)(ry * (Ga * C * 4) + rx * (C * 4) + qq * 16);
                } else if (e & (1u << 14)) {
                    src = reinterpret_cast<const char*>(ltab + (size_t)min(lo[pv] + ry, Gv - 1) * C + ch0 + 4 * qq);
                } else {
                    src = reinterpret_cast<const char*>(ptab + ((size_t)min(lo[pb] + ry, Gb - 1) * Ga + min(lo[pa] + rx, Ga - 1)) * C + ch0 + 4 * qq);
                }
                dma16(src, buf + (NT * r + 64 * wave) * 4);
            }
        }
    };
    // NDP = 1: the three 16-channel density planes and lines as one pass (its own chunk decomposition, used once)
    auto dma_density_all = [&](float* buf) {
        constexpr int NPL1 = G::DPLANE / 4, NLN1 = G::DLINE / 4;      // chunks of one plane / one line
#pragma unroll
        for (int r = 0; r < G::NR; ++r) {
            const int L = NT * r + tid;
            if (L < G::NCH) {
                const bool is_line = L >= 3 * NPL1;
                const int Lr = is_line ? L - 3 * NPL1 : L;
                const int per = is_line ? NLN1 : NPL1;
                const int i = (Lr >= per ? 1 : 0) + (Lr >= 2 * per ? 1 : 0);                 // plane / line index
                const int rem = Lr - i * per, texel = rem >> 2, qq = rem & 3;
                // the plane's / line's parameters by selects (no runtime-indexed lo[] / grid[]): plane i spans (a, b) = (0,1),(0,2),(1,2), line i runs along 2 - i
                const int Ga = i == 2 ? f.grid[1] : f.grid[0], Gb = i == 0 ? f.grid[1] : f.grid[2];
                const int la = i == 2 ? lo[1] : lo[0], lb = i == 0 ? lo[1] : lo[2];
                const int Gv = i == 0 ? f.grid[2] : (i == 1 ? f.grid[1] : f.grid[0]), lv = i == 0 ? lo[2] : (i == 1 ? lo[1] : lo[0]);
                const float* ptab = i == 0 ? f.dplane[0] : (i == 1 ? f.dplane[1] : f.dplane[2]);
                const float* ltab = i == 0 ? f.dline[0] : (i == 1 ? f.dline[1] : f.dline[2]);
                const int ry = texel / PITCH, rx = min(texel - ry * PITCH, FP - 1);
                const int row = is_line ? (inner ? lv + texel : min(lv + texel, Gv - 1)) : (inner ? lb + ry : min(lb + ry, Gb - 1));
                const int col = inner ? la + rx : min(la + rx, Ga - 1);
                const float* src = is_line ? ltab + ((size_t)row * 16 + 4 * qq) : ptab + (((size_t)row * Ga + col) * 16 + 4 * qq);
                const int ea = i == 2 ? ext[1] : ext[0], eb = i == 0 ? ext[1] : ext[2], ev = i == 0 ? ext[2] : (i == 1 ? ext[1] : ext[0]);
                if (is_line ? texel < ev : (rx < ea && ry < eb))          // the actual box only (dma_slice)
                    dma16(src, buf + (NT * r + 64 * wave) * 4);
            }
        }
    };
    // pass n of the fan into its buffer: passes 0 .. NDP-1 density, then appearance (plane i, slice js) = NDP + NSL i + js
    auto issue_pass = [&](int n) {
        float* buf = (n & 1) ? s_buf1 : s_buf0;
        if (n < NDP) {
            if (NDP == 1) dma_density_all(buf);
            else dma_slice(f.dplane[n], f.dline[n], 16, 0, mat_a(n), mat_b(n), vec_ax(n), buf);
        } else {
            const int i = (n - NDP) / NSL, js = (n - NDP) - i * NSL;
            dma_slice(f.aplane[i], f.aline[i], 48, js * SLC, mat_a(i), mat_b(i), vec_ax(i), buf);
        }
    };
    // a buffer may be read once every wave's pieces of it have landed: each wave waits for its own, then the barrier
    auto land_and_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    if (fits) issue_pass(0);
    STAMP(1);
    // ---------------------------------------------------------------------------------------------------- phase A, second half
    // the tap indices relative to the box, packed; the records to LDS
    auto write_rec = [&](int t, const Pre& r) {
        u32q rec = {0u, 0u, 0u, 0u};
        if (r.any) {
            uint32_t packed = r.flags;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) packed |= (uint32_t)((r.t0[ax] - lo[ax]) & 31) << (8 * ax);
            rec.x = packed; rec.y = __float_as_uint(r.w1[0]); rec.z = __float_as_uint(r.w1[1]); rec.w = __float_as_uint(r.w1[2]);
        }
        *reinterpret_cast<u32q*>(s_rec + t * 4) = rec;
    };
    write_rec(tid, pre0);
    if (wave == 7 && lane < NSMP - NT) write_rec(NT + lane, pre1);
    STAMP(2);
    // ---------------------------------------------------------------------------------------------------- phase B: density
    // quad p of a ray takes the samples s = 4 k + p (k = 0 .. 4): in trip k the four quads of the ray -- one ds_read_b128 lane group --
    // gather four consecutive samples.  Lane c holds the partial sum of its channel quarter over the planes, in the order of
    // density_partial (plane 0, 1, 2).
    const int t0 = (slot_on ? rs : 0) * FS + p;          // record index of the quad's first sample; then + 4 k
    // one sample's density terms of the planes staged in `buf` (all three, or plane `dp`), added to `acc` in the order of
    // density_partial (plane 0, 1, 2; channels x, y, z, w)
    auto density_terms = [&](const float* buf, const RecView& rv, int dp, float acc) {
        constexpr int NPL = (NDP == 1) ? 3 : 1;
#pragma unroll
        for (int ii = 0; ii < NPL; ++ii) {
            const int i = (NDP == 1) ? ii : dp;
            const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
            const float* P = buf + ((NDP == 1) ? ii * G::DPLANE : 0) + ((rv.r[ax_b] * PITCH + rv.r[ax_a]) * 16 + 4 * c);
            const int da = rv.d[ax_a] * 16, db = rv.d[ax_b] * (PITCH * 16);
            const f32q tnw = *reinterpret_cast<const f32q*>(P), tne = *reinterpret_cast<const f32q*>(P + da);
            const f32q tsw = *reinterpret_cast<const f32q*>(P + db), tse = *reinterpret_cast<const f32q*>(P + db + da);
            const float* L = buf + ((NDP == 1) ? 3 * G::DPLANE + ii * G::DLINE : G::DPLANE) + (rv.r[ax_v] * 16 + 4 * c);
            const f32q ll = *reinterpret_cast<const f32q*>(L), lh = *reinterpret_cast<const f32q*>(L + rv.d[ax_v] * 16);
            const float pw[4] = {rv.wt[ax_b][0] * rv.wt[ax_a][0], rv.wt[ax_b][0] * rv.wt[ax_a][1],
                                 rv.wt[ax_b][1] * rv.wt[ax_a][0], rv.wt[ax_b][1] * rv.wt[ax_a][1]};
            const float lw[2] = {rv.wt[ax_v][0], rv.wt[ax_v][1]};
            const f32q pq = lerp_plane_q(tnw, tne, tsw, tse, pw), lq = lerp_line_q(ll, lh, lw);
            acc = fmaf(pq.w, lq.w, fmaf(pq.z, lq.z, fmaf(pq.y, lq.y, fmaf(pq.x, lq.x, acc))));
        }
        return acc;
    };
    // lane c finishes the sample of trip c (round 0: trips 0 .. 3) and lane 0 the fifth (round 1): sigma and alpha (tensorBase.py:25,849)
    auto finish = [&](int k, float feat, bool valid, bool store) {
        const int s = 4 * k + p;
        const float sigma = valid ? feature2density(f, feat) : 0.0f;
        const float zs = z_of(f, 0, FS, 0.0f, s);
        const float dist = (s + 1 < FS) ? (z_of(f, 0, FS, 0.0f, s + 1) - zs) : 0.0f;       // tensorBase.py:800-803
        const float alpha = 1.0f - expf(-sigma * (dist * f.distance_scale));
        if (slot_on && store) s_w[rs * FS + s] = alpha;
    };
    if (fits && NDP == 1) {
        land_and_sync();                                  // the records are written, the density patches have landed
        issue_pass(1);
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            float feat_mine = 0.0f;
            bool valid_mine = false;
            const int nk = round == 0 ? 4 : 1;
#pragma unroll 1
            for (int j = 0; j < nk; ++j) {
                const int k = 4 * round + j;
                const RecView rv = unpack_rec(*reinterpret_cast<const u32q*>(s_rec + (t0 + 4 * k) * 4));
                const float feat = sum4_dpp(rv.valid ? density_terms(s_buf0, rv, 0, 0.0f) : 0.0f);
                if (j == c) { feat_mine = feat; valid_mine = rv.valid; }
            }
            finish(round == 0 ? c : 4, feat_mine, valid_mine, round == 0 || c == 0);
        }
    } else if (fits) {
        // one plane per pass: the five partial sums of the quad's samples stay in registers across the passes (select chains, no
        // runtime-indexed array); the record of the next sample is read one trip ahead
        float part[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int dp = 0; dp < NDP; ++dp) {
            land_and_sync();
            issue_pass(dp + 1);
            const float* buf = (dp & 1) ? s_buf1 : s_buf0;
            u32q nrec = *reinterpret_cast<const u32q*>(s_rec + t0 * 4);
#pragma unroll 1
            for (int k = 0; k < 5; ++k) {
                const RecView rv = unpack_rec(nrec);
                nrec = *reinterpret_cast<const u32q*>(s_rec + (t0 + 4 * min(k + 1, 4)) * 4);
                const float before = k == 0 ? part[0] : (k == 1 ? part[1] : (k == 2 ? part[2] : (k == 3 ? part[3] : part[4])));
                const float after = density_terms(buf, rv, dp, before);
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) part[kk] = kk == k ? after : part[kk];
            }
        }
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            // the quad's four lanes finish four different samples: each needs the sum of ITS sample's four quarter partials
            float feat = 0.0f;
            bool valid = false;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ks = round == 0 ? kk : 4;
                const float pk = ks == 0 ? part[0] : (ks == 1 ? part[1] : (ks == 2 ? part[2] : (ks == 3 ? part[3] : part[4])));
                const bool vk = (s_rec[(t0 + 4 * ks) * 4] & REC_VALID) != 0u;
                const float fk = sum4_dpp(vk ? pk : 0.0f);
                if (kk == c || round == 1) { feat = fk; valid = vk; }
            }
            finish(round == 0 ? c : 4, feat, valid, round == 0 || c == 0);
        }
    } else {
        // the gather path: the taps of a valid sample where the general kernels read them (density_partial: the same chain)
        land_and_sync();                                  // the records are written
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            float feat_mine = 0.0f;
            bool valid_mine = false;
            const int nk = round == 0 ? 4 : 1;
#pragma unroll 1
            for (int j = 0; j < nk; ++j) {
                const int k = 4 * round + j;
                const bool valid = (s_rec[(t0 + 4 * k) * 4] & REC_VALID) != 0u;
                float v = 0.0f;
                if (valid) {
                    const float* sr = s_ray + (slot_on ? rs : 0) * 8;
                    const float z = z_of(f, 0, FS, 0.0f, 4 * k + p);
                    const float pt[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                    float xn[3];
                    field_normalize(f, pt, xn);
                    v = density_partial(f, xn, c);
                }
                const float feat = sum4_dpp(v);
                if (j == c) { feat_mine = feat; valid_mine = valid; }
            }
            finish(round == 0 ? c : 4, feat_mine, valid_mine, round == 0 || c == 0);
        }
    }
    STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the alphas are in LDS; no vector-memory wait: the first appearance patch stays in flight
    // the transmittance product of the rays (tensorBase.py:27-32): one lane per ray in wave 0 -- the chain is serial per ray, and
    // run by the sixteen lanes of every slot it cost every wave its instructions; the other waves go on to the next barrier
    if (wave == 0 && lane < FR) {
        const int ry = lane, tb = ry * FS;
        const int64_t r_glob = ray0 + ry;
        const bool rlive = ry < n_live;
        float run_T = 1.0f, run_acc = 0.0f, run_depth = 0.0f;
        int run_valid = 0, run_app = 0;
        unsigned sh_bits = 0u;
        float al[FS];
#pragma unroll
        for (int s = 0; s < FS; ++s) al[s] = s_w[tb + s];
        if (a.counts) {
#pragma unroll
            for (int s = 0; s < FS; ++s) run_valid += (s_rec[(tb + s) * 4] & REC_VALID) ? 1 : 0;
        }
#pragma unroll
        for (int s = 0; s < FS; ++s) {
            const float alpha = al[s];
            const float z = z_of(f, 0, FS, 0.0f, s);
            const float w = alpha * run_T;
            run_T = run_T * ((1.0f - alpha) + 1e-10f);
            run_acc += w;
            run_depth += w * z;
            const bool sh = w > f.weight_thres;                                     // tensorBase.py:851
            run_app += sh ? 1 : 0;
            sh_bits |= (sh ? 1u : 0u) << s;
            s_w[tb + s] = w;
#ifndef FAN_STAMPS
            if (rlive && a.alpha) a.alpha[r_glob * FS + s] = alpha;
#endif
        }
        s_sh[ry] = rlive ? sh_bits : 0u;
        if (rlive) {
            if (MODE == 3) s_ray[ry * 8 + 6] = run_acc;             // phase E blends with it
            a.acc[r_glob] = run_acc;
            a.depth[r_glob] = run_depth + (1.0f - run_acc) * s_ray[ry * 8 + 7];
            if (a.counts) { a.counts[r_glob * 2] = run_valid; a.counts[r_glob * 2 + 1] = run_app; }
        }
    }
    if (wave == 0 && lane >= FR && lane < 32) s_sh[lane] = 0u;
    STAMP(5);
    // (the masks are read behind the barrier that opens the first appearance pass / the gather path)
    // ---------------------------------------------------------------------------------------------------- phase C: appearance
    // accp[12 j + 4 i + e]: the weighted product sum of channel 16 j + 4 c + e of plane i over this quad's samples
    float accp[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) accp[i] = 0.0f;
    unsigned shmask = 0u, mymask = 0u;                     // the ray's shaded samples, this quad's of them (s = p mod 4)
    f32q pre[2];                                           // basis_mat for phase D: requested during the last pass
    auto fetch_basis = [&]() {
        constexpr int NB = 27 * 144 / 4;                   // 972 pieces
        pre[0] = *reinterpret_cast<const f32q*>(f.basis + 4 * tid);
        if (tid + NT < NB) pre[1] = *reinterpret_cast<const f32q*>(f.basis + 4 * (tid + NT));
    };
    if (fits) {
#pragma unroll
        for (int ap = 0; ap < 3 * NSL; ++ap) {
            const int n = NDP + ap, i = ap / NSL, js = ap - i * NSL;
            land_and_sync();                               // the weights are written (ap = 0) / the other buffer has been read; this patch landed
            if (ap == 0) { shmask = s_sh[rs]; mymask = shmask & (0x11111u << p); }
            if (ap + 1 < 3 * NSL) issue_pass(n + 1);
            STAMP(6 + (ap < 4 ? ap : 4));
            const float* buf = (n & 1) ? s_buf1 : s_buf0;
            const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
            unsigned m = mymask;
            const int tb = (slot_on ? rs : 0) * FS;
            if constexpr (QPP == 1) {
                // one 16-channel quarter per pass: 24 registers of taps per trip, so the taps of the NEXT shaded sample are requested
                // before this one's are combined, and the record after that is on its way (three LDS round trips overlapped per trip)
                const int j = js;
                struct Trip { f32q t[6]; float pw[4], lw[2], w; };
                auto pop = [&]() { const int s = __ffs((int)m) - 1; m &= m - 1u; return s; };
                auto request = [&](const u32q rec, float w, Trip& T) {
                    const RecView rv = unpack_rec(rec);
                    const float* P = buf + ((rv.r[ax_b] * PITCH + rv.r[ax_a]) * SLC + 4 * c);
                    const int da = rv.d[ax_a] * SLC, db = rv.d[ax_b] * (PITCH * SLC);
                    const float* L = buf + G::PLANE + (rv.r[ax_v] * SLC + 4 * c);
                    T.t[0] = *reinterpret_cast<const f32q*>(P); T.t[1] = *reinterpret_cast<const f32q*>(P + da);
                    T.t[2] = *reinterpret_cast<const f32q*>(P + db); T.t[3] = *reinterpret_cast<const f32q*>(P + db + da);
                    T.t[4] = *reinterpret_cast<const f32q*>(L); T.t[5] = *reinterpret_cast<const f32q*>(L + rv.d[ax_v] * SLC);
                    T.pw[0] = rv.wt[ax_b][0] * rv.wt[ax_a][0]; T.pw[1] = rv.wt[ax_b][0] * rv.wt[ax_a][1];
                    T.pw[2] = rv.wt[ax_b][1] * rv.wt[ax_a][0]; T.pw[3] = rv.wt[ax_b][1] * rv.wt[ax_a][1];
                    // the compositing weight rides on the line taps' weights, as in the four-wave kernel (fan_march_kernels.hip, phase C):
                    // plane * (w line), one fma per channel
                    T.lw[0] = w * rv.wt[ax_v][0]; T.lw[1] = w * rv.wt[ax_v][1];
                    T.w = w;
                };
                auto combine = [&](const Trip& T) {
                    const f32q pl = lerp_plane_q(T.t[0], T.t[1], T.t[2], T.t[3], T.pw), ln = lerp_line_q(T.t[4], T.t[5], T.lw);
                    accp[12 * j + 4 * i + 0] = fmaf(pl.x, ln.x, accp[12 * j + 4 * i + 0]);
                    accp[12 * j + 4 * i + 1] = fmaf(pl.y, ln.y, accp[12 * j + 4 * i + 1]);
                    accp[12 * j + 4 * i + 2] = fmaf(pl.z, ln.z, accp[12 * j + 4 * i + 2]);
                    accp[12 * j + 4 * i + 3] = fmaf(pl.w, ln.w, accp[12 * j + 4 * i + 3]);
                };
                Trip A, B;
                u32q rec = {0u, 0u, 0u, 0u};
                float wn = 0.0f;
                bool hA = m != 0u;
                if (hA) { const int s = pop(); rec = *reinterpret_cast<const u32q*>(s_rec + (tb + s) * 4); wn = s_w[tb + s]; request(rec, wn, A); }
                bool pending = m != 0u;
                if (pending) { const int s = pop(); rec = *reinterpret_cast<const u32q*>(s_rec + (tb + s) * 4); wn = s_w[tb + s]; }
                while (hA) {
                    const bool hB = pending;
                    if (hB) {
                        request(rec, wn, B);
                        pending = m != 0u;
                        if (pending) { const int s = pop(); rec = *reinterpret_cast<const u32q*>(s_rec + (tb + s) * 4); wn = s_w[tb + s]; }
                    }
                    combine(A);
                    hA = false;
                    if (hB) {
                        hA = pending;
                        if (hA) {
                            request(rec, wn, A);
                            pending = m != 0u;
                            if (pending) { const int s = pop(); rec = *reinterpret_cast<const u32q*>(s_rec + (tb + s) * 4); wn = s_w[tb + s]; }
                        }
                        combine(B);
                    }
                }
            } else {
            // the record and weight of the next sample are read one trip ahead
            u32q na = {0u, 0u, 0u, 0u};
            float nw = 0.0f;
            if (m) {
                const int s = __ffs((int)m) - 1;
                na = *reinterpret_cast<const u32q*>(s_rec + (tb + s) * 4);
                nw = s_w[tb + s];
            }
            while (m) {
                m &= m - 1u;
                const RecView rv = unpack_rec(na);
                const float w = nw;
                if (m) {
                    const int s = __ffs((int)m) - 1;
                    na = *reinterpret_cast<const u32q*>(s_rec + (tb + s) * 4);
                    nw = s_w[tb + s];
                }
                const float* P = buf + ((rv.r[ax_b] * PITCH + rv.r[ax_a]) * SLC + 4 * c);
                const int da = rv.d[ax_a] * SLC, db = rv.d[ax_b] * (PITCH * SLC);
                const float* L = buf + G::PLANE + (rv.r[ax_v] * SLC + 4 * c);
                const int dv = rv.d[ax_v] * SLC;
                const float pw[4] = {rv.wt[ax_b][0] * rv.wt[ax_a][0], rv.wt[ax_b][0] * rv.wt[ax_a][1],
                                     rv.wt[ax_b][1] * rv.wt[ax_a][0], rv.wt[ax_b][1] * rv.wt[ax_a][1]};
                const float lw[2] = {w * rv.wt[ax_v][0], w * rv.wt[ax_v][1]};        // (the compositing weight on the line taps: see above)
#pragma unroll
                for (int jj = 0; jj < QPP; ++jj) {         // quarter c + 4 j of the 192-B texel, j = QPP js + jj
                    const int j = QPP * js + jj;
                    const f32q tnw = *reinterpret_cast<const f32q*>(P + 16 * jj), tne = *reinterpret_cast<const f32q*>(P + 16 * jj + da);
                    const f32q tsw = *reinterpret_cast<const f32q*>(P + 16 * jj + db), tse = *reinterpret_cast<const f32q*>(P + 16 * jj + db + da);
                    const f32q ll = *reinterpret_cast<const f32q*>(L + 16 * jj), lh = *reinterpret_cast<const f32q*>(L + 16 * jj + dv);
                    const f32q pl = lerp_plane_q(tnw, tne, tsw, tse, pw), ln = lerp_line_q(ll, lh, lw);
                    accp[12 * j + 4 * i + 0] = fmaf(pl.x, ln.x, accp[12 * j + 4 * i + 0]);
                    accp[12 * j + 4 * i + 1] = fmaf(pl.y, ln.y, accp[12 * j + 4 * i + 1]);
                    accp[12 * j + 4 * i + 2] = fmaf(pl.z, ln.z, accp[12 * j + 4 * i + 2]);
                    accp[12 * j + 4 * i + 3] = fmaf(pl.w, ln.w, accp[12 * j + 4 * i + 3]);
                }
            }
            }
            if (ap + 1 == 3 * NSL) fetch_basis();          // no DMA is in flight any more: plain loads
        }
    } else {
        // the gather path: a tile whose samples do not fit one patch reads its taps where the general kernels do
        __syncthreads();                                   // the weights are written
        shmask = s_sh[rs]; mymask = shmask & (0x11111u << p);
        unsigned m = mymask;
        const int tb = (slot_on ? rs : 0) * FS;
        const float* sr = s_ray + (slot_on ? rs : 0) * 8;          // (the ray from LDS: its registers are not kept through phase C)
        while (m) {
            const int s = __ffs((int)m) - 1;
            m &= m - 1u;
            const float w = s_w[tb + s];
            const float z = z_of(f, 0, FS, 0.0f, s);
            const float pt[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
            float xn[3];
            field_normalize(f, pt, xn);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float prod[12];
                app_products_lane(f, xn, c + 4 * j, prod);
#pragma unroll
                for (int qq = 0; qq < 12; ++qq) accp[12 * j + qq] = fmaf(w, prod[qq], accp[12 * j + qq]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fetch_basis();
    }
    STAMP(11);
    // ---------------------------------------------------------------------------------------------------- phase D: basis_mat
    // the four quads of a ray add their sample chains, (p0 + p1) + (p2 + p3); then F[ray][o] = sum_k basis_mat[o][k] A[ray][k] over
    // the 144 weighted products as one 32 x 32 x 144 product on the fp32 matrix cores -- the four-wave kernel's phase D: k split
    // over waves 0 .. 3, the four partial tiles added in a fixed order.
    {
        // the ray's quads are {0, 3 | 5, 6} or {1, 2 | 4, 7} of a 32-lane half (two DPP rows of four quads): first the partner inside
        // the row -- one quad to the left for quads 0, 2, 4, 6 of the half (row_ror 4), one to the right for the others (row_ror 12) --,
        // then the pair sum two quads on in the OTHER row (one ds_bpermute): (p0 + p1) + (p2 + p3) on every lane of the ray
        const bool left = ((q & 7) == 0) || ((q & 7) == 2) || ((q & 7) == 4) || ((q & 7) == 6);
        const int other = (((lane + 8) & 15) | ((lane & 16) ^ 16) | (lane & 32)) << 2;      // byte index for ds_bpermute
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const float a4 = dpp_mov<0x124>(accp[i]), a12 = dpp_mov<0x12C>(accp[i]);       // row_ror:4, row_ror:12
            accp[i] = accp[i] + (left ? a4 : a12);
        }
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(other, __builtin_bit_cast(int, accp[i])));
            accp[i] = (p < 2) ? accp[i] + o : o + accp[i];
        }
    }
    constexpr int DLD = 33;                            // operand rows padded: conflict-free ds_read_b32 down a column of k
    float* const s_A = s_pool;                         // [144][DLD]  A^T: weighted products, column = ray slot
    float* const s_B = s_pool + 144 * DLD;             // [144][DLD]  basis_mat^T, column = output feature
    float* const s_feat = s_pool + 4 * 32 * 32;        // [32][28] output rows, behind the partial tiles
    static_assert(2 * 144 * DLD <= 2 * BUF && 4 * 32 * 32 + 32 * 28 <= 2 * BUF, "phase D operands fit the two buffers");
    __syncthreads();                                   // every wave is done with the patches
    if (slot_on && p == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) s_A[(48 * i + 16 * j + 4 * c + e4) * DLD + rs] = accp[12 * j + 4 * i + e4];
    }
    if (!slot_on && p == 0) {                          // columns 27 .. 31 of the matrix operand
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) s_A[(48 * i + 16 * j + 4 * c + e4) * DLD + rs] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int chunk = tid + NT * r;
        if (chunk < 27 * 144 / 4) {
            const int o = chunk / 36, k4 = (chunk - o * 36) * 4;
            s_B[(k4 + 0) * DLD + o] = pre[r].x; s_B[(k4 + 1) * DLD + o] = pre[r].y;
            s_B[(k4 + 2) * DLD + o] = pre[r].z; s_B[(k4 + 3) * DLD + o] = pre[r].w;
        }
    }
    if (tid < 144) {                                   // columns 27 .. 31 of basis_mat^T
#pragma unroll
        for (int o = 27; o < 32; ++o) s_B[tid * DLD + o] = 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);                 // (the head's operand loads below stay behind the accumulators' last use)
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    f32q stage_pre = splat(0.0f), wrow[7], bias_pre[4];
    if (MODE == 3) {
        // the two head slices phase E keeps in LDS (from spec_w on: spec_w, spec_b, ide_mat; up to bott_w: the small heads), 16 B per thread
        const int n_tail4 = (ho.total - ho.spec_w) / 4, n_small4 = ho.bott_w / 4;
        if (tid < n_tail4) stage_pre = *reinterpret_cast<const f32q*>(f.head + ho.spec_w + 4 * tid);
        else if (tid < n_tail4 + n_small4) stage_pre = *reinterpret_cast<const f32q*>(f.head + 4 * (tid - n_tail4));
        if (wave < 4 && 32 * wave < f.feature_c) {
            // phase E's matrix operand (this wave's 32 bottleneck rows, one row per lane) and biases: in flight through phase D
            const float* wr = f.head + ho.bott_w + (32 * wave + (lane & 31)) * 28;
#pragma unroll
            for (int k4 = 0; k4 < 7; ++k4) wrow[k4] = *reinterpret_cast<const f32q*>(wr + 4 * k4);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) bias_pre[qq] = *reinterpret_cast<const f32q*>(f.head + ho.bott_b + 32 * wave + 8 * qq + 4 * (lane >> 5));
        }
    }
    __syncthreads();
    STAMP(12);
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    {
        const int lr = lane & 31, lh = lane >> 5;
        f32x16 dacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[r] = 0.0f;
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < 18; ++t) {             // this wave's 36 values of k, two per instruction
                const int k = 36 * wave + 2 * t + lh;
                dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_A[k * DLD + lr], s_B[k * DLD + lr], dacc, 0, 0, 0);
            }
        }
        __syncthreads();                               // the operands have been read: the partial tiles go over them
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_pool[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = dacc[r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < FR * 27; idx += NT) {
        const int ry = idx / 27, o = idx - ry * 27;
        const float* pp = s_pool + ry * 32 + o;
        s_feat[ry * 28 + o] = (pp[0] + pp[32 * 32]) + (pp[2 * 32 * 32] + pp[3 * 32 * 32]);
    }
    if (slot_on && p == 0 && c == 0) s_feat[rs * 28 + 27] = shmask != 0u ? 1.0f : 0.0f;
    if (MODE == 3 && tid < 5 * 28) s_feat[FR * 28 + tid] = 0.0f;             // rows 27..31 of the matrix operand
    STAMP(13);
    __syncthreads();
    if (MODE != 3) {
        if (tid < n_live * 7)
            *reinterpret_cast<f32q*>(a.feat + ray0 * 28 + 4 * tid) = *reinterpret_cast<const f32q*>(s_feat + 4 * tid);
        STAMP(14);
        return;
    }
    // ---------------------------------------------------------------------------------------------------- phase E: the Ref head
    // (models/ref.py:103-152) as in the four-wave kernel: the bottleneck as W[32 rows][28] x F^T[28][32 rays] tiles on the fp32 matrix
    // cores, one 32-row block per wave 0 .. 3; four lanes per ray then finish it (ref_head_quad) in two waves.
    constexpr int BLD = 164;
    float* const s_b = s_pool + 5120;
    static_assert(4 * 32 * 32 + 32 * 28 <= 5120 && 5120 + 32 * BLD + 680 + 296 <= 2 * BUF, "phase E operands fit the two buffers");
    const int fc = f.feature_c;
    float* const s_tail = s_b + 32 * BLD;
    float* const s_small = s_tail + 680;
    {
        const int n_tail4 = (ho.total - ho.spec_w) / 4, n_small4 = ho.bott_w / 4;         // <= 170 + 74 (fan_head_fusable)
        if (tid < n_tail4) *reinterpret_cast<f32q*>(s_tail + 4 * tid) = stage_pre;
        else if (tid < n_tail4 + n_small4) *reinterpret_cast<f32q*>(s_small + 4 * (tid - n_tail4)) = stage_pre;
    }
    if (wave < 4 && 32 * wave < fc) {
        const int lr = lane & 31, lh = lane >> 5;
        f32x16 e;
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const float av = lh ? wrow[ks >> 1][2 * (ks & 1) + 1] : wrow[ks >> 1][2 * (ks & 1)];
            float bv = s_feat[lr * 28 + 2 * ks + lh];
            if (ks == 13) bv = lh ? 0.0f : bv;                             // column 27 is the shaded flag, not a feature
            e = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, e, 0, 0, 0);
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int r0 = 32 * wave + 8 * qq + 4 * lh;
            const f32q bias = bias_pre[qq];
            f32q o4 = {e[4 * qq] + bias[0], e[4 * qq + 1] + bias[1], e[4 * qq + 2] + bias[2], e[4 * qq + 3] + bias[3]};
            *reinterpret_cast<f32q*>(s_b + lr * BLD + r0) = o4;
        }
    }
    __syncthreads();
    // four lanes per ray, 32 ray slots = 128 threads = two of the eight waves: which two rotates with the tile
    const int ew = (wave - (int)(blockIdx.x & 7u)) & 7;
    if (ew < 2) {
        const int eg = 16 * ew + (lane >> 2), sub = lane & 3;
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

}  // namespace

// Which patch size serves a field: the box of a fan is at most 2 x (ten steps in texels) + 2 per axis.  Unisphere contraction
// (utils.py:139-146, applied per axis) is monotone with slope <= 1, so a fan's box is at most that of the uncontracted step.
//   0: neither (the general kernels), 12 / 22: k4g_fan_march<12, 1> / <22, 3>
int fan8_patch_side(const FieldDev& f, int mode, int S) {
    if (mode != 0 || S != FS || f.n_density != 16 || f.n_app != 48 || f.app_dim != 27) return 0;
    float worst = 0.0f;
    for (int ax = 0; ax < 3; ++ax) {
        const float scale = f.unisphere ? 1.0f : f.inv_aabb[ax];            // d(normalised coordinate) / d(world coordinate), at most
        const float texels = 10.0f * f.step_size * scale * 0.5f * (float)(f.grid[ax] - 1);
        if (!(texels == texels)) return 0;
        worst = texels > worst ? texels : worst;
    }
    if (worst <= 5.25f) return 12;          // 2 x 5 + 2 = 12 (the reference's step_ratio 0.5 gives 5 texels per ten steps)
    if (worst <= 10.25f) return 22;         // 2 x 10 + 2 = 22 (unisphere: the step is a whole texel)
    return 0;
}

hipError_t launch_fan8_march(const FieldDev& f, const MarchArgs& a, int variant, hipStream_t s) {
    const int64_t n_tiles = (a.R + FR - 1) / FR;
    if (n_tiles == 0) return hipSuccess;
    if (n_tiles > 0x7fffffff) return hipErrorInvalidValue;
    const int fp = fan8_patch_side(f, a.mode, a.S);
    const dim3 grid((unsigned)n_tiles), block(NT);
    if (fp == 12) {
        if (variant == 3) hipLaunchKernelGGL((k4g_fan_march<12, 1, 3>), grid, block, 0, s, f, a);
        else hipLaunchKernelGGL((k4g_fan_march<12, 1, 2>), grid, block, 0, s, f, a);
    } else if (fp == 22) {
        if (variant == 3) hipLaunchKernelGGL((k4g_fan_march<22, 3, 3>), grid, block, 0, s, f, a);
        else hipLaunchKernelGGL((k4g_fan_march<22, 3, 2>), grid, block, 0, s, f, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
