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

namespace {

constexpr int FR = 27;          // rays per tile = one iso-cell fan (pose_estimation/isocell.py:6-68)
constexpr int FS = 20;          // samples per ray (pose_estimation/sampling.py:247)
constexpr int FP = 12;          // patch side, texels
constexpr int NT = 256;         // threads: 32 groups of 8 lanes (two sub-groups of 4); group g serves ray g of the tile
constexpr int REC = 8;          // dwords per sample record
constexpr int PLANE16 = FP * FP * 16, LINE16 = FP * 16;      // density patch (floats)
constexpr int PLANE48 = FP * FP * 48, LINE48 = FP * 48;      // appearance patch (floats)
constexpr int PATCH_FLOATS = PLANE48 + LINE48;               // 7488 floats = 29 952 B = 3 * (PLANE16 + LINE16)
constexpr int BASIS_FLOATS = 27 * 12 * 12;
static_assert(3 * (PLANE16 + LINE16) == PATCH_FLOATS, "the density patches fill the appearance patch exactly");
static_assert(BASIS_FLOATS <= PATCH_FLOATS, "basis_mat is staged in the patch buffer");

#ifndef FAN_WAVES
#define FAN_WAVES 3            // waves per SIMD the register budget is set for (three 256-thread workgroups per CU)
#endif
typedef uint32_t u32q __attribute__((ext_vector_type(4)));
// FAN_STAMPS (diagnostic build only): wave w of a tile stores the low word of s_memtime after each phase into the tile's
// slice of the optional alpha output ([R,20] floats: slot 16 w + k), which then carries no alphas.
#ifdef FAN_STAMPS
#define STAMP(k) do { if (a.alpha && lane == 0 && n_live == FR) reinterpret_cast<uint32_t*>(a.alpha)[ray0 * FS + 16 * wave + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
// FAN_EXIT_AFTER=k (timing / counter builds only): every tile stops after the phase that ends at stamp k -- the vector-instruction count
// of a phase is the difference of two such builds' SQ_INSTS_VALU (scripts/pmc_fan_phases.sh)
#ifdef FAN_EXIT_AFTER
#define FAN_EXIT(k) do { if (FAN_EXIT_AFTER == (k)) return; } while (0)
#else
#define FAN_EXIT(k) do { } while (0)
#endif
#if defined(FAN_STAMPS) && FAN_STAMPS == 2
#define ESTAMP(k) STAMP(k)
#else
#define ESTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ f32q splat(float v) { return (f32q)(v); }
// cross-lane moves as DPP modifiers of vector-ALU instructions (__shfl_xor compiles to ds_bpermute_b32: a round trip through
// the LDS crossbar per call)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 4 lanes of a quad, in the order of sum4 (iff_device.h): (v + xor1) then (+ xor2)
__device__ __forceinline__ float sum4_dpp(float v) {
    v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    return v;
}
// the value of lane (i xor 4): reverse inside the quad (quad_perm [3,2,1,0]), then mirror the 8-lane half row (row_half_mirror)
__device__ __forceinline__ float xor4_dpp(float v) { return dpp_mov<0x141>(dpp_mov<0x1B>(v)); }
// Bilinear / linear combination of tap quarters in the operation order of lerp_plane4 / lerp_line4 (iff_device.h): one multiply and
// fused multiply-adds per component.  THE LIBRARY IS BUILT WITHOUT PACKED FP32 INSTRUCTIONS (iffnerf_amd/build.py: -packed-fp32-ops;
// tests/test_isa_rules.py checks the shipped code objects), so these vector expressions compile to one v_mul / v_fma per component.
// Why: left to itself the compiler turns this function into v_pk_mul_f32 / v_pk_fma_f32 on register pairs with the weight broadcast
// by op_sel, and THAT code returned wrong sums for the last sixteen lanes of a wave -- rays 6, 7 (+ 8 w) of a tile, waves 0-2 --
// in 0.7 % of the steps (14 of 2 000 checked, 4 of 480), always in the last ~300 tiles of a march and only while a workgroup of
// the encoder / logits kernel (k5_trunk_h: fp16 MFMA) shared the CU: with four captured steps in flight the tail of a march runs
// next to another step's trunk.  Evaluating phase C twice in the same workgroup and comparing catches every event: a transient of
// the execution, not stale LDS.  What the investigation ("the packed-fp32 fault") excluded: missing waits (the
// s_waitcnt sequence of the faulty loop was checked load by load), waits / barriers / idle cycles around every LDS access, DPP vs
// ds_bpermute, occupancy, scratch (none), the matrix-core row order of phase D -- and the instruction FORM: the same products as
// hand-placed v_pk_mul_f32 / v_pk_fma_f32 (-DFAN_LERP_ASM=1..8 below: op_sel broadcast with a small-integer, equal or 1.0f high
// half, real (w, w) pairs, the weight as src0 or src1, every destination written over the broadcast pair, every destination
// disjoint) show 0 events in 6 400 checked steps, next to 18 in 2 480 for the compiler's own packing in the same runs.  So the
// trigger is the compiler's schedule of packed fp32 code next to MFMA work, not an operand form one could avoid by hand; no packed
// fp32 instruction is the rule that holds for all ~8 000 the compiler had placed in this library, and it costs nothing measurable
// (14 590-14 760 poses/s without, 14 590-14 810 with, same box, same run).
#if defined(FAN_LERP_ASM)
// Experiment builds of the packed-fp32 investigation (need +packed-fp32-ops; scripts/packed_fp32_forms.sh builds and runs them): the tap combination
// as hand-placed v_pk_mul_f32 / v_pk_fma_f32 with the weight operand in a chosen form.  FAN_LERP_ASM = 1: op_sel broadcast of the low half, high half = a small integer (what the
// compiler's own packing leaves there: an LDS address); 2: broadcast, high half = the weight again; 3: broadcast, high half = 1.0f;
// 4: no op_sel, a real (w, w) pair; 6: broadcast with the weight as src0 (the form of the compositing-weight accumulate);
// 7: as 1 with every result written OVER the broadcast pair (the destination overlaps the op_sel source); 8: as 1 with every
// destination disjoint from its sources.
typedef float f32p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float opq(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ f32p wpair(float w) {
#if FAN_LERP_ASM == 1 || FAN_LERP_ASM == 6 || FAN_LERP_ASM == 7 || FAN_LERP_ASM == 8
    return f32p{w, __uint_as_float(0x1200u + 16u * (threadIdx.x & 63u))};
#elif FAN_LERP_ASM == 3
    return f32p{w, opq(1.0f)};
#else
    return f32p{w, opq(w)};
#endif
}
__device__ __forceinline__ f32p pk_mul_w(f32p x, f32p w) {
    f32p r;
#if FAN_LERP_ASM == 4
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(w), "v"(x));
#elif FAN_LERP_ASM == 7          // the destination IS the broadcast source pair
    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(w) : "v"(x));
    r = w;
#elif FAN_LERP_ASM == 8          // the destination never overlaps a source (early clobber)
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=&v"(r) : "v"(w), "v"(x));
#else
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(x));
#endif
    return r;
}
__device__ __forceinline__ f32p pk_fma_w(f32p x, f32p w, f32p c) {
    f32p r;
#if FAN_LERP_ASM == 4
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(w), "v"(c));
#elif FAN_LERP_ASM == 6
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(w), "v"(x), "v"(c));
#elif FAN_LERP_ASM == 7
    asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(w) : "v"(x), "v"(c));
    r = w;
#elif FAN_LERP_ASM == 8
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(x), "v"(w), "v"(c));
#else
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(w), "v"(c));
#endif
    return r;
}
__device__ __forceinline__ f32q lerp_plane_q(f32q nw, f32q ne, f32q sw, f32q se, const float pw[4]) {
    const f32p w0 = wpair(pw[0]), w1 = wpair(pw[1]), w2 = wpair(pw[2]), w3 = wpair(pw[3]);
    f32p lo = pk_mul_w(f32p{nw.x, nw.y}, w0), hi = pk_mul_w(f32p{nw.z, nw.w}, w0);
    lo = pk_fma_w(f32p{ne.x, ne.y}, w1, lo); hi = pk_fma_w(f32p{ne.z, ne.w}, w1, hi);
    lo = pk_fma_w(f32p{sw.x, sw.y}, w2, lo); hi = pk_fma_w(f32p{sw.z, sw.w}, w2, hi);
    lo = pk_fma_w(f32p{se.x, se.y}, w3, lo); hi = pk_fma_w(f32p{se.z, se.w}, w3, hi);
    return f32q{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32q lerp_line_q(f32q lo_, f32q hi_, const float lw[2]) {
    const f32p w0 = wpair(lw[0]), w1 = wpair(lw[1]);
    f32p lo = pk_mul_w(f32p{lo_.x, lo_.y}, w0), hi = pk_mul_w(f32p{lo_.z, lo_.w}, w0);
    lo = pk_fma_w(f32p{hi_.x, hi_.y}, w1, lo); hi = pk_fma_w(f32p{hi_.z, hi_.w}, w1, hi);
    return f32q{lo.x, lo.y, hi.x, hi.y};
}
#else
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
#endif

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

// Ref.forward (models/ref.py:103-152, normals=None) for one ray by EIGHT lanes (`sub` = the lane's index in the group): the per-ray
// part of the head after the two matrix products (bottleneck rows and the ten small-head rows, phase E of the fused kernel) --
// activations, reflection, integrated directional encoding (ref_utils.py:82-112), the specular layer, sigmoid and sRGB.
// `sb` = this ray's LDS row: [0, fc) the bottleneck outputs (bias added), [fc, fc + 10) scratch for the small heads; `F` = the ray's
// feature row; `small` = the head up to bott_w (the four small heads), `tail` = the head from spec_w on (spec_w, spec_b, ide_mat),
// both in LDS.
// The arithmetic is ref_shade_group16's (iff_device.h), operation for operation: that kernel spreads the specular sum of a ray
// over 16 lanes (lane l takes the encoding pairs l, l + 16 and the bottleneck features l + 16 t) and adds the lanes by butterfly
// (xor 1, 2, 4, 8); here lane `sub` carries the partial sums l = sub and l = sub + 8, the butterfly runs over xor 1, 2, 4 on each
// and the two results are added -- the xor-8 step -- so both forms return the same bits.  The three colour channels are
// finished by sub = 0, 1, 2; the return value is this lane's channel (sub < 3).
__device__ __forceinline__ float ref_head_oct(const float* small, const HeadOff& ho, int fc, float* sb, const float* F, const float* tail,
                                             const float d[3], int sub) {
    // the ten small-head rows (normal 0-2, tint 3-5, diffuse 6-8, roughness 9), rows sub and sub + 8 on this lane: the fmaf chain,
    // bias add and activation of ref_shade_group16, computed once per ray and handed round the group through the ray's LDS row
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = sub + 8 * u;
        if (row < 10) {
            const int blk = row / 3, o = row - 3 * blk;                       // blk 0 normal, 1 tint, 2 diffuse, 3 roughness
            const int w_off = blk == 0 ? ho.normal_w : (blk == 1 ? ho.tint_w : (blk == 2 ? ho.diffuse_w : ho.rough_w));
            const int b_off = blk == 0 ? ho.normal_b : (blk == 1 ? ho.tint_b : (blk == 2 ? ho.diffuse_b : ho.rough_b));
            const float* wr = small + w_off + o * 28;
            float acc = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 28; k4 += 4) {
                const f32q w4 = *reinterpret_cast<const f32q*>(wr + k4), f4 = *reinterpret_cast<const f32q*>(F + k4);
                acc = fmaf(w4[0], f4[0], acc); acc = fmaf(w4[1], f4[1], acc); acc = fmaf(w4[2], f4[2], acc);
                acc = fmaf(w4[3], k4 + 3 == 27 ? 0.0f : f4[3], acc);           // column 27 of the row is the shaded flag, not a feature
            }
            const float raw = acc + small[b_off + o];
            const float x = raw + (blk == 2 ? -1.0986122886681098f : -1.0f);     // diffuse: - ln 3; roughness: - 1
            float mine = raw;
            if (blk == 3) mine = softplusf_(x);
            else if (blk != 0) mine = sigmoidf_(blk == 1 ? raw : x);
            sb[fc + row] = mine;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the eight lanes of a ray are lanes of one wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float nr[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) nr[o] = sb[fc + o];
    const int ch = sub < 3 ? sub : 0;                            // this lane's colour channel (sub < 3; the others repeat channel 0 and drop it)
    const float tint_c = sb[fc + 3 + ch], diff_c = sb[fc + 6 + ch];
    const float rough = sb[fc + 9];
    float nn = fmaxf(sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), 1e-12f);
    float n[3] = {-(nr[0] / nn), -(nr[1] / nn), -(nr[2] / nn)};
    float v[3] = {-d[0], -d[1], -d[2]};
    float ndv = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
    float r[3] = {2.0f * ndv * n[0] - v[0], 2.0f * ndv * n[1] - v[1], 2.0f * ndv * n[2] - v[2]};
    float dot = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    const int K = fc + 39, KL = ho.spec_ld;
    const float* spec_w = tail;                                 // [3][KL]
    const float* spec_b = tail + (ho.spec_b - ho.spec_w);
    const float* ide_mat = tail + (ho.ide_mat - ho.spec_w);     // [9][19]
    float part[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float zp[9];
    zp[0] = 1.0f;
#pragma unroll
    for (int k = 1; k < 9; ++k) zp[k] = zp[k - 1] * r[2];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = sub + 8 * u;                              // pairs sub, sub + 8, sub + 16: partial sums l = i & 15
        if (i < 19) {
            const int l = (i < 2) ? 1 : (i < 5) ? 2 : (i < 10) ? 4 : 8;
            const int m = i - ((i < 2) ? 0 : (i < 5) ? 2 : (i < 10) ? 5 : 10);
            float pr = 1.0f, pi = 0.0f;
            for (int q = 0; q < m; ++q) {
                float t = pr * r[0] - pi * r[1];
                pi = pr * r[1] + pi * r[0];
                pr = t;
            }
            float poly = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) poly = fmaf(zp[k], ide_mat[k * 19 + i], poly);
            const float att = expf(-(0.5f * (float)(l * (l + 1))) * rough);
            const float re = pr * poly * att, im = pi * poly * att;
#pragma unroll
            for (int o = 0; o < 3; ++o)
                part[o][u & 1] = fmaf(spec_w[o * KL + fc + 2 * i], re, fmaf(spec_w[o * KL + fc + 2 * i + 1], im, part[o][u & 1]));
        }
    }
    // bottleneck features j = 16 t + sub + 8 u into partial sum u, t ascending.  All LDS reads of a batch of two t are issued
    // before the batch's first fmaf (one read-to-use round trip per batch, not per feature)
#pragma unroll 4
    for (int j0 = 0; j0 < fc; j0 += 32) {
        float b[4], wv[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + 8 * q + sub;                     // q = 2 t' + u
            b[q] = sb[j];
#pragma unroll
            for (int o = 0; o < 3; ++o) wv[o][q] = spec_w[o * KL + j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < 3; ++o) part[o][q & 1] = fmaf(wv[o][q], b[q], part[o][q & 1]);
    }
    float ps[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float first = fmaf(spec_w[o * KL + K - 1], dot, part[o][0]) + spec_b[o];
        float lo = sub == 0 ? first : part[o][0], hi = part[o][1];
        lo += dpp_mov<0xB1>(lo); hi += dpp_mov<0xB1>(hi);       // xor 1
        lo += dpp_mov<0x4E>(lo); hi += dpp_mov<0x4E>(hi);       // xor 2
        lo += xor4_dpp(lo); hi += xor4_dpp(hi);                 // xor 4
        ps[o] = lo + hi;                                        // xor 8
    }
    const float sg = sigmoidf_(ch == 0 ? ps[0] : (ch == 1 ? ps[1] : ps[2]));
    float c = srgbf_(tint_c * sg + diff_c);
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return c * 1.002f - 0.001f;
}

// The same head for one ray by FOUR lanes (`sub` = the lane's index in the quad): the fused kernel's phase E runs it in TWO of the
// tile's four waves (32 ray slots x 4 lanes) and lets the other two leave -- the per-ray part of the head is mostly arithmetic every
// lane of a ray repeats (normalisation, reflection, the powers of r_z, the final sigmoid / sRGB), so a wave-level instruction serves
// 16 rays instead of 8 and the tile issues ~40 % fewer vector instructions for the phase.  Bit for bit ref_head_oct /
// ref_shade_group16: lane `sub` carries the FOUR partial sums l = sub + 4 p of the sixteen (pairs i = l, then l + 16; bottleneck
// features l + 16 t, t ascending), the butterfly runs xor 1, 2 across the quad's lanes on each, and the xor-4 and xor-8 steps are the
// additions (p0 + p1) + (p2 + p3); the ten small-head rows are rows sub, sub + 4, sub + 8.
__device__ __forceinline__ float ref_head_quad(const float* small, const HeadOff& ho, int fc, float* sb, const float* F, const float* tail,
                                              const float d[3], int sub) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int row = sub + 4 * u;
        if (row < 10) {
            const int blk = row / 3, o = row - 3 * blk;
            const int w_off = blk == 0 ? ho.normal_w : (blk == 1 ? ho.tint_w : (blk == 2 ? ho.diffuse_w : ho.rough_w));
            const int b_off = blk == 0 ? ho.normal_b : (blk == 1 ? ho.tint_b : (blk == 2 ? ho.diffuse_b : ho.rough_b));
            const float* wr = small + w_off + o * 28;
            float acc = 0.0f;
#pragma unroll
            for (int k4 = 0; k4 < 28; k4 += 4) {
                const f32q w4 = *reinterpret_cast<const f32q*>(wr + k4), f4 = *reinterpret_cast<const f32q*>(F + k4);
                acc = fmaf(w4[0], f4[0], acc); acc = fmaf(w4[1], f4[1], acc); acc = fmaf(w4[2], f4[2], acc);
                acc = fmaf(w4[3], k4 + 3 == 27 ? 0.0f : f4[3], acc);
            }
            const float raw = acc + small[b_off + o];
            const float x = raw + (blk == 2 ? -1.0986122886681098f : -1.0f);
            float mine = raw;
            if (blk == 3) mine = softplusf_(x);
            else if (blk != 0) mine = sigmoidf_(blk == 1 ? raw : x);
            sb[fc + row] = mine;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the four lanes of a ray are lanes of one wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float nr[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) nr[o] = sb[fc + o];
    const int ch = sub < 3 ? sub : 0;
    const float tint_c = sb[fc + 3 + ch], diff_c = sb[fc + 6 + ch];
    const float rough = sb[fc + 9];
    float nn = fmaxf(sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), 1e-12f);
    float n[3] = {-(nr[0] / nn), -(nr[1] / nn), -(nr[2] / nn)};
    float v[3] = {-d[0], -d[1], -d[2]};
    float ndv = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
    float r[3] = {2.0f * ndv * n[0] - v[0], 2.0f * ndv * n[1] - v[1], 2.0f * ndv * n[2] - v[2]};
    float dot = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    const int K = fc + 39, KL = ho.spec_ld;
    const float* spec_w = tail;
    const float* spec_b = tail + (ho.spec_b - ho.spec_w);
    const float* ide_mat = tail + (ho.ide_mat - ho.spec_w);
    float part[3][4];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) part[o][q] = 0.0f;
    float zp[9];
    zp[0] = 1.0f;
#pragma unroll
    for (int k = 1; k < 9; ++k) zp[k] = zp[k - 1] * r[2];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int i = sub + 4 * u;                              // pairs sub + 4 p (p = u < 4: partial p), then sub + 16 (partial 0)
        if (i < 19) {
            const int l = (i < 2) ? 1 : (i < 5) ? 2 : (i < 10) ? 4 : 8;
            const int m = i - ((i < 2) ? 0 : (i < 5) ? 2 : (i < 10) ? 5 : 10);
            float pr = 1.0f, pi = 0.0f;
            for (int q = 0; q < m; ++q) {
                float t = pr * r[0] - pi * r[1];
                pi = pr * r[1] + pi * r[0];
                pr = t;
            }
            float poly = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) poly = fmaf(zp[k], ide_mat[k * 19 + i], poly);
            const float att = expf(-(0.5f * (float)(l * (l + 1))) * rough);
            const float re = pr * poly * att, im = pi * poly * att;
#pragma unroll
            for (int o = 0; o < 3; ++o)
                part[o][u & 3] = fmaf(spec_w[o * KL + fc + 2 * i], re, fmaf(spec_w[o * KL + fc + 2 * i + 1], im, part[o][u & 3]));
        }
    }
    // bottleneck features j = 16 t + sub + 4 p into partial sum p, t ascending; the LDS reads of a t are issued before its fmafs
#pragma unroll 4
    for (int j0 = 0; j0 < fc; j0 += 16) {
        float b[4], wv[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + 4 * q + sub;
            b[q] = sb[j];
#pragma unroll
            for (int o = 0; o < 3; ++o) wv[o][q] = spec_w[o * KL + j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < 3; ++o) part[o][q] = fmaf(wv[o][q], b[q], part[o][q]);
    }
    float ps[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float first = fmaf(spec_w[o * KL + K - 1], dot, part[o][0]) + spec_b[o];
        float t[4] = {sub == 0 ? first : part[o][0], part[o][1], part[o][2], part[o][3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t[q] += dpp_mov<0xB1>(t[q]);                        // xor 1
            t[q] += dpp_mov<0x4E>(t[q]);                        // xor 2
        }
        ps[o] = (t[0] + t[1]) + (t[2] + t[3]);                  // xor 4, xor 8
    }
    const float sg = sigmoidf_(ch == 0 ? ps[0] : (ch == 1 ? ps[1] : ps[2]));
    float c = srgbf_(tint_c * sg + diff_c);
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return c * 1.002f - 0.001f;
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
    // Two experiment builds behind DESIGN.md section 4 ("where it stands"); never defined in the product:
    //   -DFAN_PAD_LDS=bytes  LDS nobody uses, so that fewer tiles fit a CU (30000: two, 50000: one): 0.98 / 1.30 / 2.35 ms per launch
    //   -DFAN_SLEEP=n        every wave idles n x 64 clocks first (holds its slot, uses no unit): +4.1 k / 8.2 k / 16.4 k clocks on a
    //                        tile of 89.8 k cost +3.5 / 6.6 / 12.2 % -- three quarters of what pure latency-boundness would cost
#ifdef FAN_PAD_LDS
    __shared__ int s_pad[FAN_PAD_LDS / 4];
    if (a.R < 0) s_pad[threadIdx.x] = 1;
    if (a.R < -1) a.counts[0] = s_pad[(threadIdx.x * 7) % (FAN_PAD_LDS / 4)];
#endif
#ifdef FAN_SLEEP
#pragma unroll
    for (int i = 0; i < FAN_SLEEP / 64; ++i) __builtin_amdgcn_s_sleep(64);
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = tid >> 3, h = (tid >> 2) & 1, c = tid & 3;     // ray of the tile, sub-group, texel quarter
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
        const int l8 = tid & 7;
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
            for (int r = 0; r < 2; ++r) *reinterpret_cast<f32q*>(s_patch + i * PLANE16 + 4 * (tid + NT * r)) = pre[2 * i + r];
        if (wave < 3) {
            *reinterpret_cast<f32q*>(s_patch + wave * PLANE16 + 4 * (2 * NT + lane)) = pre[6];
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
        const int q4 = tid >> 2;
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
                        const float* P = s_patch + i * PLANE16 + ((rv.r[ax_b] * FP + rv.r[ax_a]) * 16 + 4 * c);
                        const int da = rv.d[ax_a] * 16, db = rv.d[ax_b] * (FP * 16);
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
        const bool writer = live && (tid & 7) == 0;
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
            for (int r = 0; r < 6; ++r) *reinterpret_cast<f32q*>(s_patch + 4 * (tid + NT * r)) = pre[r];
            if (tid < FP * FP * 12 - 6 * NT) *reinterpret_cast<f32q*>(s_patch + 4 * (tid + NT * 6)) = pre[6];
            if (tid < FP * 12) *reinterpret_cast<f32q*>(s_patch + PLANE48 + 4 * tid) = pre[7];
            __syncthreads();
            if (i < 2) fetch_app(i + 1);                  // the next plane (registers) under this plane's arithmetic
            else fetch_basis();                           // basis_mat for phase D
            __builtin_amdgcn_sched_barrier(0);
            STAMP(6 + 2 * i);
            const int ax_a = mat_a(i), ax_b = mat_b(i), ax_v = vec_ax(i);
            auto accumulate = [&](float (&acc)[36]) {
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
                    acc[12 * j + 4 * i + 0] = fmaf(rv.w, pr.x, acc[12 * j + 4 * i + 0]);
                    acc[12 * j + 4 * i + 1] = fmaf(rv.w, pr.y, acc[12 * j + 4 * i + 1]);
                    acc[12 * j + 4 * i + 2] = fmaf(rv.w, pr.z, acc[12 * j + 4 * i + 2]);
                    acc[12 * j + 4 * i + 3] = fmaf(rv.w, pr.w, acc[12 * j + 4 * i + 3]);
                }
            }
            };
            accumulate(accp);
#ifdef FAN_CHECK_TWICE
            {
                float accq[36];
#pragma unroll
                for (int q = 0; q < 36; ++q) accq[q] = 0.0f;
                accumulate(accq);
                bool bad = false;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) bad = bad || (accq[12 * j + 4 * i + e4] != accp[12 * j + 4 * i + e4]);
                if (bad) accp[4 * i] = __builtin_nanf("");        // poison: the ray's features become NaN, its colour 0 (the final clamp)
            }
#endif
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
    for (int i = 0; i < 36; ++i) accp[i] = accp[i] + xor4_dpp(accp[i]);
    __syncthreads();                                   // every wave is done with the patch and the records
    if (grp_on && h == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) s_A[(48 * i + 16 * j + 4 * c + e4) * DLD + g] = accp[12 * j + 4 * i + e4];
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
    if ((tid & 7) == 0 && grp_on) s_feat[g * 28 + 27] = any ? 1.0f : 0.0f;
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
#ifdef FAN_ELIGIBLE_ALWAYS          // experiment builds: every point-centred march through the fan kernel (its gather path when a box does not fit)
    return true;
#endif
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
