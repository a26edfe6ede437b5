// fan_common.h -- device helpers shared by the two fused fan-march kernels (fan_march_kernels.hip: four waves per 27-ray fan,
// register-staged patches; fan8_march_kernels.hip: eight waves per fan, patches by global -> LDS DMA): cross-lane moves as DPP
// modifiers, the tap combination in the lookup functions' operation order, the per-ray part of the Ref head.
#pragma once
#include "iff_device.h"

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
// hand-placed v_pk_mul_f32 / v_pk_fma_f32 in eight operand forms (round 4's experiment builds, since removed) show 0 events in 6 400 checked steps, next to 18 in 2 480 for the compiler's own packing in the same runs.  So the
// trigger is the compiler's schedule of packed fp32 code next to MFMA work, not an operand form one could avoid by hand; no packed
// fp32 instruction is the rule that holds for all ~8 000 the compiler had placed in this library, and it costs nothing measurable
// (14 590-14 760 poses/s without, 14 590-14 810 with, same box, same run).
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

