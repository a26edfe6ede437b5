// iff_device.h -- device-side view of a field handle and the lookup primitives shared by the kernels.
// gfx950 only (wave64).  Compiled with -ffp-contract=off: every fused multiply-add below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IFF_WAVE 64

// VM index conventions of the reference: models/tensorBase.py:311-312
//   plane i spans axes (a,b) = matMode[i] = (0,1),(0,2),(1,2) with W<->a, H<->b ; line i runs along vecMode[i] = 2,1,0
__host__ __device__ constexpr int mat_a(int i) { return i == 2 ? 1 : 0; }
__host__ __device__ constexpr int mat_b(int i) { return i == 0 ? 1 : 2; }
__host__ __device__ constexpr int vec_ax(int i) { return 2 - i; }

struct FieldDev {
    // channels-last tables built by K0 (field_relayout): plane i [G_b][G_a][C], line i [G_v][C]
    const float* dplane[3];
    const float* dline[3];
    const float* aplane[3];
    const float* aline[3];
    const float* basis_l;       // basis_mat re-laid [app_dim][4][3*n_app/4] (lane-slice major), see field_kernels.hip
    const float* basis_l12;     // basis_mat re-laid [app_dim][n_app/4][12] for the 12-lanes-per-point gather (K4b)
    const float* basis;         // basis_mat as given [app_dim][3*n_app]
    const uint8_t* mask;        // [D][H][W] bytes in {0,1}, or nullptr
    const uint8_t* cell;        // [D+1][H+1][W+1] corner bits of the mask cell (z0, y0, x0) = index - 1: bit dx + 2 dy + 4 dz is set
                                // when corner (x0+dx, y0+dy, z0+dz) lies inside the volume and is occupied (mask_occupied below)
    const float* head;          // packed Ref head, offsets below
    int grid[3];
    int mask_dims[3];           // D,H,W
    float aabb_lo[3], aabb_hi[3], inv_aabb[3];        // inv_aabb = 2/size   (tensorBase.py:358)
    float mask_lo[3], mask_hi[3], mask_inv[3];        // mask_inv = (1/size)*2 (tensorBase.py:59)
    float density_shift, distance_scale, weight_thres, step_size, near, far;
    int n_samples, softplus, unisphere;
    int density_lanes;          // 0 auto, 1 or 4 forced (iff_field_desc.density_lanes)
    int head_lanes;             // 0 auto, 16 = the vector form of the Ref head launches (iff_field_desc.head_lanes)
    int sampler_persistent;     // 1: the surface sampler as one persistent launch (iff_field_desc.sampler_persistent)
    int fan_waves;              // 0 auto, 4 / 8: the fused fan kernel that serves the point-centred march (iff_field_desc.fan_waves)
    int n_density, n_app, app_dim, feature_c;
};

// Ref head packing (floats): nn.Linear weights row-major [out][ld] with rows zero-padded to a multiple of 4 floats
// (ld = 28 for app_dim 27, spec_ld = 168 for 128 + 39), every block 16-B aligned, so a row is read as float4s.
struct HeadOff {
    int normal_w, normal_b, tint_w, tint_b, rough_w, rough_b, diffuse_w, diffuse_b, bott_w, bott_b, spec_w, spec_b,
        ide_mat, total, ld, spec_ld;
};
__host__ __device__ inline HeadOff head_offsets(int app_dim, int feature_c) {
    HeadOff o;
    o.ld = (app_dim + 3) & ~3;
    o.spec_ld = (feature_c + 39 + 3) & ~3;
    int p = 0;
    o.normal_w = p; p += 3 * o.ld; o.normal_b = p; p += 4;
    o.tint_w = p; p += 3 * o.ld; o.tint_b = p; p += 4;
    o.rough_w = p; p += o.ld; o.rough_b = p; p += 4;
    o.diffuse_w = p; p += 3 * o.ld; o.diffuse_b = p; p += 4;
    o.bott_w = p; p += feature_c * o.ld; o.bott_b = p; p += (feature_c + 3) & ~3;
    o.spec_w = p; p += 3 * o.spec_ld; o.spec_b = p; p += 4;
    o.ide_mat = p; p += (9 * 19 + 3) & ~3;
    o.total = p;
    return o;
}

// ---------------------------------------------------------------------------------------- coordinates
// utils.py:139-146 power_transformation(x, alpha=-1.5): sign(x) * (2.5/-1.5) * ((|x|/2.5 + 1)^-1.5 - 1)
// t^-1.5 (t >= 1) as 1 / (t sqrt(t)): a correctly rounded square root, one product, one correctly rounded quotient (<= 1.5 ulp)
// in place of powf's log2 / exp2 chain (~1 ulp, three times the instructions; the reference's torch.pow and any libm differ from
// both in the last bit or two: tests/test_hip_fullsize.py coord_tol).  bicycle64k: K4b 1 432 -> 1 268 us, K4a 813 -> 770, the
// sampler's iteration 21 -> 19.9; poses/s +4 %.
__device__ inline float contract_power(float x) {
    const float na = 2.5f, alpha = -1.5f;
    float m = fabsf(x);
    float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    const float t = (m / na) + 1.0f;
    return sgn * (na / alpha) * (1.0f / (t * sqrtf(t)) - 1.0f);
}

// TensorBase.normalize_coord, tensorBase.py:389-397
__device__ inline void field_normalize(const FieldDev& f, const float p[3], float out[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (f.unisphere) {
            float c = (f.aabb_lo[a] + f.aabb_hi[a]) / 2.0f;
            out[a] = contract_power(p[a] - c);
        } else {
            out[a] = (p[a] - f.aabb_lo[a]) * f.inv_aabb[a] - 1.0f;
        }
    }
}

// AlphaGridMask.normalize_coord, tensorBase.py:74-83
__device__ inline void mask_normalize(const FieldDev& f, const float p[3], float out[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (f.unisphere) {
            float c = (f.mask_lo[a] + f.mask_hi[a]) / 2.0f;
            out[a] = contract_power(p[a] - c);
        } else {
            out[a] = (p[a] - f.mask_lo[a]) * f.mask_inv[a] - 1.0f;
        }
    }
}

__device__ inline bool inside_aabb(const FieldDev& f, const float p[3]) {
    // tensorBase.py:634-636: outside = (aabb[0] > p) | (p > aabb[1]) on any axis
    bool out = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) out = out || (f.aabb_lo[a] > p[a]) || (p[a] > f.aabb_hi[a]);
    return !out;
}

// align_corners=True un-normalisation of F.grid_sample: ((c + 1) / 2) * (size - 1)
__device__ inline float unnorm(float c, int size) { return ((c + 1.0f) / 2.0f) * (float)(size - 1); }

// 3-D trilinear read of the {0,1} occupancy bytes with zero padding: the value F.grid_sample returns for
// tensorBase.py:66-72 (corner order and weight products as ATen's grid_sampler_3d).
__device__ inline float mask_value_at(const FieldDev& f, const float g[3]) {
    const int W = f.mask_dims[2], H = f.mask_dims[1], D = f.mask_dims[0];
    float ix = unnorm(g[0], W), iy = unnorm(g[1], H), iz = unnorm(g[2], D);
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    // weights of the low corner: (i0 + 1 - i), of the high corner: (i - i0)
    float wx[2] = {(fx + 1.0f) - ix, ix - fx};
    float wy[2] = {(fy + 1.0f) - iy, iy - fy};
    float wz[2] = {(fz + 1.0f) - iz, iz - fz};
    // NaN / huge coordinates: every corner is out of range -> 0
    if (!(ix > -2.0f && ix < (float)(W + 1) && iy > -2.0f && iy < (float)(H + 1) && iz > -2.0f && iz < (float)(D + 1)))
        return 0.0f;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float acc = 0.0f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                bool in = (x >= 0) && (x < W) && (y >= 0) && (y < H) && (z >= 0) && (z < D);
                float v = 0.0f;
                if (in) v = (float)f.mask[((size_t)z * H + y) * W + x];
                acc = acc + v * (wx[dx] * wy[dy] * wz[dz]);
            }
    return acc;
}
__device__ inline float mask_value(const FieldDev& f, const float p[3]) {
    float g[3];
    mask_normalize(f, p, g);
    return mask_value_at(f, g);
}
// the same with the point's field-normalised coordinate at hand: under unisphere contraction both normalisations are
// contract_power(p - centre of the box) (three divisions and a square root per axis), and the mask's box is the field's unless the
// model was shrunk after its mask was made -- equal centres, equal bits, computed once
__device__ inline float mask_value(const FieldDev& f, const float p[3], const float xn[3]) {
    float g[3];
    if (f.unisphere) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float cm = (f.mask_lo[a] + f.mask_hi[a]) / 2.0f, cf = (f.aabb_lo[a] + f.aabb_hi[a]) / 2.0f;
            if (cm == cf) g[a] = xn[a];                    // wave-uniform
            else g[a] = contract_power(p[a] - cm);
        }
    } else {
        mask_normalize(f, p, g);
    }
    return mask_value_at(f, g);
}

// `mask_value(...) > 0` -- the only use the march, compute_alpha and the surface sampler make of the occupancy lookup
// (tensorBase.py:66-72 + :764, :832: alpha_mask = sample_alpha(xyz) > 0) -- from ONE byte of the corner-bit table instead of eight
// loads and a trilinear sum.  Exact: the mask holds {0,1} and every weight is >= 0, so the sum is positive iff some in-range
// occupied corner has a positive weight product.  A low-corner weight (i0 + 1) - i is always positive; a high-corner weight
// i - i0 is zero exactly when the coordinate sits on a texel.  Non-zero weights are >= 2^-25 (the coordinate (g + 1) / 2 (size - 1)
// is a multiple of 2^-25 near 0 and has an ulp >= 2^-24 elsewhere), so a product of three never underflows to zero and
// "product > 0" is "all three factors > 0": the corners that count are those of bit set (dx or frac_x > 0) and (dy or ...) and
// (dz or ...).  Coordinates for which mask_value_at bails out (NaN, beyond one texel outside) index no cell: false.
__device__ inline bool mask_occupied_at(const FieldDev& f, const float g[3]) {
    const int W = f.mask_dims[2], H = f.mask_dims[1], D = f.mask_dims[0];
    const float ix = unnorm(g[0], W), iy = unnorm(g[1], H), iz = unnorm(g[2], D);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    // cell index = i0 + 1 in [0, size]: i0 = -1 (low corner outside) .. size - 1 (high corner outside); NaN fails every comparison
    const bool ok = (fx >= -1.0f) && (fx <= (float)(W - 1)) && (fy >= -1.0f) && (fy <= (float)(H - 1)) && (fz >= -1.0f) && (fz <= (float)(D - 1));
    // the load is unconditional (cell 0 when the point indexes none): callers issue it together with their table taps
    const int xi = ok ? (int)fx + 1 : 0, yi = ok ? (int)fy + 1 : 0, zi = ok ? (int)fz + 1 : 0;
    const unsigned bits = f.cell[(unsigned)((zi * (H + 1) + yi) * (W + 1) + xi)];          // < 2^31 cells: checked at create
    const unsigned allow = (ix > fx ? 0xFFu : 0x55u) & (iy > fy ? 0xFFu : 0x33u) & (iz > fz ? 0xFFu : 0x0Fu);
    return ok && (bits & allow) != 0u;
}
__device__ inline bool mask_occupied(const FieldDev& f, const float p[3]) {
    float g[3];
    mask_normalize(f, p, g);
    return mask_occupied_at(f, g);
}
// with the point's field-normalised coordinate at hand (see mask_value(f, p, xn) above)
__device__ inline bool mask_occupied(const FieldDev& f, const float p[3], const float xn[3]) {
    float g[3];
    if (f.unisphere) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float cm = (f.mask_lo[a] + f.mask_hi[a]) / 2.0f, cf = (f.aabb_lo[a] + f.aabb_hi[a]) / 2.0f;
            if (cm == cf) g[a] = xn[a];                    // wave-uniform
            else g[a] = contract_power(p[a] - cm);
        }
    } else {
        mask_normalize(f, p, g);
    }
    return mask_occupied_at(f, g);
}

// ---------------------------------------------------------------------------------------- VM addressing
// Bilinear tap set of one plane (align_corners=True, zero padding): 4 texel offsets (clamped) + weights
// (zeroed when the corner is out of range), and the 2-tap linear set of the matching line.
struct Taps {
    int   p_off[4];   // texel index (y*W + x) of nw, ne, sw, se
    float p_w[4];
    int   l_off[2];   // line texel index
    float l_w[2];
};

// One axis of a lookup: the two texel indices (clamped into the table) and their weights (zero for an index outside it:
// grid_sample's zero padding).  NaN / far-out coordinates get two out-of-range taps.
struct AxisTap { int i[2]; float w[2]; };
// a * b + c with the 24-bit multiplier (full rate; a 32-bit v_mul_lo is quarter rate and LLVM turns __umul24 by a scalar
// back into one).  a, b in [0, 2^24), b wave-uniform.
__device__ inline int mad24_uniform(int a, int b, int c) {
    int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ inline AxisTap axis_tap(float xn_ax, int size) {
    const float x = unnorm(xn_ax, size);
    const float fx = floorf(x);
    const float w1 = x - fx, w0 = 1.0f - w1;
    const bool ok = (x > -2.0f) && (x < (float)(size + 1));     // keeps the int conversion defined
    const int i0 = ok ? (int)fx : -2, i1 = i0 + 1;
    AxisTap t;
    t.i[0] = min(max(i0, 0), size - 1);
    t.i[1] = min(max(i1, 0), size - 1);
    t.w[0] = (i0 >= 0 && i0 < size) ? w0 : 0.0f;
    t.w[1] = (i1 >= 0 && i1 < size) ? w1 : 0.0f;
    return t;
}

// The masking is per axis (a tap's weight is the product of its axes' weights, so one zero factor zeroes it) and the texel
// index uses the 24-bit multiplier (indices < 2^24: checked at create) -- a 32-bit v_mul_lo is a quarter-rate instruction.
__device__ inline void make_taps(const FieldDev& f, const float xn[3], int i, Taps& t) {
    const int a = mat_a(i), b = mat_b(i), v = vec_ax(i);
    const int W = f.grid[a];
    const AxisTap tx = axis_tap(xn[a], W), ty = axis_tap(xn[b], f.grid[b]), tz = axis_tap(xn[v], f.grid[v]);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            t.p_off[dy * 2 + dx] = mad24_uniform(ty.i[dy], W, tx.i[dx]);
            t.p_w[dy * 2 + dx] = ty.w[dy] * tx.w[dx];
        }
    // the line is a width-1 image sampled at x = 0 (tensoRF.py:225): x tap 0 has weight 1, tap 1 is out of range
#pragma unroll
    for (int dz = 0; dz < 2; ++dz) { t.l_off[dz] = tz.i[dz]; t.l_w[dz] = tz.w[dz]; }
}

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// channels [ch, ch + 4) of texel `texel` of a channels-last table with C channels: the 32-bit byte offset texel * 4C + 4ch
// in one 24-bit multiply-add (texel < 2^24, table < 4 GiB: checked at create); the table base stays in scalar registers
// (global_load ... v_off, s[base]) -- no 64-bit vector address arithmetic
__device__ inline float4 ld4_tex(const float* tab, int texel, int C, int ch) {
    const unsigned off = (unsigned)mad24_uniform(texel, 4 * C, 4 * ch);
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tab) + off);
}

__device__ inline float4 lerp_plane4(const float* tab, int C, const Taps& t, int ch) {
    float4 nw = ld4_tex(tab, t.p_off[0], C, ch), ne = ld4_tex(tab, t.p_off[1], C, ch);
    float4 sw = ld4_tex(tab, t.p_off[2], C, ch), se = ld4_tex(tab, t.p_off[3], C, ch);
    // one multiply and three fused multiply-adds per channel (every kernel shares this form, so their results agree bit for bit)
    float4 r;
    r.x = fmaf(se.x, t.p_w[3], fmaf(sw.x, t.p_w[2], fmaf(ne.x, t.p_w[1], nw.x * t.p_w[0])));
    r.y = fmaf(se.y, t.p_w[3], fmaf(sw.y, t.p_w[2], fmaf(ne.y, t.p_w[1], nw.y * t.p_w[0])));
    r.z = fmaf(se.z, t.p_w[3], fmaf(sw.z, t.p_w[2], fmaf(ne.z, t.p_w[1], nw.z * t.p_w[0])));
    r.w = fmaf(se.w, t.p_w[3], fmaf(sw.w, t.p_w[2], fmaf(ne.w, t.p_w[1], nw.w * t.p_w[0])));
    return r;
}

__device__ inline float4 lerp_line4(const float* tab, int C, const Taps& t, int ch) {
    float4 lo = ld4_tex(tab, t.l_off[0], C, ch), hi = ld4_tex(tab, t.l_off[1], C, ch);
    float4 r;
    r.x = fmaf(hi.x, t.l_w[1], lo.x * t.l_w[0]);
    r.y = fmaf(hi.y, t.l_w[1], lo.y * t.l_w[0]);
    r.z = fmaf(hi.z, t.l_w[1], lo.z * t.l_w[0]);
    r.w = fmaf(hi.w, t.l_w[1], lo.w * t.l_w[0]);
    return r;
}

// Partial density feature of one point for the calling lane's channel slice.
// `sub` in [0,4): the lane owns channels {4*sub + 16*j + e}; with n_density == 16 that is one float4 per texel,
// so the 4 lanes of a point read one contiguous 64-B texel per tap.  Sum over the 4 lanes = the feature.
__device__ inline float density_partial(const FieldDev& f, const float xn[3], int sub) {
    float acc = 0.0f;
    const int C = f.n_density;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Taps t;
        make_taps(f, xn, i, t);
        for (int ch = 4 * sub; ch < C; ch += 16) {
            float4 p = lerp_plane4(f.dplane[i], C, t, ch);
            float4 l = lerp_line4(f.dline[i], C, t, ch);
            acc = fmaf(p.w, l.w, fmaf(p.z, l.z, fmaf(p.y, l.y, fmaf(p.x, l.x, acc))));
        }
    }
    return acc;
}

// The whole density feature by ONE lane (the taps of a plane are computed once for all 16 channels instead of once per
// channel quarter), in the summation order of the 4-lane form: the four quarter partials are formed separately, each
// over plane 0, 1, 2 in turn, and added the way the xor butterfly of sum4 adds them, (p0 + p1) + (p2 + p3) -- the
// result is bit-identical to sum4(density_partial(.., sub)).
__device__ inline float density_full(const FieldDev& f, const float xn[3]) {
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int C = f.n_density;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Taps t;
        make_taps(f, xn, i, t);
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
            for (int ch = 4 * sub; ch < C; ch += 16) {
                float4 p = lerp_plane4(f.dplane[i], C, t, ch);
                float4 l = lerp_line4(f.dline[i], C, t, ch);
                part[sub] = fmaf(p.w, l.w, fmaf(p.z, l.z, fmaf(p.y, l.y, fmaf(p.x, l.x, part[sub]))));
            }
    }
    return (part[0] + part[1]) + (part[2] + part[3]);
}

// xor-butterfly sum over the 4 lanes that share a point (lanes 4k..4k+3)
__device__ inline float sum4(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}

// tensorBase.py:750-754 ; torch softplus: beta 1, threshold 20
__device__ inline float feature2density(const FieldDev& f, float feat) {
    if (f.softplus) {
        float x = feat + f.density_shift;
        return (x > 20.0f) ? x : log1pf(expf(x));
    }
    return fmaxf(feat, 0.0f);
}

// Appearance plane*line products of one point for the calling lane's channel slice:
// prod[i*(C/4) + 4*j + e] = plane_i[ch] * line_i[ch], ch = 16*j + 4*sub + e  (tensoRF.py:248-256 before basis_mat).
// NPL = n_app / 4 products per plane per lane (12 for n_app = 48).
template <int NPL>
__device__ inline void app_products_slice(const FieldDev& f, const float xn[3], int sub, float* prod) {
    const int C = f.n_app;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Taps t;
        make_taps(f, xn, i, t);
#pragma unroll
        for (int j = 0; j < NPL / 4; ++j) {
            int ch = 16 * j + 4 * sub;
            float4 p = lerp_plane4(f.aplane[i], C, t, ch);
            float4 l = lerp_line4(f.aline[i], C, t, ch);
            prod[i * NPL + 4 * j + 0] = p.x * l.x;
            prod[i * NPL + 4 * j + 1] = p.y * l.y;
            prod[i * NPL + 4 * j + 2] = p.z * l.z;
            prod[i * NPL + 4 * j + 3] = p.w * l.w;
        }
    }
}

// Appearance products with one lane per 16-B texel quarter: lane c in [0, n_app/4) owns channels 4c..4c+3 of every
// plane, so the n_app/4 lanes of a point read one contiguous 192-B texel per tap and each lane keeps only 12 products:
// prod[4*i + e] = plane_i[4c+e] * line_i[4c+e].
__device__ inline void app_products_lane(const FieldDev& f, const float xn[3], int c, float prod[12]) {
    const int C = f.n_app;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Taps t;
        make_taps(f, xn, i, t);
        float4 p = lerp_plane4(f.aplane[i], C, t, 4 * c);
        float4 l = lerp_line4(f.aline[i], C, t, 4 * c);
        prod[4 * i + 0] = p.x * l.x; prod[4 * i + 1] = p.y * l.y; prod[4 * i + 2] = p.z * l.z; prod[4 * i + 3] = p.w * l.w;
    }
}

__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ inline float softplusf_(float x) { return (x > 20.0f) ? x : log1pf(expf(x)); }

// models/image.py:6-13 linear_to_srgb (eps = float32 machine epsilon)
__device__ inline float srgbf_(float lin) {
    float lo = 12.92f * lin;
    float hi = (211.0f * powf(fmaxf(lin, 1.1920928955078125e-07f), 0.4166666567325592f) - 11.0f) / 200.0f;
    return (lin <= 0.0031308f) ? lo : hi;
}

__device__ inline float sum16(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// dot of one weight row with F.  Scalar indexing on purpose: rows whose address is wave-uniform (the small heads) are
// then fetched with scalar loads into SGPRs and cost no vector registers; float4 reads of the same rows made the
// scheduler keep every row in VGPRs at once (300+ registers, one wave per SIMD).
template <int LD>
__device__ inline float row_dot(const float* __restrict__ row, const float* F) {
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < LD; ++k) a = fmaf(row[k], F[k], a);
    return a;
}

// Ref.forward (models/ref.py:103-152, normals=None) evaluated by a group of 16 consecutive lanes for one ray.
// Every lane passes the same F[LD] (app_dim features, zero padded to LD = 28) and d[3]; `l16` is the lane's index in the
// group; `head` is the packed head (LDS or global).  All 16 lanes return the rgb triple.
// pointer into LDS that keeps its address space through pointer arithmetic (reads compile to ds_read, not flat_load)
typedef float f32q __attribute__((ext_vector_type(4)));       // a plain 16-B vector (HIP's float4 class cannot be read from LDS-typed pointers)
typedef const __attribute__((address_space(3))) float* lds_cfloat_p;
typedef const __attribute__((address_space(3))) f32q* lds_cfloat4_p;
template <typename P> struct quad_ptr { typedef const f32q* type; };
template <> struct quad_ptr<lds_cfloat_p> { typedef lds_cfloat4_p type; };

// HP: `const float*` (head in global memory) or `lds_cfloat_p` (head staged in LDS by the caller)
template <int LD, typename HP>
__device__ inline void ref_shade_group16(HP head, const HeadOff& ho, int feature_c, const float* F,
                                         const float d[3], int l16, float rgb[3]) {
    // small heads: ten rows (normal 0-2, tint 3-5, diffuse 6-8, roughness 9), ONE per lane -- each value is the same fmaf
    // chain, bias add and activation as before, computed once per ray instead of sixteen times and handed round the group
    float nr[3], tint[3], diff[3], rough;
    {
        const int row = l16 < 10 ? l16 : 9;
        const int blk = row / 3, o = row - 3 * blk;                       // blk 0 normal, 1 tint, 2 diffuse, 3 roughness
        const int w_off = blk == 0 ? ho.normal_w : (blk == 1 ? ho.tint_w : (blk == 2 ? ho.diffuse_w : ho.rough_w));
        const int b_off = blk == 0 ? ho.normal_b : (blk == 1 ? ho.tint_b : (blk == 2 ? ho.diffuse_b : ho.rough_b));
        const HP wr = head + w_off + o * LD;
        float acc = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < LD; k4 += 4) {
            const f32q w4 = *(typename quad_ptr<HP>::type)(wr + k4);
            acc = fmaf(w4.x, F[k4], acc); acc = fmaf(w4.y, F[k4 + 1], acc); acc = fmaf(w4.z, F[k4 + 2], acc); acc = fmaf(w4.w, F[k4 + 3], acc);
        }
        const float raw = acc + head[b_off + o];
        const float x = raw + (blk == 2 ? -1.0986122886681098f : -1.0f);     // diffuse: - ln 3; roughness: - 1
        const float sg = sigmoidf_(blk == 1 ? raw : x), sp = softplusf_(x);
        const float mine = blk == 0 ? raw : (blk == 3 ? sp : sg);
        const int g0 = (threadIdx.x & 63) & ~15;                            // first lane of this ray's group inside the wave
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            nr[q] = __shfl(mine, g0 + q, 64);
            tint[q] = __shfl(mine, g0 + 3 + q, 64);
            diff[q] = __shfl(mine, g0 + 6 + q, 64);
        }
        rough = __shfl(mine, g0 + 9, 64);
    }
    float nn = fmaxf(sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), 1e-12f);
    float n[3] = {-(nr[0] / nn), -(nr[1] / nn), -(nr[2] / nn)};   // normal_mlp: normalise then * -1
    float v[3] = {-d[0], -d[1], -d[2]};
    float ndv = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
    float r[3] = {2.0f * ndv * n[0] - v[0], 2.0f * ndv * n[1] - v[1], 2.0f * ndv * n[2] - v[2]};   // ref_utils.py:18
    float dot = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    const int K = feature_c + 39, KL = ho.spec_ld;
    float part[3] = {0.f, 0.f, 0.f};
    // integrated directional encoding (ref_utils.py:82-112): pairs i = l16, l16 + 16
    float zp[9];
    zp[0] = 1.0f;
#pragma unroll
    for (int k = 1; k < 9; ++k) zp[k] = zp[k - 1] * r[2];
    for (int i = l16; i < 19; i += 16) {
        // (m, l) of pair i: l = 1,2,4,8 with m = 0..l  -> offsets 0,2,5,10
        int l = (i < 2) ? 1 : (i < 5) ? 2 : (i < 10) ? 4 : 8;
        int m = i - ((i < 2) ? 0 : (i < 5) ? 2 : (i < 10) ? 5 : 10);
        float pr = 1.0f, pi = 0.0f;
        for (int q = 0; q < m; ++q) {
            float t = pr * r[0] - pi * r[1];
            pi = pr * r[1] + pi * r[0];
            pr = t;
        }
        float poly = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) poly = fmaf(zp[k], head[ho.ide_mat + k * 19 + i], poly);
        float att = expf(-(0.5f * (float)(l * (l + 1))) * rough);
        float re = pr * poly * att, im = pi * poly * att;
#pragma unroll
        for (int o = 0; o < 3; ++o)
            part[o] = fmaf(head[ho.spec_w + o * KL + feature_c + 2 * i], re,
                           fmaf(head[ho.spec_w + o * KL + feature_c + 2 * i + 1], im, part[o]));
    }
    // bottleneck slice j = l16 + 16 t
    // (the rows differ per lane, so these are vector loads: 16 B at a time -- one load per weight made this kernel bound by
    // load instructions, ~250 per wave)
#pragma unroll 1
    for (int j = l16; j < feature_c; j += 16) {
        const HP wr = head + ho.bott_w + j * LD;
        float b = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < LD; k4 += 4) {
            const f32q w4 = *(typename quad_ptr<HP>::type)(wr + k4);
            b = fmaf(w4.x, F[k4], b); b = fmaf(w4.y, F[k4 + 1], b); b = fmaf(w4.z, F[k4 + 2], b); b = fmaf(w4.w, F[k4 + 3], b);
        }
        b = b + head[ho.bott_b + j];
#pragma unroll
        for (int o = 0; o < 3; ++o) part[o] = fmaf(head[ho.spec_w + o * KL + j], b, part[o]);
    }
    if (l16 == 0) {
#pragma unroll
        for (int o = 0; o < 3; ++o) part[o] = fmaf(head[ho.spec_w + o * KL + K - 1], dot, part[o]) + head[ho.spec_b + o];
    }
    // the three colour channels on three lanes: one pass of sigmoid + sRGB (a powf) through the instruction stream instead
    // of three; lane o of the group finishes channel o, the results are handed round (same operations per channel: same bits)
    float ps[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) ps[o] = sum16(part[o]);
    {
        const int ch = l16 < 3 ? l16 : 0;
        const float p_ = ch == 0 ? ps[0] : (ch == 1 ? ps[1] : ps[2]);
        const float t_ = ch == 0 ? tint[0] : (ch == 1 ? tint[1] : tint[2]);
        const float d_ = ch == 0 ? diff[0] : (ch == 1 ? diff[1] : diff[2]);
        const float s = sigmoidf_(p_);
        float c = srgbf_(t_ * s + d_);
        c = fminf(fmaxf(c, 0.0f), 1.0f);
        c = c * 1.002f - 0.001f;
        const int g0 = (threadIdx.x & 63) & ~15;
#pragma unroll
        for (int o = 0; o < 3; ++o) rgb[o] = __shfl(c, g0 + o, 64);
    }
}
