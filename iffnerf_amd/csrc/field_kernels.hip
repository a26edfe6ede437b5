// field_kernels.hip -- K0 table re-layout and the per-point lookups (K1 point_alpha, K2 point_normals, ...).
// Hand-written for gfx950 (wave64).  Each point is served by 4 consecutive lanes: one lane per 16-byte quarter of a
// 64-byte (16-channel fp32) texel, so a tap is one contiguous 64-B read per point and a wave keeps 16 points x 18
// taps in flight.
#include "iff_device.h"
#include "iff_launch.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------ K0
// [C][H*W] -> [H*W][C]
__global__ void k0_channels_last(const float* __restrict__ src, float* __restrict__ dst, int C, int64_t HW) {
    int64_t n = (int64_t)C * HW;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t px = t / C;
        int c = (int)(t - px * C);
        dst[t] = src[(int64_t)c * HW + px];
    }
}

__global__ void k0_mask_bytes(const float* __restrict__ src, uint8_t* __restrict__ dst, int64_t n) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        dst[t] = (src[t] > 0.0f) ? 1 : 0;
}

// mask bytes [D][H][W] -> corner bits [D+1][H+1][W+1] (FieldDev::cell, iff_device.h mask_occupied_at): entry (zi, yi, xi) describes
// the cell whose low corner is (xi - 1, yi - 1, zi - 1)
__global__ void k0_mask_cells(const uint8_t* __restrict__ mask, uint8_t* __restrict__ cell, int D, int H, int W) {
    const int64_t n = (int64_t)(D + 1) * (H + 1) * (W + 1);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int xi = (int)(t % (W + 1)), yi = (int)((t / (W + 1)) % (H + 1)), zi = (int)(t / ((int64_t)(W + 1) * (H + 1)));
        unsigned bits = 0u;
        for (int c8 = 0; c8 < 8; ++c8) {
            const int x = xi - 1 + (c8 & 1), y = yi - 1 + ((c8 >> 1) & 1), z = zi - 1 + (c8 >> 2);
            if (x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D && mask[((size_t)z * H + y) * W + x]) bits |= 1u << c8;
        }
        cell[t] = (uint8_t)bits;
    }
}

// basis_mat [app_dim][3*n_app] -> [app_dim][4][3*n_app/4]: slice `sub` holds, in order (plane i, j, e), the weights of
// channels ch = 16 j + 4 sub + e -- the order app_products_slice() produces them in.
__global__ void k0_basis_slices(const float* __restrict__ src, float* __restrict__ dst, int app_dim, int n_app) {
    int npl = n_app / 4;
    int per = 3 * npl;
    int n = app_dim * 4 * per;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        int kk = t % per;
        int sub = (t / per) & 3;
        int o = t / (4 * per);
        int i = kk / npl, r = kk % npl, j = r >> 2, e = r & 3;
        dst[t] = src[o * 3 * n_app + i * n_app + 16 * j + 4 * sub + e];
    }
}

// basis_mat [app_dim][3*n_app] -> [app_dim][n_app/4][12]: lane c holds, in order (plane i, e), the weights of channels
// 4c+e -- the order app_products_lane() produces them in.
__global__ void k0_basis_lanes(const float* __restrict__ src, float* __restrict__ dst, int app_dim, int n_app) {
    int nl = n_app / 4;
    int n = app_dim * nl * 12;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        int kk = t % 12, c = (t / 12) % nl, o = t / (12 * nl);
        int i = kk >> 2, e = kk & 3;
        dst[t] = src[o * 3 * n_app + i * n_app + 4 * c + e];
    }
}

// ------------------------------------------------------------------------------------------------ elementwise
__global__ void k_normalize_coord(FieldDev f, const float* __restrict__ xyz, int64_t n, float* __restrict__ out) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float p[3] = {xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]}, q[3];
        field_normalize(f, p, q);
        out[3 * t] = q[0]; out[3 * t + 1] = q[1]; out[3 * t + 2] = q[2];
    }
}

__global__ void k_mask_sample(FieldDev f, const float* __restrict__ xyz, int64_t n, float* __restrict__ out) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float p[3] = {xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]};
        out[t] = f.mask ? mask_value(f, p) : 1.0f;
    }
}

__global__ void k_mask_occupied(FieldDev f, const float* __restrict__ xyz, int64_t n, uint8_t* __restrict__ out) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float p[3] = {xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2]};
        out[t] = (f.mask ? mask_occupied(f, p) : true) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ K1
// mode 0: compute_densityfeature(xn) -> feature ; mode 1: compute_alpha(xyz, length) -> alpha
template <int MODE>
__global__ void __launch_bounds__(256) k1_point_density(FieldDev f, const float* __restrict__ pts, int64_t n, float length,
                                                        float* __restrict__ out) {
    int64_t nt = n * 4;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // the trip count is wave-uniform (nt is a multiple of 4 and waves start on multiples of 64)
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < ((nt + 63) & ~(int64_t)63); t += stride) {
        bool live = t < nt;
        int64_t pi = live ? (t >> 2) : 0;
        int sub = (int)(t & 3);
        float p[3] = {pts[3 * pi], pts[3 * pi + 1], pts[3 * pi + 2]};
        float xn[3];
        bool valid = live;
        if (MODE == 1) {
            field_normalize(f, p, xn);
            if (f.mask) valid = valid && mask_occupied(f, p, xn);
        } else {
            xn[0] = p[0]; xn[1] = p[1]; xn[2] = p[2];
        }
        float part = 0.0f;
        if (valid) part = density_partial(f, xn, sub);
        float feat = sum4(part);
        if (live && sub == 0) {
            if (MODE == 0) {
                out[pi] = feat;
            } else {
                float sigma = valid ? feature2density(f, feat) : 0.0f;
                out[pi] = 1.0f - expf(-sigma * length);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ app feature / K2
// mode 0: compute_appfeature(xn) -> [n, app_dim]; mode 1: samples_points_normals(xyz) -> [n,3]
template <int MODE, int NPL, int APP>
__global__ void __launch_bounds__(256) k2_point_app(FieldDev f, const float* __restrict__ pts, int64_t n,
                                                    float* __restrict__ out) {
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    int64_t nt = n * 4;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < ((nt + 63) & ~(int64_t)63); t += stride) {
        bool live = t < nt;
        int64_t pi = live ? (t >> 2) : 0;
        int sub = (int)(t & 3);
        float p[3] = {pts[3 * pi], pts[3 * pi + 1], pts[3 * pi + 2]};
        float xn[3];
        if (MODE == 1) field_normalize(f, p, xn);
        else { xn[0] = p[0]; xn[1] = p[1]; xn[2] = p[2]; }
        float prod[3 * NPL];
        app_products_slice<NPL>(f, xn, sub, prod);
        float F[APP];
        const float* bl = f.basis_l + sub * (3 * NPL);
#pragma unroll
        for (int o = 0; o < APP; ++o) {
            float a = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 3 * NPL; ++kk) a = fmaf(bl[o * 4 * 3 * NPL + kk], prod[kk], a);
            F[o] = sum4(a);
        }
        if (!live) continue;
        if (MODE == 0) {
            for (int o = sub; o < APP; o += 4) out[pi * APP + o] = F[o];
        } else if (sub == 0) {
            // Ref.compute_normals = -normal_mlp(F) = +normalise(W F + b)   (models/ref.py:85-89,154-155)
            float nr[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < APP; ++k) a = fmaf(f.head[ho.normal_w + o * ho.ld + k], F[k], a);
                nr[o] = a + f.head[ho.normal_b + o];
            }
            float nn = fmaxf(sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), 1e-12f);
            out[3 * pi] = nr[0] / nn; out[3 * pi + 1] = nr[1] / nn; out[3 * pi + 2] = nr[2] / nn;
        }
    }
}

// Ref.forward for explicit (viewdirs, features): 16 lanes per row.  With BLEND it is also the march epilogue (K4c,
// tensorBase.py:886-908): features come with a row stride and a "has shaded samples" flag in column APP, and the colour
// is blended with the background by the accumulated opacity.
struct ShadeArgs {
    const float* dirs; int dir_stride;     // viewdirs: row r at dirs + r * dir_stride
    const float* feat; int feat_stride;    // features: row r at feat + r * feat_stride
    const float* acc;                      // [n] (BLEND only)
    float bg[3];
    int64_t n;
    float* rgb;
};
// The packed head (~5.3 k floats, 21 KB) is copied to LDS once per workgroup and every weight is read from there: with the
// weights in global memory a wave issued ~130 vector loads per 4 rays, and the CU's vector-memory pipeline spends ~16 cycles
// on a wave-level load whatever its width -- that, not arithmetic, was this kernel's time (profiles/README.md).  Workgroups
// loop over 16-ray tiles; the LDS pointer is made opaque per tile so that LLVM does not hoist the (loop-invariant) weight
// reads out of the loop into 300+ registers.
template <int APP, bool BLEND>
__global__ void __launch_bounds__(256) k_ref_shade(FieldDev f, ShadeArgs a, int64_t n_tiles) {
    extern __shared__ __align__(16) float s_head[];
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    for (int i = threadIdx.x; i < ho.total / 4; i += 256)
        reinterpret_cast<float4*>(s_head)[i] = reinterpret_cast<const float4*>(f.head)[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t rr = tile * 16 + (threadIdx.x >> 4);
        const bool live = rr < a.n;
        const int64_t ri = live ? rr : 0;
        constexpr int LD = (APP + 3) & ~3;
        static_assert(!BLEND || LD == 28, "the march's feature rows are 28 floats: 27 features + the shaded flag");
        const float* dp = a.dirs + ri * a.dir_stride;
        const float* fp = a.feat + ri * a.feat_stride;
        float F[LD], d[3] = {dp[0], dp[1], dp[2]}, c[3];
        float flag = 0.0f;
        if (BLEND) {        // rows of 28 floats, 16-B aligned (march workspace): seven 16-B loads instead of 28 scalar ones
#pragma unroll
            for (int k4 = 0; k4 < LD; k4 += 4) {
                const float4 v = *reinterpret_cast<const float4*>(fp + k4);
                F[k4] = v.x; F[k4 + 1] = v.y; F[k4 + 2] = v.z; F[k4 + 3] = v.w;
            }
            flag = F[APP];
            F[APP] = 0.0f;
        } else {
#pragma unroll
            for (int k = 0; k < LD; ++k) F[k] = (k < APP) ? fp[k] : 0.0f;
        }
        int opaque0 = 0;
        asm volatile("" : "+v"(opaque0));            // a zero the optimiser cannot see through (see the comment above the kernel)
        const lds_cfloat_p hp = (lds_cfloat_p)s_head + opaque0;
        ref_shade_group16<LD, lds_cfloat_p>(hp, ho, f.feature_c, F, d, l16, c);
        if (live && l16 == 0) {
            if (BLEND) {
                const bool any = flag != 0.0f;
                const float acc = a.acc[ri];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float v = any ? c[ch] : 0.0f;
                    v = v * acc + a.bg[ch] * (1.0f - acc);
                    a.rgb[3 * ri + ch] = fminf(fmaxf(v, 0.0f), 1.0f);
                }
            } else {
                a.rgb[3 * ri] = c[0]; a.rgb[3 * ri + 1] = c[1]; a.rgb[3 * ri + 2] = c[2];
            }
        }
    }
}

// Ref.compute_normals(features) = +normalise(W F + b)   (models/ref.py:85-89,154-155)
template <int APP>
__global__ void k_ref_normals(FieldDev f, const float* __restrict__ feat, int64_t n, float* __restrict__ out) {
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float nr[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < APP; ++k) a = fmaf(f.head[ho.normal_w + o * ho.ld + k], feat[t * APP + k], a);
            nr[o] = a + f.head[ho.normal_b + o];
        }
        float nn = fmaxf(sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), 1e-12f);
        out[3 * t] = nr[0] / nn; out[3 * t + 1] = nr[1] / nn; out[3 * t + 2] = nr[2] / nn;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static inline int grid_for(int64_t threads, int block = 256, int cap = 256 * 8) {
    int64_t g = (threads + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

hipError_t launch_k0_channels_last(const float* src, float* dst, int C, int64_t HW, hipStream_t s) {
    hipLaunchKernelGGL(k0_channels_last, dim3(grid_for((int64_t)C * HW)), dim3(256), 0, s, src, dst, C, HW);
    return hipGetLastError();
}
hipError_t launch_k0_mask_bytes(const float* src, uint8_t* dst, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k0_mask_bytes, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_k0_mask_cells(const uint8_t* mask, uint8_t* cell, int D, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(k0_mask_cells, dim3(grid_for((int64_t)(D + 1) * (H + 1) * (W + 1))), dim3(256), 0, s, mask, cell, D, H, W);
    return hipGetLastError();
}
hipError_t launch_k0_basis_slices(const float* src, float* dst, int app_dim, int n_app, hipStream_t s) {
    hipLaunchKernelGGL(k0_basis_slices, dim3(grid_for(app_dim * 3 * n_app)), dim3(256), 0, s, src, dst, app_dim, n_app);
    return hipGetLastError();
}
hipError_t launch_k0_basis_lanes(const float* src, float* dst, int app_dim, int n_app, hipStream_t s) {
    hipLaunchKernelGGL(k0_basis_lanes, dim3(grid_for(app_dim * 3 * n_app)), dim3(256), 0, s, src, dst, app_dim, n_app);
    return hipGetLastError();
}
hipError_t launch_normalize_coord(const FieldDev& f, const float* xyz, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_normalize_coord, dim3(grid_for(n)), dim3(256), 0, s, f, xyz, n, out);
    return hipGetLastError();
}
hipError_t launch_mask_sample(const FieldDev& f, const float* xyz, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_mask_sample, dim3(grid_for(n)), dim3(256), 0, s, f, xyz, n, out);
    return hipGetLastError();
}
hipError_t launch_mask_occupied(const FieldDev& f, const float* xyz, int64_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_mask_occupied, dim3(grid_for(n)), dim3(256), 0, s, f, xyz, n, out);
    return hipGetLastError();
}
hipError_t launch_density_feature(const FieldDev& f, const float* xn, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k1_point_density<0>, dim3(grid_for(n * 4)), dim3(256), 0, s, f, xn, n, 1.0f, out);
    return hipGetLastError();
}
hipError_t launch_point_alpha(const FieldDev& f, const float* xyz, int64_t n, float length, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k1_point_density<1>, dim3(grid_for(n * 4)), dim3(256), 0, s, f, xyz, n, length, out);
    return hipGetLastError();
}
hipError_t launch_app_feature(const FieldDev& f, const float* xn, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL((k2_point_app<0, 12, 27>), dim3(grid_for(n * 4)), dim3(256), 0, s, f, xn, n, out);
    return hipGetLastError();
}
hipError_t launch_point_normals(const FieldDev& f, const float* xyz, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL((k2_point_app<1, 12, 27>), dim3(grid_for(n * 4)), dim3(256), 0, s, f, xyz, n, out);
    return hipGetLastError();
}
template <bool BLEND>
static hipError_t launch_shade_kernel(const FieldDev& f, const ShadeArgs& a, int64_t n, hipStream_t s) {
    // the reference's head shape runs in the 8-lanes-per-ray form with the bottleneck on the matrix cores (fan_march_kernels.hip:
    // same bits); iff_field_desc.head_lanes = 16 keeps this file's kernel, the form other head shapes use
    if (fan_head_fusable(f) && f.head_lanes != 16)
        return launch_ref_shade_oct(f, a.dirs, a.dir_stride, a.feat, a.feat_stride, BLEND ? a.acc : nullptr, a.bg, n, a.rgb, s);
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    const size_t lds = (size_t)((ho.total + 3) & ~3) * sizeof(float);
    const int64_t n_tiles = (n + 15) / 16;
    const int64_t grid = n_tiles < 256 * 5 ? n_tiles : 256 * 5;
    hipLaunchKernelGGL((k_ref_shade<27, BLEND>), dim3((unsigned)grid), dim3(256), lds, s, f, a, n_tiles);
    return hipGetLastError();
}
hipError_t launch_ref_shade(const FieldDev& f, const float* dirs, const float* feat, int64_t n, float* rgb, hipStream_t s) {
    ShadeArgs a;
    a.dirs = dirs; a.dir_stride = 3; a.feat = feat; a.feat_stride = f.app_dim; a.acc = nullptr;
    a.bg[0] = a.bg[1] = a.bg[2] = 0.0f; a.n = n; a.rgb = rgb;
    return launch_shade_kernel<false>(f, a, n, s);
}
hipError_t launch_shade_blend(const FieldDev& f, const float* rays, int ray_cols, const float* feat28, const float* acc,
                              const float* bg, int64_t n, float* rgb, hipStream_t s) {
    ShadeArgs a;
    a.dirs = rays + 3; a.dir_stride = ray_cols; a.feat = feat28; a.feat_stride = 28; a.acc = acc;
    a.bg[0] = bg[0]; a.bg[1] = bg[1]; a.bg[2] = bg[2]; a.n = n; a.rgb = rgb;
    return launch_shade_kernel<true>(f, a, n, s);
}
hipError_t launch_ref_normals(const FieldDev& f, const float* feat, int64_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL((k_ref_normals<27>), dim3(grid_for(n)), dim3(256), 0, s, f, feat, n, out);
    return hipGetLastError();
}
