// march_kernels.hip -- K3 isocell_emit and K4 march_composite_shade (TensorBase.forward, models/tensorBase.py:775-917).
//
// K4 work decomposition (gfx950, 256-thread workgroups = 4 waves, persistent over 16-ray tiles):
//   phase 1  density: every (ray, sample) point gets 4 lanes (one 16-B quarter of each 64-B texel), all tiles' taps in
//            flight together; sigma lands in LDS                                  [HBM/L2 gather, the dominant bytes]
//   phase 2  one lane per ray walks its samples: alpha, transmittance product, weights, acc, depth; records which
//            samples pass weight > rayMarch_weight_thres                          [sequential by definition]
//   phase 3  appearance: 16 lanes per ray = 4 sub-groups x 4 channel-quarter lanes; sub-group q takes the ray's passing
//            samples q, q+4, ... and accumulates weight * (plane*line) for its 36 channels in registers; a fixed-order
//            xor butterfly merges the 4 sub-groups (deterministic, no atomics)
//   phase 4  basis_mat once per ray on the weighted sum (linear, so equal to the reference's per-sample basis_mat up to
//            fp32 rounding), then the Ref head by the same 16 lanes, background blend, clamp.
// Samples are processed in chunks of CH (LDS holds one chunk), so the 20-sample point-centred sampler and the
// ~1000-sample slab sampler share the code.
#include "iff_device.h"
#include "iff_launch.h"

// ------------------------------------------------------------------------------------------------ K3
// 27 iso-cell directions (pose_estimation/isocell.py:6-68, N0=3, isrand=-1) are passed in by the host mirror, which
// evaluates the closed form once; the kernel applies rotate_isocell (isocell.py:144-171) and renormalises
// (sampling.py:455-457).
struct IsoCells { float v[27][3]; };

__global__ void k3_isocell_emit(IsoCells cells, const float* __restrict__ pts, const float* __restrict__ nrm, int64_t P,
                                float* __restrict__ ori, float* __restrict__ dirs) {
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
        ori[3 * t] = pts[3 * pi]; ori[3 * t + 1] = pts[3 * pi + 1]; ori[3 * t + 2] = pts[3 * pi + 2];
    }
}

hipError_t launch_isocell_emit(const float* cells27x3_host, const float* pts, const float* nrm, int64_t P, float* ori,
                               float* dirs, hipStream_t s) {
    IsoCells c;
    for (int i = 0; i < 27; ++i)
        for (int j = 0; j < 3; ++j) c.v[i][j] = cells27x3_host[3 * i + j];
    int64_t n = P * 27;
    int grid = (int)((n + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k3_isocell_emit, dim3(grid), dim3(256), 0, s, c, pts, nrm, P, ori, dirs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K4
constexpr int RPB = 16;   // rays per workgroup tile (256 threads / 16 lanes per ray in phases 3-4)
constexpr int CH = 32;    // samples per chunk held in LDS

struct MarchArgs {
    const float* rays;
    int ray_cols;
    int64_t R;
    int mode;        // 0 point-centred, 1 slab
    int S;           // samples per ray
    float bg[3];
    float* rgb; float* depth; float* acc;
    float* alpha;    // nullable [R,S]
    int* counts;     // nullable [R,2]
    int64_t n_tiles;
};

// sample position parameter z_s (tensorBase.py:628-631 / :504-529), float ops in the reference's order
__device__ inline float z_of(const FieldDev& f, int mode, int S, float t0, int s) {
    if (mode == 0) return f.step_size * (float)(s - S / 2);
    return t0 + f.step_size * (float)s;
}

template <int NPL, int APP>
__global__ void __launch_bounds__(256) k4_march(FieldDev f, MarchArgs a) {
    extern __shared__ __align__(16) float smem[];
    const HeadOff ho = head_offsets(f.app_dim, f.feature_c);
    const int per = 3 * NPL;                       // products per lane slice (36)
    float* s_basis = smem;                         // [APP][4][per]
    float* s_head = s_basis + APP * 4 * per;       // packed head
    float* s_sigma = s_head + ho.total;            // [RPB][CH]
    float* s_w = s_sigma + RPB * CH;               // [RPB][CH]
    float* s_feat = s_w + RPB * CH;                // [RPB][APP+1]
    float* s_ray = s_feat + RPB * (APP + 1);       // [RPB][8]: o(3) d(3) t0 last
    int* s_napp = (int*)(s_ray + RPB * 8);         // [RPB]
    unsigned char* s_list = (unsigned char*)(s_napp + RPB);   // [RPB][CH]

    const int tid = threadIdx.x;
    for (int i = tid; i < APP * 4 * per; i += 256) s_basis[i] = f.basis_l[i];
    for (int i = tid; i < ho.total; i += 256) s_head[i] = f.head[i];

    const int S = a.S;
    const int n_chunks = (S + CH - 1) / CH;

    for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t ray0 = tile * RPB;
        __syncthreads();   // previous tile's readers are done with s_ray / s_feat; weights are in place
        if (tid < RPB) {
            int64_t r = ray0 + tid;
            bool live = r < a.R;
            float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f}, last = 0.f;
            if (live) {
                const float* rp = a.rays + r * a.ray_cols;
                o[0] = rp[0]; o[1] = rp[1]; o[2] = rp[2]; d[0] = rp[3]; d[1] = rp[4]; d[2] = rp[5];
                last = rp[a.ray_cols - 1];
            }
            float t0 = 0.f;
            if (a.mode == 1) {
                // tensorBase.py:499-502: slab entry, clamped to [near, far]
                float tmax = -INFINITY;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    float v = (d[ax] == 0.0f) ? 1e-6f : d[ax];
                    float ra = (f.aabb_hi[ax] - o[ax]) / v, rb = (f.aabb_lo[ax] - o[ax]) / v;
                    tmax = fmaxf(tmax, fminf(ra, rb));
                }
                t0 = fminf(fmaxf(tmax, f.near), f.far);
            }
            float* sr = s_ray + tid * 8;
            sr[0] = o[0]; sr[1] = o[1]; sr[2] = o[2]; sr[3] = d[0]; sr[4] = d[1]; sr[5] = d[2]; sr[6] = t0; sr[7] = last;
        }
        // per-ray running state lives in the registers of thread `tid < RPB`
        float run_T = 1.0f, run_acc = 0.0f, run_depth = 0.0f;
        int run_valid = 0, run_app = 0;
        // phase-3 accumulators: 16 lanes per ray
        const int ray_l = tid >> 4, l16 = tid & 15, q = l16 >> 2, sub = l16 & 3;
        float accp[3 * NPL];
#pragma unroll
        for (int i = 0; i < 3 * NPL; ++i) accp[i] = 0.0f;
        __syncthreads();

        for (int c = 0; c < n_chunks; ++c) {
            const int s_base = c * CH;
            const int ns = min(CH, S - s_base);
            // ---------------- phase 1: sigma for RPB x ns points, 4 lanes per point
            const int n_lane_tasks = RPB * ns * 4;
            for (int t = tid; t < ((n_lane_tasks + 63) & ~63); t += 256) {
                bool live = t < n_lane_tasks;
                int ps = live ? (t >> 2) : 0;
                int lsub = t & 3;
                int rl = ps / ns, sl = ps - rl * ns;
                const float* sr = s_ray + rl * 8;
                float z = z_of(f, a.mode, S, sr[6], s_base + sl);
                float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z};
                bool valid = live && (ray0 + rl < a.R) && inside_aabb(f, p);
                if (valid && f.mask) valid = mask_value(f, p) > 0.0f;
                float part = 0.0f;
                if (valid) {
                    float xn[3];
                    field_normalize(f, p, xn);
                    part = density_partial(f, xn, lsub);
                }
                float feat = sum4(part);
                if (live && lsub == 0) s_sigma[rl * CH + sl] = valid ? feature2density(f, feat) : -1.0f;  // -1: invalid
            }
            __syncthreads();
            // ---------------- phase 2: compositing, one lane per ray
            if (tid < RPB) {
                const float* sr = s_ray + tid * 8;
                int64_t r = ray0 + tid;
                int napp = 0;
                for (int sl = 0; sl < ns; ++sl) {
                    int s = s_base + sl;
                    float sg = s_sigma[tid * CH + sl];
                    bool valid = sg >= 0.0f;
                    float sigma = valid ? sg : 0.0f;
                    float z = z_of(f, a.mode, S, sr[6], s);
                    float dist = (s + 1 < S) ? (z_of(f, a.mode, S, sr[6], s + 1) - z) : 0.0f;   // tensorBase.py:800-803
                    float alpha = 1.0f - expf(-sigma * (dist * f.distance_scale));              // tensorBase.py:25,849
                    float w = alpha * run_T;
                    run_T = run_T * ((1.0f - alpha) + 1e-10f);                                   // tensorBase.py:27-32
                    run_acc += w;
                    run_depth += w * z;
                    run_valid += valid ? 1 : 0;
                    bool shade = w > f.weight_thres;                                              // tensorBase.py:851
                    s_w[tid * CH + sl] = w;
                    if (shade) s_list[tid * CH + napp++] = (unsigned char)sl;
                    if (a.alpha && r < a.R) a.alpha[r * S + s] = alpha;
                }
                run_app += napp;
                s_napp[tid] = napp;
            }
            __syncthreads();
            // ---------------- phase 3: appearance gather for the passing samples
            {
                const float* sr = s_ray + ray_l * 8;
                const int napp = s_napp[ray_l];
                for (int j = q; j < napp; j += 4) {
                    int sl = s_list[ray_l * CH + j];
                    float w = s_w[ray_l * CH + sl];
                    float z = z_of(f, a.mode, S, sr[6], s_base + sl);
                    float p[3] = {sr[0] + sr[3] * z, sr[1] + sr[4] * z, sr[2] + sr[5] * z}, xn[3];
                    field_normalize(f, p, xn);
                    float prod[3 * NPL];
                    app_products_slice<NPL>(f, xn, sub, prod);
#pragma unroll
                    for (int i = 0; i < 3 * NPL; ++i) accp[i] = fmaf(w, prod[i], accp[i]);
                }
            }
            __syncthreads();   // s_sigma / s_w / s_list are rewritten by the next chunk
        }
        // merge the 4 sub-groups (fixed order), then basis_mat: sub-group q produces outputs o = q, q+4, ...
#pragma unroll
        for (int i = 0; i < 3 * NPL; ++i) {
            float v = accp[i];
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            accp[i] = v;
        }
        for (int o = q; o < APP; o += 4) {
            const float* bl = s_basis + (o * 4 + sub) * per;
            float v = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 3 * NPL; ++kk) v = fmaf(bl[kk], accp[kk], v);
            v = sum4(v);
            if (sub == 0) s_feat[ray_l * (APP + 1) + o] = v;
        }
        if (tid < RPB) {
            // park the per-ray scalars next to the features for the 16-lane epilogue
            s_w[tid * CH + 0] = run_acc; s_w[tid * CH + 1] = run_depth;
            s_napp[tid] = run_app;
            if (a.counts && ray0 + tid < a.R) { a.counts[(ray0 + tid) * 2] = run_valid; a.counts[(ray0 + tid) * 2 + 1] = run_app; }
        }
        __syncthreads();
        // ---------------- phase 4: Ref head + blend (tensorBase.py:886-908), 16 lanes per ray
        {
            const float* sr = s_ray + ray_l * 8;
            float F[APP];
#pragma unroll
            for (int k = 0; k < APP; ++k) F[k] = s_feat[ray_l * (APP + 1) + k];
            float d[3] = {sr[3], sr[4], sr[5]}, c[3];
            ref_shade_group16<APP>(s_head, ho, f.feature_c, F, d, l16, c);
            int64_t r = ray0 + ray_l;
            if (l16 == 0 && r < a.R) {
                float acc = s_w[ray_l * CH + 0], depth = s_w[ray_l * CH + 1];
                bool any = s_napp[ray_l] > 0;
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    float v = any ? c[o] : 0.0f;
                    v = v * acc + a.bg[o] * (1.0f - acc);
                    a.rgb[3 * r + o] = fminf(fmaxf(v, 0.0f), 1.0f);
                }
                a.acc[r] = acc;
                a.depth[r] = depth + (1.0f - acc) * sr[7];
            }
        }
    }
}

size_t march_lds_bytes(int app_dim, int feature_c, int n_app) {
    HeadOff ho = head_offsets(app_dim, feature_c);
    size_t fl = (size_t)app_dim * 3 * n_app + ho.total + 2 * RPB * CH + RPB * (app_dim + 1) + RPB * 8;
    return fl * 4 + RPB * 4 + RPB * CH;
}

hipError_t launch_march(const FieldDev& f, const float* rays, int ray_cols, int64_t R, int mode, int S, const float* bg,
                        float* rgb, float* depth, float* acc, float* alpha, int* counts, hipStream_t s) {
    MarchArgs a;
    a.rays = rays; a.ray_cols = ray_cols; a.R = R; a.mode = mode; a.S = S;
    a.bg[0] = bg[0]; a.bg[1] = bg[1]; a.bg[2] = bg[2];
    a.rgb = rgb; a.depth = depth; a.acc = acc; a.alpha = alpha; a.counts = counts;
    a.n_tiles = (R + RPB - 1) / RPB;
    if (a.n_tiles == 0) return hipSuccess;
    size_t lds = march_lds_bytes(f.app_dim, f.feature_c, f.n_app);
    int64_t grid = a.n_tiles;
    const int64_t cap = 256 * 4;   // 4 workgroups per CU fit the ~36 KB LDS footprint
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((k4_march<12, 27>), dim3((unsigned)grid), dim3(256), lds, s, f, a);
    return hipGetLastError();
}
