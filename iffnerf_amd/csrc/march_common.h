// march_common.h -- argument block and sample-position helpers shared by the march kernels (march_kernels.hip: the general
// forms; fan_march_kernels.hip: the fused 27-ray-fan form of the point-centred sampler).
#pragma once
#include "iff_device.h"

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
    float* weights;  // [R,S] compositing weights, written by K4a and read by K4b (caller's workspace)
    float* feat;     // [R,28] per-ray weighted features + shaded flag, written by K4b and read by K4c (workspace)
    int64_t n_tiles;
};

// tensorBase.py:499-502: slab entry parameter, clamped to [near, far]
__device__ inline float slab_entry(const FieldDev& f, const float o[3], const float d[3]) {
    float tmax = -INFINITY;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        float v = (d[ax] == 0.0f) ? 1e-6f : d[ax];
        float ra = (f.aabb_hi[ax] - o[ax]) / v, rb = (f.aabb_lo[ax] - o[ax]) / v;
        tmax = fmaxf(tmax, fminf(ra, rb));
    }
    return fminf(fmaxf(tmax, f.near), f.far);
}

// sample position parameter z_s (tensorBase.py:628-631 / :504-529), float ops in the reference's order
__device__ inline float z_of(const FieldDev& f, int mode, int S, float t0, int s) {
    if (mode == 0) return f.step_size * (float)(s - S / 2);
    return t0 + f.step_size * (float)s;
}
