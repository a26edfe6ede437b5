// api.hip -- the extern "C" surface of libiffnerf_hip.so (include/iffnerf_hip.h): argument checking, handle ownership,
// error reporting.  No torch types; plain pointers and sizes.
#include "../../include/iffnerf_hip.h"
#include "iff_device.h"
#include "iff_launch.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <sys/stat.h>
#include <exception>

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int hip_fail(hipError_t e, const char* what) {
    return fail((int)e, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
}
#define IFF_HIP(call)                                     \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) return hip_fail(e__, #call); \
    } while (0)
#define IFF_REQUIRE(cond, ...)                            \
    do {                                                  \
        if (!(cond)) return fail(IFF_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)

extern "C" const char* iff_last_error(void) { return g_err; }
extern "C" int iff_abi_version(void) { return IFF_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------ field handle
struct iff_field {
    FieldDev dev;
    void* slab = nullptr;        // every table lives in one allocation
    size_t slab_bytes = 0;
    int* occ_list = nullptr;     // occupied mask voxels (ascending), for the surface sampler
    int n_occ = 0;
    int n_cus = 256;
};

extern "C" void iff_field_destroy(iff_field* f) {
    if (!f) return;
    if (f->slab) (void)hipFree(f->slab);
    if (f->occ_list) (void)hipFree(f->occ_list);
    delete f;
}

extern "C" size_t iff_field_table_bytes(const iff_field* f) { return f ? f->slab_bytes : 0; }

static size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// Where every table of a field handle sits in its slab, as a function of the descriptor's dimensions alone: iff_field_create
// lays the slab out with it and iff_field_load checks a table file against it (a file is accepted only if it is exactly what
// iff_field_create would have produced for the dimensions it claims).
struct FieldLayout { size_t dp[3], dl[3], ap[3], al[3], basis, basis_l, basis_l12, head, mask, n_mask, cell, n_cell, total; };
static FieldLayout field_layout(const int G[3], int n_density, int n_app, int app_dim, int feature_c, const int mask_dims[3], bool has_mask) {
    FieldLayout L;
    size_t off = 0;
    for (int i = 0; i < 3; ++i) {
        size_t hw = (size_t)G[mat_a(i)] * G[mat_b(i)], l = (size_t)G[vec_ax(i)];
        L.dp[i] = off; off = up256(off + hw * n_density * 4);
        L.dl[i] = off; off = up256(off + l * n_density * 4);
        L.ap[i] = off; off = up256(off + hw * n_app * 4);
        L.al[i] = off; off = up256(off + l * n_app * 4);
    }
    L.basis = off; off = up256(off + (size_t)app_dim * 3 * n_app * 4);
    L.basis_l = off; off = up256(off + (size_t)app_dim * 3 * n_app * 4);
    L.basis_l12 = off; off = up256(off + (size_t)app_dim * 3 * n_app * 4);
    L.head = off; off = up256(off + (size_t)head_offsets(app_dim, feature_c).total * 4);
    L.n_mask = has_mask ? (size_t)mask_dims[0] * mask_dims[1] * mask_dims[2] : 0;
    L.mask = off; off = up256(off + L.n_mask);
    L.n_cell = has_mask ? (size_t)(mask_dims[0] + 1) * (mask_dims[1] + 1) * (mask_dims[2] + 1) : 0;
    L.cell = off; off = up256(off + L.n_cell);
    L.total = off;
    return L;
}
// the range checks iff_field_create applies to a descriptor's dimensions (shared with iff_field_load)
static int check_field_dims(const int G[3], int n_density, int n_app, int app_dim, int feature_c, int density_lanes, const int mask_dims[3],
                            bool has_mask) {
    for (int i = 0; i < 3; ++i) IFF_REQUIRE(G[i] >= 2 && G[i] <= 4096, "gridSize[%d] = %d out of range", i, G[i]);
    if (n_app != 48 || app_dim != 27)
        return fail(IFF_ERR_UNSUPPORTED, "only appearance_n_comp = 48 and app_dim = 27 are built (got %d, %d)", n_app, app_dim);
    IFF_REQUIRE(n_density >= 4 && n_density % 4 == 0 && n_density <= 64, "density_n_comp = %d unsupported", n_density);
    IFF_REQUIRE(density_lanes == 0 || density_lanes == 4 || (density_lanes == 1 && n_density == 16),
                "density_lanes = %d: must be 0 (auto), 4, or 1 with density_n_comp = 16", density_lanes);
    IFF_REQUIRE(feature_c >= 16 && feature_c <= 512 && feature_c % 16 == 0, "featureC = %d unsupported", feature_c);
    if (has_mask) {
        for (int i = 0; i < 3; ++i) IFF_REQUIRE(mask_dims[i] >= 1 && mask_dims[i] <= 4096, "mask dim %d out of range", i);
        // the corner-bit table is indexed with 32-bit arithmetic (iff_device.h mask_occupied_at)
        IFF_REQUIRE((uint64_t)(mask_dims[0] + 1) * (uint64_t)(mask_dims[1] + 1) * (uint64_t)(mask_dims[2] + 1) < (1ull << 31),
                    "occupancy mask %d x %d x %d: 2^31 cells and more are not addressable", mask_dims[0], mask_dims[1], mask_dims[2]);
    }
    for (int i = 0; i < 3; ++i) {
        // the gathers address a table by a 32-bit byte offset from its base (iff_device.h ld4_tex)
        const uint64_t texels = (uint64_t)G[mat_a(i)] * (uint64_t)G[mat_b(i)];
        IFF_REQUIRE(texels < (1ull << 24) && texels * (uint64_t)std::max(n_density, n_app) * 4u < (1ull << 32),
                    "VM plane %d has %llu texels: 2^24 texels / 4 GiB per table and more are not addressable", i,
                    (unsigned long long)texels);
    }
    return 0;
}

extern "C" int iff_field_create(const iff_field_desc* d, void* stream, iff_field** out) {
    IFF_REQUIRE(d && out, "iff_field_create: null argument");
    *out = nullptr;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < 3; ++i)
        IFF_REQUIRE(d->density_plane[i] && d->density_line[i] && d->app_plane[i] && d->app_line[i], "null VM table %d", i);
    IFF_REQUIRE(d->basis && d->normal_w && d->normal_b && d->tint_w && d->tint_b && d->rough_w && d->rough_b && d->diffuse_w &&
                    d->diffuse_b && d->bottleneck_w && d->bottleneck_b && d->specular_w && d->specular_b && d->ide_mat,
                "null Ref-head / basis tensor");
    const int G[3] = {d->grid[0], d->grid[1], d->grid[2]};
    if (int rc = check_field_dims(G, d->n_density, d->n_app, d->app_dim, d->feature_c, d->density_lanes, d->mask_dims, d->mask_volume != nullptr))
        return rc;
    IFF_REQUIRE(d->head_lanes == 0 || d->head_lanes == 16, "head_lanes = %d: must be 0 (auto) or 16", d->head_lanes);
    IFF_REQUIRE(d->fan_waves == 0 || d->fan_waves == 4 || d->fan_waves == 8, "fan_waves = %d: must be 0 (auto), 4 or 8", d->fan_waves);

    iff_field* f = new iff_field();
    FieldDev& v = f->dev;
    memset(&v, 0, sizeof(v));
    const HeadOff ho = head_offsets(d->app_dim, d->feature_c);
    const FieldLayout L = field_layout(G, d->n_density, d->n_app, d->app_dim, d->feature_c, d->mask_dims, d->mask_volume != nullptr);
    const size_t* o_dp = L.dp; const size_t* o_dl = L.dl; const size_t* o_ap = L.ap; const size_t* o_al = L.al;
    const size_t o_basis = L.basis, o_basis_l = L.basis_l, o_basis_l12 = L.basis_l12, o_head = L.head, o_mask = L.mask, n_mask = L.n_mask;
    const size_t off = L.total;
    f->slab_bytes = off;
    hipError_t e = hipMalloc(&f->slab, off);
    if (e != hipSuccess) { delete f; return hip_fail(e, "hipMalloc(field tables)"); }
    char* base = (char*)f->slab;
#define IFF_CREATE_HIP(call)                                           \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) { iff_field_destroy(f); return hip_fail(e__, #call); } \
    } while (0)
    IFF_CREATE_HIP(hipMemsetAsync(f->slab, 0, off, s));
    for (int i = 0; i < 3; ++i) {
        int64_t hw = (int64_t)G[mat_a(i)] * G[mat_b(i)], l = G[vec_ax(i)];
        IFF_CREATE_HIP(launch_k0_channels_last(d->density_plane[i], (float*)(base + o_dp[i]), d->n_density, hw, s));
        IFF_CREATE_HIP(launch_k0_channels_last(d->density_line[i], (float*)(base + o_dl[i]), d->n_density, l, s));
        IFF_CREATE_HIP(launch_k0_channels_last(d->app_plane[i], (float*)(base + o_ap[i]), d->n_app, hw, s));
        IFF_CREATE_HIP(launch_k0_channels_last(d->app_line[i], (float*)(base + o_al[i]), d->n_app, l, s));
        v.dplane[i] = (const float*)(base + o_dp[i]); v.dline[i] = (const float*)(base + o_dl[i]);
        v.aplane[i] = (const float*)(base + o_ap[i]); v.aline[i] = (const float*)(base + o_al[i]);
    }
    IFF_CREATE_HIP(hipMemcpyAsync(base + o_basis, d->basis, (size_t)d->app_dim * 3 * d->n_app * 4, hipMemcpyDeviceToDevice, s));
    IFF_CREATE_HIP(launch_k0_basis_slices(d->basis, (float*)(base + o_basis_l), d->app_dim, d->n_app, s));
    v.basis = (const float*)(base + o_basis);
    v.basis_l = (const float*)(base + o_basis_l);
    IFF_CREATE_HIP(launch_k0_basis_lanes(d->basis, (float*)(base + o_basis_l12), d->app_dim, d->n_app, s));
    v.basis_l12 = (const float*)(base + o_basis_l12);
    {
        float* h = (float*)(base + o_head);
        struct { const float* src; int off; int rows; int cols; int ld; } parts[] = {
            {d->normal_w, ho.normal_w, 3, d->app_dim, ho.ld}, {d->normal_b, ho.normal_b, 1, 3, 4},
            {d->tint_w, ho.tint_w, 3, d->app_dim, ho.ld}, {d->tint_b, ho.tint_b, 1, 3, 4},
            {d->rough_w, ho.rough_w, 1, d->app_dim, ho.ld}, {d->rough_b, ho.rough_b, 1, 1, 4},
            {d->diffuse_w, ho.diffuse_w, 3, d->app_dim, ho.ld}, {d->diffuse_b, ho.diffuse_b, 1, 3, 4},
            {d->bottleneck_w, ho.bott_w, d->feature_c, d->app_dim, ho.ld}, {d->bottleneck_b, ho.bott_b, 1, d->feature_c, d->feature_c},
            {d->specular_w, ho.spec_w, 3, d->feature_c + 39, ho.spec_ld}, {d->specular_b, ho.spec_b, 1, 3, 4},
            {d->ide_mat, ho.ide_mat, 1, 9 * 19, 9 * 19}};
        for (auto& p : parts)   // rows are zero-padded to `ld` floats (the slab was memset to 0)
            IFF_CREATE_HIP(hipMemcpy2DAsync(h + p.off, (size_t)p.ld * 4, p.src, (size_t)p.cols * 4, (size_t)p.cols * 4,
                                            (size_t)p.rows, hipMemcpyDeviceToDevice, s));
        v.head = h;
    }
    if (d->mask_volume) {
        IFF_CREATE_HIP(launch_k0_mask_bytes(d->mask_volume, (uint8_t*)(base + o_mask), (int64_t)n_mask, s));
        v.mask = (const uint8_t*)(base + o_mask);
        IFF_CREATE_HIP(launch_k0_mask_cells(v.mask, (uint8_t*)(base + L.cell), d->mask_dims[0], d->mask_dims[1], d->mask_dims[2], s));
        v.cell = (const uint8_t*)(base + L.cell);
        // occupied-voxel list for the sampler's seeds (pose_estimation/sampling.py:82-102), built once on the host
        std::vector<uint8_t> hm(n_mask);
        IFF_CREATE_HIP(hipMemcpyAsync(hm.data(), base + o_mask, n_mask, hipMemcpyDeviceToHost, s));
        IFF_CREATE_HIP(hipStreamSynchronize(s));
        std::vector<int> occ;
        occ.reserve(n_mask / 2 + 1);
        for (size_t i = 0; i < n_mask; ++i) if (hm[i]) occ.push_back((int)i);
        f->n_occ = (int)occ.size();
        if (f->n_occ > 0) {
            IFF_CREATE_HIP(hipMalloc((void**)&f->occ_list, occ.size() * sizeof(int)));
            IFF_CREATE_HIP(hipMemcpyAsync(f->occ_list, occ.data(), occ.size() * sizeof(int), hipMemcpyHostToDevice, s));
            IFF_CREATE_HIP(hipStreamSynchronize(s));
        }
    }
    for (int i = 0; i < 3; ++i) {
        v.grid[i] = G[i];
        v.mask_dims[i] = d->mask_volume ? d->mask_dims[i] : 0;
        v.aabb_lo[i] = d->aabb[i]; v.aabb_hi[i] = d->aabb[3 + i];
        v.inv_aabb[i] = 2.0f / (v.aabb_hi[i] - v.aabb_lo[i]);                 // tensorBase.py:357-358
        v.mask_lo[i] = d->mask_aabb[i]; v.mask_hi[i] = d->mask_aabb[3 + i];
        v.mask_inv[i] = 1.0f / (v.mask_hi[i] - v.mask_lo[i]) * 2.0f;         // tensorBase.py:58-59
    }
    v.density_shift = d->density_shift; v.distance_scale = d->distance_scale; v.weight_thres = d->weight_thres;
    v.step_size = d->step_size; v.near = d->near_far[0]; v.far = d->near_far[1];
    v.n_samples = d->n_samples; v.softplus = d->softplus; v.unisphere = d->unisphere;
    v.density_lanes = d->density_lanes; v.head_lanes = d->head_lanes; v.sampler_persistent = d->sampler_persistent ? 1 : 0;
    v.fan_waves = d->fan_waves;
    v.n_density = d->n_density; v.n_app = d->n_app; v.app_dim = d->app_dim; v.feature_c = d->feature_c;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) f->n_cus = prop.multiProcessorCount;
    IFF_CREATE_HIP(hipStreamSynchronize(s));   // the source tensors may be released by the caller after return
    *out = f;
    return 0;
}

#define IFF_FIELD_ARGS(f, p, n) IFF_REQUIRE((f) != nullptr && ((n) == 0 || (p) != nullptr) && (n) >= 0, "%s: bad argument", __func__)

extern "C" int iff_normalize_coord(const iff_field* f, const float* xyz, int64_t n, float* out, void* stream) {
    IFF_FIELD_ARGS(f, xyz, n);
    if (n == 0) return 0;
    IFF_HIP(launch_normalize_coord(f->dev, xyz, n, out, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_mask_sample(const iff_field* f, const float* xyz, int64_t n, float* out, void* stream) {
    IFF_FIELD_ARGS(f, xyz, n);
    if (n == 0) return 0;
    IFF_HIP(launch_mask_sample(f->dev, xyz, n, out, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_mask_occupied(const iff_field* f, const float* xyz, int64_t n, uint8_t* out, void* stream) {
    IFF_FIELD_ARGS(f, xyz, n);
    if (n == 0) return 0;
    IFF_REQUIRE(out != nullptr, "iff_mask_occupied: null output");
    IFF_HIP(launch_mask_occupied(f->dev, xyz, n, out, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_density_feature(const iff_field* f, const float* xn, int64_t n, float* out, void* stream) {
    IFF_FIELD_ARGS(f, xn, n);
    if (n == 0) return 0;
    IFF_HIP(launch_density_feature(f->dev, xn, n, out, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_app_feature(const iff_field* f, const float* xn, int64_t n, float* out, void* stream) {
    IFF_FIELD_ARGS(f, xn, n);
    if (n == 0) return 0;
    IFF_HIP(launch_app_feature(f->dev, xn, n, out, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_point_alpha(const iff_field* f, const float* xyz, int64_t n, float length, float* alpha, void* stream) {
    IFF_FIELD_ARGS(f, xyz, n);
    if (n == 0) return 0;
    IFF_HIP(launch_point_alpha(f->dev, xyz, n, length, alpha, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_point_normals(const iff_field* f, const float* xyz, int64_t n, float* normals, void* stream) {
    IFF_FIELD_ARGS(f, xyz, n);
    if (n == 0) return 0;
    IFF_HIP(launch_point_normals(f->dev, xyz, n, normals, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_ref_shade(const iff_field* f, const float* viewdirs, const float* features, int64_t n, float* rgb,
                             void* stream) {
    IFF_FIELD_ARGS(f, viewdirs, n);
    if (n == 0) return 0;
    IFF_REQUIRE(features && rgb, "iff_ref_shade: null buffer");
    IFF_HIP(launch_ref_shade(f->dev, viewdirs, features, n, rgb, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_ref_normals(const iff_field* f, const float* features, int64_t n, float* normals, void* stream) {
    IFF_FIELD_ARGS(f, features, n);
    if (n == 0) return 0;
    IFF_REQUIRE(normals, "iff_ref_normals: null buffer");
    IFF_HIP(launch_ref_normals(f->dev, features, n, normals, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_isocell_emit(const float* cells_host, const float* points, const float* normals, int64_t P, float* ori,
                                float* dirs, float* rays6_opt, void* stream) {
    IFF_REQUIRE(P >= 0, "iff_isocell_emit: negative P");
    if (P == 0) return 0;
    IFF_REQUIRE(cells_host && points && normals && ori && dirs, "iff_isocell_emit: null buffer");
    IFF_HIP(launch_isocell_emit(cells_host, points, normals, P, ori, dirs, rays6_opt, (hipStream_t)stream));
    return 0;
}

static int march_samples(const iff_field* f, int32_t mode, int32_t n_samples) {
    return n_samples > 0 ? n_samples : (mode == IFF_MARCH_POINT_CENTRED ? 20 : f->dev.n_samples);
}

extern "C" int32_t iff_march_default_samples(const iff_field* f, int32_t mode) { return f ? march_samples(f, mode, 0) : 0; }

extern "C" int32_t iff_march_plan(const iff_field* f, int32_t mode, int32_t n_samples) {
    if (!f) return IFF_MARCH_PLAN_GENERAL;
    const int plan = march_plan(f->dev, mode, march_samples(f, mode, n_samples));
    return (plan == IFF_MARCH_PLAN_FAN && march_head_fused(f->dev)) ? IFF_MARCH_PLAN_FAN_HEAD : plan;
}

extern "C" int iff_march_fan_kernel(const iff_field* f, int32_t mode, int32_t n_samples, int32_t* waves, int32_t* patch_side) {
    IFF_REQUIRE(f && waves && patch_side, "iff_march_fan_kernel: null argument");
    const int S = march_samples(f, mode, n_samples);
    const int plan = march_plan(f->dev, mode, S);
    *waves = plan == IFF_MARCH_PLAN_FAN ? fan_kernel_for(f->dev, mode, S) : 0;
    *patch_side = *waves == 8 ? fan8_patch_side(f->dev, mode, S) : (*waves == 4 ? 12 : 0);
    return 0;
}

extern "C" size_t iff_march_workspace(const iff_field* f, int64_t R, int32_t mode, int32_t n_samples) {
    if (!f || R <= 0) return 0;
    return march_workspace_bytes(R, march_samples(f, mode, n_samples));
}

static int march_impl(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                      int32_t n_samples, const float* bg_host, float* rgb, float* depth, float* acc, float* alpha_opt,
                      int32_t* counts_opt, void* workspace, size_t workspace_bytes, float* stage_ms_host, void* stream,
                      float* feat_out = nullptr) {
    IFF_REQUIRE(f != nullptr && R >= 0, "iff_march_shade: bad argument");
    if (R == 0) return 0;
    IFF_REQUIRE(rays && (rgb || feat_out) && depth && acc && bg_host, "iff_march_shade: null buffer");
    IFF_REQUIRE(ray_cols == 6 || ray_cols == 7, "iff_march_shade: rays must have 6 or 7 columns (got %d)", ray_cols);
    IFF_REQUIRE(mode == IFF_MARCH_POINT_CENTRED || mode == IFF_MARCH_SLAB, "iff_march_shade: unknown mode %d", mode);
    int S = march_samples(f, mode, n_samples);
    IFF_REQUIRE(S >= 1 && S <= (1 << 20), "iff_march_shade: n_samples = %d out of range", S);
    if (!workspace || workspace_bytes < march_workspace_bytes(R, S))
        return fail(IFF_ERR_WORKSPACE, "iff_march_shade: workspace %zu < %zu bytes", workspace_bytes, march_workspace_bytes(R, S));
    if (mode == IFF_MARCH_SLAB && f->dev.unisphere)
        return fail(IFF_ERR_UNSUPPORTED, "slab sampler with contraction_type='unisphere' is not built "
                                         "(the reference's own branch is unfinished: models/tensorBase.py:511-525)");
    IFF_HIP(launch_march(f->dev, rays, ray_cols, R, mode, S, bg_host, rgb, depth, acc, alpha_opt, counts_opt, feat_out,
                         workspace, workspace_bytes, stage_ms_host, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_march_features(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                                  int32_t n_samples, float* feat28, float* depth, float* acc, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(feat28 != nullptr, "iff_march_features: feat28 is null");
    const float bg[3] = {0.0f, 0.0f, 0.0f};
    return march_impl(f, rays, ray_cols, R, mode, n_samples, bg, nullptr, depth, acc, nullptr, nullptr, workspace,
                      workspace_bytes, nullptr, stream, feat28);
}

extern "C" size_t iff_march_grad_workspace(const iff_field* f, int64_t R, int32_t mode, int32_t n_samples) {
    if (!f || R <= 0) return 0;
    return march_grad_workspace_bytes(R, march_samples(f, mode, n_samples));
}

extern "C" int iff_march_grad(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                              int32_t n_samples, const float* g_feat28, const float* g_acc, float* g_rays6,
                              void* workspace, size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(f != nullptr && R >= 0, "iff_march_grad: bad argument");
    if (R == 0) return 0;
    IFF_REQUIRE(rays && g_feat28 && g_acc && g_rays6, "iff_march_grad: null buffer");
    IFF_REQUIRE(ray_cols == 6 || ray_cols == 7, "iff_march_grad: rays must have 6 or 7 columns (got %d)", ray_cols);
    IFF_REQUIRE(mode == IFF_MARCH_POINT_CENTRED || mode == IFF_MARCH_SLAB, "iff_march_grad: unknown mode %d", mode);
    int S = march_samples(f, mode, n_samples);
    IFF_REQUIRE(S >= 1 && S <= (1 << 20), "iff_march_grad: n_samples = %d out of range", S);
    IFF_REQUIRE(f->dev.n_density % 4 == 0 && f->dev.n_app % 4 == 0 && f->dev.basis != nullptr && f->dev.dplane[0] != nullptr,
                "iff_march_grad: the handle carries no VM tables");
    if (f->dev.unisphere)
        return fail(IFF_ERR_UNSUPPORTED, "iff_march_grad: contraction_type='unisphere' is not built (the reference's slab "
                                         "sampler for it is unfinished: models/tensorBase.py:511-525)");
    if (!workspace || workspace_bytes < march_grad_workspace_bytes(R, S))
        return fail(IFF_ERR_WORKSPACE, "iff_march_grad: workspace %zu < %zu bytes", workspace_bytes, march_grad_workspace_bytes(R, S));
    IFF_HIP(launch_march_grad(f->dev, rays, ray_cols, R, mode, S, g_feat28, 28, g_acc, g_rays6, workspace, workspace_bytes,
                              (hipStream_t)stream));
    return 0;
}

extern "C" int iff_march_shade(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                               int32_t n_samples, const float* bg_host, float* rgb, float* depth, float* acc, float* alpha_opt,
                               int32_t* counts_opt, void* workspace, size_t workspace_bytes, void* stream) {
    return march_impl(f, rays, ray_cols, R, mode, n_samples, bg_host, rgb, depth, acc, alpha_opt, counts_opt, workspace,
                      workspace_bytes, nullptr, stream);
}

extern "C" int iff_march_shade_timed(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                                     int32_t n_samples, const float* bg_host, float* rgb, float* depth, float* acc,
                                     float* alpha_opt, int32_t* counts_opt, void* workspace, size_t workspace_bytes,
                                     float* stage_ms_host, void* stream) {
    IFF_REQUIRE(stage_ms_host != nullptr, "iff_march_shade_timed: stage_ms_host is null");
    stage_ms_host[0] = stage_ms_host[1] = stage_ms_host[2] = 0.0f;
    return march_impl(f, rays, ray_cols, R, mode, n_samples, bg_host, rgb, depth, acc, alpha_opt, counts_opt, workspace,
                      workspace_bytes, stage_ms_host, stream);
}

extern "C" size_t iff_surface_sample_workspace(int64_t P) { return P > 0 ? sampler_workspace_bytes(P) : 0; }

extern "C" int iff_surface_sample_residency(const iff_field* f, int32_t B, int64_t P, int32_t* wgs_per_run, int32_t* device_capacity) {
    IFF_REQUIRE(f && wgs_per_run && device_capacity && B >= 1, "iff_surface_sample_residency: bad argument");
    int w = 0, c = 0;
    IFF_HIP(sampler_residency(P, f->n_cus, sampler_lpc(f->dev, B), B, &w, &c));
    if (sampler_stepped(f->dev)) {       // no workgroup waits for another one: nothing has to be resident together
        const int64_t want = (5 * P * (f->dev.density_lanes == 0 ? 1 : sampler_lpc(f->dev, B)) + 255) / 256;     // quad form: one lane per candidate
        w = (int)(want > 1024 ? 1024 : want);
        c = 0x7fffffff;
    }
    *wgs_per_run = w; *device_capacity = c;
    return 0;
}

extern "C" int iff_surface_sample(const iff_field* f, int64_t P, int32_t n_epochs, int32_t max_iterations, uint64_t seed,
                                  const uint64_t* seed_dev_opt, float rho, float* samples, float* alpha, int32_t* stats, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return iff_surface_sample_batched(f, 1, P, n_epochs, max_iterations, seed, seed_dev_opt, rho, samples, alpha, stats, workspace,
                                      workspace_bytes, stream);
}

extern "C" int iff_surface_sample_batched(const iff_field* f, int32_t B, int64_t P, int32_t n_epochs, int32_t max_iterations,
                                          uint64_t seed, const uint64_t* seed_dev_opt, float rho, float* samples, float* alpha,
                                          int32_t* stats, void* workspace, size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(f && samples && alpha && stats && workspace, "iff_surface_sample: null argument");
    IFF_REQUIRE(P >= 1 && B >= 1, "iff_surface_sample: P and the batch size must be >= 1");
    if (workspace_bytes < sampler_workspace_bytes(P) * (size_t)B)
        return fail(IFF_ERR_WORKSPACE, "iff_surface_sample: workspace %zu < %zu bytes", workspace_bytes,
                    sampler_workspace_bytes(P) * (size_t)B);
    if (f->dev.mask && f->n_occ == 0) return fail(IFF_ERR_INVALID_ARGUMENT, "iff_surface_sample: the occupancy mask is empty");
    IFF_HIP(launch_surface_sample_occ(f->dev, f->occ_list, f->n_occ, B, P, n_epochs, max_iterations, seed, seed_dev_opt, rho, samples,
                                      alpha, stats, workspace, workspace_bytes, f->n_cus, (hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ identification net
struct iff_idnet {
    IdNetDev dev;
    void* slab = nullptr;
    size_t slab_bytes = 0;
};

extern "C" void iff_idnet_destroy(iff_idnet* n) {
    if (!n) return;
    if (n->slab) (void)hipFree(n->slab);
    delete n;
}

// q_proj, k_proj and the last encoder Linear have no nonlinearity between them (ray_preprocessor.py:24-25,38;
// multihead_attention.py:60-63), so with h3 the encoder's last hidden activation
//   q . k = (Wq t + bq) . (Wk (W4 h3 + b4) + bk) = (H t + hb) . h3 + (r . t + r0),
//   G = Wk W4, g = Wk b4 + bk, H = G^T Wq, hb = G^T bq, r = Wq^T g, r0 = bq . g.
// The products are formed once, in double, and rounded to fp32: the same class of error as re-associating an fp32 sum.
// Output: wqf [KQ][ld] k-major (column c < C: H row c; column C: r; others 0), bqf [ld].
static void fold_heads(const float* W4, const float* b4, const float* Wk, const float* bk, const float* Wq, const float* bq, int C,
                       int Fe, int IF, int KQ, int ld, std::vector<float>& wqf, std::vector<float>& bqf) {
    std::vector<double> G((size_t)Fe * C, 0.0), g(Fe, 0.0);
    for (int e = 0; e < Fe; ++e) {
        double* Ge = &G[(size_t)e * C];
        double acc = bk[e];
        for (int f = 0; f < Fe; ++f) {
            const double w = Wk[(size_t)e * Fe + f];
            const float* w4 = W4 + (size_t)f * C;
            for (int c = 0; c < C; ++c) Ge[c] += w * (double)w4[c];
            acc += w * (double)b4[f];
        }
        g[e] = acc;
    }
    std::vector<double> H((size_t)KQ * ld, 0.0), hb(ld, 0.0);
    for (int e = 0; e < Fe; ++e) {
        const double* Ge = &G[(size_t)e * C];
        for (int i = 0; i < IF; ++i) {
            const double w = Wq[(size_t)e * IF + i];
            double* Hi = &H[(size_t)i * ld];
            for (int c = 0; c < C; ++c) Hi[c] += w * Ge[c];
            Hi[C] += w * g[e];
        }
        for (int c = 0; c < C; ++c) hb[c] += (double)bq[e] * Ge[c];
        hb[C] += (double)bq[e] * g[e];
    }
    wqf.resize((size_t)KQ * ld);
    bqf.resize(ld);
    for (size_t t = 0; t < H.size(); ++t) wqf[t] = (float)H[t];
    for (int c = 0; c < ld; ++c) bqf[c] = (float)hb[c];
}

// IFF_GEMM_F16X2: powers of two for the fp16 hi/lo planes of the fused encoder kernel (trunk_f16_kernels.hip).  fp16 tops out
// at 65504, so every operand is scaled such that a worst-case bound of its magnitude stays <= 2^15.  Bounds: the encoder
// input x has raw ray origins in columns 0..2 (|o| <= F16_ORIGIN_BOUND, scene units), directions / colours (|.| <= 2) and
// sines / cosines (<= 1) elsewhere; a layer's output is bounded by max_row(sum_j |W[r][j]| bound_j + |b[r]|).  The bounds
// are loose (L1 norms), typical activations sit a few powers of two below them, which is exactly the headroom the lo term
// needs to stay a normal fp16.  Returns false (the handle then keeps the 3xBF16 kernel) when a scale would have to drop below
// 2^-3, i.e. when fp16's range cannot hold the network's worst case with useful precision.
static const float F16_ORIGIN_BOUND = 64.0f;
static int exp_below(double bound, double top = 32768.0) {      // largest e with bound * 2^e <= top
    if (!(bound > 0.0)) return 15;
    int e = (int)std::floor(std::log2(top / bound));
    return e > 20 ? 20 : e;
}
static bool plan_f16_scales(const float* W1, const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, int C,
                            IdNetDev& v) {
    const int IN = IFF_RAY_INPUT;
    std::vector<double> xb(IN, 1.0);
    for (int j = 0; j < 3; ++j) xb[j] = F16_ORIGIN_BOUND;
    for (int j = 3; j < 9; ++j) xb[j] = 2.0;
    auto maxabs = [](const float* w, size_t n) { double m = 0; for (size_t i = 0; i < n; ++i) m = std::fmax(m, std::fabs((double)w[i])); return m; };
    double H1 = 0, H2 = 0, H3 = 0, w3h_max = 0, w3x_max = 0;
    for (int r = 0; r < C; ++r) {
        double s = std::fabs((double)b1[r]);
        for (int j = 0; j < IN; ++j) s += std::fabs((double)W1[(size_t)r * IN + j]) * xb[j];
        H1 = std::fmax(H1, s);
    }
    for (int r = 0; r < C; ++r) {
        double s = 0;
        for (int j = 0; j < C; ++j) s += std::fabs((double)W2[(size_t)r * C + j]);
        H2 = std::fmax(H2, s * H1 + std::fabs((double)b2[r]));
    }
    for (int r = 0; r < C; ++r) {
        const float* row = W3 + (size_t)r * (C + IN);
        double sh = 0, sx = 0;
        for (int j = 0; j < C; ++j) { sh += std::fabs((double)row[j]); w3h_max = std::fmax(w3h_max, std::fabs((double)row[j])); }
        for (int j = 0; j < IN; ++j) { sx += std::fabs((double)row[C + j]) * xb[j]; w3x_max = std::fmax(w3x_max, std::fabs((double)row[C + j])); }
        H3 = std::fmax(H3, sh * H2 + sx + std::fabs((double)b3[r]));
    }
    v.e_x = exp_below(F16_ORIGIN_BOUND);
    v.e_h1 = exp_below(H1); v.e_h2 = exp_below(H2); v.e_h3 = exp_below(H3);
    v.e_w1 = exp_below(maxabs(W1, (size_t)C * IN));
    v.e_w2 = exp_below(maxabs(W2, (size_t)C * C));
    // layer 3 accumulates its h-part and its x-part together: e_w3h + e_h2 == e_w3x + e_x
    int e3h = exp_below(w3h_max), e3x = exp_below(w3x_max);
    if (e3h + v.e_h2 - v.e_x > e3x) e3h = e3x - v.e_h2 + v.e_x;
    v.e_w3h = e3h; v.e_w3x = e3h + v.e_h2 - v.e_x;
    return std::isfinite(H3) && v.e_h1 >= -3 && v.e_h2 >= -3 && v.e_h3 >= -3 && v.e_w3x >= -8 && v.e_w3h >= -8;
}

extern "C" int iff_idnet_dims(const iff_idnet* n, int32_t* feature_c, int32_t* fea, int32_t* img_fea) {
    IFF_REQUIRE(n && feature_c && fea && img_fea, "iff_idnet_dims: null argument");
    *feature_c = n->dev.feature_c; *fea = n->dev.fea; *img_fea = n->dev.img_fea;
    return 0;
}

extern "C" int32_t iff_idnet_gemm_mode(const iff_idnet* n) {
    if (!n) return -1;
    if (n->dev.trunk_f16) return n->dev.trunk_f16 == 2 ? IFF_GEMM_F16X1 : IFF_GEMM_F16X2;
    return n->dev.gemm_mode == 0 ? IFF_GEMM_F32 : (n->dev.fused_trunk ? IFF_GEMM_BF16X3 : IFF_GEMM_BF16X3_LAYERED);
}

// Slab layout of an identification-net handle as a function of its widths and of which optional table sets it carries
// (fused: the fragment-ordered bf16 planes of a 256-wide encoder; f16: the fp16 hi/lo planes of IFF_GEMM_F16X2) -- shared by
// iff_idnet_create and the table-file check of iff_idnet_load.
struct IdLayout { size_t w1, b1, w2, b2, w3, b3, w4, b4, wk, bk, wq, bq, p1, p2, p3, p4, pk, wqf, bqf, f1, f2, f3, h1, h2, h3h, h3x, total; };
static IdLayout idnet_layout(int C, int Fe, int IF, bool fused, bool want_f16) {
    const int KQ = (IF + 15) / 16 * 16, XW = 160, QLD = C + 16, KXS = (IFF_RAY_INPUT + 15) / 16;
    IdLayout L;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off = up256(off + floats * 4); return o; };
    // bf16 planes: 3 x out x K_pad halves = 1.5 floats per element
    auto take_planes = [&](size_t out_f, size_t kpad) { return take((out_f * kpad * 3 + 1) / 2); };
    L.w1 = take((size_t)XW * C); L.b1 = take(C); L.w2 = take((size_t)C * C); L.b2 = take(C);
    L.w3 = take((size_t)(C + XW) * C); L.b3 = take(C); L.w4 = take((size_t)C * Fe); L.b4 = take(Fe);
    L.wk = take((size_t)Fe * Fe); L.bk = take(Fe); L.wq = take((size_t)KQ * Fe); L.bq = take(Fe);
    L.p1 = take_planes(C, XW); L.p2 = take_planes(C, C); L.p3 = take_planes(C, C + XW); L.p4 = take_planes(Fe, C); L.pk = take_planes(Fe, Fe);
    L.wqf = take((size_t)KQ * QLD); L.bqf = take(QLD);
    L.f1 = fused ? take_planes(C, XW) : 0; L.f2 = fused ? take_planes(C, C) : 0; L.f3 = fused ? take_planes(C, C + XW) : 0;
    // fp16 hi/lo planes in fragment order (IFF_GEMM_F16X2): [k-steps][2][256][16] halves = k-steps * 4096 floats
    L.h1 = want_f16 ? take((size_t)KXS * 4096) : 0; L.h2 = want_f16 ? take((size_t)(C / 16) * 4096) : 0;
    L.h3h = want_f16 ? take((size_t)(C / 16) * 4096) : 0; L.h3x = want_f16 ? take((size_t)KXS * 4096) : 0;
    L.total = off;
    return L;
}
static int check_idnet_dims(int C, int Fe, int IF) {
    IFF_REQUIRE(C >= 16 && C % 16 == 0 && Fe >= 16 && Fe % 16 == 0 && IF >= 1 && C <= 4096 && Fe <= 4096 && IF <= 4096,
                "identification net: widths %d/%d/%d unsupported", C, Fe, IF);
    IFF_REQUIRE(C % 32 == 0 && Fe % 32 == 0, "identification net: widths must be multiples of 32 (got %d, %d)", C, Fe);
    return 0;
}

extern "C" int iff_idnet_create(const iff_idnet_desc* d, void* stream, iff_idnet** out) {
    IFF_REQUIRE(d && out, "iff_idnet_create: null argument");
    *out = nullptr;
    hipStream_t s = (hipStream_t)stream;
    IFF_REQUIRE(d->l1_w && d->l1_b && d->l2_w && d->l2_b && d->l3_w && d->l3_b && d->l4_w && d->l4_b && d->q_w && d->q_b &&
                    d->k_w && d->k_b, "iff_idnet_create: null weight");
    const int C = d->feature_c, Fe = d->fea, IF = d->img_fea;
    if (int rc = check_idnet_dims(C, Fe, IF)) return rc;
    IFF_REQUIRE(d->gemm_mode >= 0 && d->gemm_mode <= 4, "iff_idnet_create: gemm_mode must be one of IFF_GEMM_* (0..4)");
    IFF_REQUIRE(d->trunk_variant >= 0 && d->trunk_variant <= 4, "iff_idnet_create: trunk_variant must be 0 (choose) or 1..4");
    const int KQ = (IF + 15) / 16 * 16;
    const int XW = 160;                  // encoder input 141 padded to a multiple of 32 (identify_kernels.hip)
    const int QLD = C + 16;
    const int KXS = (IFF_RAY_INPUT + 15) / 16;
    const bool fused = (C == 256);
    const bool want_f16 = fused && (d->gemm_mode == IFF_GEMM_F16X2 || d->gemm_mode == IFF_GEMM_F16X1);
    const IdLayout L = idnet_layout(C, Fe, IF, fused, want_f16);
    const size_t o_w1 = L.w1, o_b1 = L.b1, o_w2 = L.w2, o_b2 = L.b2, o_w3 = L.w3, o_b3 = L.b3, o_w4 = L.w4, o_b4 = L.b4, o_wk = L.wk,
                 o_bk = L.bk, o_wq = L.wq, o_bq = L.bq, o_p1 = L.p1, o_p2 = L.p2, o_p3 = L.p3, o_p4 = L.p4, o_pk = L.pk, o_wqf = L.wqf,
                 o_bqf = L.bqf, o_f1 = L.f1, o_f2 = L.f2, o_f3 = L.f3, o_h1 = L.h1, o_h2 = L.h2, o_h3h = L.h3h, o_h3x = L.h3x;
    const size_t off = L.total;
    iff_idnet* n = new iff_idnet();
    n->slab_bytes = off;
    hipError_t e = hipMalloc(&n->slab, off);
    if (e != hipSuccess) { delete n; return hip_fail(e, "hipMalloc(idnet weights)"); }
    char* b = (char*)n->slab;
#define IFF_NET_HIP(call)                                              \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) { iff_idnet_destroy(n); return hip_fail(e__, #call); } \
    } while (0)
    IFF_NET_HIP(hipMemsetAsync(n->slab, 0, off, s));
    IFF_NET_HIP(launch_transpose_pad(d->l1_w, (float*)(b + o_w1), C, IFF_RAY_INPUT, XW, 0, s));
    IFF_NET_HIP(launch_transpose_pad(d->l2_w, (float*)(b + o_w2), C, C, C, 0, s));
    // layer 3 consumes [h (C) | x (141)]: its weight columns 0..C-1 go to rows 0..C-1, columns C.. to rows C..C+140
    IFF_NET_HIP(launch_transpose_pad(d->l3_w, (float*)(b + o_w3), C, C + IFF_RAY_INPUT, C + XW, 0, s));
    IFF_NET_HIP(launch_transpose_pad(d->l4_w, (float*)(b + o_w4), Fe, C, C, 0, s));
    IFF_NET_HIP(launch_transpose_pad(d->k_w, (float*)(b + o_wk), Fe, Fe, Fe, 0, s));
    IFF_NET_HIP(launch_transpose_pad(d->q_w, (float*)(b + o_wq), Fe, IF, KQ, 0, s));
    IFF_NET_HIP(launch_split_rows(d->l1_w, b + o_p1, C, IFF_RAY_INPUT, XW, s));
    IFF_NET_HIP(launch_split_rows(d->l2_w, b + o_p2, C, C, C, s));
    IFF_NET_HIP(launch_split_rows(d->l3_w, b + o_p3, C, C + IFF_RAY_INPUT, C + XW, s));
    IFF_NET_HIP(launch_split_rows(d->l4_w, b + o_p4, Fe, C, C, s));
    IFF_NET_HIP(launch_split_rows(d->k_w, b + o_pk, Fe, Fe, Fe, s));
    struct { const float* src; size_t off; int n; } bs[] = {{d->l1_b, o_b1, C}, {d->l2_b, o_b2, C}, {d->l3_b, o_b3, C},
                                                            {d->l4_b, o_b4, Fe}, {d->k_b, o_bk, Fe}, {d->q_b, o_bq, Fe}};
    for (auto& p : bs) IFF_NET_HIP(hipMemcpyAsync(b + p.off, p.src, (size_t)p.n * 4, hipMemcpyDeviceToDevice, s));
    IdNetDev& v = n->dev;
    v.w1 = (const float*)(b + o_w1); v.b1 = (const float*)(b + o_b1); v.w2 = (const float*)(b + o_w2); v.b2 = (const float*)(b + o_b2);
    v.w3 = (const float*)(b + o_w3); v.b3 = (const float*)(b + o_b3); v.w4 = (const float*)(b + o_w4); v.b4 = (const float*)(b + o_b4);
    v.wk = (const float*)(b + o_wk); v.bk = (const float*)(b + o_bk); v.wq = (const float*)(b + o_wq); v.bq = (const float*)(b + o_bq);
    v.p1 = b + o_p1; v.p2 = b + o_p2; v.p3 = b + o_p3; v.p4 = b + o_p4; v.pk = b + o_pk;
    v.gemm_mode = d->gemm_mode == 0 ? 0 : 1;
    v.h1 = v.h2 = v.h3h = v.h3x = nullptr;
    v.trunk_f16 = 0; v.trunk_variant = 0;
    v.e_x = v.e_h1 = v.e_h2 = v.e_h3 = v.e_w1 = v.e_w2 = v.e_w3h = v.e_w3x = 0;
    v.feature_c = C; v.fea = Fe; v.img_fea = IF;
    v.f1 = v.f2 = v.f3 = nullptr;
    v.fused_trunk = 0;
    if (fused) {
        IFF_NET_HIP(launch_frag_order(b + o_p1, b + o_f1, XW, s));
        IFF_NET_HIP(launch_frag_order(b + o_p2, b + o_f2, C, s));
        IFF_NET_HIP(launch_frag_order(b + o_p3, b + o_f3, C + XW, s));
        v.f1 = b + o_f1; v.f2 = b + o_f2; v.f3 = b + o_f3;
        v.fused_trunk = (d->gemm_mode == IFF_GEMM_BF16X3 || d->gemm_mode == IFF_GEMM_F16X2 || d->gemm_mode == IFF_GEMM_F16X1) ? 1 : 0;
    }
    if (want_f16) {
        std::vector<float> W1((size_t)C * IFF_RAY_INPUT), b1(C), W2((size_t)C * C), b2(C), W3((size_t)C * (C + IFF_RAY_INPUT)), b3(C);
        struct { std::vector<float>* dst; const float* src; } dl[] = {{&W1, d->l1_w}, {&b1, d->l1_b}, {&W2, d->l2_w},
                                                                      {&b2, d->l2_b}, {&W3, d->l3_w}, {&b3, d->l3_b}};
        IFF_NET_HIP(hipStreamSynchronize(s));
        for (auto& p : dl) IFF_NET_HIP(hipMemcpy(p.dst->data(), p.src, p.dst->size() * 4, hipMemcpyDeviceToHost));
        if (plan_f16_scales(W1.data(), b1.data(), W2.data(), b2.data(), W3.data(), b3.data(), C, v)) {
            IFF_NET_HIP(launch_frag_order_h(d->l1_w, IFF_RAY_INPUT, 0, IFF_RAY_INPUT, KXS, ldexpf(1.0f, v.e_w1), b + o_h1, s));
            IFF_NET_HIP(launch_frag_order_h(d->l2_w, C, 0, C, C / 16, ldexpf(1.0f, v.e_w2), b + o_h2, s));
            IFF_NET_HIP(launch_frag_order_h(d->l3_w, C + IFF_RAY_INPUT, 0, C, C / 16, ldexpf(1.0f, v.e_w3h), b + o_h3h, s));
            IFF_NET_HIP(launch_frag_order_h(d->l3_w, C + IFF_RAY_INPUT, C, IFF_RAY_INPUT, KXS, ldexpf(1.0f, v.e_w3x), b + o_h3x, s));
            v.h1 = b + o_h1; v.h2 = b + o_h2; v.h3h = b + o_h3h; v.h3x = b + o_h3x;
            v.trunk_f16 = d->gemm_mode == IFF_GEMM_F16X1 ? 2 : 1;          // 2: one product per block (trunk_f16_kernels.hip, MODE + 4)
            v.trunk_variant = d->trunk_variant > 0 ? d->trunk_variant - 1 : 0;
        }
    }
    IFF_NET_HIP(hipStreamSynchronize(s));
    {
        std::vector<float> W4((size_t)Fe * C), b4(Fe), Wk((size_t)Fe * Fe), bk(Fe), Wq((size_t)Fe * IF), bq(Fe), wqf, bqf;
        struct { std::vector<float>* dst; const float* src; } dl[] = {{&W4, d->l4_w}, {&b4, d->l4_b}, {&Wk, d->k_w},
                                                                      {&bk, d->k_b}, {&Wq, d->q_w}, {&bq, d->q_b}};
        for (auto& p : dl) IFF_NET_HIP(hipMemcpy(p.dst->data(), p.src, p.dst->size() * 4, hipMemcpyDeviceToHost));
        fold_heads(W4.data(), b4.data(), Wk.data(), bk.data(), Wq.data(), bq.data(), C, Fe, IF, KQ, QLD, wqf, bqf);
        IFF_NET_HIP(hipMemcpy(b + o_wqf, wqf.data(), wqf.size() * 4, hipMemcpyHostToDevice));
        IFF_NET_HIP(hipMemcpy(b + o_bqf, bqf.data(), bqf.size() * 4, hipMemcpyHostToDevice));
        v.wqf = (const float*)(b + o_wqf); v.bqf = (const float*)(b + o_bqf); v.qf_ld = QLD;
    }
    *out = n;
    return 0;
}

extern "C" size_t iff_ray_encode_workspace(const iff_idnet* n, int64_t N) {
    return (n && N > 0) ? ray_encode_workspace_bytes(n->dev, N) : 0;
}

extern "C" int iff_ray_encode(const iff_idnet* n, const float* o, const float* d, const float* rgb, int64_t N, float* feat_opt,
                              float* k_out, void* workspace, size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(n && N >= 0, "iff_ray_encode: bad argument");
    if (N == 0) return 0;
    IFF_REQUIRE(o && d && rgb && workspace, "iff_ray_encode: null buffer");
    IFF_REQUIRE(feat_opt || k_out, "iff_ray_encode: no output requested");
    if (workspace_bytes < ray_encode_workspace_bytes(n->dev, N))
        return fail(IFF_ERR_WORKSPACE, "iff_ray_encode: workspace %zu < %zu bytes", workspace_bytes, ray_encode_workspace_bytes(n->dev, N));
    IFF_HIP(launch_ray_encode(n->dev, o, d, rgb, N, feat_opt, k_out, workspace, workspace_bytes, (hipStream_t)stream));
    return 0;
}

extern "C" size_t iff_ray_trunk_workspace(const iff_idnet* n, int64_t N) {
    return (n && N > 0) ? ray_trunk_workspace_bytes(n->dev, N) : 0;
}

extern "C" int iff_ray_trunk(const iff_idnet* n, const float* o, const float* d, const float* rgb, int64_t N, float* h3,
                             void* workspace, size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(n && N >= 0, "iff_ray_trunk: bad argument");
    if (N == 0) return 0;
    IFF_REQUIRE(o && d && rgb && h3 && workspace, "iff_ray_trunk: null buffer");
    if (workspace_bytes < ray_trunk_workspace_bytes(n->dev, N))
        return fail(IFF_ERR_WORKSPACE, "iff_ray_trunk: workspace %zu < %zu bytes", workspace_bytes, ray_trunk_workspace_bytes(n->dev, N));
    IFF_HIP(launch_ray_trunk(n->dev, o, d, rgb, N, h3, workspace, workspace_bytes, (hipStream_t)stream));
    return 0;
}

extern "C" int32_t iff_q_fold_width(const iff_idnet* n) { return n ? n->dev.qf_ld : 0; }

extern "C" int iff_q_fold(const iff_idnet* n, const float* img, int32_t M, float* qf, void* stream) {
    IFF_REQUIRE(n && M >= 0, "iff_q_fold: bad argument");
    if (M == 0) return 0;
    IFF_REQUIRE(img && qf, "iff_q_fold: null buffer");
    IFF_HIP(launch_q_fold(n->dev, img, M, qf, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_attn_logits_folded(const iff_idnet* n, const float* qf, const float* h3, int32_t M, int64_t N, float divisor,
                                      float* logits, float* row_max, float* row_sumexp, void* stream) {
    IFF_REQUIRE(n && M >= 0 && N >= 0, "iff_attn_logits_folded: bad argument");
    if (M == 0 || N == 0) return 0;
    IFF_REQUIRE(qf && h3 && logits, "iff_attn_logits_folded: null buffer");
    IFF_REQUIRE((row_max == nullptr) == (row_sumexp == nullptr), "iff_attn_logits_folded: pass both row statistics or neither");
    IFF_HIP(launch_attn_logits_folded(qf, n->dev.qf_ld, h3, M, N, n->dev.feature_c, divisor, logits, row_max, row_sumexp,
                                      (hipStream_t)stream));
    return 0;
}

extern "C" size_t iff_ray_logits_folded_workspace(const iff_idnet* n, int64_t N, int32_t M) {
    return (n && N > 0 && M > 0) ? ray_logits_workspace_bytes(n->dev, N, M, 1) : 0;
}
extern "C" size_t iff_ray_logits_folded_batched_workspace(const iff_idnet* n, int32_t B, int64_t N, int32_t M) {
    return (n && N > 0 && M > 0 && B > 0) ? ray_logits_workspace_bytes(n->dev, N, M, B) : 0;
}

static int ray_logits_common(const char* who, const iff_idnet* n, int32_t B, const float* o, const float* d, const float* rgb, int64_t N,
                             const float* qf, int32_t M, float divisor, float* logits, float* row_max, float* row_sumexp,
                             void* workspace, size_t workspace_bytes, float* trunk_ms_host, void* stream) {
    IFF_REQUIRE(n && N >= 0 && M >= 0 && B >= 0, "%s: bad argument", who);
    if (N == 0 || M == 0 || B == 0) return 0;
    IFF_REQUIRE(o && d && rgb && qf && logits && workspace, "%s: null buffer", who);
    IFF_REQUIRE((row_max == nullptr) == (row_sumexp == nullptr), "%s: pass both row statistics or neither", who);
    if (workspace_bytes < ray_logits_workspace_bytes(n->dev, N, M, B))
        return fail(IFF_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, ray_logits_workspace_bytes(n->dev, N, M, B));
    IFF_HIP(launch_ray_logits_folded(n->dev, o, d, rgb, N, qf, M, B, divisor, logits, row_max, row_sumexp, workspace, workspace_bytes,
                                     trunk_ms_host, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_ray_logits_folded(const iff_idnet* n, const float* o, const float* d, const float* rgb, int64_t N, const float* qf,
                                     int32_t M, float divisor, float* logits, float* row_max, float* row_sumexp, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    return ray_logits_common("iff_ray_logits_folded", n, 1, o, d, rgb, N, qf, M, divisor, logits, row_max, row_sumexp, workspace,
                             workspace_bytes, nullptr, stream);
}

extern "C" int iff_ray_logits_folded_batched(const iff_idnet* n, int32_t B, const float* o, const float* d, const float* rgb, int64_t N,
                                             const float* qf, int32_t M, float divisor, float* logits, float* row_max,
                                             float* row_sumexp, void* workspace, size_t workspace_bytes, void* stream) {
    return ray_logits_common("iff_ray_logits_folded_batched", n, B, o, d, rgb, N, qf, M, divisor, logits, row_max, row_sumexp,
                             workspace, workspace_bytes, nullptr, stream);
}

extern "C" int iff_ray_logits_folded_timed(const iff_idnet* n, int32_t B, const float* o, const float* d, const float* rgb, int64_t N,
                                           const float* qf, int32_t M, float divisor, float* logits, float* row_max,
                                           float* row_sumexp, void* workspace, size_t workspace_bytes, float* trunk_ms_host,
                                           void* stream) {
    IFF_REQUIRE(trunk_ms_host && N >= 1 && M >= 1 && B >= 1, "iff_ray_logits_folded_timed: bad argument");
    return ray_logits_common("iff_ray_logits_folded_timed", n, B, o, d, rgb, N, qf, M, divisor, logits, row_max, row_sumexp, workspace,
                             workspace_bytes, trunk_ms_host, stream);
}

// ---- image side of stage C
extern "C" int iff_token_assemble(const float* patch_tokens, int32_t Q, int32_t gh, int32_t gw, int32_t C, const float* mask_grid_opt,
                                  float mask_thres, const float* lin_h_host, const float* lin_w_host, float* tokens_out,
                                  uint8_t* keep_out, void* stream) {
    IFF_REQUIRE(Q >= 0 && gh >= 1 && gh <= 32 && gw >= 1 && gw <= 32 && C >= 1, "iff_token_assemble: bad shape Q=%d grid=%dx%d C=%d", Q, gh, gw, C);
    if (Q == 0) return 0;
    IFF_REQUIRE(patch_tokens && lin_h_host && lin_w_host && tokens_out && keep_out, "iff_token_assemble: null buffer");
    IFF_HIP(launch_token_assemble(patch_tokens, Q, gh, gw, C, mask_grid_opt, mask_thres, lin_h_host, lin_w_host, tokens_out, keep_out,
                                  (hipStream_t)stream));
    return 0;
}
extern "C" int iff_token_assemble_compact(const float* patch_tokens, int32_t Q, int32_t gh, int32_t gw, int32_t C, const float* mask_grid_opt,
                                          float mask_thres, const float* lin_h_host, const float* lin_w_host, float* tokens_out,
                                          uint8_t* keep_out, int32_t* rows_out, void* stream) {
    IFF_REQUIRE(Q >= 0 && gh >= 1 && gh <= 32 && gw >= 1 && gw <= 32 && C >= 1, "iff_token_assemble_compact: bad shape Q=%d grid=%dx%d C=%d", Q, gh, gw, C);
    if (Q == 0) return 0;
    IFF_REQUIRE(patch_tokens && lin_h_host && lin_w_host && tokens_out && keep_out && rows_out, "iff_token_assemble_compact: null buffer");
    IFF_HIP(launch_token_assemble_compact(patch_tokens, Q, gh, gw, C, mask_grid_opt, mask_thres, lin_h_host, lin_w_host, tokens_out, keep_out,
                                          rows_out, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_mask_token_rows(const uint8_t* keep, int64_t rows, float* row_max, float* row_sumexp, void* stream) {
    IFF_REQUIRE(rows >= 0, "iff_mask_token_rows: bad argument");
    if (rows == 0) return 0;
    IFF_REQUIRE(keep && row_max && row_sumexp, "iff_mask_token_rows: null buffer");
    IFF_HIP(launch_mask_token_rows(keep, rows, row_max, row_sumexp, (hipStream_t)stream));
    return 0;
}

// ---- per-model encoder cache
static size_t up256a(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t iff_ray_cache_bytes(const iff_idnet* n, int64_t N) {
    if (!n || N <= 0) return 0;
    // F16X2: fp16 hi/lo planes [2][N][C]; otherwise the fp32 activation [N][C]
    return (size_t)N * n->dev.feature_c * 4;
}
extern "C" size_t iff_ray_cache_workspace(const iff_idnet* n, int64_t N) {
    return (n && N > 0 && !n->dev.trunk_f16) ? ray_trunk_workspace_bytes(n->dev, N) : 0;
}
extern "C" int iff_ray_cache_build(const iff_idnet* n, const float* o, const float* d, const float* rgb, int64_t N, void* cache,
                                   size_t cache_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(n && N >= 0, "iff_ray_cache_build: bad argument");
    if (N == 0) return 0;
    IFF_REQUIRE(o && d && rgb && cache, "iff_ray_cache_build: null buffer");
    if (cache_bytes < iff_ray_cache_bytes(n, N)) return fail(IFF_ERR_WORKSPACE, "iff_ray_cache_build: cache %zu < %zu bytes", cache_bytes, iff_ray_cache_bytes(n, N));
    if (n->dev.trunk_f16) {
        IFF_HIP(launch_trunk_h_cache(n->dev, o, d, rgb, N, cache, (hipStream_t)stream));
        return 0;
    }
    if (!workspace || workspace_bytes < ray_trunk_workspace_bytes(n->dev, N))
        return fail(IFF_ERR_WORKSPACE, "iff_ray_cache_build: workspace %zu < %zu bytes", workspace_bytes, ray_trunk_workspace_bytes(n->dev, N));
    IFF_HIP(launch_ray_trunk(n->dev, o, d, rgb, N, (float*)cache, workspace, workspace_bytes, (hipStream_t)stream));
    return 0;
}
extern "C" size_t iff_logits_from_cache_workspace(const iff_idnet* n, int64_t N, int32_t M) {
    if (!n || N <= 0 || M <= 0 || !n->dev.trunk_f16) return 0;
    const size_t n_tb = (size_t)(M + 255) / 256, n_blk = (size_t)(N + 63) / 64;
    return up256a(n_tb * (n->dev.feature_c / 16) * 2 * 8 * 64 * 16) + up256a(n_blk * n_tb * 256 * 8) + up256a(n_tb * 256 * 4) + 256;
}
static int logits_from_cache(const iff_idnet* n, const void* cache, int64_t N, const float* qf, int32_t M, const int32_t* rows, float divisor,
                             float* logits, float* row_max, float* row_sumexp, void* workspace, size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(n && N >= 0 && M >= 0, "iff_logits_from_cache: bad argument");
    if (N == 0 || M == 0) return 0;
    IFF_REQUIRE(cache && qf && logits, "iff_logits_from_cache: null buffer");
    IFF_REQUIRE((row_max == nullptr) == (row_sumexp == nullptr), "iff_logits_from_cache: pass both row statistics or neither");
    if (!n->dev.trunk_f16) {
        // (the other arithmetic modes compute every row; the rows behind a block's count then hold ordinary logits and statistics,
        // which a caller that stops at the counts never reads)
        IFF_HIP(launch_attn_logits_folded(qf, n->dev.qf_ld, (const float*)cache, M, N, n->dev.feature_c, divisor, logits, row_max,
                                          row_sumexp, (hipStream_t)stream));
        return 0;
    }
    if (!workspace || workspace_bytes < iff_logits_from_cache_workspace(n, N, M))
        return fail(IFF_ERR_WORKSPACE, "iff_logits_from_cache: workspace %zu < %zu bytes", workspace_bytes, iff_logits_from_cache_workspace(n, N, M));
    const size_t n_tb = (size_t)(M + 255) / 256, n_blk64 = (size_t)(N + 63) / 64;
    char* base = (char*)workspace;
    void* Qf = base;
    float2* part = (float2*)(base + up256a(n_tb * (n->dev.feature_c / 16) * 2 * 8 * 64 * 16));
    float* qscale = (float*)((char*)part + up256a(n_blk64 * n_tb * 256 * 8));
    IFF_HIP(launch_trunk_h_logits_cached(n->dev, cache, N, qf, M, divisor, logits, Qf, qscale, part, rows, (hipStream_t)stream));
    if (row_max) {
        // (the partials are per 64-ray block whatever tile the launch used)
        IFF_HIP(launch_merge_stats(part, (int)n_blk64, (int)(n_tb * 256), M, 1, row_max, row_sumexp, rows, (hipStream_t)stream));
    }
    return 0;
}
extern "C" int iff_logits_from_cache(const iff_idnet* n, const void* cache, int64_t N, const float* qf, int32_t M, float divisor,
                                     float* logits, float* row_max, float* row_sumexp, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return logits_from_cache(n, cache, N, qf, M, nullptr, divisor, logits, row_max, row_sumexp, workspace, workspace_bytes, stream);
}
extern "C" int iff_logits_from_cache_rows(const iff_idnet* n, const void* cache, int64_t N, const float* qf, int32_t M,
                                          const int32_t* rows_per_block, float divisor, float* logits, float* row_max, float* row_sumexp,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(rows_per_block, "iff_logits_from_cache_rows: null row counts (iff_logits_from_cache is the call without them)");
    IFF_REQUIRE(M > 0 && M % 256 == 0, "iff_logits_from_cache_rows: M = %d is not a whole number of 256-row token blocks (one block per "
                "image, rows_per_block holds M / 256 counts)", M);
    return logits_from_cache(n, cache, N, qf, M, rows_per_block, divisor, logits, row_max, row_sumexp, workspace, workspace_bytes, stream);
}

extern "C" int iff_k_proj(const iff_idnet* n, const float* ray_features, int64_t N, float* k_out, void* stream) {
    IFF_REQUIRE(n && N >= 0, "iff_k_proj: bad argument");
    if (N == 0) return 0;
    IFF_REQUIRE(ray_features && k_out, "iff_k_proj: null buffer");
    IFF_HIP(launch_k_proj(n->dev, ray_features, N, k_out, (hipStream_t)stream));
    return 0;
}

extern "C" size_t iff_q_proj_workspace(const iff_idnet* n, int32_t M) {
    if (!n || M <= 0) return 0;
    return (size_t)M * ((n->dev.img_fea + 15) / 16 * 16) * sizeof(float);
}

extern "C" int iff_q_proj(const iff_idnet* n, const float* img, int32_t M, float* q, void* workspace, size_t workspace_bytes,
                          void* stream) {
    IFF_REQUIRE(n && M >= 0, "iff_q_proj: bad argument");
    if (M == 0) return 0;
    IFF_REQUIRE(img && q && workspace, "iff_q_proj: null buffer");
    if (workspace_bytes < iff_q_proj_workspace(n, M)) return fail(IFF_ERR_WORKSPACE, "iff_q_proj: workspace too small");
    IFF_HIP(launch_q_proj(n->dev, img, M, q, workspace, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_attn_logits(const float* q, const float* k, int32_t M, int64_t N, int32_t D, float divisor, float* logits,
                               float* row_max, float* row_sumexp, int32_t gemm_mode, void* stream) {
    IFF_REQUIRE(M >= 0 && N >= 0 && D > 0 && D % 32 == 0, "iff_attn_logits: bad shape M=%d N=%lld D=%d", M, (long long)N, D);
    IFF_REQUIRE(gemm_mode == 0 || gemm_mode == 1, "iff_attn_logits: gemm_mode must be 0 or 1");
    if (M == 0 || N == 0) return 0;
    IFF_REQUIRE(q && k && logits, "iff_attn_logits: null buffer");
    IFF_REQUIRE((row_max == nullptr) == (row_sumexp == nullptr), "iff_attn_logits: pass both row statistics or neither");
    IFF_HIP(launch_attn_logits(q, k, M, N, D, divisor, logits, row_max, row_sumexp, gemm_mode, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_attn_colsum(float* logits_inout, int32_t M, int64_t N, const float* row_max, const float* row_sumexp,
                               int32_t write_attention, float* score, void* stream) {
    IFF_REQUIRE(M >= 1 && M <= 8064 && N >= 0, "iff_attn_colsum: bad shape (1 <= M <= 8064: the row statistics live in 64 KiB of LDS)");
    if (N == 0) return 0;
    IFF_REQUIRE(logits_inout && row_max && row_sumexp && score, "iff_attn_colsum: null buffer");
    IFF_HIP(launch_attn_colsum(logits_inout, 1, M, N, row_max, row_sumexp, write_attention, score, nullptr, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_attn_colsum_batched(float* logits_inout, int32_t Q, int32_t M, int64_t N, const float* row_max,
                                       const float* row_sumexp, int32_t write_attention, float* score, void* stream) {
    IFF_REQUIRE(Q >= 0 && Q <= 65535 && M >= 1 && M <= 8064 && N >= 0, "iff_attn_colsum_batched: bad shape (1 <= M <= 8064)");
    if (N == 0 || Q == 0) return 0;
    IFF_REQUIRE(logits_inout && row_max && row_sumexp && score, "iff_attn_colsum_batched: null buffer");
    IFF_HIP(launch_attn_colsum(logits_inout, Q, M, N, row_max, row_sumexp, write_attention, score, nullptr, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_attn_colsum_rows(float* logits_inout, int32_t Q, int32_t M, int64_t N, const float* row_max, const float* row_sumexp,
                                    const int32_t* rows_per_query, int32_t write_attention, float* score, void* stream) {
    IFF_REQUIRE(Q >= 0 && Q <= 65535 && M >= 1 && M <= 8064 && N >= 0, "iff_attn_colsum_rows: bad shape (1 <= M <= 8064)");
    if (N == 0 || Q == 0) return 0;
    IFF_REQUIRE(logits_inout && row_max && row_sumexp && score && rows_per_query, "iff_attn_colsum_rows: null buffer");
    IFF_HIP(launch_attn_colsum(logits_inout, Q, M, N, row_max, row_sumexp, write_attention, score, rows_per_query, (hipStream_t)stream));
    return 0;
}

extern "C" size_t iff_topk_workspace(int64_t N, int32_t k) { return topk_workspace_bytes(N, k); }

extern "C" int iff_topk(const float* score, int64_t N, int32_t k, int64_t* idx, float* val, void* workspace,
                        size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(score && idx && val, "iff_topk: null buffer");
    IFF_REQUIRE(k >= 1 && k <= 1024, "iff_topk: k = %d outside [1, 1024]", k);
    IFF_REQUIRE(N >= k, "iff_topk: k = %d exceeds N = %lld (torch.topk raises here too)", k, (long long)N);
    IFF_HIP(launch_topk(score, 1, N, k, idx, val, workspace, workspace_bytes, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_topk_batched(const float* score, int32_t Q, int64_t N, int32_t k, int64_t* idx, float* val, void* stream) {
    IFF_REQUIRE(Q >= 0, "iff_topk_batched: bad batch");
    if (Q == 0) return 0;
    IFF_REQUIRE(score && idx && val, "iff_topk_batched: null buffer");
    IFF_REQUIRE(k >= 1 && k <= 1024, "iff_topk_batched: k = %d outside [1, 1024]", k);
    IFF_REQUIRE(N >= k, "iff_topk_batched: k = %d exceeds N = %lld", k, (long long)N);
    IFF_HIP(launch_topk(score, Q, N, k, idx, val, nullptr, 0, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_pose_from_topk(const int64_t* idx, const float* val, int32_t k, const float* rays_o, const float* rays_d,
                                  int64_t N, const float* up_host, float* c2w, float* parts_opt, void* stream) {
    IFF_REQUIRE(idx && val && rays_o && rays_d && up_host && c2w, "iff_pose_from_topk: null buffer");
    IFF_REQUIRE(k >= 1 && k <= 1024, "iff_pose_from_topk: k = %d outside [1, 1024]", k);
    IFF_HIP(launch_pose(idx, val, 1, k, rays_o, rays_d, N, 0, up_host, c2w, parts_opt, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_pose_from_topk_batched(const int64_t* idx, const float* val, int32_t Q, int32_t k, const float* rays_o,
                                          const float* rays_d, int64_t N, int64_t ray_batch_stride, const float* up_host,
                                          float* c2w, float* parts_opt, void* stream) {
    IFF_REQUIRE(Q >= 0 && ray_batch_stride >= 0, "iff_pose_from_topk_batched: bad batch");
    if (Q == 0) return 0;
    IFF_REQUIRE(idx && val && rays_o && rays_d && up_host && c2w, "iff_pose_from_topk_batched: null buffer");
    IFF_REQUIRE(k >= 1 && k <= 1024, "iff_pose_from_topk_batched: k = %d outside [1, 1024]", k);
    IFF_HIP(launch_pose(idx, val, Q, k, rays_o, rays_d, N, ray_batch_stride, up_host, c2w, parts_opt, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_pose_errors(const float* c2w, const float* gt_c2w, const float* parts_opt, int32_t Q, int32_t k, float* summary, void* stream) {
    IFF_REQUIRE(Q >= 0 && k >= 0 && k <= 1024, "iff_pose_errors: bad argument");
    if (Q == 0) return 0;
    IFF_REQUIRE(c2w && gt_c2w && summary, "iff_pose_errors: null buffer");
    IFF_HIP(launch_pose_errors(c2w, gt_c2w, parts_opt, Q, k, summary, (hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ ray-sharded merges
extern "C" int iff_merge_row_stats(const float* stats_all, int32_t G, int64_t R, float* gmax, float* gsum, void* stream) {
    IFF_REQUIRE(G >= 1 && R >= 0, "iff_merge_row_stats: bad argument");
    if (R == 0) return 0;
    IFF_REQUIRE(stats_all && gmax && gsum, "iff_merge_row_stats: null buffer");
    IFF_HIP(launch_merge_row_stats(stats_all, G, R, gmax, gsum, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_pack_candidates(const int64_t* idx, const float* val, const float* rays_o, const float* rays_d, int64_t ray_stride, int32_t Q,
                                   int32_t kl, int32_t k, int64_t first_ray, float* msg, void* stream) {
    IFF_REQUIRE(Q >= 0 && kl >= 0 && k >= 1 && kl <= k && ray_stride >= 0 && first_ray >= 0, "iff_pack_candidates: bad argument");
    if (Q == 0) return 0;
    IFF_REQUIRE(msg && (kl == 0 || (idx && val && rays_o && rays_d)), "iff_pack_candidates: null buffer");
    IFF_HIP(launch_pack_candidates(idx, val, rays_o, rays_d, ray_stride, Q, kl, k, first_ray, msg, (hipStream_t)stream));
    return 0;
}
extern "C" int iff_merge_candidates(const float* cand_all, int32_t G, int32_t Qt, int32_t q0, int32_t Q, int32_t k, float* val, int64_t* idx,
                                    float* rays_o_out, float* rays_d_out, void* stream) {
    IFF_REQUIRE(G >= 1 && Qt >= 0 && q0 >= 0 && Q >= 0 && q0 + Q <= Qt && k >= 1, "iff_merge_candidates: bad argument");
    IFF_REQUIRE((int64_t)G * k * 8 <= 64 * 1024, "iff_merge_candidates: %d lists of %d candidates exceed the 8192 a workgroup merges", G, k);
    if (Q == 0) return 0;
    IFF_REQUIRE(cand_all && val && idx && rays_o_out && rays_d_out, "iff_merge_candidates: null buffer");
    IFF_HIP(launch_merge_candidates(cand_all, G, Qt, q0, Q, k, val, idx, rays_o_out, rays_d_out, (hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ table files
// A handle's slab (tables already in kernel layout) written to / read from one file, so that a serving process skips the
// reference-layout checkpoint and the K0 re-layout (models/tensorBase.py:424-458 is the reference's on-disk format; this is
// its pre-laid-out counterpart).  Layout, little-endian:
//   header (64 B) | device-side descriptor struct with every pointer replaced by (offset into the slab + 1, 0 = null) |
//   slab bytes | extra (field: the occupied-voxel list, int32 each)
// The descriptor struct's size is stored and checked, so a file written by a build with another struct layout is refused
// rather than misread; IFF_TABLE_FILE_VERSION changes whenever a table's layout does.
#define IFF_TABLE_FILE_VERSION 3          // 2: the field slab carries the corner-bit occupancy table (FieldDev::cell); 3: FieldDev::fan_waves
struct TableFileHeader {
    char magic[8];              // "IFFTABLE"
    uint32_t version, kind;     // kind 1 = field, 2 = idnet
    uint32_t abi, struct_bytes;
    uint64_t slab_bytes, extra_count;
    uint64_t reserved[3];
};
static_assert(sizeof(TableFileHeader) == 64, "header is 64 bytes");

template <typename Dev>
static void rebase(Dev& v, const std::vector<const void**>& ptrs, const char* from, const char* to) {
    // from != null: pointer -> offset + 1;  to != null: offset + 1 -> pointer
    for (const void** pp : ptrs) {
        if (from) *pp = *pp ? (const void*)(uintptr_t)((const char*)*pp - from + 1) : nullptr;
        else *pp = *pp ? (const void*)(to + ((uintptr_t)*pp - 1)) : nullptr;
    }
    (void)v;
}
static std::vector<const void**> field_ptrs(FieldDev& v) {
    std::vector<const void**> p;
    for (int i = 0; i < 3; ++i) {
        p.push_back((const void**)&v.dplane[i]); p.push_back((const void**)&v.dline[i]);
        p.push_back((const void**)&v.aplane[i]); p.push_back((const void**)&v.aline[i]);
    }
    p.push_back((const void**)&v.basis_l); p.push_back((const void**)&v.basis_l12); p.push_back((const void**)&v.basis);
    p.push_back((const void**)&v.mask); p.push_back((const void**)&v.cell); p.push_back((const void**)&v.head);
    return p;
}
static std::vector<const void**> idnet_ptrs(IdNetDev& v) {
    std::vector<const void**> p;
    const void** all[] = {(const void**)&v.w1, (const void**)&v.b1, (const void**)&v.w2, (const void**)&v.b2, (const void**)&v.w3,
                          (const void**)&v.b3, (const void**)&v.w4, (const void**)&v.b4, (const void**)&v.wk, (const void**)&v.bk,
                          (const void**)&v.wq, (const void**)&v.bq, &v.p1, &v.p2, &v.p3, &v.p4, &v.pk, (const void**)&v.wqf,
                          (const void**)&v.bqf, &v.f1, &v.f2, &v.f3, &v.h1, &v.h2, &v.h3h, &v.h3x};
    for (auto q : all) p.push_back(q);
    return p;
}

static int write_table_file(const char* path, uint32_t kind, const void* dev_struct, size_t struct_bytes, const void* slab,
                            size_t slab_bytes, const int* extra_dev, size_t extra_count, hipStream_t s) {
    std::vector<char> host(slab_bytes);
    std::vector<int> extra(extra_count);
    IFF_HIP(hipStreamSynchronize(s));
    IFF_HIP(hipMemcpy(host.data(), slab, slab_bytes, hipMemcpyDeviceToHost));
    if (extra_count) IFF_HIP(hipMemcpy(extra.data(), extra_dev, extra_count * sizeof(int), hipMemcpyDeviceToHost));
    FILE* fh = fopen(path, "wb");
    if (!fh) return fail(IFF_ERR_INVALID_ARGUMENT, "cannot open %s for writing", path);
    TableFileHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "IFFTABLE", 8);
    h.version = IFF_TABLE_FILE_VERSION; h.kind = kind; h.abi = IFF_ABI_VERSION; h.struct_bytes = (uint32_t)struct_bytes;
    h.slab_bytes = slab_bytes; h.extra_count = extra_count;
    bool ok = fwrite(&h, sizeof(h), 1, fh) == 1 && fwrite(dev_struct, struct_bytes, 1, fh) == 1 &&
              (slab_bytes == 0 || fwrite(host.data(), slab_bytes, 1, fh) == 1) &&
              (extra_count == 0 || fwrite(extra.data(), extra_count * sizeof(int), 1, fh) == 1);
    ok = (fclose(fh) == 0) && ok;
    return ok ? 0 : fail(IFF_ERR_INVALID_ARGUMENT, "short write to %s", path);
}

static int read_table_file(const char* path, uint32_t kind, void* dev_struct, size_t struct_bytes, std::vector<char>& slab,
                           std::vector<int>& extra) {
    FILE* fh = fopen(path, "rb");
    if (!fh) return fail(IFF_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    TableFileHeader h;
    int rc = 0;
    struct stat st;
    if (fread(&h, sizeof(h), 1, fh) != 1 || memcmp(h.magic, "IFFTABLE", 8) != 0) rc = fail(IFF_ERR_INVALID_ARGUMENT, "%s is not an IFFTABLE file", path);
    else if (h.version != IFF_TABLE_FILE_VERSION || h.kind != kind || h.struct_bytes != struct_bytes)
        rc = fail(IFF_ERR_UNSUPPORTED, "%s: table file version %u kind %u descriptor %u B; this build reads version %u kind %u descriptor %zu B",
                  path, h.version, h.kind, h.struct_bytes, IFF_TABLE_FILE_VERSION, kind, struct_bytes);
    else if (h.abi != IFF_ABI_VERSION)
        rc = fail(IFF_ERR_UNSUPPORTED, "%s was written by ABI version %u of the library; this build is ABI version %d (re-save the tables)",
                  path, h.abi, IFF_ABI_VERSION);
    else if (h.slab_bytes > ((uint64_t)1 << 40) || h.extra_count > ((uint64_t)1 << 32)) rc = fail(IFF_ERR_INVALID_ARGUMENT, "%s: implausible sizes", path);
    else if (fstat(fileno(fh), &st) != 0 || (uint64_t)st.st_size != 64 + (uint64_t)struct_bytes + h.slab_bytes + 4 * h.extra_count)
        rc = fail(IFF_ERR_INVALID_ARGUMENT, "%s is truncated or carries trailing bytes: its header promises %llu bytes", path,
                  (unsigned long long)(64 + (uint64_t)struct_bytes + h.slab_bytes + 4 * h.extra_count));
    else {
        try {
            slab.resize(h.slab_bytes);
            extra.resize(h.extra_count);
        } catch (const std::exception&) {
            fclose(fh);
            return fail(IFF_ERR_INVALID_ARGUMENT, "%s: cannot allocate %llu bytes of host memory for its tables", path,
                        (unsigned long long)(h.slab_bytes + 4 * h.extra_count));
        }
        bool ok = fread(dev_struct, struct_bytes, 1, fh) == 1 && (h.slab_bytes == 0 || fread(slab.data(), h.slab_bytes, 1, fh) == 1) &&
                  (h.extra_count == 0 || fread(extra.data(), h.extra_count * sizeof(int), 1, fh) == 1);
        if (!ok) rc = fail(IFF_ERR_INVALID_ARGUMENT, "%s is truncated", path);
    }
    fclose(fh);
    return rc;
}

// a stored pointer is (offset into the slab + 1), 0 = null: does it name exactly the table `want_off` (or null when !present)?
static bool stored_is(const void* stored, bool present, size_t want_off) {
    return present ? (uintptr_t)stored == want_off + 1 : stored == nullptr;
}

// A field descriptor read from a file, pointers still in stored form: every dimension in the range iff_field_create accepts,
// every table exactly where iff_field_create would have put it, the slab exactly that long, the occupied-voxel list inside the
// mask -- so no kernel can be sent outside the slab by a corrupted or foreign file.
static int validate_field_file(const char* path, const FieldDev& v, size_t slab_bytes, const std::vector<int>& occ) {
    const bool has_mask = v.mask != nullptr;
    if (int rc = check_field_dims(v.grid, v.n_density, v.n_app, v.app_dim, v.feature_c, v.density_lanes, v.mask_dims, has_mask)) return rc;
    const FieldLayout L = field_layout(v.grid, v.n_density, v.n_app, v.app_dim, v.feature_c, v.mask_dims, has_mask);
    bool ok = slab_bytes == L.total;
    for (int i = 0; i < 3; ++i)
        ok = ok && stored_is(v.dplane[i], true, L.dp[i]) && stored_is(v.dline[i], true, L.dl[i]) && stored_is(v.aplane[i], true, L.ap[i]) &&
             stored_is(v.aline[i], true, L.al[i]);
    ok = ok && stored_is(v.basis, true, L.basis) && stored_is(v.basis_l, true, L.basis_l) && stored_is(v.basis_l12, true, L.basis_l12) &&
         stored_is(v.head, true, L.head) && stored_is(v.mask, has_mask, L.mask) && stored_is(v.cell, has_mask, L.cell);
    if (!ok) return fail(IFF_ERR_INVALID_ARGUMENT, "%s: the table offsets / slab size do not match the dimensions the file claims", path);
    IFF_REQUIRE(v.n_samples >= 1 && (v.softplus == 0 || v.softplus == 1) && (v.unisphere == 0 || v.unisphere == 1) && v.step_size > 0.0f &&
                    v.step_size < 1e30f && (v.head_lanes == 0 || v.head_lanes == 16) && (v.sampler_persistent == 0 || v.sampler_persistent == 1) && (v.fan_waves == 0 || v.fan_waves == 4 || v.fan_waves == 8),
                "%s: implausible march parameters", path);
    IFF_REQUIRE(occ.size() <= L.n_mask, "%s: %zu occupied voxels listed for a mask of %zu", path, occ.size(), L.n_mask);
    for (int i : occ) IFF_REQUIRE(i >= 0 && (size_t)i < L.n_mask, "%s: occupied-voxel index %d outside the mask", path, i);
    return 0;
}

static int validate_idnet_file(const char* path, const IdNetDev& v, size_t slab_bytes) {
    if (int rc = check_idnet_dims(v.feature_c, v.fea, v.img_fea)) return rc;
    const bool fused = v.feature_c == 256, f16 = v.h1 != nullptr;
    IFF_REQUIRE(!f16 || fused, "%s: fp16 planes without a 256-wide encoder", path);
    const IdLayout L = idnet_layout(v.feature_c, v.fea, v.img_fea, fused, f16);
    const bool ok = slab_bytes == L.total && stored_is(v.w1, true, L.w1) && stored_is(v.b1, true, L.b1) && stored_is(v.w2, true, L.w2) &&
                    stored_is(v.b2, true, L.b2) && stored_is(v.w3, true, L.w3) && stored_is(v.b3, true, L.b3) && stored_is(v.w4, true, L.w4) &&
                    stored_is(v.b4, true, L.b4) && stored_is(v.wk, true, L.wk) && stored_is(v.bk, true, L.bk) && stored_is(v.wq, true, L.wq) &&
                    stored_is(v.bq, true, L.bq) && stored_is(v.p1, true, L.p1) && stored_is(v.p2, true, L.p2) && stored_is(v.p3, true, L.p3) &&
                    stored_is(v.p4, true, L.p4) && stored_is(v.pk, true, L.pk) && stored_is(v.wqf, true, L.wqf) && stored_is(v.bqf, true, L.bqf) &&
                    stored_is(v.f1, fused, L.f1) && stored_is(v.f2, fused, L.f2) && stored_is(v.f3, fused, L.f3) && stored_is(v.h1, f16, L.h1) &&
                    stored_is(v.h2, f16, L.h2) && stored_is(v.h3h, f16, L.h3h) && stored_is(v.h3x, f16, L.h3x);
    if (!ok) return fail(IFF_ERR_INVALID_ARGUMENT, "%s: the table offsets / slab size do not match the widths the file claims", path);
    auto small = [](int e) { return e >= -64 && e <= 64; };
    IFF_REQUIRE(v.qf_ld == v.feature_c + 16 && (v.gemm_mode == 0 || v.gemm_mode == 1) && (v.trunk_f16 == 0 || v.trunk_f16 == 1 || v.trunk_f16 == 2) &&
                    (v.trunk_f16 == 0 || f16) && (v.fused_trunk == 0 || (v.fused_trunk == 1 && fused)) && v.trunk_variant >= 0 &&
                    v.trunk_variant <= 3 && small(v.e_x) && small(v.e_h1) && small(v.e_h2) && small(v.e_h3) && small(v.e_w1) &&
                    small(v.e_w2) && small(v.e_w3h) && small(v.e_w3x),
                "%s: implausible identification-net parameters", path);
    return 0;
}

extern "C" int iff_field_save(const iff_field* f, const char* path, void* stream) {
    IFF_REQUIRE(f && path, "iff_field_save: null argument");
    FieldDev v = f->dev;
    rebase(v, field_ptrs(v), (const char*)f->slab, nullptr);
    return write_table_file(path, 1, &v, sizeof(v), f->slab, f->slab_bytes, f->occ_list, (size_t)f->n_occ, (hipStream_t)stream);
}

extern "C" int iff_field_load(const char* path, void* stream, iff_field** out) {
    IFF_REQUIRE(path && out, "iff_field_load: null argument");
    *out = nullptr;
    hipStream_t s = (hipStream_t)stream;
    FieldDev v;
    std::vector<char> slab;
    std::vector<int> occ;
    int rc = read_table_file(path, 1, &v, sizeof(v), slab, occ);
    if (rc) return rc;
    if ((rc = validate_field_file(path, v, slab.size(), occ)) != 0) return rc;
    iff_field* f = new iff_field();
    f->slab_bytes = slab.size();
    f->n_occ = (int)occ.size();
    IFF_CREATE_HIP(hipMalloc(&f->slab, slab.size()));
    IFF_CREATE_HIP(hipMemcpyAsync(f->slab, slab.data(), slab.size(), hipMemcpyHostToDevice, s));
    if (f->n_occ > 0) {
        IFF_CREATE_HIP(hipMalloc((void**)&f->occ_list, occ.size() * sizeof(int)));
        IFF_CREATE_HIP(hipMemcpyAsync(f->occ_list, occ.data(), occ.size() * sizeof(int), hipMemcpyHostToDevice, s));
    }
    IFF_CREATE_HIP(hipStreamSynchronize(s));
    rebase(v, field_ptrs(v), nullptr, (const char*)f->slab);
    f->dev = v;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) f->n_cus = prop.multiProcessorCount;
    *out = f;
    return 0;
}

extern "C" int iff_idnet_save(const iff_idnet* n, const char* path, void* stream) {
    IFF_REQUIRE(n && path, "iff_idnet_save: null argument");
    IdNetDev v = n->dev;
    rebase(v, idnet_ptrs(v), (const char*)n->slab, nullptr);
    return write_table_file(path, 2, &v, sizeof(v), n->slab, n->slab_bytes, nullptr, 0, (hipStream_t)stream);
}

extern "C" int iff_idnet_load(const char* path, void* stream, iff_idnet** out) {
    IFF_REQUIRE(path && out, "iff_idnet_load: null argument");
    *out = nullptr;
    hipStream_t s = (hipStream_t)stream;
    IdNetDev v;
    std::vector<char> slab;
    std::vector<int> none;
    int rc = read_table_file(path, 2, &v, sizeof(v), slab, none);
    if (rc) return rc;
    if ((rc = validate_idnet_file(path, v, slab.size())) != 0) return rc;
    iff_idnet* n = new iff_idnet();
    n->slab_bytes = slab.size();
    IFF_NET_HIP(hipMalloc(&n->slab, slab.size()));
    IFF_NET_HIP(hipMemcpyAsync(n->slab, slab.data(), slab.size(), hipMemcpyHostToDevice, s));
    IFF_NET_HIP(hipStreamSynchronize(s));
    rebase(v, idnet_ptrs(v), nullptr, (const char*)n->slab);
    n->dev = v;
    *out = n;
    return 0;
}


// ------------------------------------------------------------------------------------------------ image backbone (ViT-S/14)
struct iff_vit {
    VitDev dev;
    void* slab = nullptr;
    size_t slab_bytes = 0;
};

extern "C" void iff_vit_destroy(iff_vit* v) {
    if (!v) return;
    if (v->slab) (void)hipFree(v->slab);
    delete v;
}

extern "C" int iff_vit_create(const iff_vit_desc* d, void* stream, iff_vit** out) {
    IFF_REQUIRE(d && out, "iff_vit_create: null argument");
    *out = nullptr;
    hipStream_t s = (hipStream_t)stream;
    IFF_REQUIRE(d->dim == 384 && d->heads == 6, "iff_vit_create: built for ViT-S (dim 384, 6 heads of 64); got dim %d, heads %d", d->dim, d->heads);
    IFF_REQUIRE(d->depth >= 1 && d->depth <= 64 && d->mlp >= 128 && d->mlp % 128 == 0 && d->mlp <= 8192, "iff_vit_create: depth %d / mlp %d unsupported", d->depth, d->mlp);
    IFF_REQUIRE(d->patch >= 1 && d->patch <= 32 && d->grid_h >= 1 && d->grid_w >= 1 && 1 + d->grid_h * d->grid_w <= 288,
                "iff_vit_create: patch %d grid %dx%d unsupported (at most 288 tokens)", d->patch, d->grid_h, d->grid_w);
    IFF_REQUIRE(d->patch_w && d->patch_b && d->cls && d->pos && d->ln1_w && d->ln1_b && d->qkv_w && d->qkv_b && d->proj_w && d->proj_b &&
                    d->ls1 && d->ln2_w && d->ln2_b && d->fc1_w && d->fc1_b && d->fc2_w && d->fc2_b && d->ls2 && d->norm_w && d->norm_b,
                "iff_vit_create: null weight");
    const size_t D = d->dim, L = d->depth, F = d->mlp, T = 1 + (size_t)d->grid_h * d->grid_w;
    const size_t kraw = 3 * (size_t)d->patch * d->patch, kp = (kraw + 63) / 64 * 64;
    IFF_REQUIRE(d->precision == IFF_VIT_FP32 || d->precision == IFF_VIT_BF16, "iff_vit_create: precision %d is neither IFF_VIT_FP32 nor IFF_VIT_BF16", d->precision);
    IFF_REQUIRE(d->depth <= VIT_MAX_DEPTH, "iff_vit_create: depth %d exceeds %d", d->depth, VIT_MAX_DEPTH);
    iff_vit* v = new iff_vit();                     // every descriptor check is above this line: nothing below returns without freeing v
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
    IFF_REQUIRE(d->gemm_form >= 0 && d->gemm_form <= 10, "iff_vit_create: gemm_form %d is not one of 0 .. 10", d->gemm_form);
    const bool split = d->precision == IFF_VIT_FP32;
    const size_t eb = split ? 4 : 2;               // bytes per weight: bf16, or fp16 hi + lo planes
    const size_t o_pw = take(D * kp * eb), o_qkv = take(L * 3 * D * D * eb), o_proj = take(L * D * D * eb), o_fc1 = take(L * F * D * eb),
                 o_fc2 = take(L * D * F * eb);
    struct { const float* src; size_t n; size_t off; } vecs[] = {
        {d->patch_b, D, 0}, {d->cls, D, 0}, {d->pos, T * D, 0}, {d->ln1_w, L * D, 0}, {d->ln1_b, L * D, 0}, {d->ln2_w, L * D, 0},
        {d->ln2_b, L * D, 0}, {d->qkv_b, L * 3 * D, 0}, {d->proj_b, L * D, 0}, {d->fc1_b, L * F, 0}, {d->fc2_b, L * D, 0}, {d->ls1, L * D, 0},
        {d->ls2, L * D, 0}, {d->norm_w, D, 0}, {d->norm_b, D, 0}};
    for (auto& x : vecs) x.off = take(x.n * 4);
    v->slab_bytes = off;
    hipError_t e = hipMalloc(&v->slab, off);
    if (e != hipSuccess) { delete v; return hip_fail(e, "hipMalloc(ViT weights)"); }
    char* b = (char*)v->slab;
#define IFF_VIT_HIP(call)                                              \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) { iff_vit_destroy(v); return hip_fail(e__, #call); } \
    } while (0)
    VitDev& w = v->dev;
    memset(&w, 0, sizeof(w));
    w.prec = split ? 1 : 0;
    w.gemm_form = d->gemm_form;
    w.s_patch = 1.0f;
    for (int l = 0; l < VIT_MAX_DEPTH; ++l) w.s_qkv[l] = w.s_proj[l] = w.s_fc1[l] = w.s_fc2[l] = 1.0f;
    if (!split) {
        IFF_VIT_HIP(launch_vit_pad_rows(d->patch_w, (int)D, (int)kraw, (int)kp, b + o_pw, s));
        IFF_VIT_HIP(launch_vit_to_bf16(d->qkv_w, (int64_t)(L * 3 * D * D), b + o_qkv, s));
        IFF_VIT_HIP(launch_vit_to_bf16(d->proj_w, (int64_t)(L * D * D), b + o_proj, s));
        IFF_VIT_HIP(launch_vit_to_bf16(d->fc1_w, (int64_t)(L * F * D), b + o_fc1, s));
        IFF_VIT_HIP(launch_vit_to_bf16(d->fc2_w, (int64_t)(L * D * F), b + o_fc2, s));
    } else {
        // fp16 hi / lo planes of every matrix times a power of two: the largest |weight| of the matrix lands in [2^13, 2^14), so
        // the hi piece uses fp16's normal range and the lo piece (2^-11 of it) stays normal down to weights 2^-16 of the largest;
        // the GEMM epilogue multiplies the accumulator by the exact inverse (VitDev::s_*)
        std::vector<float> host;
        auto convert = [&](const float* src, size_t rows, size_t cols, size_t KP, size_t layers, char* dst, float* scales) -> hipError_t {
            try { host.resize(layers * rows * cols); } catch (const std::exception&) { return hipErrorOutOfMemory; }
            hipError_t e2 = hipMemcpyAsync(host.data(), src, host.size() * 4, hipMemcpyDeviceToHost, s);
            if (e2 != hipSuccess) return e2;
            if ((e2 = hipStreamSynchronize(s)) != hipSuccess) return e2;
            _Float16* hi = (_Float16*)dst;
            _Float16* lo = hi + layers * rows * KP;
            for (size_t l = 0; l < layers; ++l) {
                float mx = 0.0f;
                for (size_t i = 0; i < rows * cols; ++i) { const float a = fabsf(host[l * rows * cols + i]); if (a > mx) mx = a; }
                if (!(mx < INFINITY)) return hipErrorInvalidValue;
                int ex = 0;
                if (mx > 0.0f) { (void)frexpf(mx, &ex); ex = 14 - ex; }              // mx * 2^ex in [2^13, 2^14)
                ex = ex > 40 ? 40 : (ex < -40 ? -40 : ex);
                scales[l] = ldexpf(1.0f, -ex);
                if ((e2 = launch_vit_to_f16_planes(src + l * rows * cols, (int64_t)rows, (int)cols, (int)KP, ldexpf(1.0f, ex), hi + l * rows * KP,
                                                   lo + l * rows * KP, s)) != hipSuccess) return e2;
            }
            return hipSuccess;
        };
        IFF_VIT_HIP(convert(d->patch_w, D, kraw, kp, 1, b + o_pw, &w.s_patch));
        IFF_VIT_HIP(convert(d->qkv_w, 3 * D, D, D, L, b + o_qkv, w.s_qkv));
        IFF_VIT_HIP(convert(d->proj_w, D, D, D, L, b + o_proj, w.s_proj));
        IFF_VIT_HIP(convert(d->fc1_w, F, D, D, L, b + o_fc1, w.s_fc1));
        IFF_VIT_HIP(convert(d->fc2_w, D, F, F, L, b + o_fc2, w.s_fc2));
    }
    for (auto& x : vecs) IFF_VIT_HIP(hipMemcpyAsync(b + x.off, x.src, x.n * 4, hipMemcpyDeviceToDevice, s));
    w.patch_w = b + o_pw; w.qkv_w = b + o_qkv; w.proj_w = b + o_proj; w.fc1_w = b + o_fc1; w.fc2_w = b + o_fc2;
    const float** dst[] = {&w.patch_b, &w.cls, &w.pos, &w.ln1_w, &w.ln1_b, &w.ln2_w, &w.ln2_b, &w.qkv_b, &w.proj_b, &w.fc1_b, &w.fc2_b,
                           &w.ls1, &w.ls2, &w.norm_w, &w.norm_b};
    for (size_t i = 0; i < sizeof(dst) / sizeof(dst[0]); ++i) *dst[i] = (const float*)(b + vecs[i].off);
    w.dim = d->dim; w.depth = d->depth; w.heads = d->heads; w.mlp = d->mlp; w.patch = d->patch; w.gh = d->grid_h; w.gw = d->grid_w;
    w.T = (int)T; w.kp = (int)kp; w.eps = d->ln_eps;
    IFF_VIT_HIP(hipStreamSynchronize(s));      // the source tensors may be released by the caller after return
    *out = v;
    return 0;
}

extern "C" size_t iff_vit_workspace(const iff_vit* v, int32_t Q) { return (v && Q > 0) ? vit_workspace_bytes(v->dev, Q) : 0; }

extern "C" int iff_vit_forward(const iff_vit* v, const float* images, int32_t Q, float* patch_tokens, float* cls_opt, void* workspace,
                               size_t workspace_bytes, void* stream) {
    IFF_REQUIRE(v && Q >= 0, "iff_vit_forward: bad argument");
    if (Q == 0) return 0;
    IFF_REQUIRE(images && patch_tokens, "iff_vit_forward: null buffer");
    if (!workspace || workspace_bytes < vit_workspace_bytes(v->dev, Q))
        return fail(IFF_ERR_WORKSPACE, "iff_vit_forward: workspace %zu < %zu bytes", workspace_bytes, vit_workspace_bytes(v->dev, Q));
    IFF_HIP(launch_vit_forward(v->dev, images, Q, v->dev.gh * v->dev.patch, v->dev.gw * v->dev.patch, patch_tokens, cls_opt, workspace,
                               workspace_bytes, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_image_resize_crop(const float* src, int32_t Q, int32_t H, int32_t W, int32_t C, int32_t rh, int32_t rw, int32_t top,
                                     int32_t left, int32_t ch, int32_t cw, int32_t cubic, const float* mean, const float* std, float* dst,
                                     void* stream) {
    IFF_REQUIRE(Q >= 0 && H >= 1 && W >= 1 && C >= 1 && C <= 4 && rh >= 1 && rw >= 1, "iff_image_resize_crop: bad shape");
    IFF_REQUIRE(top >= 0 && left >= 0 && ch >= 1 && cw >= 1 && top + ch <= rh && left + cw <= rw, "iff_image_resize_crop: crop window outside the resized image");
    if (Q == 0) return 0;
    IFF_REQUIRE(src && dst, "iff_image_resize_crop: null buffer");
    const float sup = cubic ? 2.0f : 1.0f;
    const float sy = std::max((float)H / rh, 1.0f), sx = std::max((float)W / rw, 1.0f);
    if (2.0f * sup * std::max(sx, sy) + 2.0f > 32.0f)
        return fail(IFF_ERR_UNSUPPORTED, "iff_image_resize_crop: scale factor %.2f needs more than 32 filter taps", std::max(sx, sy));
    IFF_HIP(launch_resize_crop(src, Q, H, W, C, 0, rh, rw, top, left, ch, cw, cubic ? 1 : 0, mean, std, dst, (hipStream_t)stream));
    return 0;
}

extern "C" int iff_image_resize_crop_rgba(const float* src, int32_t Q, int32_t H, int32_t W, int32_t mode, int32_t rh, int32_t rw, int32_t top,
                                          int32_t left, int32_t ch, int32_t cw, int32_t cubic, const float* mean, const float* std, float* dst,
                                          void* stream) {
    IFF_REQUIRE(mode == IFF_RESIZE_RGB_ON_WHITE || mode == IFF_RESIZE_ALPHA, "iff_image_resize_crop_rgba: mode %d is neither IFF_RESIZE_RGB_ON_WHITE nor IFF_RESIZE_ALPHA", mode);
    IFF_REQUIRE(Q >= 0 && H >= 1 && W >= 1 && rh >= 1 && rw >= 1, "iff_image_resize_crop_rgba: bad shape");
    IFF_REQUIRE(top >= 0 && left >= 0 && ch >= 1 && cw >= 1 && top + ch <= rh && left + cw <= rw, "iff_image_resize_crop_rgba: crop window outside the resized image");
    if (Q == 0) return 0;
    IFF_REQUIRE(src && dst, "iff_image_resize_crop_rgba: null buffer");
    const float sup = cubic ? 2.0f : 1.0f;
    const float sy = std::max((float)H / rh, 1.0f), sx = std::max((float)W / rw, 1.0f);
    if (2.0f * sup * std::max(sx, sy) + 2.0f > 32.0f)
        return fail(IFF_ERR_UNSUPPORTED, "iff_image_resize_crop_rgba: scale factor %.2f needs more than 32 filter taps", std::max(sx, sy));
    IFF_HIP(launch_resize_crop(src, Q, H, W, mode == IFF_RESIZE_RGB_ON_WHITE ? 3 : 1, mode, rh, rw, top, left, ch, cw, cubic ? 1 : 0, mean, std, dst,
                               (hipStream_t)stream));
    return 0;
}
