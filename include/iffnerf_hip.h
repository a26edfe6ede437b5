/*
 * iffnerf_hip.h -- C ABI of libiffnerf_hip.so: the MI355X (gfx950) implementation of IFFNeRF's
 * per-query inference hot path.
 *
 * This is the drop-in boundary.  The reference is pure Python/PyTorch and has no
 * FFI of its own; each entry point below replaces the aten-op chain of the reference function it cites
 * (paths relative to the upstream repository), and is what the Python classes that keep the reference's
 * module paths and signatures (iffnerf_amd/models, iffnerf_amd/pose_estimation) bind through ctypes.
 * INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers (HBM) unless a parameter says "host".  fp32, C-contiguous.
 *   - The caller allocates and owns every input, output and workspace buffer.  Outputs are fully
 *     overwritten.  Functions never allocate except inside *_create, never free except in *_destroy.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on it.
 *   - Return value: 0 on success, non-zero on error (a hipError_t, or IFF_ERR_* below); the message is
 *     available from iff_last_error() (thread-local).  Numerical degeneracies are values (NaN / identity
 *     pose), never errors, exactly as in the reference.
 *   - No torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef IFFNERF_HIP_H
#define IFFNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IFF_ABI_VERSION 12

#define IFF_ERR_INVALID_ARGUMENT 1001
#define IFF_ERR_UNSUPPORTED      1002
#define IFF_ERR_WORKSPACE        1003

#define IFF_MARCH_POINT_CENTRED 0   /* sampler = TensorBase.sample_point_color (models/tensorBase.py:623-638) */
#define IFF_MARCH_SLAB          1   /* sampler = TensorBase.sample_ray, is_train=False (models/tensorBase.py:494-536) */

/* matrix-product arithmetic of the ray encoder / attention logits (both keep fp32 accuracy):
 *   F32     v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fp32 fmaf chain
 *   BF16X3  every fp32 operand split exactly into 3 bf16 pieces, 6 bf16 MFMAs per product block, fp32 accumulation;
 *           dropped terms < 2^-25 relative -- the error class of an fp32 re-association, at ~2.7x the throughput
 *   BF16X3_LAYERED  the same arithmetic with one launch per encoder layer (BF16X3 runs the three ReLU layers of a
 *           256-wide encoder as one launch with the activations kept in LDS; other widths always run layered)
 *   F16X2   every fp32 operand scaled by a power of two and split into two fp16 pieces (22 significant bits), 3 fp16 MFMAs
 *           per product block: the fused encoder + logits launch of a 256-wide encoder at half the matrix-core work of
 *           BF16X3 and the same accuracy class; power-of-two scales are planned at iff_idnet_create from worst-case
 *           activation bounds (ray origins must satisfy |o| <= 64 scene units), and a network whose bounds do not fit
 *           fp16's range keeps BF16X3 -- iff_idnet_gemm_mode() reports which arithmetic the handle runs.  Every other
 *           entry point (per-layer calls, unfolded projections) runs BF16X3 in this mode. */
#define IFF_GEMM_F32    0
#define IFF_GEMM_BF16X3 1
#define IFF_GEMM_BF16X3_LAYERED 2
#define IFF_GEMM_F16X2  3
#define IFF_GEMM_F16X1  4   /* ONE fp16 product per block (11 significant bits per operand): a throughput class, NOT the reference's accuracy
                             * class (logits to ~1e-2 instead of 1e-4) -- opt-in, never a default; measured next to the default by bench.py */

#define IFF_ISOCELL_DIRS 27         /* pose_estimation/sampling.py:229-234, isocell.py:6-68 (27 targets, N0=3) */
#define IFF_RAY_FEATURES 384        /* DINOv2 ViT-S/14 width: pose_estimation/backbone.py:12-14 */
#define IFF_RAY_INPUT    141        /* pose_estimation/ray_preprocessor.py:8 (3*3 + 2*3*(8+8+6)) */

const char* iff_last_error(void);
int iff_abi_version(void);

/* ------------------------------------------------------------------------------------------------ field
 * A tensorial radiance field (TensorVMSplit + AlphaGridMask + Ref head) re-laid-out for the kernels.
 * Replaces: the parameter tensors of models/tensoRF.py:155-170, models/tensorBase.py:50-64,
 * models/ref.py:48-101 as consumed by the lookups below.  Source tensors are read once by
 * iff_field_create (device pointers, reference layouts) and are not retained.
 */
typedef struct iff_field iff_field;

typedef struct iff_field_desc {
    int32_t grid[3];                 /* gridSize (x, y, z): kwargs["gridSize"], tensorBase.py:405 */
    float   aabb[6];                 /* aabb[0] xyz, aabb[1] xyz */
    int32_t n_density;               /* channels per density plane/line (16) */
    int32_t n_app;                   /* channels per appearance plane/line (48) */
    int32_t app_dim;                 /* 27 */
    int32_t feature_c;               /* Ref bottleneck width (128) */
    const float* density_plane[3];   /* [n_density, G_b, G_a], (a,b) = matMode[i]  (tensoRF.py:166)  */
    const float* density_line[3];    /* [n_density, G_v],      v = vecMode[i]      (tensoRF.py:168)  */
    const float* app_plane[3];       /* [n_app, G_b, G_a] */
    const float* app_line[3];        /* [n_app, G_v] */
    const float* basis;              /* basis_mat.weight [app_dim, 3*n_app] (tensoRF.py:158) */
    const float* mask_volume;        /* AlphaGridMask.alpha_volume [D,H,W] in {0,1}, or NULL (tensorBase.py:60) */
    int32_t mask_dims[3];            /* D, H, W */
    float   mask_aabb[6];
    float   density_shift;           /* tensorBase.py:296 */
    float   distance_scale;          /* tensorBase.py:298 */
    float   weight_thres;            /* rayMarch_weight_thres, tensorBase.py:299 */
    float   step_size;               /* self.stepSize as computed by the host mirror (tensorBase.py:366) */
    int32_t n_samples;               /* self.nSamples (tensorBase.py:368) */
    float   near_far[2];
    int32_t softplus;                /* 1: fea2denseAct == "softplus", 0: relu (tensorBase.py:750-754) */
    int32_t unisphere;               /* 1: contraction_type == "unisphere" (tensorBase.py:390-396) */
    int32_t density_lanes;           /* lanes that share one density lookup (compute_densityfeature, tensoRF.py:216-235) in the
                                        march and the surface sampler: 0 = choose (one lane per point when a density texel
                                        is one 64-B line, and for sampler batches < 8 four), 1 or 4 = force that form.  Both
                                        forms produce the same bits; the knob exists for A/B tests and tuning.  A non-zero value
                                        also keeps the point-centred march on the general kernels (K4a / K4b / head launch). */
    int32_t head_lanes;              /* lanes that share one ray in the Ref head launches (Ref.forward, models/ref.py:103-152):
                                        0 = choose (8, the bottleneck on the fp32 matrix cores, for the reference's head shape),
                                        16 = the 16-lane vector form every head shape can take.  Same bits; for parity tests. */
    int32_t sampler_persistent;      /* 0: the surface sampler (pose_estimation/sampling.py:509-532) as a chain of short launches;
                                        1: ONE persistent launch with in-kernel grid barriers (the same samples bit for bit; then
                                        iff_surface_sample_residency bounds the launches that may be in flight).  For parity tests. */
    int32_t fan_waves;               /* which fused fan kernel serves the point-centred march (TensorBase.forward with sample_point_color,
                                        tensorBase.py:775-917, :623-638) where one applies: 0 = choose (boxes of <= 12 texels: the form
                                        measured faster on MI355X; boxes of <= 22 texels -- unisphere scenes, tensorBase.py:361-365 --:
                                        the eight-wave kernel, the only one that stages them), 4 = the four-wave kernel with
                                        register-staged patches (12-texel boxes only), 8 = the eight-wave kernel with DMA-staged
                                        patches.  alpha / acc / depth / sample counters are the same bits in every form; for A/B and
                                        parity tests. */
    /* Ref head, models/ref.py:69-101 (nn.Linear layouts [out,in]) */
    const float* normal_w;  const float* normal_b;    /* [3,app_dim],[3] */
    const float* tint_w;    const float* tint_b;      /* [3,app_dim],[3] */
    const float* rough_w;   const float* rough_b;     /* [1,app_dim],[1] */
    const float* diffuse_w; const float* diffuse_b;   /* [3,app_dim],[3] */
    const float* bottleneck_w; const float* bottleneck_b; /* [feature_c,app_dim],[feature_c] */
    const float* specular_w;   const float* specular_b;   /* [3,feature_c+39],[3] */
    const float* ide_mat;            /* dir_enc_fn.mat [9,19] (models/ref_utils.py:72-80) */
} iff_field_desc;

int  iff_field_create(const iff_field_desc* desc, void* stream, iff_field** out);
void iff_field_destroy(iff_field* f);
/* bytes of HBM the handle's tables occupy (for DESIGN.md / roofline bookkeeping) */
size_t iff_field_table_bytes(const iff_field* f);
/* Pre-laid-out table file: the handle's tables exactly as the kernels read them (channels-last planes / lines, byte
 * occupancy, packed Ref head, occupied-voxel list) behind a versioned header, so a serving process builds its handle with
 * one read and one copy instead of TensorBase.load + the re-layout of iff_field_create.  Counterpart of the reference's
 * checkpoint file, models/tensorBase.py:424-458 (save / load).  `path` is a host string.  A handle loaded from a file is
 * bit-for-bit the handle it was saved from; files of another IFF_TABLE_FILE_VERSION / descriptor layout are refused. */
int iff_field_save(const iff_field* f, const char* path, void* stream);
int iff_field_load(const char* path, void* stream, iff_field** out);

/* TensorBase.normalize_coord, models/tensorBase.py:389-397.  xyz,out: [n,3] */
int iff_normalize_coord(const iff_field* f, const float* xyz, int64_t n, float* out, void* stream);
/* AlphaGridMask.sample_alpha, models/tensorBase.py:66-72.  xyz [n,3] world -> value [n] */
int iff_mask_sample(const iff_field* f, const float* xyz, int64_t n, float* out, void* stream);
/* `AlphaGridMask.sample_alpha(xyz) > 0` -- the only use the path makes of the occupancy lookup (compute_alpha
 * models/tensorBase.py:762-764, forward :830-833; the surface sampler reaches it through compute_alpha) -- from the corner-bit
 * table of the handle: one byte per point, exactly the truth value of the trilinear sum.  xyz [n,3] world -> flag [n] uint8 */
int iff_mask_occupied(const iff_field* f, const float* xyz, int64_t n, uint8_t* out, void* stream);
/* TensorVMSplit.compute_densityfeature, models/tensoRF.py:216-235.  xn [n,3] normalised -> [n] */
int iff_density_feature(const iff_field* f, const float* xn, int64_t n, float* out, void* stream);
/* TensorVMSplit.compute_appfeature, models/tensoRF.py:237-256.  xn [n,3] normalised -> [n,app_dim] */
int iff_app_feature(const iff_field* f, const float* xn, int64_t n, float* out, void* stream);
/* TensorBase.compute_alpha, models/tensorBase.py:756-773.  xyz [n,3] world -> alpha [n] */
int iff_point_alpha(const iff_field* f, const float* xyz, int64_t n, float length, float* alpha, void* stream);
/* samples_points_normals, pose_estimation/sampling.py:535-541 (+ Ref.compute_normals, models/ref.py:154).
 * xyz [n,3] world -> normals [n,3] */
int iff_point_normals(const iff_field* f, const float* xyz, int64_t n, float* normals, void* stream);
/* Ref.forward, models/ref.py:103-152 (normals=None).  viewdirs [n,3], features [n,app_dim] -> rgb [n,3] */
int iff_ref_shade(const iff_field* f, const float* viewdirs, const float* features, int64_t n, float* rgb,
                  void* stream);

/* Ref.compute_normals, models/ref.py:154-155.  features [n,app_dim] -> normals [n,3] */
int iff_ref_normals(const iff_field* f, const float* features, int64_t n, float* normals, void* stream);

/* rotate_isocell + renormalise + origin broadcast: pose_estimation/isocell.py:144-171,
 * pose_estimation/sampling.py:449-461.  cells_host [27,3] = isocell_distribution(27) (host floats, isocell.py:6-68);
 * points,normals [P,3] -> ori,dirs [27*P,3] (point-major, 27 directions contiguous); rays6_opt [27*P,6] (nullable)
 * receives the same rays as (o, d) rows, the layout TensorBase.forward takes (sampling.py:246). */
int iff_isocell_emit(const float* cells_host, const float* points, const float* normals, int64_t P, float* ori, float* dirs,
                     float* rays6_opt, void* stream);

/* TensorBase.forward, models/tensorBase.py:775-917 (is_train=False, ndc_ray=False).
 *   rays [R, ray_cols] (ray_cols 6 or 7: o, d, [radius]); mode IFF_MARCH_*; n_samples <= 0 -> default
 *   (20 for point-centred, field n_samples for slab); bg [3] host floats.
 *   rgb [R,3], depth [R], acc [R] required; alpha [R,S] and counts [R,2] (valid, shaded samples) optional.
 *   Workspace (the [R,S] compositing weights between the two launches): iff_march_workspace(f, R, mode, n_samples).
 * Also what renderer.OctreeRender_trilinear_fast (renderer.py:12-25) calls per chunk. */
size_t iff_march_workspace(const iff_field* f, int64_t R, int32_t mode, int32_t n_samples);
/* samples per ray when n_samples <= 0: 20 for the point-centred sampler (pose_estimation/sampling.py:247), the handle's
 * nSamples (models/tensorBase.py:368) for the slab sampler */
int32_t iff_march_default_samples(const iff_field* f, int32_t mode);
/* Which kernels serve TensorBase.forward (models/tensorBase.py:775-917) for this (mode, n_samples) on this handle:
 *   IFF_MARCH_PLAN_GENERAL  three launches: density + compositing per sample from the vector caches, appearance gather, Ref head
 *   IFF_MARCH_PLAN_FAN      two launches: the fused fan kernel + Ref head (models/ref.py:103-152).  Rays are taken 27 at a time (one iso-cell fan of
 *                           pose_estimation/sampling.py:442-488 when they come from iff_isocell_emit); the table patches the 540
 *                           samples of a tile touch are staged once in LDS and density, compositing, appearance and basis_mat run
 *                           from there.  Chosen for the point-centred 20-sample march of a field whose ten steps span about five
 *                           texels per axis (every aabb config of the reference; under unisphere contraction the reference halves
 *                           the grid in the step, tensorBase.py:361, so its fans span ~21 texels and keep the general plan) and
 *                           whose descriptor leaves density_lanes at 0; any rays are accepted -- a tile
 *                           whose samples do not fit one patch is gathered from global memory by the same kernel, same results.
 *   IFF_MARCH_PLAN_FAN_HEAD one launch: the fan kernel also runs the Ref head and the background blend of its 27 rays (the
 *                           bottleneck rows on the fp32 matrix cores) -- iff_march_shade under the conditions of
 *                           IFF_MARCH_PLAN_FAN when the head has the reference's shape (27 features, feature_c a multiple of 32
 *                           up to 128) and the descriptor leaves head_lanes at 0.  iff_march_features (no colours) runs the same kernel without that phase.
 * All plans produce the same alpha / acc / depth / sample counters bit for bit; the fan plans' colours are bit-identical to each
 * other and agree with the general plan's to fp32 summation order. */
#define IFF_MARCH_PLAN_GENERAL  0
#define IFF_MARCH_PLAN_FAN      2
#define IFF_MARCH_PLAN_FAN_HEAD 3
int32_t iff_march_plan(const iff_field* f, int32_t mode, int32_t n_samples);
/* Which fused fan kernel iff_march_plan's IFF_MARCH_PLAN_FAN(_HEAD) means on this handle (TensorBase.forward with sample_point_color,
 * models/tensorBase.py:775-917, :623-638): *waves = 4 (k4f_fan_march: four waves per 27-ray fan, register-staged 12-texel patches),
 * 8 (k4g_fan_march: eight waves per fan, patches by global -> LDS DMA) or 0 (the general kernels); *patch_side = 12 or 22 texels
 * (22: unisphere scenes, tensorBase.py:361-365).  iff_field_desc.fan_waves names one; this reports the choice (bench.py's roofline). */
int iff_march_fan_kernel(const iff_field* f, int32_t mode, int32_t n_samples, int32_t* waves, int32_t* patch_side);
int iff_march_shade(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                    int32_t n_samples, const float* bg_host, float* rgb, float* depth, float* acc,
                    float* alpha_opt, int32_t* counts_opt, void* workspace, size_t workspace_bytes, void* stream);
/* Same call (models/tensorBase.py:775-917), but SYNCHRONOUS and instrumented: stage_ms_host[3] receives the durations of
 * its launches from hipEvents on `stream`: (density+compositing, appearance gather, Ref shading) under
 * IFF_MARCH_PLAN_GENERAL, (0, fused fan kernel, Ref shading) under IFF_MARCH_PLAN_FAN, (0, fused fan kernel incl. the head, 0)
 * under IFF_MARCH_PLAN_FAN_HEAD.  Measurement aid for bench.py's
 * roofline; not for the timed path. */
int iff_march_shade_timed(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                          int32_t n_samples, const float* bg_host, float* rgb, float* depth, float* acc,
                          float* alpha_opt, int32_t* counts_opt, void* workspace, size_t workspace_bytes,
                          float* stage_ms_host, void* stream);

/* The march up to, not including, the Ref head: the differentiable quantities of TensorBase.forward
 * (models/tensorBase.py:886-888: acc_map and the weighted feature sum).  feat28 [R,28] = the 27 summed features + a
 * "has shaded samples" flag (rays_to_consider, :887); depth, acc [R].  Workspace: iff_march_workspace.
 * With iff_march_grad it carries the gradient path of inerf/estimate_pose_inerf.py:151-176 (`model(rays_chunk, ...)`
 * followed by loss.backward() on the camera pose): the caller runs the Ref head (models/ref.py:103-152, a per-ray MLP) in
 * its autograd framework between the two. */
int iff_march_features(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode,
                       int32_t n_samples, float* feat28, float* depth, float* acc, void* workspace,
                       size_t workspace_bytes, void* stream);
/* Gradient of (feat28[:, :27], acc) with respect to the rays (models/tensorBase.py:775-888 as autograd differentiates it
 * for inerf/estimate_pose_inerf.py:176: through sample_ray's slab entry :499-502, the plane/line grid_sample coordinates
 * tensoRF.py:216-256, feature2density :750-754 and raw2alpha :23-35; masks and the weight threshold are constants).
 * g_feat28 [R,28] (column 27 ignored), g_acc [R] -> g_rays6 [R,6] = dL/d(o, d).  'aabb' contraction only (IFF_ERR_UNSUPPORTED
 * otherwise).  Workspace: iff_march_grad_workspace (3 floats per sample). */
size_t iff_march_grad_workspace(const iff_field* f, int64_t R, int32_t mode, int32_t n_samples);
int iff_march_grad(const iff_field* f, const float* rays, int32_t ray_cols, int64_t R, int32_t mode, int32_t n_samples,
                   const float* g_feat28, const float* g_acc, float* g_rays6, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ------------------------------------------------------------------------------------- surface sampler
 * iterative_surface_sampling_process, pose_estimation/sampling.py:509-532 (+ :78-116,131-213,35-67):
 * P seeds in occupied mask voxels, then n_epochs epochs of "jitter <= 5P candidates, accept alpha >
 * quantile_0.6, pick one uniformly".  Device-side counter-based RNG (Philox4x32-10) keyed by `seed`;
 * no host synchronisation.  seed_dev_opt (nullable, device): a 64-bit word added to `seed` when the kernel starts, so a
 * captured hipGraph can draw a fresh stream on every replay.  rho = jitter scale (sampling.py:518-523, computed by the caller).  samples [P,3],
 * alpha [P]; stats [n_epochs,4] int32 = (iterations run, samples left invalid, float bits of the threshold, last
 * candidates-per-sample); stats[3] == -1 reports an in-kernel barrier timeout.
 * Workspace: iff_surface_sample_workspace(P). */
size_t iff_surface_sample_workspace(int64_t P);
int iff_surface_sample(const iff_field* f, int64_t P, int32_t n_epochs, int32_t max_iterations, uint64_t seed,
                       const uint64_t* seed_dev_opt, float rho, float* samples, float* alpha, int32_t* stats, void* workspace, size_t workspace_bytes,
                       void* stream);
/* B independent runs of the same process (pose_estimation/sampling.py:509-532) in ONE launch, for batches of cold
 * queries: run b draws with seed + b * IFF_SAMPLER_SEED_STRIDE (mod 2^64) and is bit-identical to the single call
 * with that seed.  samples [B,P,3], alpha [B,P], stats [B,n_epochs,4]; workspace B * iff_surface_sample_workspace(P). */
#define IFF_SAMPLER_SEED_STRIDE 0x9E3779B97F4A7C15ull
/* The sampler runs as a chain of short launches (seeds; per epoch 8 (first epoch) or 5 iteration launches, a finisher for the rare run that needs
 * more, the apply step), the kernel boundary being the grid barrier of sampling.py:143-213's loop: no workgroup waits for another
 * one and nothing has to be resident together -- device_capacity is then INT32_MAX and wgs_per_run the workgroups of one run's
 * iteration launch.  On a handle made with iff_field_desc.sampler_persistent = 1 the sampler is ONE persistent launch whose workgroups meet
 * at in-kernel barriers; all of them must then be resident together: device_capacity = sampler workgroups the device holds at
 * once (from the kernel's register / LDS footprint), and a caller that keeps several sampler launches in flight (streams, graphs)
 * must keep sum(B * wgs_per_run) <= device_capacity; one batched launch clamps itself.  Both forms draw the same samples bit for bit. */
int iff_surface_sample_residency(const iff_field* f, int32_t B, int64_t P, int32_t* wgs_per_run, int32_t* device_capacity);
int iff_surface_sample_batched(const iff_field* f, int32_t B, int64_t P, int32_t n_epochs, int32_t max_iterations,
                               uint64_t seed, const uint64_t* seed_dev_opt, float rho, float* samples, float* alpha,
                               int32_t* stats, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------ identification
 * Ray encoder + attention projections (weights of id_module.th, nn.Linear layouts [out,in]).
 * Replaces pose_estimation/ray_preprocessor.py:4-39 and multihead_attention.py:44-45. */
typedef struct iff_idnet iff_idnet;

typedef struct iff_idnet_desc {
    int32_t feature_c;              /* 256 (identification_module.py:66-68) */
    int32_t fea;                    /* 384 */
    int32_t img_fea;                /* 398 = 384 + 14 */
    int32_t gemm_mode;              /* IFF_GEMM_* */
    int32_t trunk_variant;          /* work split of the F16X2 logits launches: 0 = choose (the fused launch: form 1; the launch
                                       against cached encoder planes, iff_logits_from_cache: form 3, which halves the query-plane
                                       stream per logit); 1 = 8 waves x 64 rays per workgroup (cached planes: as 0), 2 = 4 waves
                                       x 64 rays, 3 = 8 waves x 128 rays, 4 = sixteen waves on two 64-ray tiles one stage apart
                                       (matrix-core and vector stages side by side).  Logits, softmax statistics and scores are
                                       the same bits in every form: the partials are per 64-ray block whatever the tile. */
    const float* l1_w; const float* l1_b;   /* ray_preprocessor.mlp.0   [feature_c,141] */
    const float* l2_w; const float* l2_b;   /* ray_preprocessor.mlp.2   [feature_c,feature_c] */
    const float* l3_w; const float* l3_b;   /* ray_preprocessor.mlp2.0  [feature_c,feature_c+141] */
    const float* l4_w; const float* l4_b;   /* ray_preprocessor.mlp2.2  [fea,feature_c] */
    const float* q_w;  const float* q_b;    /* attention.q_proj [fea,img_fea] */
    const float* k_w;  const float* k_b;    /* attention.k_proj [fea,fea] */
} iff_idnet_desc;

int  iff_idnet_create(const iff_idnet_desc* desc, void* stream, iff_idnet** out);
void iff_idnet_destroy(iff_idnet* net);
/* the IFF_GEMM_* arithmetic the handle's fused encoder / logits launch actually runs (F16X2 / F16X1 may have fallen back to
 * BF16X3 at create time; ray_preprocessor.py:29-39 is computed to fp32 accuracy in every mode but F16X1) */
int32_t iff_idnet_gemm_mode(const iff_idnet* net);
/* layer widths of the handle: featureC of RayPreprocessor (pose_estimation/ray_preprocessor.py:6), its output width and the
 * image-token width of MultiHeadAttention (multihead_attention.py:31-45) */
int iff_idnet_dims(const iff_idnet* net, int32_t* feature_c, int32_t* fea, int32_t* img_fea);
/* Table file of the identification net (every Linear in its MFMA layouts: k-major fp32, bf16 planes, fragment-ordered
 * bf16 / fp16 planes with their planned scales, the folded token-side Linear), as iff_field_save / iff_field_load.
 * Counterpart of id_module.th (pose_estimation/train.py:226-234, reloaded at train_eval_pose_est.py:59-66). */
int iff_idnet_save(const iff_idnet* net, const char* path, void* stream);
int iff_idnet_load(const char* path, void* stream, iff_idnet** out);

/* RayPreprocessor.forward (ray_preprocessor.py:29-39) and, when k_out != NULL, k_proj
 * (multihead_attention.py:61).  o,d,rgb [N,3] -> feat_opt [N,fea] (nullable), k_out [N,fea] (nullable). */
size_t iff_ray_encode_workspace(const iff_idnet* net, int64_t N);
int iff_ray_encode(const iff_idnet* net, const float* o, const float* d, const float* rgb, int64_t N,
                   float* feat_opt, float* k_out, void* workspace, size_t workspace_bytes, void* stream);
/* k_proj alone (multihead_attention.py:61), for callers that hold encoded rays.  ray_features [N,fea] -> k [N,fea] */
int iff_k_proj(const iff_idnet* net, const float* ray_features, int64_t N, float* k_out, void* stream);
/* q_proj (multihead_attention.py:60).  img [M,img_fea] -> q [M,fea].  Workspace: iff_q_proj_workspace(net, M). */
size_t iff_q_proj_workspace(const iff_idnet* net, int32_t M);
int iff_q_proj(const iff_idnet* net, const float* img, int32_t M, float* q, void* workspace, size_t workspace_bytes,
               void* stream);

/* The same logits with the three bias-only-separated Linears folded (no nonlinearity lies between
 * ray_preprocessor.py:24-25 mlp2.2, multihead_attention.py:60 q_proj and :61 k_proj):
 *   q . k = (Wq t + bq) . (Wk (W4 h3 + b4) + bk) = qf[0:C] . h3 + qf[C]
 * with h3 [N,feature_c] the encoder's last hidden activation (after mlp2.0 + ReLU, ray_preprocessor.py:38) and
 * qf [M, iff_q_fold_width] one token-side Linear whose weights are formed once, in double, at iff_idnet_create.
 * Removes two of the five per-ray GEMMs and a third of the logits GEMM; results agree with the unfolded chain to fp32
 * re-association error (tests/test_hip_identify.py::test_folded_heads).
 *   iff_ray_trunk:          o,d,rgb [N,3] -> h3 [N,feature_c]          (ray_preprocessor.py:29-38 up to the last ReLU)
 *   iff_q_fold:             img [M,img_fea] -> qf [M, width]           (multihead_attention.py:60, folded)
 *   iff_attn_logits_folded: logits[M,N] = (qf[:, :C] h3^T + qf[:, C]) / divisor, row statistics as iff_attn_logits
 *                           (multihead_attention.py:6-8) */
size_t iff_ray_trunk_workspace(const iff_idnet* net, int64_t N);
int iff_ray_trunk(const iff_idnet* net, const float* o, const float* d, const float* rgb, int64_t N, float* h3,
                  void* workspace, size_t workspace_bytes, void* stream);
int32_t iff_q_fold_width(const iff_idnet* net);
int iff_q_fold(const iff_idnet* net, const float* img, int32_t M, float* qf, void* stream);
int iff_attn_logits_folded(const iff_idnet* net, const float* qf, const float* h3, int32_t M, int64_t N, float divisor,
                           float* logits, float* row_max, float* row_sumexp, void* stream);
/* iff_ray_trunk + iff_attn_logits_folded as one call (ray_preprocessor.py:29-38 then multihead_attention.py:6-8,
 * folded): o,d,rgb [N,3] and qf [M, iff_q_fold_width] -> logits [M,N], row_max [M], row_sumexp [M].  For a 256-wide
 * encoder in IFF_GEMM_BF16X3 mode the encoder's hidden activations stay in LDS and the logits are their "fourth
 * layer"; other configurations run the two calls above internally. */
size_t iff_ray_logits_folded_workspace(const iff_idnet* net, int64_t N, int32_t M);
int iff_ray_logits_folded(const iff_idnet* net, const float* o, const float* d, const float* rgb, int64_t N,
                          const float* qf, int32_t M, float divisor, float* logits, float* row_max, float* row_sumexp,
                          void* workspace, size_t workspace_bytes, void* stream);
/* B cold queries at once, each with its own ray set and its own M tokens (the per-image loop of
 * pose_estimation/test.py:67-91 around identification_module.py:162-165): o,d,rgb [B,N,3], qf [B*M, width] ->
 * logits [B,M,N], row_max / row_sumexp [B,M].  One launch for the whole batch (grid.y = query). */
size_t iff_ray_logits_folded_batched_workspace(const iff_idnet* net, int32_t B, int64_t N, int32_t M);
int iff_ray_logits_folded_batched(const iff_idnet* net, int32_t B, const float* o, const float* d, const float* rgb,
                                  int64_t N, const float* qf, int32_t M, float divisor, float* logits, float* row_max,
                                  float* row_sumexp, void* workspace, size_t workspace_bytes, void* stream);
/* The batched call (ray_preprocessor.py:29-38 + multihead_attention.py:6-8), but SYNCHRONOUS and instrumented:
 * trunk_ms_host[1] receives the duration of the fused encoder/logits launch from hipEvents on `stream` (-1 when the
 * configuration has no fused launch).  Measurement aid for bench.py's roofline; not for the timed path. */
int iff_ray_logits_folded_timed(const iff_idnet* net, int32_t B, const float* o, const float* d, const float* rgb, int64_t N,
                                const float* qf, int32_t M, float divisor, float* logits, float* row_max,
                                float* row_sumexp, void* workspace, size_t workspace_bytes, float* trunk_ms_host,
                                void* stream);

/* ------------------------------------------------------------------------------------------- image backbone
 * DINOv2 ViT-S/14 (pose_estimation/backbone.py:12-14: torch.hub "dinov2_vits14"), the network IdentificationModule.image_processing
 * runs on every query image (pose_estimation/identification_module.py:137-146, forward_features(img)["x_norm_patchtokens"]):
 * 14 x 14 patch embedding, class token, position embedding, `depth` pre-norm blocks (LayerNorm eps 1e-6, 6-head attention, MLP with
 * exact GELU, LayerScale), final LayerNorm.  bf16 operands on the matrix cores, fp32 accumulation; the residual stream, LayerNorms,
 * softmax statistics and outputs are fp32.  Weights are read once at create (device pointers, fp32, nn.Linear layouts, the per-block
 * tensors stacked along a leading `depth` axis); `pos` is the position embedding ALREADY interpolated to the 1 + grid_h * grid_w
 * tokens of this input size (DINOv2's interpolate_pos_encoding; the host mirror does it).  Built for dim = 384, heads = 6 (ViT-S)
 * and at most 288 tokens (224 x 224 inputs: 257). */
typedef struct iff_vit iff_vit;
typedef struct iff_vit_desc {
    int32_t dim, depth, heads, mlp, patch, grid_h, grid_w;
    float   ln_eps;
    const float* patch_w;  const float* patch_b;      /* patch_embed.proj.{weight [dim,3,patch,patch], bias [dim]} */
    const float* cls;      const float* pos;          /* cls_token [dim]; position embedding [1 + grid_h*grid_w, dim] */
    const float* ln1_w;    const float* ln1_b;        /* blocks.*.norm1.{weight,bias}    [depth, dim] */
    const float* qkv_w;    const float* qkv_b;        /* blocks.*.attn.qkv.{weight,bias} [depth, 3 dim, dim], [depth, 3 dim] */
    const float* proj_w;   const float* proj_b;       /* blocks.*.attn.proj              [depth, dim, dim], [depth, dim] */
    const float* ls1;                                 /* blocks.*.ls1.gamma              [depth, dim] */
    const float* ln2_w;    const float* ln2_b;        /* blocks.*.norm2 */
    const float* fc1_w;    const float* fc1_b;        /* blocks.*.mlp.fc1                [depth, mlp, dim], [depth, mlp] */
    const float* fc2_w;    const float* fc2_b;        /* blocks.*.mlp.fc2                [depth, dim, mlp], [depth, dim] */
    const float* ls2;                                 /* blocks.*.ls2.gamma */
    const float* norm_w;   const float* norm_b;       /* norm.{weight,bias} [dim] */
    int32_t precision;                                /* IFF_VIT_FP32 / IFF_VIT_BF16 below */
    int32_t gemm_form;                                /* which kernels multiply a batch of >= 24 images (smaller batches: 64-token register-staged
                                                       * tiles).  0: the library's choice (today 10); 1: register-staged 128 x 128 tiles; 2-10: operand
                                                       * tiles by LDS-DMA in several shapes and buffer counts (csrc/vit_kernels.hip gemm()).  Every form
                                                       * returns the same bits (one k order per accumulator): the others exist for A/B runs and tests */
} iff_vit_desc;
/* Arithmetic of the backbone's matrix products.  The reference runs DINOv2 in fp32 (pose_estimation/identification_module.py:137-142,
 * backbone.py:12-14); IFF_VIT_FP32 is that accuracy class on the fp16 matrix cores: every operand split exactly into two fp16
 * pieces (22 significant bits), three MFMA products per block, fp32 accumulation -- tokens agree with the fp32 torch module to
 * ~1e-5 relative.  IFF_VIT_BF16 rounds the operands to bf16 (8 significant bits: ~1e-2 relative on the tokens) and is ~2x
 * faster: a throughput option, never the default of the drop-in. */
#define IFF_VIT_FP32 0
#define IFF_VIT_BF16 1
/* builds the handle from the parameters of torch.hub's dinov2_vits14 (pose_estimation/backbone.py:12-14) */
int  iff_vit_create(const iff_vit_desc* desc, void* stream, iff_vit** out);
void iff_vit_destroy(iff_vit* vit);
/* forward_features (backbone.py:12-14 / identification_module.py:141): images [Q,3,patch*grid_h,patch*grid_w] fp32, already resized,
 * cropped and normalised -> patch_tokens [Q, grid_h*grid_w, dim] (x_norm_patchtokens) and, when cls_opt != NULL, cls_opt [Q, dim]
 * (x_norm_clstoken).  Workspace: iff_vit_workspace(vit, Q). */
size_t iff_vit_workspace(const iff_vit* vit, int32_t Q);
int iff_vit_forward(const iff_vit* vit, const float* images, int32_t Q, float* patch_tokens, float* cls_opt, void* workspace,
                    size_t workspace_bytes, void* stream);

/* Image preprocessing of IdentificationModule (pose_estimation/identification_module.py:36-61: torchvision Resize(256, BICUBIC) /
 * Resize(BILINEAR), CenterCrop(224), Normalize): an antialiased separable resize -- the triangle (cubic = 0) or cubic a = -0.5
 * (cubic = 1) filter stretched by the scale factor, as F.interpolate(antialias=True, align_corners=False) computes it -- of
 * channels-last images src [Q,H,W,C] (C <= 4) to a virtual [resized_h, resized_w] image, of which only the crop window
 * [crop_top, crop_left, crop_h, crop_w] is evaluated; every channel is then normalised, (v - mean[c]) / std[c] (host arrays of C
 * floats, NULL = no normalisation), and written channels-first: dst [Q,C,crop_h,crop_w].  Scale factors up to 7.5 (cubic) / 15
 * (triangle) per axis (32 filter taps); larger ones return IFF_ERR_UNSUPPORTED. */
int iff_image_resize_crop(const float* src, int32_t Q, int32_t H, int32_t W, int32_t C, int32_t resized_h, int32_t resized_w,
                          int32_t crop_top, int32_t crop_left, int32_t crop_h, int32_t crop_w, int32_t cubic, const float* mean_host_opt,
                          const float* std_host_opt, float* dst, void* stream);

/* The same resize / crop / normalise straight from the RGBA query images the evaluation loop holds (dataset.all_rgbs [n,H,W,4],
 * pose_estimation/test.py:75-81): src [Q,H,W,4];
 *   IFF_RESIZE_RGB_ON_WHITE  dst [Q,3,crop_h,crop_w]: the colour composited on white, rgb * a + (1 - a) (test.py:78-80: the same
 *                            three fp32 roundings as the reference's multiply / subtract / add), evaluated per input pixel, then
 *                            filtered exactly as iff_image_resize_crop filters a [Q,H,W,3] image holding those values
 *   IFF_RESIZE_ALPHA         dst [Q,1,crop_h,crop_w]: the alpha channel (mask_img = obs_img[..., -1], test.py:77; the mask side of
 *                            identification_module.py:49-61, :133-136) */
#define IFF_RESIZE_RGB_ON_WHITE 1
#define IFF_RESIZE_ALPHA        2
int iff_image_resize_crop_rgba(const float* src, int32_t Q, int32_t H, int32_t W, int32_t mode, int32_t resized_h, int32_t resized_w,
                               int32_t crop_top, int32_t crop_left, int32_t crop_h, int32_t crop_w, int32_t cubic,
                               const float* mean_host_opt, const float* std_host_opt, float* dst, void* stream);

/* Image tokens for stage C: what IdentificationModule.image_processing does after the backbone
 * (pose_estimation/identification_module.py:149-160) -- append the 14-channel position code of get_img_position_encoding
 * (:76-99: grid position in [-1,1]^2, 'ij' indexing, then sin / cos of it at octaves 1, 2, 4) to every patch token, and turn
 * the row selection `[mask > mask_thres]` (:157-160, mask = the resized alpha channel) into one keep flag per token.
 *   patch_tokens [Q, gh*gw, C] (backbone output, row-major over the grid), mask_grid_opt [Q, gh*gw] or NULL (keep all),
 *   lin_h_host / lin_w_host: torch.linspace(-1, 1, gh / gw) as host floats -> tokens_out [Q, gh*gw, C+14], keep_out [Q, gh*gw].
 * Rows are NOT compacted (shapes stay static, nothing is read back to the host): run the logits on all gh*gw rows, then
 * iff_mask_token_rows on the row statistics before iff_attn_colsum -- a dropped row then contributes exp(l - inf) / 1 = 0 to
 * every column sum, which is exactly what deleting it (identification_module.py:157-160 before :167) does. */
int iff_token_assemble(const float* patch_tokens, int32_t Q, int32_t gh, int32_t gw, int32_t C, const float* mask_grid_opt,
                       float mask_thres, const float* lin_h_host, const float* lin_w_host, float* tokens_out, uint8_t* keep_out,
                       void* stream);
/* iff_token_assemble with the KEPT ROWS FIRST: every image's rows are partitioned stably (kept rows in grid order, then the dropped
 * ones) and rows_out[q] = how many it kept; keep_out becomes 1 ... 1 0 ... 0.  identification_module.py:157-160 deletes the dropped rows
 * before the attention (:164-167), so the reference's work per image follows the kept count; with the kept rows in front
 * iff_logits_from_cache_rows and iff_attn_colsum_rows stop at rows_out[q] -- the same saving with static shapes and no host read --
 * and the first rows_out[q] rows of an image's logits ARE the reference's attention-map rows, in its order. */
int iff_token_assemble_compact(const float* patch_tokens, int32_t Q, int32_t gh, int32_t gw, int32_t C, const float* mask_grid_opt,
                               float mask_thres, const float* lin_h_host, const float* lin_w_host, float* tokens_out, uint8_t* keep_out,
                               int32_t* rows_out, void* stream);
/* keep [rows] (iff_token_assemble), statistics [rows]: rows with keep == 0 get (row_max, row_sumexp) = (+inf, 1)
 * (the mask select of pose_estimation/identification_module.py:157-160 applied to the softmax rows of :165-167) */
int iff_mask_token_rows(const uint8_t* keep, int64_t rows, float* row_max, float* row_sumexp, void* stream);

/* The encoder cached per resident ray set.  The reference re-runs RayPreprocessor + k_proj for every query image
 * (pose_estimation/identification_module.py:164 inside the loop of pose_estimation/test.py:67-91) although the rays of a model
 * do not change between images (train_eval_pose_est.py:131-149) and the weights are frozen in eval: here the encoder's last
 * hidden activation is computed ONCE per ray set (iff_ray_cache_build: ray_preprocessor.py:29-38 up to the last ReLU) and
 * every later batch of token rows only pays the folded logits product against it (iff_logits_from_cache:
 * multihead_attention.py:6-8).  The cache is opaque: fp16 hi/lo planes [2][N][feature_c] under IFF_GEMM_F16X2, the fp32
 * activation [N][feature_c] otherwise; iff_ray_cache_bytes sizes it.  Results equal the uncached calls
 * (iff_ray_logits_folded) bit for bit under F16X2.  The caller rebuilds the cache when it re-emits rays or changes weights. */
size_t iff_ray_cache_bytes(const iff_idnet* net, int64_t N);
size_t iff_ray_cache_workspace(const iff_idnet* net, int64_t N);
int iff_ray_cache_build(const iff_idnet* net, const float* o, const float* d, const float* rgb, int64_t N, void* cache,
                        size_t cache_bytes, void* workspace, size_t workspace_bytes, void* stream);
size_t iff_logits_from_cache_workspace(const iff_idnet* net, int64_t N, int32_t M);
int iff_logits_from_cache(const iff_idnet* net, const void* cache, int64_t N, const float* qf, int32_t M, float divisor,
                          float* logits, float* row_max, float* row_sumexp, void* workspace, size_t workspace_bytes,
                          void* stream);
/* iff_logits_from_cache for token rows laid out by iff_token_assemble_compact: rows_per_block[b] (device, M / 256 entries, M a multiple
 * of 256: one block per image of a 16 x 16 grid) = the kept rows of block b, which come first -- the rows the mask select of
 * pose_estimation/identification_module.py:157-160 leaves for the attention of :164-167.  Under IFF_GEMM_F16X2 the groups of 32 rows
 * that lie wholly behind the count are neither multiplied nor written (their logits are left as they were); the statistics of EVERY
 * row behind the count become (+inf, 1) (rows of the last, partly kept group hold ordinary logits); the other arithmetic modes
 * compute every row.  A caller reads the first rows_per_block[b] rows of a block only (iff_attn_colsum_rows does).  Kept rows: the bits
 * of iff_logits_from_cache.  Workspace: iff_logits_from_cache_workspace. */
int iff_logits_from_cache_rows(const iff_idnet* net, const void* cache, int64_t N, const float* qf, int32_t M,
                               const int32_t* rows_per_block, float divisor, float* logits, float* row_max, float* row_sumexp,
                               void* workspace, size_t workspace_bytes, void* stream);

/* scaled_attention_product (multihead_attention.py:4-12, mask=None), split so that ray shards on several
 * GPUs can exchange row statistics between the two halves:
 *   iff_attn_logits: logits[M,N] = q k^T / divisor (divisor = sqrt(d_k)), row_max[M], row_sumexp[M]
 *                    (= sum_j exp(l_ij - row_max_i)); the two statistics are optional (both or neither)
 *   iff_attn_colsum: attention = exp(l - row_max)/row_sumexp written in place when write_attention != 0,
 *                    score[N] = sum_i attention_ij   (identification_module.py:167) */
int iff_attn_logits(const float* q, const float* k, int32_t M, int64_t N, int32_t D, float divisor, float* logits,
                    float* row_max, float* row_sumexp, int32_t gemm_mode, void* stream);
int iff_attn_colsum(float* logits_inout, int32_t M, int64_t N, const float* row_max, const float* row_sumexp,
                    int32_t write_attention, float* score, void* stream);
/* iff_attn_colsum for Q queries at once: logits [Q,M,N], statistics [Q,M] -> score [Q,N]
 * (the per-image loop of pose_estimation/test.py:67-91 around identification_module.py:167) */
int iff_attn_colsum_batched(float* logits_inout, int32_t Q, int32_t M, int64_t N, const float* row_max,
                            const float* row_sumexp, int32_t write_attention, float* score, void* stream);
/* iff_attn_colsum_batched over the first rows_per_query[q] (device) rows of every query only -- attention_map.sum(0) over the rows
 * identification_module.py:157-160 keeps; the other rows are neither read nor written */
int iff_attn_colsum_rows(float* logits_inout, int32_t Q, int32_t M, int64_t N, const float* row_max, const float* row_sumexp,
                         const int32_t* rows_per_query, int32_t write_attention, float* score, void* stream);

/* ---- the merge steps of the ray-sharded path (one process per GPU, rays sharded by surface point; the exchanges themselves are
 * RCCL all_gathers issued by the caller).  The reference is single-process; over G column shards these reproduce its ONE softmax
 * over the ray axis (pose_estimation/multihead_attention.py:11) and its ONE torch.topk (identification_module.py:207). */
/* stats_all [G,R,2] = every rank's per-row (max, sum exp(l - max)) over its own columns, in rank order ->
 * gmax [R], gsum [R] over all columns (ranks added in rank order: reproducible) */
int iff_merge_row_stats(const float* stats_all, int32_t G, int64_t R, float* gmax, float* gsum, void* stream);
/* a rank's local top-kl (idx [Q,kl] int64 into its own rays, val [Q,kl]; iff_topk_batched) -> its message msg [Q,k,8] =
 * (score, bits of the GLOBAL ray index idx + first_ray, origin, direction); slots kl..k-1 are padding that sorts last.
 * ray_stride = 0: all queries index one ray set rays_o / rays_d [n,3]; otherwise query q reads rays_o + q * ray_stride floats. */
int iff_pack_candidates(const int64_t* idx, const float* val, const float* rays_o, const float* rays_d, int64_t ray_stride, int32_t Q,
                        int32_t kl, int32_t k, int64_t first_ray, float* msg, void* stream);
/* cand_all [G,Qt,k,8] (every rank's message, each list in torch.topk's order) -> for the queries q0 .. q0+Q-1 the global top-k
 * of identification_module.py:207: val [Q,k] descending, lower ray index first on ties, idx [Q,k] (global), and the winners'
 * origins / directions [Q,k,3] (what the pose solve of pose_estimation/test.py:133-174 reads) */
int iff_merge_candidates(const float* cand_all, int32_t G, int32_t Qt, int32_t q0, int32_t Q, int32_t k, float* val, int64_t* idx,
                         float* rays_o_out, float* rays_d_out, void* stream);

/* torch.topk(scores, k) (identification_module.py:207): values descending, ties by lower index first.
 * idx [k] int64, val [k].  Workspace: iff_topk_workspace(N, k). */
size_t iff_topk_workspace(int64_t N, int32_t k);
int iff_topk(const float* score, int64_t N, int32_t k, int64_t* idx, float* val, void* workspace,
             size_t workspace_bytes, void* stream);
/* Q score rows at once (identification_module.py:207 inside the per-image loop of test.py:67-91):
 * score [Q,N] -> idx [Q,k], val [Q,k]; one workgroup per query */
int iff_topk_batched(const float* score, int32_t Q, int64_t N, int32_t k, int64_t* idx, float* val, void* stream);

/* Per-image pose solve, pose_estimation/test.py:133-174,192-194 with pose_geometry.py:42-95,175-204:
 * unique-origin filter, LS line intersection, negative exclusion, look-at rotation, NaN/singular -> identity.
 * idx [k] (indices into rays), val [k]; rays_o, rays_d [N,3]; up_host[3] (normalised inside, test.py:29).
 * c2w [16] row-major; parts_opt [8 + k] = centre(3), watch(3), n_kept, spare, weights[k] (nullable). */
int iff_pose_from_topk(const int64_t* idx, const float* val, int32_t k, const float* rays_o, const float* rays_d,
                       int64_t N, const float* up_host, float* c2w, float* parts_opt, void* stream);
/* Q pose solves at once (the per-image loop of pose_estimation/test.py:67-91 around :133-174): idx, val [Q,k] ->
 * c2w [Q,16], parts_opt [Q, 8 + k] (nullable; per query as iff_pose_from_topk).  ray_batch_stride = 0: all queries index ONE ray set
 * rays_o/rays_d [N,3]; otherwise query q reads rays_o + q * ray_batch_stride (floats), e.g. k*3 for per-query gathered
 * candidates [Q,k,3]. */
int iff_pose_from_topk_batched(const int64_t* idx, const float* val, int32_t Q, int32_t k, const float* rays_o,
                               const float* rays_d, int64_t N, int64_t ray_batch_stride, const float* up_host,
                               float* c2w, float* parts_opt, void* stream);

/* The error metrics the evaluation loop computes per image (pose_estimation/test.py:213-232 with errors.py:3-9, and the "loss"
 * entry of test.py:241), for Q images in one launch and without a host round trip per image: c2w, gt_c2w [Q,16] row-major,
 * parts_opt [Q, 8 + k] from iff_pose_from_topk(_batched) or NULL -> summary [Q,4] = (mean weight of the rays the origin filter
 * kept -- NaN when parts_opt is NULL --, translation error || gt[:3,3] - c2w[:3,3] ||, angular error in degrees
 * rad2deg(acos(clamp((trace(R_gt R^-1) - 1) / 2, -1, 1))) with the inverse by pivoted LU as torch.linalg.inv, rays kept). */
int iff_pose_errors(const float* c2w, const float* gt_c2w, const float* parts_opt, int32_t Q, int32_t k, float* summary, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IFFNERF_HIP_H */
