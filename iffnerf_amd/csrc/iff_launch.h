// iff_launch.h -- host-side launcher prototypes shared between the kernel translation units and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "iff_device.h"

// field_kernels.hip
hipError_t launch_k0_channels_last(const float* src, float* dst, int C, int64_t HW, hipStream_t s);
hipError_t launch_k0_mask_bytes(const float* src, uint8_t* dst, int64_t n, hipStream_t s);
hipError_t launch_k0_mask_cells(const uint8_t* mask, uint8_t* cell, int D, int H, int W, hipStream_t s);
hipError_t launch_k0_basis_slices(const float* src, float* dst, int app_dim, int n_app, hipStream_t s);
hipError_t launch_k0_basis_lanes(const float* src, float* dst, int app_dim, int n_app, hipStream_t s);
hipError_t launch_normalize_coord(const FieldDev& f, const float* xyz, int64_t n, float* out, hipStream_t s);
hipError_t launch_mask_sample(const FieldDev& f, const float* xyz, int64_t n, float* out, hipStream_t s);
hipError_t launch_mask_occupied(const FieldDev& f, const float* xyz, int64_t n, uint8_t* out, hipStream_t s);
hipError_t launch_density_feature(const FieldDev& f, const float* xn, int64_t n, float* out, hipStream_t s);
hipError_t launch_point_alpha(const FieldDev& f, const float* xyz, int64_t n, float length, float* out, hipStream_t s);
hipError_t launch_app_feature(const FieldDev& f, const float* xn, int64_t n, float* out, hipStream_t s);
hipError_t launch_point_normals(const FieldDev& f, const float* xyz, int64_t n, float* out, hipStream_t s);
hipError_t launch_ref_normals(const FieldDev& f, const float* feat, int64_t n, float* out, hipStream_t s);
hipError_t launch_shade_blend(const FieldDev& f, const float* rays, int ray_cols, const float* feat28, const float* acc,
                              const float* bg, int64_t n, float* rgb, hipStream_t s);
hipError_t launch_ref_shade(const FieldDev& f, const float* dirs, const float* feat, int64_t n, float* rgb, hipStream_t s);

// march_kernels.hip
hipError_t launch_isocell_emit(const float* cells27x3_host, const float* pts, const float* nrm, int64_t P, float* ori,
                               float* dirs, float* rays6, hipStream_t s);
size_t march_workspace_bytes(int64_t R, int S);
int march_plan(const FieldDev& f, int mode, int S);
hipError_t launch_march(const FieldDev& f, const float* rays, int ray_cols, int64_t R, int mode, int S, const float* bg,
                        float* rgb, float* depth, float* acc, float* alpha, int* counts, float* feat_out, void* ws,
                        size_t ws_bytes, float* stage_ms_host, hipStream_t s);
// fan_march_kernels.hip (the fused form of the point-centred march for 27-ray fans; MarchArgs: march_common.h)
struct MarchArgs;
bool fan_head_fusable(const FieldDev& f);
hipError_t launch_ref_shade_oct(const FieldDev& f, const float* dirs, int dir_stride, const float* feat, int feat_stride,
                                const float* acc, const float* bg, int64_t n, float* rgb, hipStream_t s);
bool march_head_fused(const FieldDev& f);
bool fan_march_eligible(const FieldDev& f, int mode, int S);
hipError_t launch_fan_march(const FieldDev& f, const MarchArgs& a, int variant, hipStream_t s);
// fan8_march_kernels.hip (the eight-wave form: DMA-staged patches, 12- and 22-texel boxes)
int fan8_patch_side(const FieldDev& f, int mode, int S);
hipError_t launch_fan8_march(const FieldDev& f, const MarchArgs& a, int variant, hipStream_t s);
int fan_kernel_for(const FieldDev& f, int mode, int S);
size_t march_grad_workspace_bytes(int64_t R, int S);
hipError_t launch_march_grad(const FieldDev& f, const float* rays, int ray_cols, int64_t R, int mode, int S,
                             const float* g_feat, int g_feat_ld, const float* g_acc, float* g_rays, void* ws, size_t ws_bytes,
                             hipStream_t s);

// sampler_kernels.hip
size_t sampler_workspace_bytes(int64_t P);
int sampler_lpc(const FieldDev& f, int B);
bool sampler_stepped(const FieldDev& f);
hipError_t sampler_residency(int64_t P, int n_cus, int lpc, int B, int* wgs_per_query, int* capacity);
hipError_t launch_surface_sample_occ(const FieldDev& f, const int* occ_list, int n_occ, int B, int64_t P, int n_epochs,
                                     int max_iterations, uint64_t seed, const uint64_t* seed_dev, float rho, float* samples,
                                     float* alpha, int* stats, void* ws, size_t ws_bytes, int n_cus, hipStream_t s);

// identify_kernels.hip
struct IdNetDev {
    // weights transposed to [in][out] (k-major), K zero-padded to a multiple of 32
    const float* w1; const float* b1;   // [160][256]   (141 inputs padded to 160)
    const float* w2; const float* b2;   // [256][256]
    const float* w3; const float* b3;   // [416][256]   rows 0..255 <- h, rows 256..396 <- x (141), then zero rows
    const float* w4; const float* b4;   // [256][384]
    const float* wk; const float* bk;   // [384][384]
    const float* wq; const float* bq;   // [400][384]   (398 inputs padded to 400)
    // the same weights pre-split into three bf16 planes [3][out][K_pad] (nn.Linear row layout) for the 3xBF16 GEMM
    const void* p1; const void* p2; const void* p3; const void* p4; const void* pk;
    // q_proj, k_proj and mlp2.2 folded into one token-side Linear (api.hip: fold_heads): wqf [KQ][qf_ld] k-major,
    // bqf [qf_ld]; columns 0..C-1 give the folded query, column C the per-token constant, the rest are zero
    const float* wqf; const float* bqf; int qf_ld;
    // layers 1-3 in the fragment order of the fused trunk kernel (k5_trunk; only when feature_c == 256), else null
    const void* f1; const void* f2; const void* f3;
    int fused_trunk;                    // 1: one launch for the three ReLU layers; 0: one 3xBF16 GEMM launch per layer
    // IFF_GEMM_F16X2 (trunk_f16_kernels.hip): layers 1-3 as fp16 hi/lo planes in fragment order, each times a power of
    // two (exponents below, planned at create time: api.hip plan_f16_scales); null / 0 when that mode is not in use
    const void* h1; const void* h2; const void* h3h; const void* h3x;
    int trunk_f16;                      // 1: the fused launch is k5_trunk_h
    int trunk_variant;                  // k5_trunk_h work split: 0 = 8 waves x 64 rays, 1 = 4 waves x 64 rays, 2 = 8 waves x 128 rays
    int e_x, e_h1, e_h2, e_h3;          // activations are stored times 2^e
    int e_w1, e_w2, e_w3h, e_w3x;       // weights are stored times 2^e  (e_w3h + e_h2 == e_w3x + e_x)
    int gemm_mode;                      // 0: fp32-input MFMA (k-ordered fmaf chain), 1: 3xBF16 split on the bf16 MFMA
    int feature_c, fea, img_fea;
};
size_t ray_encode_workspace_bytes(const IdNetDev& n, int64_t N);
hipError_t launch_ray_encode(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, float* feat,
                             float* kout, void* ws, size_t ws_bytes, hipStream_t s);
size_t ray_trunk_workspace_bytes(const IdNetDev& n, int64_t N);
hipError_t launch_ray_trunk(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, float* h3,
                            void* ws, size_t ws_bytes, hipStream_t s);
size_t ray_logits_workspace_bytes(const IdNetDev& n, int64_t N, int M, int B);
hipError_t launch_ray_logits_folded(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, const float* qf,
                                    int M, int B, float divisor, float* logits, float* row_max, float* row_sumexp, void* ws,
                                    size_t ws_bytes, float* trunk_ms_host, hipStream_t s);
hipError_t launch_q_fold(const IdNetDev& n, const float* img, int M, float* qf, hipStream_t s);
hipError_t launch_attn_logits_folded(const float* qf, int ldq, const float* h3, int M, int64_t N, int C, float divisor,
                                     float* logits, float* row_max, float* row_sumexp, hipStream_t s);
hipError_t launch_k_proj(const IdNetDev& n, const float* feat, int64_t N, float* kout, hipStream_t s);
hipError_t launch_q_proj(const IdNetDev& n, const float* img, int M, float* q, void* scratch, hipStream_t s);
hipError_t launch_transpose_pad(const float* w_out_in, float* dst_in_out, int out_f, int in_f, int in_pad, int row_off,
                                hipStream_t s);
hipError_t launch_attn_logits(const float* q, const float* k, int M, int64_t N, int D, float divisor, float* logits,
                              float* row_max, float* row_sumexp, int gemm_mode, hipStream_t s);
hipError_t launch_frag_order(const void* Wp, void* Wf, int Kp, hipStream_t s);
// trunk_f16_kernels.hip
hipError_t launch_frag_order_h(const float* W, int ld, int col0, int ncols, int nks, float scale, void* Wf, hipStream_t s);
int trunk_h_rays_per_wg(int variant);
int trunk_h_cached_variant(int variant);
hipError_t launch_trunk_h_features(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, int B, float* h3,
                                   hipStream_t s);
hipError_t launch_trunk_h_cache(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, void* planes,
                                hipStream_t s);
hipError_t launch_trunk_h_logits_cached(const IdNetDev& n, const void* planes, int64_t N, const float* qf, int M, float divisor,
                                        float* logits, void* Qf, float* qscale, float2* part, const int* rows, hipStream_t s);
hipError_t launch_merge_stats(const float2* part, int n_blk, int Mpad, int M, int B, float* row_max, float* row_sumexp, const int* rows,
                              hipStream_t s);
hipError_t launch_trunk_h_logits(const IdNetDev& n, const float* o, const float* d, const float* rgb, int64_t N, const float* qf, int M,
                                 int B, float divisor, float* logits, void* Qf, float* qscale, float2* part, hipStream_t s);
hipError_t launch_split_rows(const float* w, void* planes, int out_f, int in_f, int in_pad, hipStream_t s);
hipError_t launch_attn_colsum(float* logits, int Q, int M, int64_t N, const float* row_max, const float* row_sumexp,
                              int write_attention, float* score, const int* rows, hipStream_t s);
hipError_t launch_token_assemble(const float* tok, int Q, int gh, int gw, int C, const float* mask, float thres, const float* lin_h,
                                 const float* lin_w, float* out, uint8_t* keep, hipStream_t s);
hipError_t launch_token_assemble_compact(const float* tok, int Q, int gh, int gw, int C, const float* mask, float thres, const float* lin_h,
                                         const float* lin_w, float* out, uint8_t* keep, int* rows_out, hipStream_t s);
hipError_t launch_mask_token_rows(const uint8_t* keep, int64_t rows, float* row_max, float* row_sumexp, hipStream_t s);
size_t topk_workspace_bytes(int64_t N, int k);
hipError_t launch_topk(const float* score, int Q, int64_t N, int k, int64_t* idx, float* val, void* ws, size_t ws_bytes,
                       hipStream_t s);

// shard_kernels.hip -- merges of the ray-sharded path
hipError_t launch_merge_row_stats(const float* stats_all, int G, int64_t R, float* gmax, float* gsum, hipStream_t s);
hipError_t launch_pack_candidates(const int64_t* idx, const float* val, const float* ori, const float* dirs, int64_t ray_stride, int Q, int kl,
                                  int k, int64_t first_ray, float* msg, hipStream_t s);
hipError_t launch_merge_candidates(const float* cand_all, int G, int Qt, int q0, int Q, int k, float* val, int64_t* idx, float* ori, float* dir,
                                   hipStream_t s);

// pose_kernels.hip
hipError_t launch_pose(const int64_t* idx, const float* val, int Q, int k, const float* rays_o, const float* rays_d, int64_t N,
                       int64_t ray_batch_stride, const float* up3, float* c2w, float* parts, hipStream_t s);
hipError_t launch_pose_errors(const float* c2w, const float* gt, const float* parts, int Q, int k, float* out, hipStream_t s);

// vit_kernels.hip -- the ViT-S/14 image backbone (pose_estimation/backbone.py:12-14)
constexpr int VIT_MAX_DEPTH = 64;
struct VitDev {
    // GEMM weights in nn.Linear layout [out][in], stacked over the blocks: bf16 (prec 0), or fp16 hi plane followed by the lo
    // plane of the whole stack, each layer's weights times the power of two 1 / s_*[layer] (prec 1: the fp32-accurate mode)
    const void* patch_w;                  // [dim][kp]           patch_embed.proj.weight, k = c P P + dy P + dx, zero padded to kp
    const void* qkv_w;                    // [depth][3 dim][dim]
    const void* proj_w;                   // [depth][dim][dim]
    const void* fc1_w;                    // [depth][mlp][dim]
    const void* fc2_w;                    // [depth][dim][mlp]
    // fp32 vectors
    const float* patch_b; const float* cls; const float* pos;          // [dim], [dim], [T][dim] (position embedding at this grid)
    const float* ln1_w; const float* ln1_b; const float* ln2_w; const float* ln2_b;      // [depth][dim]
    const float* qkv_b; const float* proj_b; const float* fc1_b; const float* fc2_b;     // [depth][3 dim], [depth][dim], [depth][mlp], [depth][dim]
    const float* ls1; const float* ls2;   // LayerScale gammas [depth][dim]
    const float* norm_w; const float* norm_b;
    int dim, depth, heads, mlp, patch, gh, gw, T, kp;
    float eps;
    int prec;                             // 0: bf16 operands; 1: split fp16 operands (hi + lo), three products per block
    int gemm_form;                        // iff_vit_desc.gemm_form: which GEMM kernels serve a large batch (0: the product's choice)
    float s_patch, s_qkv[VIT_MAX_DEPTH], s_proj[VIT_MAX_DEPTH], s_fc1[VIT_MAX_DEPTH], s_fc2[VIT_MAX_DEPTH];      // accumulator scales (prec 1)
};
hipError_t launch_resize_crop(const float* src, int Q, int H, int W, int C, int mode, int rh, int rw, int top, int left, int ch, int cw, int cubic,
                              const float* mean, const float* std, float* dst, hipStream_t s);
size_t vit_workspace_bytes(const VitDev& v, int Q);
hipError_t launch_vit_to_bf16(const float* src, int64_t n, void* dst, hipStream_t s);
hipError_t launch_vit_to_f16_planes(const float* src, int64_t rows, int cols, int KP, float scale, void* hi, void* lo, hipStream_t s);
hipError_t launch_vit_pad_rows(const float* src, int rows, int cols, int KP, void* dst, hipStream_t s);
hipError_t launch_vit_forward(const VitDev& v, const float* images, int Q, int H, int W, float* patch_tokens, float* cls_opt, void* ws,
                              size_t ws_bytes, hipStream_t s);
