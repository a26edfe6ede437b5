// march_grad_kernels.hip -- gradient of TensorBase.forward's compositing (models/tensorBase.py:775-917) with respect to
// the rays: what inerf/estimate_pose_inerf.py:151-176 back-propagates through `model(rays_chunk, ...)` to the camera pose.
//
// The differentiable outputs of the march before the Ref head are the per-ray weighted feature F = sum_s w_s f_s [27] and
// acc = sum_s w_s (depth is computed under no_grad, tensorBase.py:903-905; the aabb test, the occupancy mask and the
// weight > rayMarch_weight_thres selection are comparisons and carry no gradient).  Given dL/dF and dL/dacc per ray this
// kernel returns dL/d(o, d):
//   x_s = o + d z_s, z_s = t0(o, d) + step s             (sample_ray, tensorBase.py:494-536; point-centred: z_s constant)
//   sigma_s = softplus(feat(xn_s) + shift), alpha_s = 1 - exp(-sigma_s dist_s scale), a_s = 1 - alpha_s + 1e-10
//   T_s = prod_{j<s} a_j, w_s = alpha_s T_s                  (raw2alpha, tensorBase.py:23-35)
//   G_s = dL/dw_s = [w_s > thres] (dL/dF . f_s) + dL/dacc
//   dL/dalpha_s = G_s T_s - (sum_{m>s} G_m w_m) / a_s        (cumprod's gradient)
//   dL/dxn_s = dL/dalpha_s (1 - alpha_s) dist_s scale sigmoid(feat + shift) grad feat(xn_s)  +  [shaded] w_s grad (dL/dF . f)(xn_s)
// with the coordinate gradients of the bilinear plane / linear line reads as ATen's grid_sampler_2d_backward forms them
// (align_corners=True, zero padding: out-of-range corners contribute neither value nor slope).
// One wave per ray, one lane per sample, 64 samples per pass:
//   pass A, front to back: alpha, the transmittance by a wave prefix product, the shaded samples' feature term and its
//           coordinate gradient; (alpha, +-T, G) go to a [R][S][3] record (sign of T = sample valid);
//   pass B, back to front: the suffix sums by a wave scan, the density coordinate gradients of the valid samples.
// The two passes re-gather the density tables instead of keeping [R][S] gradients; a 4096-ray iNeRF batch is ~4 M samples.
#include "iff_device.h"
#include "iff_launch.h"

namespace {

struct GradArgs {
    const float* rays; int ray_cols; int64_t R; int mode; int S;
    const float* g_feat; int g_feat_ld;   // [R][ld]: dL/dF in columns [0, app_dim)
    const float* g_acc;                   // [R]
    float* rec;                           // [R][S][3]
    float* g_rays;                        // [R][6]
};

__device__ inline float slab_entry_g(const FieldDev& f, const float o[3], const float d[3]) {
    float tmax = -INFINITY;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        float v = (d[ax] == 0.0f) ? 1e-6f : d[ax];
        float ra = (f.aabb_hi[ax] - o[ax]) / v, rb = (f.aabb_lo[ax] - o[ax]) / v;
        tmax = fmaxf(tmax, fminf(ra, rb));
    }
    return tmax;
}

__device__ inline float z_at(const FieldDev& f, int mode, int S, float t0, int s) {
    if (mode == 0) return f.step_size * (float)(s - S / 2);
    return t0 + f.step_size * (float)s;
}

// value and coordinate gradient (w.r.t. the normalised coordinate xn) of  sum_i sum_c h[i*C + c] plane_i,c(xn) line_i,c(xn)
// (h == nullptr: all ones -- the density feature of compute_densityfeature, tensoRF.py:216-235; with h = basis_mat^T dL/dF
// the appearance term of compute_appfeature, :237-256, contracted with dL/dF)
template <bool WEIGHTED>
__device__ inline void vm_value_grad(const FieldDev& f, const float* const* planes, const float* const* lines, int C,
                                     const float xn[3], const float* h, float& val, float g[3]) {
    val = 0.0f; g[0] = g[1] = g[2] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int a = mat_a(i), b = mat_b(i), v = vec_ax(i);
        const int W = f.grid[a], H = f.grid[b], L = f.grid[v];
        const float x = unnorm(xn[a], W), y = unnorm(xn[b], H), z = unnorm(xn[v], L);
        const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
        const float wx1 = x - fx, wy1 = y - fy, wz1 = z - fz;
        const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1, wz0 = 1.0f - wz1;
        const bool ok = (x > -2.0f) && (x < (float)(W + 1)) && (y > -2.0f) && (y < (float)(H + 1));
        const int x0 = ok ? (int)fx : -2, y0 = ok ? (int)fy : -2;
        const bool okz = (z > -2.0f) && (z < (float)(L + 1));
        const int z0 = okz ? (int)fz : -2;
        int off[4]; float in[4];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int xx = x0 + dx, yy = y0 + dy;
                const bool inside = (xx >= 0) && (xx < W) && (yy >= 0) && (yy < H);
                off[dy * 2 + dx] = inside ? (yy * W + xx) : 0;
                in[dy * 2 + dx] = inside ? 1.0f : 0.0f;
            }
        int loff[2]; float lin[2];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const int zz = z0 + dz;
            const bool inside = (zz >= 0) && (zz < L);
            loff[dz] = inside ? zz : 0;
            lin[dz] = inside ? 1.0f : 0.0f;
        }
        const float* pt = planes[i];
        const float* lt = lines[i];
        float vx = 0.0f, vy = 0.0f, vz = 0.0f;
        for (int ch = 0; ch < C; ch += 4) {
            const float4 nw4 = ld4(pt + (size_t)off[0] * C + ch), ne4 = ld4(pt + (size_t)off[1] * C + ch);
            const float4 sw4 = ld4(pt + (size_t)off[2] * C + ch), se4 = ld4(pt + (size_t)off[3] * C + ch);
            const float4 lo4 = ld4(lt + (size_t)loff[0] * C + ch), hi4 = ld4(lt + (size_t)loff[1] * C + ch);
            const float nwv[4] = {nw4.x, nw4.y, nw4.z, nw4.w}, nev[4] = {ne4.x, ne4.y, ne4.z, ne4.w};
            const float swv[4] = {sw4.x, sw4.y, sw4.z, sw4.w}, sev[4] = {se4.x, se4.y, se4.z, se4.w};
            const float lov[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, hiv[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float nw = nwv[e] * in[0], ne = nev[e] * in[1], sw = swv[e] * in[2], se = sev[e] * in[3];
                const float lo = lov[e] * lin[0], hi = hiv[e] * lin[1];
                const float P = nw * (wy0 * wx0) + ne * (wy0 * wx1) + sw * (wy1 * wx0) + se * (wy1 * wx1);
                const float Px = (ne - nw) * wy0 + (se - sw) * wy1;
                const float Py = (sw - nw) * wx0 + (se - ne) * wx1;
                const float Lv = lo * wz0 + hi * wz1;
                const float Lz = hi - lo;
                const float hw = WEIGHTED ? h[i * C + ch + e] : 1.0f;
                val = fmaf(hw * P, Lv, val);
                vx = fmaf(hw * Px, Lv, vx);
                vy = fmaf(hw * Py, Lv, vy);
                vz = fmaf(hw * P, Lz, vz);
            }
        }
        // grid_sample's un-normalisation (align_corners=True): texel = (xn + 1) / 2 * (size - 1)
        g[a] = fmaf(vx, 0.5f * (float)(W - 1), g[a]);
        g[b] = fmaf(vy, 0.5f * (float)(H - 1), g[b]);
        g[v] = fmaf(vz, 0.5f * (float)(L - 1), g[v]);
    }
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int GRAD_WAVES = 4;

__global__ void __launch_bounds__(64 * GRAD_WAVES) k_march_grad(FieldDev f, GradArgs a) {
    extern __shared__ float s_h[];                       // [GRAD_WAVES][3 * n_app]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int S = a.S, CA = f.n_app, NK = 3 * CA;
    float* h = s_h + wave * NK;
    const float scale = f.distance_scale;
    for (int64_t r = (int64_t)blockIdx.x * GRAD_WAVES + wave; r < a.R; r += (int64_t)gridDim.x * GRAD_WAVES) {
        const float* rp = a.rays + r * a.ray_cols;
        const float o[3] = {rp[0], rp[1], rp[2]}, d[3] = {rp[3], rp[4], rp[5]};
        const float traw = (a.mode == 1) ? slab_entry_g(f, o, d) : 0.0f;
        const float t0 = (a.mode == 1) ? fminf(fmaxf(traw, f.near), f.far) : 0.0f;
        // h = basis_mat^T dL/dF  (cum_app_features = sum_s w_s basis_mat (plane*line)_s, tensoRF.py:256)
        const float* gf = a.g_feat + r * a.g_feat_ld;
        for (int k = lane; k < NK; k += 64) {
            float v = 0.0f;
            for (int oo = 0; oo < f.app_dim; ++oo) v = fmaf(f.basis[(size_t)oo * NK + k], gf[oo], v);
            h[k] = v;
        }
        __threadfence_block();                               // h: written by some lanes, read by all lanes of this wave
        __builtin_amdgcn_wave_barrier();
        const float gacc = a.g_acc[r];
        float go[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gt = 0.0f;
        float* rec = a.rec + (size_t)r * S * 3;
        // ------------------------------------------------------------------ pass A: front to back
        float carryT = 1.0f;
        for (int base = 0; base < S; base += 64) {
            const int s = base + lane;
            const bool live = s < S;
            const float z = z_at(f, a.mode, S, t0, s);
            const float p[3] = {o[0] + d[0] * z, o[1] + d[1] * z, o[2] + d[2] * z};
            bool valid = live && inside_aabb(f, p);
            float xn[3] = {0.f, 0.f, 0.f};
            float feat = 0.0f;
            if (valid) {
                field_normalize(f, p, xn);
                if (f.mask) valid = mask_occupied(f, p, xn);
                if (valid) feat = density_full(f, xn);
            }
            const float sigma = valid ? feature2density(f, feat) : 0.0f;
            const float dist = (s + 1 < S) ? (z_at(f, a.mode, S, t0, s + 1) - z) : 0.0f;
            const float alpha = live ? 1.0f - expf(-sigma * (dist * scale)) : 0.0f;
            const float aa = live ? (1.0f - alpha) + 1e-10f : 1.0f;
            float incl = aa;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t = __shfl_up(incl, off, 64);
                if (lane >= off) incl *= t;
            }
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = carryT * excl;
            carryT = carryT * __shfl(incl, 63, 64);
            const float w = alpha * T;
            float q = 0.0f;
            if (live && w > f.weight_thres) {                    // tensorBase.py:851
                float gq[3];
                vm_value_grad<true>(f, f.aplane, f.aline, CA, xn, h, q, gq);
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const float gx = w * gq[ax] * f.inv_aabb[ax];
                    go[ax] += gx; gd[ax] = fmaf(gx, z, gd[ax]); gt = fmaf(gx, d[ax], gt);
                }
            }
            if (live) {
                rec[(size_t)s * 3 + 0] = alpha;
                rec[(size_t)s * 3 + 1] = valid ? T : -T;
                rec[(size_t)s * 3 + 2] = q + gacc;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ------------------------------------------------------------------ pass B: back to front
        float carry = 0.0f;
        for (int base = ((S - 1) / 64) * 64; base >= 0; base -= 64) {
            const int s = base + lane;
            const bool live = s < S;
            const float alpha = live ? rec[(size_t)s * 3 + 0] : 0.0f;
            const float Ts = live ? rec[(size_t)s * 3 + 1] : 0.0f;
            const float G = live ? rec[(size_t)s * 3 + 2] : 0.0f;
            const float T = fabsf(Ts);
            const float gw = G * (alpha * T);
            float x = gw;                                     // x[lane] = sum_{l >= lane} gw[l]
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t = __shfl_down(x, off, 64);
                if (lane + off < 64) x += t;
            }
            float after = __shfl_down(x, 1, 64);
            if (lane == 63) after = 0.0f;
            const float suffix = carry + after;
            carry = carry + __shfl(x, 0, 64);
            const float z = z_at(f, a.mode, S, t0, s);
            const float dist = (s + 1 < S) ? (z_at(f, a.mode, S, t0, s + 1) - z) : 0.0f;
            const float aa = (1.0f - alpha) + 1e-10f;
            const float dalpha = G * T - suffix / aa;
            const float dsigma = dalpha * ((1.0f - alpha) * (dist * scale));
            if (live && Ts > 0.0f && dsigma != 0.0f) {
                const float p[3] = {o[0] + d[0] * z, o[1] + d[1] * z, o[2] + d[2] * z};
                float xn[3], feat, gfe[3];
                field_normalize(f, p, xn);
                vm_value_grad<false>(f, f.dplane, f.dline, f.n_density, xn, nullptr, feat, gfe);
                float ds;                                        // d sigma / d feat (feature2density, tensorBase.py:750-754)
                if (f.softplus) {
                    const float xs = feat + f.density_shift;
                    ds = (xs > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-xs));
                } else {
                    ds = feat > 0.0f ? 1.0f : 0.0f;
                }
                const float gfeat = dsigma * ds;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const float gx = gfeat * gfe[ax] * f.inv_aabb[ax];
                    go[ax] += gx; gd[ax] = fmaf(gx, z, gd[ax]); gt = fmaf(gx, d[ax], gt);
                }
            }
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { go[ax] = wave_sum(go[ax]); gd[ax] = wave_sum(gd[ax]); }
        gt = wave_sum(gt);
        if (lane == 0) {
            // t0 = clamp(max_ax min(rate_a, rate_b), near, far): the gradient reaches o and d through the selected axis
            // when the clamp is inactive (torch.clamp passes it on [near, far] inclusive)
            if (a.mode == 1 && traw >= f.near && traw <= f.far) {
                int sel = 0; float best = -INFINITY, rate = 0.0f, vsel = 1.0f;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    const float v = (d[ax] == 0.0f) ? 1e-6f : d[ax];
                    const float ra = (f.aabb_hi[ax] - o[ax]) / v, rb = (f.aabb_lo[ax] - o[ax]) / v;
                    const float m = fminf(ra, rb);
                    if (m > best) { best = m; sel = ax; rate = m; vsel = v; }
                }
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
                    if (ax == sel) {
                        go[ax] += gt * (-1.0f / vsel);
                        if (d[ax] != 0.0f) gd[ax] += gt * (-rate / vsel);
                    }
            }
            float* out = a.g_rays + r * 6;
            out[0] = go[0]; out[1] = go[1]; out[2] = go[2]; out[3] = gd[0]; out[4] = gd[1]; out[5] = gd[2];
        }
        __builtin_amdgcn_wave_barrier();                     // h is rewritten by this wave's next ray
    }
}

}  // namespace

size_t march_grad_workspace_bytes(int64_t R, int S) { return (size_t)R * (size_t)S * 3 * sizeof(float); }

hipError_t launch_march_grad(const FieldDev& f, const float* rays, int ray_cols, int64_t R, int mode, int S,
                             const float* g_feat, int g_feat_ld, const float* g_acc, float* g_rays, void* ws, size_t ws_bytes,
                             hipStream_t s) {
    if (ws_bytes < march_grad_workspace_bytes(R, S)) return hipErrorInvalidValue;
    if (R == 0) return hipSuccess;
    GradArgs a;
    a.rays = rays; a.ray_cols = ray_cols; a.R = R; a.mode = mode; a.S = S;
    a.g_feat = g_feat; a.g_feat_ld = g_feat_ld; a.g_acc = g_acc; a.rec = (float*)ws; a.g_rays = g_rays;
    const int64_t wgs = (R + GRAD_WAVES - 1) / GRAD_WAVES;
    const int64_t grid = wgs < 256 * 8 ? wgs : 256 * 8;
    const size_t lds = (size_t)GRAD_WAVES * 3 * f.n_app * sizeof(float);
    hipLaunchKernelGGL(k_march_grad, dim3((unsigned)grid), dim3(64 * GRAD_WAVES), lds, s, f, a);
    return hipGetLastError();
}
