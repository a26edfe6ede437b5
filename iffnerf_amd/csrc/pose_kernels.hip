// pose_kernels.hip -- closed-form camera pose from the top-k rays, one workgroup, no host round trip.
// Follows pose_estimation/test.py:133-174,192-194 and pose_geometry.py:42-95,175-204 of the reference, quirks included:
//   * the "unique origin" filter is torch.isin over SCALARS (any coordinate of the ray origin equal to any coordinate of
//     any origin that occurs exactly once among the k), test.py:133-136;
//   * both least-squares solves are unweighted (the weights argument is commented out, test.py:146,154), so the second
//     one repeats the first; only the watch direction uses the (exclusion-masked) scores.
// 3x3 determinant / solve / inverse use LU with partial pivoting like the LAPACK routines behind torch.linalg.
#include "iff_device.h"
#include "iff_launch.h"
#include <cmath>

constexpr int PK_MAX = 1024;

struct Lu3 { float a[3][3]; int piv[3]; float sign; };

__device__ inline void lu3(const float m[3][3], Lu3& f) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) f.a[i][j] = m[i][j];
    f.sign = 1.0f;
    for (int c = 0; c < 3; ++c) {
        int p = c;
        float best = fabsf(f.a[c][c]);
        for (int r = c + 1; r < 3; ++r) if (fabsf(f.a[r][c]) > best) { best = fabsf(f.a[r][c]); p = r; }
        f.piv[c] = p;
        if (p != c) {
            for (int j = 0; j < 3; ++j) { float t = f.a[c][j]; f.a[c][j] = f.a[p][j]; f.a[p][j] = t; }
            f.sign = -f.sign;
        }
        if (f.a[c][c] != 0.0f) {
            for (int r = c + 1; r < 3; ++r) {
                f.a[r][c] = f.a[r][c] / f.a[c][c];
                for (int j = c + 1; j < 3; ++j) f.a[r][j] = f.a[r][j] - f.a[r][c] * f.a[c][j];
            }
        }
    }
}
__device__ inline float lu3_det(const Lu3& f) { return f.sign * f.a[0][0] * f.a[1][1] * f.a[2][2]; }
__device__ inline void lu3_solve(const Lu3& f, const float b[3], float x[3]) {
    float y[3] = {b[0], b[1], b[2]};
    for (int c = 0; c < 3; ++c) if (f.piv[c] != c) { float t = y[c]; y[c] = y[f.piv[c]]; y[f.piv[c]] = t; }
    y[1] = y[1] - f.a[1][0] * y[0];
    y[2] = y[2] - f.a[2][0] * y[0] - f.a[2][1] * y[1];
    x[2] = y[2] / f.a[2][2];
    x[1] = (y[1] - f.a[1][2] * x[2]) / f.a[1][1];
    x[0] = (y[0] - f.a[0][1] * x[1] - f.a[0][2] * x[2]) / f.a[0][0];
}

__global__ void __launch_bounds__(256) k_pose(const int64_t* __restrict__ idx, const float* __restrict__ val, int k,
                                              const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t N,
                                              float up0, float up1, float up2, int isin_direct_limit,
                                              float* __restrict__ c2w, float* __restrict__ parts) {
    __shared__ float so[PK_MAX][3], sd[PK_MAX][3], sw[PK_MAX];
    __shared__ unsigned char once[PK_MAX], keep[PK_MAX];
    __shared__ float red[256][12];
    __shared__ float centre[3];
    __shared__ float wsum;
    const int tid = threadIdx.x;
    for (int i = tid; i < k; i += 256) {
        int64_t r = idx[i];
        bool ok = r >= 0 && r < N;
        for (int c = 0; c < 3; ++c) { so[i][c] = ok ? rays_o[3 * r + c] : NAN; sd[i][c] = ok ? rays_d[3 * r + c] : NAN; }
        sw[i] = val[i];
    }
    __syncthreads();
    // rows occurring exactly once (torch.unique(dim=0, return_counts=True); NaN never equals itself)
    for (int i = tid; i < k; i += 256) {
        int cnt = 0;
        for (int j = 0; j < k; ++j)
            cnt += (so[i][0] == so[j][0] && so[i][1] == so[j][1] && so[i][2] == so[j][2]) ? 1 : 0;
        once[i] = (cnt == 1);
    }
    __syncthreads();
    // torch.isin(origins, scalars of once-rows, assume_unique=True).any(dim=1), test.py:134-136.
    // aten picks between two algorithms (TensorCompare.cpp isin_Tensor_Tensor_out): a direct membership test when the
    // test set is small, else a stable-sort scan that -- because assume_unique=True is passed although the 3k scalars
    // are NOT unique -- also flags every scalar that has an equal scalar LATER in the flattened [k,3] array.  Both
    // behaviours are reproduced so the kept set equals the reference's.
    __shared__ int n_once_s;
    if (tid == 0) {
        int c = 0;
        for (int j = 0; j < k; ++j) c += once[j];
        n_once_s = c;
    }
    __syncthreads();
    const bool sort_mode = !(3 * n_once_s < isin_direct_limit);
    for (int i = tid; i < k; i += 256) {
        bool hit = false;
        for (int c = 0; c < 3 && !hit; ++c) {
            const float v = so[i][c];
            for (int j = 0; j < k && !hit; ++j) {
                if (once[j]) hit = (v == so[j][0]) || (v == so[j][1]) || (v == so[j][2]);
                if (sort_mode && !hit) {
                    for (int e = 0; e < 3; ++e) hit = hit || ((3 * j + e > 3 * i + c) && (v == so[j][e]));
                }
            }
        }
        keep[i] = hit;
    }
    __syncthreads();
    // weights = weights / sum(weights) over kept rays ; R = sum(I - d d^T), q = sum((I - d d^T) o)
    float loc[12];
    for (int c = 0; c < 12; ++c) loc[c] = 0.0f;
    float lw = 0.0f;
    for (int i = tid; i < k; i += 256) {
        if (!keep[i]) continue;
        lw += sw[i];
        float P[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) P[a][b] = ((a == b) ? 1.0f : 0.0f) - sd[i][a] * sd[i][b];
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) loc[a * 3 + b] += P[a][b];
            loc[9 + a] += (P[a][0] * so[i][0] + P[a][1] * so[i][1]) + P[a][2] * so[i][2];
        }
    }
    for (int c = 0; c < 12; ++c) red[tid][c] = loc[c];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) for (int c = 0; c < 12; ++c) red[tid][c] += red[tid + off][c];
        __syncthreads();
    }
    float Rm[3][3], qv[3];
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) Rm[a][b] = red[0][a * 3 + b]; qv[a] = red[0][9 + a]; }
    __syncthreads();
    red[tid][0] = lw;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) red[tid][0] += red[tid + off][0];
        __syncthreads();
    }
    if (tid == 0) {
        wsum = red[0][0];
        Lu3 f;
        lu3(Rm, f);
        float c[3] = {NAN, NAN, NAN};
        if (!(lu3_det(f) < 1.e-7f)) lu3_solve(f, qv, c);        // pose_geometry.py:82-84
        centre[0] = c[0]; centre[1] = c[1]; centre[2] = c[2];
    }
    __syncthreads();
    // exclusion of rays pointing away from the centre, renormalise, watch direction
    float lsum = 0.0f;
    for (int i = tid; i < k; i += 256) {
        float w = 0.0f;
        if (keep[i]) {
            w = sw[i] / wsum;
            float dp = ((centre[0] - so[i][0]) * sd[i][0] + (centre[1] - so[i][1]) * sd[i][1]) + (centre[2] - so[i][2]) * sd[i][2];
            w = w * ((dp > 0.0f) ? 1.0f : 0.0f);
        }
        sw[i] = w;
        lsum += w;
    }
    __syncthreads();
    red[tid][0] = lsum;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) red[tid][0] += red[tid + off][0];
        __syncthreads();
    }
    float w2 = red[0][0];
    __syncthreads();
    float lwd[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < k; i += 256) {
        float w = keep[i] ? (sw[i] / w2) : 0.0f;
        sw[i] = w;
        if (keep[i]) for (int c = 0; c < 3; ++c) lwd[c] += sd[i][c] * w;
    }
    for (int c = 0; c < 3; ++c) red[tid][c] = lwd[c];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) for (int c = 0; c < 3; ++c) red[tid][c] += red[tid + off][c];
        __syncthreads();
    }
    if (tid == 0) {
        float wd[3] = {red[0][0], red[0][1], red[0][2]};
        float wn = sqrtf(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
        float watch[3] = {wd[0] / wn, wd[1] / wn, wd[2] / wn};
        float un = sqrtf(up0 * up0 + up1 * up1 + up2 * up2);                 // test.py:29
        float up[3] = {up0 / un, up1 / un, up2 / un};
        float dir[3] = {-watch[0], -watch[1], -watch[2]};
        // make_rotation_mat(direction, up): x = up x dir, y = dir x x (both normalised), rows (x, y, dir)
        float x[3] = {up[1] * dir[2] - up[2] * dir[1], up[2] * dir[0] - up[0] * dir[2], up[0] * dir[1] - up[1] * dir[0]};
        float xn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        for (int c = 0; c < 3; ++c) x[c] = x[c] / xn;
        float y[3] = {dir[1] * x[2] - dir[2] * x[1], dir[2] * x[0] - dir[0] * x[2], dir[0] * x[1] - dir[1] * x[0]};
        float yn = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
        for (int c = 0; c < 3; ++c) y[c] = y[c] / yn;
        float rot[3][3] = {{x[0], x[1], x[2]}, {y[0], y[1], y[2]}, {dir[0], dir[1], dir[2]}};
        Lu3 f;
        lu3(rot, f);
        float inv[3][3];
        if (lu3_det(f) < 1.0e-7f) {                                           // test.py:169-171 (NaN compares false -> not taken)
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) inv[a][b] = (a == b) ? 1.0f : 0.0f;
        } else {
            for (int b = 0; b < 3; ++b) {
                float e[3] = {b == 0 ? 1.f : 0.f, b == 1 ? 1.f : 0.f, b == 2 ? 1.f : 0.f}, col[3];
                lu3_solve(f, e, col);
                inv[0][b] = col[0]; inv[1][b] = col[1]; inv[2][b] = col[2];
            }
        }
        float M[16] = {inv[0][0], inv[0][1], inv[0][2], centre[0], inv[1][0], inv[1][1], inv[1][2], centre[1],
                       inv[2][0], inv[2][1], inv[2][2], centre[2], 0.f, 0.f, 0.f, 1.f};
        bool bad = false;
        for (int i = 0; i < 16; ++i) bad = bad || (M[i] != M[i]);
        for (int i = 0; i < 16; ++i) c2w[i] = bad ? ((i % 5 == 0) ? 1.0f : 0.0f) : M[i];   // test.py:192-194
        if (parts) {
            parts[0] = centre[0]; parts[1] = centre[1]; parts[2] = centre[2];
            parts[3] = watch[0]; parts[4] = watch[1]; parts[5] = watch[2];
            int nk = 0;
            for (int i = 0; i < k; ++i) nk += keep[i];
            parts[6] = (float)nk; parts[7] = 0.0f;
        }
    }
    if (parts) {
        __syncthreads();
        for (int i = tid; i < k; i += 256) parts[8 + i] = keep[i] ? sw[i] : -1.0f;   // -1 marks a filtered ray
    }
}

hipError_t launch_pose(const int64_t* idx, const float* val, int k, const float* rays_o, const float* rays_d, int64_t N,
                       const float* up3, float* c2w, float* parts, hipStream_t s) {
    if (k < 1 || k > PK_MAX) return hipErrorInvalidValue;
    // aten's heuristic (taken from numpy): direct membership test iff n_test < 10 * n_elements^0.145
    const int lim = (int)(int64_t)(10.0f * std::pow((double)(3 * k), 0.145));
    hipLaunchKernelGGL(k_pose, dim3(1), dim3(256), 0, s, idx, val, k, rays_o, rays_d, N, up3[0], up3[1], up3[2], lim, c2w, parts);
    return hipGetLastError();
}
