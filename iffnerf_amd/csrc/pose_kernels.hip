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

// 3x3 LU with partial pivoting on scalar members only (runtime-indexed or struct-array storage ends up in scratch
// memory on this compiler, which costs microseconds per access in a one-thread epilogue).
struct Lu3 {
    float a00, a01, a02, a10, a11, a12, a20, a21, a22;   // L below the diagonal, U on and above, rows already permuted
    int p0, p1;                                           // pivot row picked at step 0 (0..2) and step 1 (1..2)
    float sign;
};
#define IFF_SWAP(x, y) { float t__ = (x); (x) = (y); (y) = t__; }

__device__ inline void lu3(float m00, float m01, float m02, float m10, float m11, float m12, float m20, float m21,
                           float m22, Lu3& f) {
    f.a00 = m00; f.a01 = m01; f.a02 = m02; f.a10 = m10; f.a11 = m11; f.a12 = m12; f.a20 = m20; f.a21 = m21; f.a22 = m22;
    f.sign = 1.0f;
    // column 0: largest |a_i0| (first maximum wins, as LAPACK's isamax)
    f.p0 = 0;
    float best = fabsf(f.a00);
    if (fabsf(f.a10) > best) { best = fabsf(f.a10); f.p0 = 1; }
    if (fabsf(f.a20) > best) { f.p0 = 2; }
    if (f.p0 == 1) { IFF_SWAP(f.a00, f.a10) IFF_SWAP(f.a01, f.a11) IFF_SWAP(f.a02, f.a12) f.sign = -f.sign; }
    if (f.p0 == 2) { IFF_SWAP(f.a00, f.a20) IFF_SWAP(f.a01, f.a21) IFF_SWAP(f.a02, f.a22) f.sign = -f.sign; }
    if (f.a00 != 0.0f) {
        f.a10 = f.a10 / f.a00;
        f.a20 = f.a20 / f.a00;
        f.a11 = f.a11 - f.a10 * f.a01; f.a12 = f.a12 - f.a10 * f.a02;
        f.a21 = f.a21 - f.a20 * f.a01; f.a22 = f.a22 - f.a20 * f.a02;
    }
    // column 1
    f.p1 = 1;
    if (fabsf(f.a21) > fabsf(f.a11)) {
        f.p1 = 2;
        IFF_SWAP(f.a10, f.a20) IFF_SWAP(f.a11, f.a21) IFF_SWAP(f.a12, f.a22)
        f.sign = -f.sign;
    }
    if (f.a11 != 0.0f) {
        f.a21 = f.a21 / f.a11;
        f.a22 = f.a22 - f.a21 * f.a12;
    }
}
__device__ inline float lu3_det(const Lu3& f) { return f.sign * f.a00 * f.a11 * f.a22; }
__device__ inline void lu3_solve(const Lu3& f, float b0, float b1, float b2, float& x0, float& x1, float& x2) {
    float y0 = b0, y1 = b1, y2 = b2;
    if (f.p0 == 1) IFF_SWAP(y0, y1)
    if (f.p0 == 2) IFF_SWAP(y0, y2)
    if (f.p1 == 2) IFF_SWAP(y1, y2)
    y1 = y1 - f.a10 * y0;
    y2 = y2 - f.a20 * y0 - f.a21 * y1;
    x2 = y2 / f.a22;
    x1 = (y1 - f.a12 * x2) / f.a11;
    x0 = (y0 - f.a01 * x1 - f.a02 * x2) / f.a00;
}

// sum NV per-thread values over the 256-thread workgroup: wave xor-shuffles, then the 4 wave partials in fixed order.
// Every thread returns the totals.  `buf` needs 4*NV floats of LDS.
template <int NV>
__device__ inline void block_sum(float* v, float* buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        float x = v[c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        v[c] = x;
    }
    __syncthreads();                       // previous users of buf are done
    if ((tid & 63) == 0)
#pragma unroll
        for (int c = 0; c < NV; ++c) buf[(tid >> 6) * NV + c] = v[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] = (buf[c] + buf[NV + c]) + (buf[2 * NV + c] + buf[3 * NV + c]);
}

__global__ void __launch_bounds__(256) k_pose(const int64_t* __restrict__ idx, const float* __restrict__ val, int k,
                                              const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t N,
                                              int64_t ray_batch_stride, float up0, float up1, float up2, int isin_direct_limit,
                                              float* __restrict__ c2w, float* __restrict__ parts) {
    {   // blockIdx.x = query of a batch (the per-image loop of pose_estimation/test.py:67-91)
        const int64_t qb = blockIdx.x;
        idx += qb * k; val += qb * k; c2w += qb * 16;
        rays_o += qb * ray_batch_stride; rays_d += qb * ray_batch_stride;
        if (parts) parts += qb * (8 + k);
    }
    __shared__ float4 so4[PK_MAX];                  // origin xyz + "occurs exactly once" flag in .w
    __shared__ float sd[PK_MAX * 3], sw[PK_MAX];
    __shared__ unsigned char keep[PK_MAX], flag[PK_MAX * 3];
    __shared__ float red[4 * 12];
    __shared__ float centre[3];
    __shared__ int n_once_s;
    const int tid = threadIdx.x;
    if (tid == 0) n_once_s = 0;
    for (int i = tid; i < k; i += 256) {
        int64_t r = idx[i];
        bool ok = r >= 0 && r < N;
        so4[i] = ok ? make_float4(rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2], 0.0f) : make_float4(NAN, NAN, NAN, 0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) sd[3 * i + c] = ok ? rays_d[3 * r + c] : NAN;
        sw[i] = val[i];
    }
    __syncthreads();
    // rows occurring exactly once (torch.unique(dim=0, return_counts=True); NaN never equals itself).
    // One 16-byte LDS read per row, 4 rows in flight.
    int my_cnt[4] = {0, 0, 0, 0};                   // k <= 1024 -> at most 4 rows per thread
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int i = tid + 256 * u;
        if (i < k) {
            const float4 me = so4[i];
            int cnt = 0;
#pragma unroll 4
            for (int j = 0; j < k; ++j) {
                const float4 ot = so4[j];
                cnt += (int)(me.x == ot.x) & (int)(me.y == ot.y) & (int)(me.z == ot.z);
            }
            my_cnt[u] = cnt;
        }
    }
    __syncthreads();                                // all reads done before the flags are written into .w
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int i = tid + 256 * u;
        if (i < k && my_cnt[u] == 1) { so4[i].w = 1.0f; atomicAdd(&n_once_s, 1); }
    }
    __syncthreads();
    // torch.isin(origins, scalars of once-rows, assume_unique=True).any(dim=1), test.py:134-136.
    // aten picks between two algorithms (TensorCompare.cpp isin_Tensor_Tensor_out): a direct membership test when the
    // test set is small, else a stable-sort scan that -- because assume_unique=True is passed although the 3k scalars
    // are NOT unique -- also flags every scalar that has an equal scalar LATER in the flattened [k,3] array.  Both
    // behaviours are reproduced so the kept set equals the reference's.  One thread per scalar, broadcast LDS reads.
    const bool sort_mode = !(3 * n_once_s < isin_direct_limit);
    for (int p = tid; p < 3 * k; p += 256) {
        const float4 mine = so4[p / 3];
        const int pe = p - 3 * (p / 3);
        const float v = (pe == 0) ? mine.x : (pe == 1) ? mine.y : mine.z;
        int hit = 0;                                 // branch-free: short-circuit operators here compile to divergent jumps
        const int sm = sort_mode ? 1 : 0;
#pragma unroll 4
        for (int j = 0; j < k; ++j) {
            const float4 ot = so4[j];
            const int oj = (ot.w != 0.0f) ? 1 : 0;
            const int q = 3 * j;
            hit |= (int)(v == ot.x) & (oj | (sm & (int)(q > p)));
            hit |= (int)(v == ot.y) & (oj | (sm & (int)(q + 1 > p)));
            hit |= (int)(v == ot.z) & (oj | (sm & (int)(q + 2 > p)));
        }
        flag[p] = (unsigned char)hit;
    }
    __syncthreads();
    for (int i = tid; i < k; i += 256) keep[i] = flag[3 * i] | flag[3 * i + 1] | flag[3 * i + 2];
    __syncthreads();
    // weights = weights / sum(weights) over kept rays ; R = sum(I - d d^T), q = sum((I - d d^T) o)
    float loc[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) loc[c] = 0.0f;
    for (int i = tid; i < k; i += 256) {
        if (!keep[i]) continue;
        loc[12] += sw[i];
        const float* d = sd + 3 * i;
        const float4 o4 = so4[i];
        const float o[3] = {o4.x, o4.y, o4.z};
        float P[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) P[a][b] = ((a == b) ? 1.0f : 0.0f) - d[a] * d[b];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) loc[a * 3 + b] += P[a][b];
            loc[9 + a] += (P[a][0] * o[0] + P[a][1] * o[1]) + P[a][2] * o[2];
        }
    }
    __shared__ float red13[4 * 13];
    block_sum<13>(loc, red13);
    const float wsum = loc[12];
    if (tid == 0) {
        Lu3 f;
        lu3(loc[0], loc[1], loc[2], loc[3], loc[4], loc[5], loc[6], loc[7], loc[8], f);
        float c0 = NAN, c1 = NAN, c2 = NAN;
        if (!(lu3_det(f) < 1.e-7f)) lu3_solve(f, loc[9], loc[10], loc[11], c0, c1, c2);        // pose_geometry.py:82-84
        centre[0] = c0; centre[1] = c1; centre[2] = c2;
    }
    __syncthreads();
    // exclusion of rays pointing away from the centre, renormalise, watch direction
    float lsum[1] = {0.0f};
    for (int i = tid; i < k; i += 256) {
        float w = 0.0f;
        if (keep[i]) {
            const float* d = sd + 3 * i;
            const float4 o4 = so4[i];
            const float o[3] = {o4.x, o4.y, o4.z};
            w = sw[i] / wsum;
            float dp = ((centre[0] - o[0]) * d[0] + (centre[1] - o[1]) * d[1]) + (centre[2] - o[2]) * d[2];
            w = w * ((dp > 0.0f) ? 1.0f : 0.0f);
        }
        sw[i] = w;
        lsum[0] += w;
    }
    block_sum<1>(lsum, red);
    const float w2 = lsum[0];
    float lwd[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < k; i += 256) {
        float w = keep[i] ? (sw[i] / w2) : 0.0f;
        sw[i] = w;
        if (keep[i]) for (int c = 0; c < 3; ++c) lwd[c] += sd[3 * i + c] * w;
    }
    block_sum<3>(lwd, red);
    if (tid == 0) {
        float wd[3] = {lwd[0], lwd[1], lwd[2]};
        float wn = sqrtf(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
        float watch[3] = {wd[0] / wn, wd[1] / wn, wd[2] / wn};
        float un = sqrtf(up0 * up0 + up1 * up1 + up2 * up2);                 // test.py:29
        float up[3] = {up0 / un, up1 / un, up2 / un};
        float dir[3] = {-watch[0], -watch[1], -watch[2]};
        // make_rotation_mat(direction, up): x = up x dir, y = dir x x (both normalised), rows (x, y, dir)
        float x[3] = {up[1] * dir[2] - up[2] * dir[1], up[2] * dir[0] - up[0] * dir[2], up[0] * dir[1] - up[1] * dir[0]};
        float xn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = x[c] / xn;
        float y[3] = {dir[1] * x[2] - dir[2] * x[1], dir[2] * x[0] - dir[0] * x[2], dir[0] * x[1] - dir[1] * x[0]};
        float yn = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) y[c] = y[c] / yn;
        Lu3 f;
        lu3(x[0], x[1], x[2], y[0], y[1], y[2], dir[0], dir[1], dir[2], f);
        float i00 = 1.f, i01 = 0.f, i02 = 0.f, i10 = 0.f, i11 = 1.f, i12 = 0.f, i20 = 0.f, i21 = 0.f, i22 = 1.f;
        if (!(lu3_det(f) < 1.0e-7f)) {                            // test.py:169-171: singular -> identity rotation
            lu3_solve(f, 1.f, 0.f, 0.f, i00, i10, i20);           // column b of the inverse solves R x = e_b
            lu3_solve(f, 0.f, 1.f, 0.f, i01, i11, i21);
            lu3_solve(f, 0.f, 0.f, 1.f, i02, i12, i22);
        }
        float M[16] = {i00, i01, i02, centre[0], i10, i11, i12, centre[1], i20, i21, i22, centre[2], 0.f, 0.f, 0.f, 1.f};
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) bad = bad || (M[i] != M[i]);
#pragma unroll
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

hipError_t launch_pose(const int64_t* idx, const float* val, int Q, int k, const float* rays_o, const float* rays_d, int64_t N,
                       int64_t ray_batch_stride, const float* up3, float* c2w, float* parts, hipStream_t s) {
    if (k < 1 || k > PK_MAX || Q < 1) return hipErrorInvalidValue;
    // aten's heuristic (taken from numpy): direct membership test iff n_test < 10 * n_elements^0.145
    const int lim = (int)(int64_t)(10.0f * std::pow((double)(3 * k), 0.145));
    hipLaunchKernelGGL(k_pose, dim3((unsigned)Q), dim3(256), 0, s, idx, val, k, rays_o, rays_d, N, ray_batch_stride, up3[0], up3[1],
                       up3[2], lim, c2w, parts);
    return hipGetLastError();
}

// Error metrics of one estimated pose against the ground truth, pose_estimation/test.py:213-232 with errors.py:3-9:
//   translation error = || gt[:3,3] - pred[:3,3] ||_2  (the camera positions [0,0,0,1] @ c2w[:3,:].T of test.py:213-225 ARE those columns)
//   angular error     = rad2deg(acos(clamp((trace(R_gt R_pred^-1) - 1) / 2, -1, 1))), the inverse by LU with partial pivoting like torch.linalg.inv
//   loss              = weights.mean() over the rays the origin filter kept (test.py:241), from the solver's `parts`
// One thread per query: summary [Q,4] = (loss, translation error, angular error in degrees, rays kept).
__global__ void k_pose_errors(const float* __restrict__ c2w, const float* __restrict__ gt, const float* __restrict__ parts, int Q, int k,
                              float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const float* P = c2w + 16 * (int64_t)q;
    const float* G = gt + 16 * (int64_t)q;
    const float dx = G[3] - P[3], dy = G[7] - P[7], dz = G[11] - P[11];
    const float terr = sqrtf((dx * dx + dy * dy) + dz * dz);
    Lu3 f;
    lu3(P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10], f);
    float i00, i01, i02, i10, i11, i12, i20, i21, i22;
    lu3_solve(f, 1.f, 0.f, 0.f, i00, i10, i20);
    lu3_solve(f, 0.f, 1.f, 0.f, i01, i11, i21);
    lu3_solve(f, 0.f, 0.f, 1.f, i02, i12, i22);
    const float t0 = (G[0] * i00 + G[1] * i10) + G[2] * i20;
    const float t1 = (G[4] * i01 + G[5] * i11) + G[6] * i21;
    const float t2 = (G[8] * i02 + G[9] * i12) + G[10] * i22;
    float c = (((t0 + t1) + t2) - 1.0f) / 2.0f;
    c = (c != c) ? c : fminf(fmaxf(c, -1.0f), 1.0f);                     // torch.clamp keeps NaN
    const float aerr = acosf(c) * 57.29577951308232f;
    float loss = NAN, nk = 0.0f;
    if (parts) {
        const float* w = parts + (int64_t)q * (8 + k) + 8;
        float sum = 0.0f;
        for (int i = 0; i < k; ++i)
            if (w[i] >= 0.0f) { sum += w[i]; nk += 1.0f; }
        loss = sum / nk;                                                  // no ray kept: 0 / 0 = NaN, as torch's mean of an empty tensor
    }
    out[4 * q] = loss; out[4 * q + 1] = terr; out[4 * q + 2] = aerr; out[4 * q + 3] = nk;
}

hipError_t launch_pose_errors(const float* c2w, const float* gt, const float* parts, int Q, int k, float* out, hipStream_t s) {
    if (Q < 1) return hipSuccess;
    hipLaunchKernelGGL(k_pose_errors, dim3((unsigned)((Q + 63) / 64)), dim3(64), 0, s, c2w, gt, parts, Q, k, out);
    return hipGetLastError();
}
