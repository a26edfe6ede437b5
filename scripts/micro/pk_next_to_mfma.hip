// Dev micro-benchmark for DESIGN.md section 4, lesson 9: does a packed fp32 tap combination (v_pk_mul_f32 / v_pk_fma_f32 with the
// weight broadcast by op_sel, operands straight from ds_read_b128) return the bits of its one-instruction-per-component form when a
// workgroup of fp16-MFMA waves shares the CU?  Two kernels on two streams:
//   victim   256 threads, 47 KB of LDS (three per CU, like the fan kernel): every lane walks `trips` samples -- the lanes of the
//            first three quarters of a wave stop early when `ragged` (a divergent loop: the last quarter runs on alone) -- reads
//            six 16-B taps per sample from LDS, combines them packed AND per component, counts the lanes whose two sums differ;
//   neighbour 512 threads, 67.5 KB of LDS (two per CU, like k5_trunk_h): chains of v_mfma_f32_32x32x16_f16 with LDS traffic.
// Prints the number of differing lanes with and without the neighbour.
// RESULT (one MI355X): 0 differing lanes in all four cases (60 launches of 19 000 workgroups each) -- this sequence alone does NOT
// reproduce the fault the fan kernel shows once in ~30 launches; the trigger is narrower than "packed fp32 next to fp16 MFMA".
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/pk_next_to_mfma.hip -o /tmp/pk_next_to_mfma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32q __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float opq(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ f32q splat(float v) { return (f32q)(v); }
__device__ __forceinline__ f32q plane_pk(f32q nw, f32q ne, f32q sw, f32q se, const float pw[4]) {
    f32q r = nw * splat(pw[0]);
    r = __builtin_elementwise_fma(ne, splat(pw[1]), r);
    r = __builtin_elementwise_fma(sw, splat(pw[2]), r);
    r = __builtin_elementwise_fma(se, splat(pw[3]), r);
    return r;
}
__device__ __forceinline__ f32q line_pk(f32q lo, f32q hi, const float lw[2]) {
    f32q r = lo * splat(lw[0]);
    return __builtin_elementwise_fma(hi, splat(lw[1]), r);
}
__device__ __forceinline__ f32q plane_sc(f32q nw, f32q ne, f32q sw, f32q se, const float pw[4]) {
    f32q r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = opq(fmaf(se[e], pw[3], opq(fmaf(sw[e], pw[2], opq(fmaf(ne[e], pw[1], opq(nw[e] * pw[0])))))));
    return r;
}
__device__ __forceinline__ f32q line_sc(f32q lo, f32q hi, const float lw[2]) {
    f32q r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = opq(fmaf(hi[e], lw[1], opq(lo[e] * lw[0])));
    return r;
}

constexpr int PATCH = 12 * 12 * 48 + 12 * 48;          // floats: one appearance patch + its line, as in the fan kernel

__global__ void __launch_bounds__(256, 3) victim(const float* __restrict__ src, int trips, int ragged, unsigned long long* bad) {
    __shared__ __align__(16) float s_patch[PATCH + 4320];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < PATCH + 4320; i += 256) s_patch[i] = src[(i + 977 * blockIdx.x) % (1 << 20)];
    __syncthreads();
    const int c = tid & 3;
    unsigned h = 2654435761u * (tid + 256 * blockIdx.x + 1);
    float acc_p[12], acc_s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc_p[i] = acc_s[i] = 0.0f;
    const int mine = (ragged && lane < 48) ? trips / 3 + (lane & 7) : trips;          // the last quarter of a wave runs on alone
    for (int t = 0; t < mine; ++t) {
        h = h * 1664525u + 1013904223u;
        const int ra = (h >> 8) % 11, rb = (h >> 12) % 11, rv = (h >> 16) % 11;
        const float fa = (float)((h >> 20) & 255) / 256.0f, fb = (float)((h >> 4) & 255) / 256.0f, fv = (float)(h & 255) / 256.0f;
        const float pw[4] = {(1 - fb) * (1 - fa), (1 - fb) * fa, fb * (1 - fa), fb * fa};
        const float lw[2] = {1 - fv, fv};
        const float w = 0.01f + fa * fb;
        const float* P = s_patch + ((rb * 12 + ra) * 48 + 4 * c);
        const float* L = s_patch + 12 * 12 * 48 + (rv * 48 + 4 * c);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32q nw = *reinterpret_cast<const f32q*>(P + 16 * j), ne = *reinterpret_cast<const f32q*>(P + 16 * j + 48);
            const f32q sw = *reinterpret_cast<const f32q*>(P + 16 * j + 576), se = *reinterpret_cast<const f32q*>(P + 16 * j + 624);
            const f32q ll = *reinterpret_cast<const f32q*>(L + 16 * j), lh = *reinterpret_cast<const f32q*>(L + 16 * j + 48);
            const f32q pp = plane_pk(nw, ne, sw, se, pw) * line_pk(ll, lh, lw);
            const f32q ps = plane_sc(nw, ne, sw, se, pw), ls = line_sc(ll, lh, lw);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc_p[4 * j + e] = fmaf(w, pp[e], acc_p[4 * j + e]);
                acc_s[4 * j + e] = opq(fmaf(w, opq(ps[e] * ls[e]), acc_s[4 * j + e]));
            }
        }
    }
    bool differ = false;
#pragma unroll
    for (int i = 0; i < 12; ++i) differ = differ || (__float_as_uint(acc_p[i]) != __float_as_uint(acc_s[i]));
    if (differ) atomicAdd(bad, 1ull);
}

__global__ void __launch_bounds__(512, 4) neighbour(float* out, int iters, float seed) {
    __shared__ __align__(16) _Float16 s_planes[2 * 64 * 264];          // 67.5 KB
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * 64 * 264; i += 512) s_planes[i] = (_Float16)(seed + (i & 255) * 1e-3f);
    __syncthreads();
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(seed * 0.5f + i);
    for (int it = 0; it < iters; ++it) {
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(&s_planes[((tid * 8) + 264 * (it & 31)) % (2 * 64 * 264 - 8) & ~7]);
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(&s_planes[((tid * 8) + 264 * ((it + 7) & 31)) % (2 * 64 * 264 - 8) & ~7]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc[1], 0, 0, 0);
        }
        s_planes[(tid + 512 * (it & 15)) % (2 * 64 * 264)] = (_Float16)acc[0][it & 15];          // 2-byte LDS stores, as the trunk's splits
    }
    float s = 0.0f;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 512 + tid] = s;
}

int main() {
    float *src, *out;
    unsigned long long* bad;
    hipMalloc(&src, (1 << 20) * 4); hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&bad, 8);
    std::vector<float> h(1 << 20);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (float)((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipStream_t sv, sn;
    hipStreamCreate(&sv); hipStreamCreate(&sn);
    for (int ragged = 0; ragged < 2; ++ragged)
        for (int with = 0; with < 2; ++with) {
            hipMemset(bad, 0, 8);
            hipDeviceSynchronize();
            const int rounds = 60;
            for (int r = 0; r < rounds; ++r) {
                if (with) hipLaunchKernelGGL(neighbour, dim3(512), dim3(512), 0, sn, out, 1500, 0.25f + r);
                hipLaunchKernelGGL(victim, dim3(19000), dim3(256), 0, sv, src, 54, ragged, bad);
            }
            hipDeviceSynchronize();
            unsigned long long n = 0;
            hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost);
            printf("ragged loop %d, fp16-MFMA neighbour %d: %llu lanes differ in %d launches of 19000 x 256 lanes (%s)\n", ragged, with, n,
                   rounds, hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
