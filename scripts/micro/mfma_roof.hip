// Dev micro-benchmark: what the fp16 matrix pipe of one MI355X delivers on the trunk's instruction pattern.
//   chains of three DEPENDENT v_mfma_f32_32x32x16_f16 per accumulator (the hi/lo split product), NACC accumulators per wave,
//   W waves per SIMD, optionally V independent vector-ALU instructions after every MFMA (same wave).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_roof.hip -o build/mfma_roof ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int V>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 1e-3f); b[i] = (_Float16)(seed * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int c = 0; c < NACC; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < V; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
            }
        if (V > 0) {
#pragma unroll
            for (int p = 0; p < 3 * NACC; ++p) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
            }
        }
    }
    float s = 0.0f;
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = __builtin_amdgcn_s_memtime() - t0;
}

template <int NACC, int V>
void run(const char* name, int waves_per_simd, float* out) {
    static unsigned long long* clk = nullptr;
    if (!clk) hipHostMalloc(&clk, 8);
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;           // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, V>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, clk);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NACC, V>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, clk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipDeviceSynchronize();
    const double ghz = (double)clk[0] / (best * 1e-3) * 1e-9;
    const double flop = (double)blocks * 4 * iters * 3 * NACC * 32768.0;
    const double clk_per_mfma = best * 1e-3 * 2.4e9 / ((double)iters * 3 * NACC * waves_per_simd);
    printf("%-28s waves/SIMD %d  %.3f ms  %.0f TFLOP/s  (%.1f clk @2.4GHz per MFMA per SIMD)  s_memtime/wall = %.2f GHz\n", name, waves_per_simd, best, flop / best * 1e-9, clk_per_mfma, ghz);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<2, 0>("2 acc x3 dep, no valu", w, out);
        run<4, 0>("4 acc x3 dep, no valu", w, out);
        run<2, 4>("2 acc, 4 valu/mfma", w, out);
        run<2, 7>("2 acc, 7 valu/mfma", w, out);
        run<2, 10>("2 acc, 10 valu/mfma", w, out);
    }
    return 0;
}
