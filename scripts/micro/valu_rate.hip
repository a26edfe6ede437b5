// Dev micro-benchmark: issue rate of v_fma_f32 against hand-placed v_pk_fma_f32 / v_pk_mul_f32 on gfx950 (wave64), W waves per SIMD.
//   16 independent accumulator PAIRS per lane, so neither form is latency-bound.  Prints SIMD clocks per wave-level instruction.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o build/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FORM>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long* clk) {
    f32x2 acc[16];
    for (int c = 0; c < 16; ++c) { acc[c][0] = seed + c + threadIdx.x; acc[c][1] = seed - c; }
    f32x2 w; w[0] = 1.0f + seed * 1e-7f; w[1] = 1.0f - seed * 1e-7f;
    f32x2 b; b[0] = seed * 1e-3f; b[1] = -seed * 1e-3f;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (FORM == 0) {          // two scalar FMAs
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c][0]) : "v"(w[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c][1]) : "v"(w[1]), "v"(b[1]));
            } else if (FORM == 1) {   // one packed FMA
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(w), "v"(b));
            } else if (FORM == 2) {   // one packed MUL
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[c]) : "v"(w));
            } else {                  // packed FMA, weight broadcast from one register through op_sel
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(acc[c]) : "v"(w), "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int c = 0; c < 16; ++c) s += acc[c][0] + acc[c][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int FORM>
void run(const char* name, int waves_per_simd, float* out, unsigned long long* clk) {
    const int iters = 4000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FORM>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, clk);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<FORM>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, clk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double per_wave = (FORM == 0 ? 32.0 : 16.0) * iters;      // wave-level instructions per wave
    const double fma_lanes = 32.0 * iters * 64 * 4 * blocks;         // scalar-equivalent lane operations
    // s_memtime ticks at 100 MHz on this part; report the event time instead, in SIMD clocks at 2.4 GHz
    printf("%-22s waves/SIMD %d  %.3f ms  %.2f clk(2.4GHz)/wave-instr/SIMD  %.1f T lane-ops/s  (memtime %llu)\n", name, waves_per_simd, best,
           best * 1e-3 * 2.4e9 / (per_wave * waves_per_simd), fma_lanes / (best * 1e-3) * 1e-12, clk[0]);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long* clk; hipHostMalloc(&clk, 8);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        run<0>("2 x v_fma_f32", w, out, clk);
        run<1>("v_pk_fma_f32", w, out, clk);
        run<2>("v_pk_mul_f32", w, out, clk);
        run<3>("v_pk_fma_f32 op_sel", w, out, clk);
    }
    return 0;
}
