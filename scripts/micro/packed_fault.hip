// Stand-alone victim for the packed-fp32 fault of DESIGN.md section 4: the appearance loop of the fan march (csrc/fan_march_kernels.hip,
// phase C) on synthetic LDS contents, every sample evaluated TWICE and compared in the kernel.  One long-running 256-thread workgroup
// per CU leaves room for a workgroup of the encoder / logits kernel beside it; scripts/packed_fault_repro.py runs that kernel
// (k5_trunk_h of the product library: fp16 MFMA) on a second stream meanwhile and reads the mismatch counter.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC [-Xclang -target-feature -Xclang +packed-fp32-ops | -packed-fp32-ops]
//         scripts/micro/packed_fault.hip -o build/libpacked_fault_{on,off}.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32q __attribute__((ext_vector_type(4)));
constexpr int FP = 12, PLANE48 = FP * FP * 48, LINE48 = FP * 48, PATCH = PLANE48 + LINE48;

__device__ __forceinline__ f32q splat(float v) { return (f32q)(v); }
__device__ __forceinline__ f32q lerp_plane_q(f32q nw, f32q ne, f32q sw, f32q se, const float pw[4]) {
    f32q r = nw * splat(pw[0]);
    r = __builtin_elementwise_fma(ne, splat(pw[1]), r);
    r = __builtin_elementwise_fma(sw, splat(pw[2]), r);
    r = __builtin_elementwise_fma(se, splat(pw[3]), r);
    return r;
}
__device__ __forceinline__ f32q lerp_line_q(f32q lo, f32q hi, const float lw[2]) {
    f32q r = lo * splat(lw[0]);
    return __builtin_elementwise_fma(hi, splat(lw[1]), r);
}
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// the sample list of a quad: `n` samples drawn from a hash; taps inside the 12 x 12 patch; weights in (0, 1)
struct Smp { int ra, rb, rv, da, db, dv; float pw[4], lw[2], w; };
__device__ __forceinline__ Smp draw(uint32_t key) {
    const uint32_t h = mix(key), h2 = mix(h + 0x9e3779b9u);
    Smp s;
    s.ra = h % 11u; s.rb = (h >> 8) % 11u; s.rv = (h >> 16) % 11u;
    s.da = (h >> 24) & 1u; s.db = (h >> 25) & 1u; s.dv = (h >> 26) & 1u;
    const float fa = (h2 & 1023u) * (1.0f / 1024.0f), fb = ((h2 >> 10) & 1023u) * (1.0f / 1024.0f), fv = ((h2 >> 20) & 1023u) * (1.0f / 1024.0f);
    s.pw[0] = (1.0f - fb) * (1.0f - fa); s.pw[1] = (1.0f - fb) * fa; s.pw[2] = fb * (1.0f - fa); s.pw[3] = fb * fa;
    s.lw[0] = 1.0f - fv; s.lw[1] = fv;
    s.w = 0.25f + (h2 >> 30) * 0.125f;
    return s;
}

__global__ void __launch_bounds__(256, 3) victim(const float* __restrict__ pattern, int rounds, int samples, unsigned* __restrict__ mism,
                                                 float* __restrict__ sink) {
    __shared__ __align__(16) float s_patch[PATCH + 4320];            // 47 KB, the fan tile's pool: three such workgroups per CU at most
    const int tid = threadIdx.x, c = tid & 3, quad = tid >> 2;
    for (int i = tid; i < PATCH; i += 256) s_patch[i] = pattern[i];
    __syncthreads();
    float total = 0.0f;
    unsigned bad = 0u;
    for (int r = 0; r < rounds; ++r) {
        auto accumulate = [&](float (&acc)[12], int opaque) {
            // (as in the march: every quad has its own number of samples, so the loop runs under partial EXEC masks)
            const int n = samples / 2 + (int)(mix((blockIdx.x * 64u + quad) * 977u + r) % (unsigned)samples);
            for (int k = 0; k < n; ++k) {
                const Smp s = draw(((blockIdx.x * 64u + quad) * 4099u + r) * 131u + k);
                const float* P = s_patch + opaque + ((s.rb * FP + s.ra) * 48 + 4 * c);
                const int da = s.da * 48, db = s.db * (FP * 48);
                const float* L = s_patch + opaque + PLANE48 + (s.rv * 48 + 4 * c);
                const int dv = s.dv * 48;
#pragma unroll
                for (int j = 0; j < 3; ++j) {             // quarter c + 4 j of the 192-B texel
                    const f32q nw = *reinterpret_cast<const f32q*>(P + 16 * j), ne = *reinterpret_cast<const f32q*>(P + 16 * j + da);
                    const f32q sw = *reinterpret_cast<const f32q*>(P + 16 * j + db), se = *reinterpret_cast<const f32q*>(P + 16 * j + db + da);
                    const f32q ll = *reinterpret_cast<const f32q*>(L + 16 * j), lh = *reinterpret_cast<const f32q*>(L + 16 * j + dv);
                    const f32q pr = lerp_plane_q(nw, ne, sw, se, s.pw) * lerp_line_q(ll, lh, s.lw);
                    acc[4 * j + 0] = fmaf(s.w, pr.x, acc[4 * j + 0]);
                    acc[4 * j + 1] = fmaf(s.w, pr.y, acc[4 * j + 1]);
                    acc[4 * j + 2] = fmaf(s.w, pr.z, acc[4 * j + 2]);
                    acc[4 * j + 3] = fmaf(s.w, pr.w, acc[4 * j + 3]);
                }
            }
        };
        float a1[12], a2[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) a1[i] = a2[i] = 0.0f;
        int z1 = 0, z2 = 0;
        asm volatile("" : "+v"(z1));                     // two evaluations the compiler cannot merge
        accumulate(a1, z1);
        asm volatile("" : "+v"(z2));
        accumulate(a2, z2);
#pragma unroll
        for (int i = 0; i < 12; ++i) { bad += (__float_as_uint(a1[i]) != __float_as_uint(a2[i])) ? 1u : 0u; total += a1[i]; }
    }
    if (bad) atomicAdd(mism, bad);
    if (bad) atomicAdd(mism + 1 + (tid >> 4), 1u);       // which sixteen lanes of the workgroup (0..15): lanes 48-63 of wave w = slot 4 w + 3
    sink[blockIdx.x * 256 + tid] = total;
}

extern "C" int run_victim(void* stream, const float* pattern, int blocks, int rounds, int samples, unsigned* mism, float* sink) {
    hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pattern, rounds, samples, mism, sink);
    return (int)hipGetLastError();
}
