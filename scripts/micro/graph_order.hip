// Dev check: are consecutive kernel nodes of a captured hipGraph strictly ordered when several graphs replay concurrently on their
// own streams?  Kernel k of a chain: every workgroup first checks that ALL workgroups of kernel k-1 have arrived (counter[k-1] ==
// gridDim.x), waits a pseudo-random few microseconds, then arrives on counter[k].  Violations are counted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_step(unsigned* counters, unsigned* violations, int k, int chain) {
    if (threadIdx.x == 0) {
        if (k > 0 && __hip_atomic_load(&counters[k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gridDim.x) atomicAdd(violations, 1u);
        const unsigned spin = 20u + ((blockIdx.x * 2654435761u + k * 40503u) >> 26);     // 20 .. 83 x s_sleep(16)
        for (unsigned i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(16);
        atomicAdd(&counters[k], 1u);
    }
}
__global__ void k_zero(unsigned* counters, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) counters[i] = 0; }
int main() {
    const int NG = 4, CHAIN = 42, WGS = 192, ROUNDS = 200;
    hipStream_t st[NG]; hipGraph_t g[NG]; hipGraphExec_t ge[NG]; unsigned* cnt[NG]; unsigned* viol;
    hipMalloc(&viol, 4); hipMemset(viol, 0, 4);
    for (int i = 0; i < NG; ++i) {
        hipStreamCreate(&st[i]); hipMalloc(&cnt[i], CHAIN * 4);
        hipStreamBeginCapture(st[i], hipStreamCaptureModeGlobal);
        hipLaunchKernelGGL(k_zero, dim3(1), dim3(64), 0, st[i], cnt[i], CHAIN);
        for (int k = 0; k < CHAIN; ++k) hipLaunchKernelGGL(k_step, dim3(WGS), dim3(256), 12000, st[i], cnt[i], viol, k, CHAIN);
        hipStreamEndCapture(st[i], &g[i]);
        hipGraphInstantiate(&ge[i], g[i], nullptr, nullptr, 0);
    }
    for (int r = 0; r < ROUNDS; ++r)
        for (int i = 0; i < NG; ++i) hipGraphLaunch(ge[i], st[i]);
    hipDeviceSynchronize();
    unsigned v = 0; hipMemcpy(&v, viol, 4, hipMemcpyDeviceToHost);
    printf("graphs %d chain %d rounds %d: ordering violations %u\n", NG, CHAIN, ROUNDS, v);
    return v ? 1 : 0;
}
