// clock_calib.hip - what does s_memtime (__builtin_readcyclecounter) count, and what is the shader clock under MFMA load?
// Dev tool:  hipcc --offload-arch=gfx950 -O3 -o clock_calib clock_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void spin_mfma(long long* out, int iters, float* sink) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const float a = threadIdx.x * 1e-3f, b = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    const long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = c1 - c0; }
    if (acc0[0] + acc1[0] == 123.f) sink[0] = acc0[1];
}
int main() {
    long long* d; float* sink;
    hipMalloc(&d, 4096 * sizeof(long long)); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {100, 1000, 20000})
    for (int grid : {1, 100, 256}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin_mfma, dim3(grid), dim3(512), 0, 0, d, iters, sink);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            const double mfma_per_simd = (double)iters * 16 * 2;   // 2 waves per SIMD
            printf("iters %5d grid %3d: wall %8.1f us  s_memtime delta %10lld (%.1f ticks/us)  clock64 delta %10lld (%.1f /us)  -> %.1f cycles per MFMA at 2.4 GHz nominal, implied clock if 32 cyc/MFMA: %.2f GHz\n",
                   iters, grid, ms * 1e3, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), ms * 1e-3 * 2.4e9 / mfma_per_simd, mfma_per_simd * 32 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
