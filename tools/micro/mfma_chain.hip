// mfma_chain.hip - issue rate of v_mfma_f32_16x16x4_f32 as a function of waves per SIMD and independent accumulator chains
// per wave (s_memtime ticks per MFMA per SIMD).  Dev tool:  hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NCH, int THREADS>
__global__ __launch_bounds__(THREADS) void spin(long long* out, int iters, float* sink) {
    f32x4 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = (f32x4){0, 0, 0, 0};
    const float a = threadIdx.x * 1e-3f, b = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NCH], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += acc[c][0];
    if (s == 123.f) sink[0] = s;
}
template <int NCH, int THREADS> void run(long long* d, float* sink, int grid) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((spin<NCH, THREADS>), dim3(grid), dim3(THREADS), 0, 0, d, iters, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((spin<NCH, THREADS>), dim3(grid), dim3(THREADS), 0, 0, d, iters, sink);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double per_wave = (double)iters * 16, waves_per_simd = THREADS / 256.0;
    printf("grid %3d  %d wave(s)/SIMD  %d chain(s): %6.1f ticks per MFMA of one wave, %6.1f ticks per MFMA per SIMD, %6.2f ns per MFMA per SIMD, %.0f ticks/us\n", grid,
           (int)waves_per_simd, NCH, h / per_wave, h / (per_wave * waves_per_simd), ms * 1e6 / (per_wave * waves_per_simd), h / (ms * 1e3));
}
int main() {
    long long* d; float* sink;
    hipMalloc(&d, 4096 * sizeof(long long)); hipMalloc(&sink, 64);
    for (int grid : {1, 100}) {
        run<1, 256>(d, sink, grid); run<2, 256>(d, sink, grid); run<4, 256>(d, sink, grid); run<8, 256>(d, sink, grid);
        run<1, 512>(d, sink, grid); run<2, 512>(d, sink, grid); run<4, 512>(d, sink, grid);
    }
    return 0;
}
