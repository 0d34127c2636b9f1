// launch_boundary.hip - cost of a dependent kernel boundary on one stream as a function of the launch shape
// (grid, block size, dynamic LDS, argument-block size).  Dev tool:  hipcc --offload-arch=gfx950 -O3 -o launch_boundary launch_boundary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args64 { float* p; int v[14]; };
struct Args2k { float* p; int v[480]; };
template <int T> __global__ __launch_bounds__(T) void k_small(Args64 a) {
    extern __shared__ float sm[];
    if (a.v[0] == 12345) { sm[threadIdx.x] = 1.f; a.p[blockIdx.x] = sm[0]; }
}
template <int T> __global__ __launch_bounds__(T) void k_big(Args2k a) {
    extern __shared__ float sm[];
    if (a.v[0] == 12345) { sm[threadIdx.x] = 1.f; a.p[blockIdx.x] = sm[a.v[479] & 7]; }
}
// a kernel that touches 16 B per thread (dependent data: forces the boundary's cache write-back / invalidate to matter)
template <int T> __global__ __launch_bounds__(T) void k_touch(Args64 a) {
    extern __shared__ float sm[];
    float4* p = (float4*)a.p;
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    float4 v = p[i];
    v.x += 1.f;
    p[i] = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename K, typename A> static float run(K kern, A a, int grid, int block, size_t lds, int n, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, a);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f / n;
}
int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float* buf;
    CK(hipMalloc(&buf, 64 << 20));
    CK(hipMemset(buf, 0, 64 << 20));
    Args64 a{}; a.p = buf;
    Args2k b{}; b.p = buf;
    CK(hipFuncSetAttribute((const void*)k_small<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_small<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_small<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_big<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_touch<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int N = 4000;
    printf("us per launch, %d back-to-back dependent launches on one stream (empty kernels unless noted)\n", N);
    for (int grid : {1, 100, 200, 256, 512, 1024}) {
        printf("grid %4d:", grid);
        printf("  64thr/0LDS %5.2f", run(k_small<64>, a, grid, 64, 0, N, st));
        printf("  256thr/0LDS %5.2f", run(k_small<256>, a, grid, 256, 0, N, st));
        printf("  512thr/0LDS %5.2f", run(k_small<512>, a, grid, 512, 0, N, st));
        printf("  512thr/48K %5.2f", run(k_small<512>, a, grid, 512, 48 << 10, N, st));
        printf("  512thr/96K %5.2f", run(k_small<512>, a, grid, 512, 96 << 10, N, st));
        printf("  512thr/150K %5.2f", run(k_small<512>, a, grid, 512, 150 << 10, N, st));
        printf("  512thr/96K/2KB-args %5.2f", run(k_big<512>, b, grid, 512, 96 << 10, N, st));
        printf("  512thr/96K/touch16B %5.2f\n", run(k_touch<512>, a, grid, 512, 96 << 10, N, st));
    }
    // hipGraph replay of the 200 x 512 / 96K shape
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_small<512>, dim3(200), dim3(512), 96 << 10, st, a);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st); CK(hipGraphLaunch(ge, st)); hipEventRecord(e1, st); CK(hipStreamSynchronize(st));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("hipGraph replay, 1000 x (200 WG x 512 thr, 96K LDS): %5.2f us per launch\n", ms);
    }
    return 0;
}
