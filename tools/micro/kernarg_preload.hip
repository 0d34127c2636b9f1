// kernarg_preload.hip - what a kernel pays between its first instruction and its first global load, as a function of how its
// arguments arrive: (a) one by-value struct (every field behind an s_load from the kernarg segment, cold scalar cache),
// (b) leading pointer arguments, (c) the same with  -mllvm -amdgpu-kernarg-preload-count=N  (the dispatcher places the first
// N argument dwords in user SGPRs: no s_load before the first address is formed).  Build both ways and compare:
//   hipcc --offload-arch=gfx950 -O3 -o kp_plain kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o kp_preload kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args { const float4* p; float4* out; long long* stamps; int n; int pad[60]; };
#define STAMP() __builtin_readcyclecounter()
__global__ __launch_bounds__(256) void k_struct(Args a) {
    const long long t0 = STAMP();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v = a.p[i];
    const long long t1 = STAMP();
    v.x += 1.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = STAMP();
    a.out[i] = v;
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) { a.stamps[0] = t0; a.stamps[1] = t1; a.stamps[2] = t2; }
}
__global__ __launch_bounds__(256) void k_ptr(const float4* __restrict__ p, float4* __restrict__ out, long long* stamps, int n, Args rest) {
    const long long t0 = STAMP();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4 v = p[i];
    const long long t1 = STAMP();
    v.x += 1.f + rest.pad[7];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = STAMP();
    out[i] = v;
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t0; stamps[1] = t1; stamps[2] = t2; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float4 *a, *b;
    long long* stamps;
    CK(hipMalloc(&a, 16 << 20)); CK(hipMalloc(&b, 16 << 20)); CK(hipMalloc(&stamps, 64));
    CK(hipMemset(a, 0, 16 << 20)); CK(hipMemset(b, 0, 16 << 20));
    const int N = 4000;
    for (int grid : {1, 100, 256}) {
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            Args s{}; s.stamps = stamps; s.n = grid * 256;
            auto launch = [&](int it) {
                const float4* src = (it & 1) ? b : a;
                float4* dst = (it & 1) ? a : b;
                if (mode == 0) { s.p = src; s.out = dst; hipLaunchKernelGGL(k_struct, dim3(grid), dim3(256), 0, st, s); }
                else hipLaunchKernelGGL(k_ptr, dim3(grid), dim3(256), 0, st, src, dst, stamps, grid * 256, s);
            };
            for (int i = 0; i < 100; ++i) launch(i);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int i = 0; i < N; ++i) launch(i);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            long long h[3];
            CK(hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost));
            printf("grid %3d  %-22s %6.2f us per dependent launch   entry->load issued %5lld ticks   issued->data %5lld ticks\n", grid,
                   mode == 0 ? "by-value struct" : "leading pointer args", ms * 1e3f / N, h[1] - h[0], h[2] - h[1]);
        }
    }
    // the same chain replayed as a hipGraph (GPU-side boundary only)
    for (int mode = 0; mode < 2; ++mode) {
        const int grid = 100, G = 200;
        hipGraph_t g; hipGraphExec_t ge;
        Args s{}; s.stamps = nullptr; s.n = grid * 256;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int it = 0; it < G; ++it) {
            const float4* src = (it & 1) ? b : a;
            float4* dst = (it & 1) ? a : b;
            if (mode == 0) { s.p = src; s.out = dst; hipLaunchKernelGGL(k_struct, dim3(grid), dim3(256), 0, st, s); }
            else hipLaunchKernelGGL(k_ptr, dim3(grid), dim3(256), 0, st, src, dst, (long long*)nullptr, grid * 256, s);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        CK(hipGraphLaunch(ge, st)); hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, st));
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("hipGraph, grid 100  %-22s %6.2f us per node\n", mode == 0 ? "by-value struct" : "leading pointer args", ms * 1e3f / (20 * G));
    }
    return 0;
}
