// Micro-benchmark (dev tool): cost of a hand-rolled device-wide barrier on MI355X, one 512-thread workgroup per CU,
// with a cross-workgroup data exchange between barriers (checks the release/acquire really makes other XCDs' writes visible).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int FENCE>
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& target, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nwg;
        if (FENCE) __threadfence();  // release: write back this XCD's dirty L2 lines
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        if (FENCE) __threadfence();  // acquire: invalidate L1 / non-local L2 lines
    }
    __syncthreads();
}

template <int FENCE>
__global__ __launch_bounds__(512) void bar_kernel(unsigned* ctr, float* buf, int iters, int words, int* bad, long long* cyc) {
    unsigned target = 0;
    const unsigned nwg = gridDim.x;
    const int b = blockIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        // write my slab of the ping-pong buffer
        float* dst = buf + (size_t)(it & 1) * nwg * words + (size_t)b * words;
        for (int i = threadIdx.x; i < words; i += blockDim.x) { if (FENCE) dst[i] = (float)(it * 1000 + b); else __builtin_nontemporal_store((float)(it * 1000 + b), dst + i); }
        if (!FENCE) __builtin_amdgcn_s_waitcnt(0); grid_barrier<FENCE>(ctr, target, nwg);
        // read the slab of a workgroup on another XCD
        const int nb = (b + 1 + (it % 7)) % nwg;
        const float* src = buf + (size_t)(it & 1) * nwg * words + (size_t)nb * words;
        int wrong = 0;
        for (int i = threadIdx.x; i < words; i += blockDim.x) wrong += ((FENCE ? src[i] : __builtin_nontemporal_load(src + i)) != (float)(it * 1000 + nb));
        if (wrong) atomicAdd(bad, wrong);
    }
    if (threadIdx.x == 0 && b == 0) *cyc = __builtin_readcyclecounter() - t0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d  clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    unsigned* ctr; float* buf; int* bad; long long* cyc;
    const int maxwords = 4096;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, (size_t)2 * 1024 * maxwords * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int fence : {1, 0}) for (int nwg : {100, 256}) for (int words : {0, 1024, 4096}) {
        int iters = 200;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(ctr, 0, 4)); CK(hipMemset(bad, 0, 4));
            void* args[] = {&ctr, &buf, &iters, &words, &bad, &cyc};
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel(fence ? (const void*)bar_kernel<1> : (const void*)bar_kernel<0>, dim3(nwg), dim3(512), args, 0, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int hbad; long long hc; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
            if (rep) printf("fence %d nwg %3d  slab %5d B : %.2f us per (write+barrier+read)   cycles/iter %lld   mismatches %d\n", fence, nwg, words * 4, ms * 1000 / iters, hc / iters, hbad);
        }
    }
    return 0;
}
