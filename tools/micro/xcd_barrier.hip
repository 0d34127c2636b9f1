// Micro-benchmark (dev tool): XCD-LOCAL barrier + data exchange between the workgroups that landed on the same XCD
// (same L2).  Question: can a persistent kernel whose layer-to-layer dependencies stay inside one XCD synchronise more
// cheaply than a kernel boundary (~3 us), and which fences does the exchange need to be correct?
//   mode 0: plain stores / plain loads, no fences (expected: stale L1 reads)
//   mode 1: acquire-only  (agent-scope acquire fence after the barrier: L1 invalidate; stores already sit in the shared L2)
//   mode 2: release + acquire at agent scope (what a chip-wide exchange needs: L2 write-back + invalidate)
//   mode 3: plain stores, loads that bypass L1 (__builtin_nontemporal_load), no fences
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(512) void xcd_kernel(unsigned* reg_cnt /*[8]*/, unsigned* bar /*[8]*/, unsigned* all_in, int* slot_of /*[grid]*/,
                                                  int* members /*[8][64]*/, float* buf, int iters, int words, int* bad, long long* cyc, int* hist) {
    __shared__ unsigned s_x, s_slot, s_n;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        const unsigned slot = atomicAdd(reg_cnt + x, 1u);
        members[x * 64 + slot] = b;
        __threadfence();
        atomicAdd(all_in, 1u);
        while (__hip_atomic_load(all_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        __threadfence();
        s_x = x; s_slot = slot; s_n = __hip_atomic_load(reg_cnt + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot == 0) hist[x] = (int)s_n;
    }
    __syncthreads();
    const unsigned x = s_x, slot = s_slot, n = s_n;
    unsigned target = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float* dst = buf + ((size_t)(it & 1) * gridDim.x + b) * words;
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = (float)(it * 1000 + b);
        // ---- XCD-local barrier
        __syncthreads();   // this workgroup's stores are issued and acknowledged (s_waitcnt vmcnt(0) precedes the barrier)
        if (threadIdx.x == 0) {
            target += n;
            if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(bar + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            if (MODE == 1 || MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // ---- read the slab of another member of the same XCD
        const int nb = members[x * 64 + (slot + 1 + (it % 5)) % n];
        const float* src = buf + ((size_t)(it & 1) * gridDim.x + nb) * words;
        int wrong = 0;
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            const float v = (MODE == 3) ? __builtin_nontemporal_load(src + i) : src[i];
            wrong += (v != (float)(it * 1000 + nb));
        }
        if (wrong) atomicAdd(bad, wrong);
    }
    if (threadIdx.x == 0 && b == 0) *cyc = __builtin_readcyclecounter() - t0;
}

template <int MODE>
int run(int nwg, int words, unsigned* reg, unsigned* bar, unsigned* all_in, int* slot_of, int* members, float* buf, int* bad, long long* cyc, int* hist) {
    int iters = 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(reg, 0, 32)); CK(hipMemset(bar, 0, 32)); CK(hipMemset(all_in, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(hist, 0, 32));
        void* args[] = {&reg, &bar, &all_in, &slot_of, &members, &buf, &iters, &words, &bad, &cyc, &hist};
        CK(hipEventRecord(e0, 0));
        CK(hipLaunchCooperativeKernel((const void*)xcd_kernel<MODE>, dim3(nwg), dim3(512), args, 0, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hbad, hh[8]; long long hc;
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost));
        if (rep) printf("mode %d  nwg %3d  slab %5d B : %.2f us per (write + XCD barrier + read)  mismatches %d   workgroups per XCD: %d %d %d %d %d %d %d %d\n",
                        MODE, nwg, words * 4, ms * 1000 / iters, hbad, hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6], hh[7]);
    }
    return 0;
}

int main() {
    unsigned *reg, *bar, *all_in; int *slot_of, *members, *bad, *hist; float* buf; long long* cyc;
    CK(hipMalloc(&reg, 32)); CK(hipMalloc(&bar, 32)); CK(hipMalloc(&all_in, 4)); CK(hipMalloc(&slot_of, 1024 * 4)); CK(hipMalloc(&members, 8 * 64 * 4));
    CK(hipMalloc(&buf, (size_t)2 * 256 * 8192 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&hist, 32));
    for (int nwg : {200, 256}) for (int words : {1024, 8192}) {
        if (run<0>(nwg, words, reg, bar, all_in, slot_of, members, buf, bad, cyc, hist)) return 1;
        if (run<1>(nwg, words, reg, bar, all_in, slot_of, members, buf, bad, cyc, hist)) return 1;
        if (run<2>(nwg, words, reg, bar, all_in, slot_of, members, buf, bad, cyc, hist)) return 1;
        if (run<3>(nwg, words, reg, bar, all_in, slot_of, members, buf, bad, cyc, hist)) return 1;
    }
    return 0;
}
