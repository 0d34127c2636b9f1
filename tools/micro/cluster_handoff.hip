// Micro-benchmark (dev tool): the layer-to-layer hand-off of a PERSISTENT inner-level program (downs[3] + mid + ups[0]).
//
// Geometry of the real thing: an N-tile (4 trajectories x 8 positions) is owned by a CLUSTER of 8 workgroups, workgroup m computes
// output channels [32m, 32m+32) of every layer, so after each layer every workgroup needs the 4-KB tiles of its 7 peers.
// K is split over the 8 waves BY PRODUCER: wave w of every workgroup consumes the tile of peer w - eight independent 1-to-1
// hand-offs per workgroup and layer, no workgroup-wide wait, no grid/XCD barrier.
//
// Forms measured (MI355X_MICROARCH.md price list: handoff-1to1 / handoff-flag):
//   G  data-tagged granules: {tag = epoch, value} pairs, two per 16-B sc1 store / sc1 load; the data is the flag
//   F  sc1 payload (16-B stores) -> every storing wave drains vmcnt -> workgroup barrier -> one sc1 flag store;
//      the consumer polls the flag (relaxed, agent scope), then reads the payload with sc1 loads (no acquire fence)
// Placement: block = c*8 + m (members of a cluster on 8 different XCDs, the weights of channel tile m stay in ONE L2) or
//            block = m*NC + c ... remapped so that a cluster shares an XCD.
// Load: none, or a weight-streaming MFMA loop between hand-offs (what the real kernel does while it waits).
// Every word of every tile is checked in every layer; every spin is bounded (a give-up code ends the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_F = 1024;                 // floats per tile (32 positions x 32 channels)
constexpr int SPIN_MAX = 400000;             // ~0.1-0.3 s worst case, then give up

struct Args {
    unsigned* xg;        // granule exchange  [2][nwg][TILE_F] x 8 B
    float* xf;           // flag-form payload [2][nwg][TILE_F]
    unsigned* flags;     // [2][nwg]
    const float* w;      // weight stream (L2/MALL resident), >= 2 MB
    int* bad;            // mismatch count
    int* gaveup;         // spin give-ups
    long long* cyc;      // [nwg] kernel cycles
    int layers, nc, same_xcd, mfma_per_wave, epoch0;
};

__device__ __forceinline__ float expect_val(int layer, int prod, int i) { return (float)((layer * 131 + prod) * 1024 + i) * 0.25f; }

template <int FORM, bool LOAD>
__global__ __launch_bounds__(512) void handoff_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c, m;
    if (a.same_xcd) {   // cluster c lives on XCD c % 8 (block b runs on XCD b % 8): b = (c % 8) + 8 * (m + 8 * (c / 8))
        const int b = blockIdx.x, x = b & 7, r = b >> 3;
        m = r & 7; c = (r >> 3) * 8 + x;
        if (c >= a.nc) return;
    } else { c = blockIdx.x >> 3; m = blockIdx.x & 7; }
    const int me = c * 8 + m, nwg = a.nc * 8;
    const long long t0 = __builtin_readcyclecounter();
    if (tid < 256) smem[tid] = 0.f;   // the MFMA loop's B operand
    __syncthreads();
    auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.xg, 0, 0x7fffffff, 0x00020000);
    auto rf = __builtin_amdgcn_make_buffer_rsrc((void*)a.xf, 0, 0x7fffffff, 0x00020000);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    int wrong = 0, gave = 0;
    for (int layer = 0; layer < a.layers; ++layer) {
        const unsigned epoch = (unsigned)(a.epoch0 + layer + 1);
        const int par = layer & 1;
        // ---- "compute": stream weights, MFMA (keeps the CU's memory queue as busy as the real k-loop does)
        if (LOAD) {
            const float* wb = a.w + ((size_t)((layer * 8 + m) & 15) * 8 + wave) * 8192 + lane * 4;
            for (int g = 0; g < a.mfma_per_wave / 16; ++g) {
                const f32x4 a0 = *(const f32x4*)(wb + (g & 31) * 256), a1 = *(const f32x4*)(wb + (g & 31) * 256 + 128 * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], smem[lane + e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], smem[lane + e], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], smem[lane + 64 + e], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], smem[lane + 64 + e], acc[3], 0, 0, 0);
                }
            }
            __syncthreads();   // the real kernel reduces K-partials here
        }
        // ---- publish this workgroup's tile: waves 0..3 hold it as 4 floats per lane (the epilogue's register image)
        if (wave < 4) {
            const int i0 = (wave * 64 + lane) * 4;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = expect_val(layer, me, i0 + e) + (LOAD ? acc[e][0] * 0.f : 0.f);
            if (FORM == 0) {
                const unsigned off = (unsigned)(((size_t)par * nwg + me) * TILE_F + i0) * 8u;
                u32x4 g0 = {epoch, __float_as_uint(v[0]), epoch, __float_as_uint(v[1])};
                u32x4 g1 = {epoch, __float_as_uint(v[2]), epoch, __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(g0, rg, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(g1, rg, off + 16, 0, 16);
            } else {
                const unsigned off = (unsigned)(((size_t)par * nwg + me) * TILE_F + i0) * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rf, off, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (FORM == 1) {
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.flags + (size_t)par * nwg + me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- consume: wave w takes the tile of peer w of its cluster (its own tile too: same path, simplest to verify)
        const int prod = c * 8 + wave;
        float* win = smem + 256 + wave * TILE_F;   // wave-private LDS window
        if (FORM == 0) {
            const unsigned base = (unsigned)(((size_t)par * nwg + prod) * TILE_F) * 8u;
            u32x4 g[8];
            int spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    g[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, base + (unsigned)(k * 64 + lane) * 16u, 0, 16);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) ok &= (g[k][0] == epoch) & (g[k][2] == epoch);
                if (__all(ok)) break;
                if (++spins > SPIN_MAX) { gave = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = (k * 64 + lane) * 2;
                *(float2*)(win + i) = make_float2(__uint_as_float(g[k][1]), __uint_as_float(g[k][3]));
            }
        } else {
            int spins = 0;
            while (__hip_atomic_load(a.flags + (size_t)par * nwg + prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                if (++spins > SPIN_MAX) { gave = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            const unsigned base = (unsigned)(((size_t)par * nwg + prod) * TILE_F) * 4u;
            u32x4 g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = __builtin_amdgcn_raw_buffer_load_b128(rf, base + (unsigned)(k * 64 + lane) * 16u, 0, 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(u32x4*)(win + (k * 64 + lane) * 4) = g[k];
        }
        // ---- verify every word (through LDS, as the k-loop would read it)
        for (int i = lane; i < TILE_F; i += 64) wrong += (win[i] != expect_val(layer, prod, i));
        if (gave) break;
    }
    if (wrong) atomicAdd(a.bad, wrong);
    if (gave && lane == 0) atomicAdd(a.gaveup, 1);
    if (tid == 0) a.cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (LOAD && acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) a.bad[1] = 1;
}

template <int FORM, bool LOAD>
static int run(Args a, int mfma_per_wave, int same_xcd, const char* label) {
    a.mfma_per_wave = mfma_per_wave; a.same_xcd = same_xcd;
    const int nwg = same_xcd ? ((a.nc + 7) / 8) * 64 : a.nc * 8;
    const size_t lds = (256 + 8 * TILE_F) * sizeof(float);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f; int reps = 6, hbad = 0, hgave = 0;
    for (int rep = 0; rep < reps; ++rep) {
        a.epoch0 = 1000 * (rep + 1) + (FORM * 2 + (LOAD ? 1 : 0)) * 100000 + same_xcd * 1000000 + mfma_per_wave * 16;
        CK(hipMemsetAsync(a.bad, 0, 8, 0)); CK(hipMemsetAsync(a.gaveup, 0, 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((handoff_kernel<FORM, LOAD>), dim3(nwg), dim3(512), lds, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int b2[2], g; CK(hipMemcpy(b2, a.bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&g, a.gaveup, 4, hipMemcpyDeviceToHost));
        hbad += b2[0]; hgave += g;
        if (rep) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-34s form %c  %s  nc %2d  mfma/wave %4d : %6.2f us per layer (best %6.2f)  mismatches %d  give-ups %d\n", label, FORM ? 'F' : 'G',
           same_xcd ? "cluster-on-XCD " : "cluster-spread ", a.nc, mfma_per_wave, sum / (reps - 1) * 1000 / a.layers, best * 1000 / a.layers, hbad, hgave);
    return 0;
}


// ---- PING-PONG variant: the cluster's N-tile is split into two HALF tiles (2 trajectories each) that alternate; the granule
// sweep for the input of half-step s+1 (= output of half-step s-1, published by the peers one k-loop ago) is issued SPECULATIVELY
// right after the k-loop of half-step s, so that it flies under the K-reduction barrier + epilogue + publish of half-step s;
// the tags are checked at the start of half-step s+1 (re-sweep on a mismatch).  The layer's weights are streamed once per half.
constexpr int HALF_F = TILE_F / 2;
template <bool SPEC>
__global__ __launch_bounds__(512) void pingpong_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x >> 3, m = blockIdx.x & 7;
    const int me = c * 8 + m, nwg = a.nc * 8;
    if (tid < 256) smem[tid] = 0.f;
    __syncthreads();
    auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.xg, 0, 0x7fffffff, 0x00020000);
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    int wrong = 0, gave = 0;
    const int prod = c * 8 + wave;
    u32x4 g[4];
    // granule buffer of (half h, parity par, workgroup q): ((h*2 + par) * nwg + q) * HALF_F granules of 8 B
    auto gbase = [&](int h, int par, int q) { return (unsigned)((((size_t)h * 2 + par) * nwg + q) * HALF_F) * 8u; };
    auto sweep_issue = [&](int s) {   // loads of the output of half-step s
        const unsigned base = gbase(s & 1, (s >> 1) & 1, prod);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, base + (unsigned)(k * 64 + lane) * 16u, 0, 16);
    };
    const int nsteps = a.layers * 2;
    for (int s = 0; s < nsteps; ++s) {
        const int layer = s >> 1, h = s & 1;
        float* win = smem + 256 + (h * 8 + wave) * HALF_F;
        if (s >= 2) {
            const unsigned epoch = (unsigned)(a.epoch0 + (s - 2) + 1);
            if (!SPEC) sweep_issue(s - 2);
            int spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) ok &= (g[k][0] == epoch) & (g[k][2] == epoch);
                if (__all(ok)) break;
                if (++spins > SPIN_MAX) { gave = 1; break; }
                __builtin_amdgcn_s_sleep(1);
                sweep_issue(s - 2);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) *(float2*)(win + (k * 64 + lane) * 2) = make_float2(__uint_as_float(g[k][1]), __uint_as_float(g[k][3]));
            for (int i = lane; i < HALF_F; i += 64) wrong += (win[i] != expect_val(s - 2, prod, i));
            if (gave) break;
        }
        // ---- k-loop of this half-step: 2 A fragments (2 KiB) per 8 MFMAs per wave
        const float* wb = a.w + ((size_t)((layer * 8 + m) & 15) * 8 + wave) * 8192 + lane * 4;
        for (int gi = 0; gi < a.mfma_per_wave / 8; ++gi) {
            const f32x4 a0 = *(const f32x4*)(wb + (gi & 31) * 256), a1 = *(const f32x4*)(wb + (gi & 31) * 256 + 128 * 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], smem[lane + e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], smem[lane + e], acc[1], 0, 0, 0);
            }
        }
        if (SPEC && s >= 1 && s + 1 < nsteps) sweep_issue(s - 1);
        __syncthreads();   // K-partials
        if (wave < 2) {    // the half tile's 512 floats: 2 waves x 64 lanes x 4
            const unsigned epoch = (unsigned)(a.epoch0 + s + 1);
            const int i0 = (wave * 64 + lane) * 4;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = expect_val(s, me, i0 + e) + acc[e & 1][0] * 0.f;
            const unsigned off = gbase(h, layer & 1, me) + (unsigned)i0 * 8u;
            u32x4 g0 = {epoch, __float_as_uint(v[0]), epoch, __float_as_uint(v[1])};
            u32x4 g1 = {epoch, __float_as_uint(v[2]), epoch, __float_as_uint(v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(g0, rg, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(g1, rg, off + 16, 0, 16);
        }
    }
    if (wrong) atomicAdd(a.bad, wrong);
    if (gave && lane == 0) atomicAdd(a.gaveup, 1);
    if (acc[0][0] + acc[1][0] == 123.456f) a.bad[1] = 1;
}

template <bool SPEC>
static int run_pp(Args a, int mfma_per_wave) {
    a.mfma_per_wave = mfma_per_wave; a.same_xcd = 0;
    const size_t lds = (256 + 16 * HALF_F) * sizeof(float);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f; int reps = 6, hbad = 0, hgave = 0;
    for (int rep = 0; rep < reps; ++rep) {
        a.epoch0 = 50000000 + 1000 * (rep + 1) + (SPEC ? 1 : 0) * 100000 + mfma_per_wave * 16 + a.nc * 2000000;
        CK(hipMemsetAsync(a.bad, 0, 8, 0)); CK(hipMemsetAsync(a.gaveup, 0, 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((pingpong_kernel<SPEC>), dim3(a.nc * 8), dim3(512), lds, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int b2[2], g; CK(hipMemcpy(b2, a.bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&g, a.gaveup, 4, hipMemcpyDeviceToHost));
        hbad += b2[0]; hgave += g;
        if (rep) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("ping-pong halves, %s sweep    form G  cluster-spread   nc %2d  mfma/wave 2x%3d : %6.2f us per layer (best %6.2f)  mismatches %d  give-ups %d\n",
           SPEC ? "speculative" : "blocking   ", a.nc, mfma_per_wave, sum / (reps - 1) * 1000 / a.layers, best * 1000 / a.layers, hbad, hgave);
    return 0;
}

// reference: the same per-layer work as separate dependent launches (what the product does today)
__global__ __launch_bounds__(512) void layer_kernel(const Args a, int layer) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x >> 3, m = blockIdx.x & 7, me = blockIdx.x, nwg = gridDim.x, par = layer & 1;
    // stage the 8 tiles of the previous layer (plain loads: a kernel boundary made them visible)
    if (tid < 256) smem[tid] = 0.f;
    if (layer > 0) {
        const float* src = a.xf + ((size_t)(par ^ 1) * nwg + c * 8 + wave) * TILE_F;
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)(smem + 256 + wave * TILE_F + (k * 64 + lane) * 4) = *(const f32x4*)(src + (k * 64 + lane) * 4);
    }
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float* wb = a.w + ((size_t)((layer * 8 + m) & 15) * 8 + wave) * 8192 + lane * 4;
    for (int g = 0; g < a.mfma_per_wave / 16; ++g) {
        const f32x4 a0 = *(const f32x4*)(wb + (g & 31) * 256), a1 = *(const f32x4*)(wb + (g & 31) * 256 + 128 * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], smem[lane + e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], smem[lane + e], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], smem[lane + 64 + e], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], smem[lane + 64 + e], acc[3], 0, 0, 0);
        }
    }
    __syncthreads();
    if (wave < 4) {
        const int i0 = (wave * 64 + lane) * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = expect_val(layer, me, i0 + e) + acc[e][0] * 0.f;
        *(f32x4*)(a.xf + ((size_t)par * nwg + me) * TILE_F + i0) = v;
    }
}


// ---- EARLY-LAUNCH variant of the per-layer chain: consecutive launches ALTERNATE between two streams, so launch n+1 starts while launch n
// still runs (each stream keeps its own barrier semantics: at most two launches are resident), prefetches what does not depend on
// launch n (here: the weight fragments of its k-loop), then waits on DEVICE-SIDE arrival counters of launch n (one per XCD-slot, eight
// in all: a workgroup's lanes 0..7 poll them relaxed) and stages launch n's tiles with sc1 loads; every workgroup publishes its
// tile with sc1 (write-through) stores, drains, and arrives on counter[blockIdx % 8].  No host event between launches.
__global__ __launch_bounds__(512) void layer_early_kernel(const Args a, int layer, unsigned* counters /*[layers][8]*/, unsigned expect_per_slot, int* gaveup) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x >> 3, m = blockIdx.x & 7, me = blockIdx.x, nwg = gridDim.x, par = layer & 1;
    if (tid < 256) smem[tid] = 0.f;
    // (1) independent of the predecessor: this wave's weight fragments (the real kernels: first ring fill, parameters, halo zeros)
    const float* wb = a.w + ((size_t)((layer * 8 + m) & 15) * 8 + wave) * 8192 + lane * 4;
    f32x4 a0 = *(const f32x4*)(wb), a1 = *(const f32x4*)(wb + 128 * 64);
    // (2) wait for the predecessor launch: all of its workgroups have arrived
    if (layer > 0) {
        if (tid < 8) {
            const unsigned* cnt = counters + (size_t)(layer - 1) * 8 + tid;
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect_per_slot) {
                if (++spins > SPIN_MAX) { atomicAdd(gaveup, 1); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        auto rf = __builtin_amdgcn_make_buffer_rsrc((void*)a.xf, 0, 0x7fffffff, 0x00020000);
        const unsigned base = (unsigned)(((size_t)(par ^ 1) * nwg + c * 8 + wave) * TILE_F) * 4u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rf, base + (unsigned)(k * 64 + lane) * 16u, 0, 16);
            *(u32x4*)(smem + 256 + wave * TILE_F + (k * 64 + lane) * 4) = g;
        }
    }
    __syncthreads();
    int wrong = 0;
    if (layer > 0) for (int i = lane; i < TILE_F; i += 64) wrong += (smem[256 + wave * TILE_F + i] != expect_val(layer - 1, c * 8 + wave, i));
    if (wrong) atomicAdd(a.bad, wrong);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int g = 0; g < a.mfma_per_wave / 16; ++g) {
        if (g) { a0 = *(const f32x4*)(wb + (g & 31) * 256); a1 = *(const f32x4*)(wb + (g & 31) * 256 + 128 * 64); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], smem[lane + e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], smem[lane + e], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], smem[lane + 64 + e], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], smem[lane + 64 + e], acc[3], 0, 0, 0);
        }
    }
    __syncthreads();
    if (wave < 4) {
        const int i0 = (wave * 64 + lane) * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = expect_val(layer, me, i0 + e) + acc[e][0] * 0.f;
        auto rf = __builtin_amdgcn_make_buffer_rsrc((void*)a.xf, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rf, (unsigned)(((size_t)par * nwg + me) * TILE_F + i0) * 4u, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(counters + (size_t)layer * 8 + (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static int run_early(Args a, int mpw, int nstreams) {
    a.mfma_per_wave = mpw;
    const int layers = a.layers, nwg = a.nc * 8;
    unsigned* counters; int* gave;
    CK(hipMalloc(&counters, (size_t)layers * 8 * 4)); CK(hipMalloc(&gave, 4));
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    hipEvent_t e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
    float sum = 0.f; int hbad = 0, hg = 0;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(counters, 0, (size_t)layers * 8 * 4, st[0])); CK(hipMemsetAsync(a.bad, 0, 8, st[0])); CK(hipMemsetAsync(gave, 0, 4, st[0]));
        CK(hipEventRecord(e0, st[0]));
        CK(hipStreamWaitEvent(st[1], e0, 0));
        for (int l = 0; l < layers; ++l)
            hipLaunchKernelGGL(layer_early_kernel, dim3(nwg), dim3(512), (256 + 8 * TILE_F) * sizeof(float), st[nstreams == 2 ? (l & 1) : 0], a, l, counters,
                               (unsigned)(nwg / 8), gave);
        CK(hipEventRecord(ej, st[1])); CK(hipStreamWaitEvent(st[0], ej, 0));
        CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int b2[2], g; CK(hipMemcpy(b2, a.bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&g, gave, 4, hipMemcpyDeviceToHost));
        hbad += b2[0]; hg += g;
        if (rep) sum += ms;
    }
    printf("%-34s %d stream(s), device-side deps  nc %2d  mfma/wave %4d : %6.2f us per layer  mismatches %d  give-ups %d\n", "one launch per layer, EARLY launch", nstreams,
           a.nc, mpw, sum / 5 * 1000 / layers, hbad, hg);
    CK(hipFree(counters)); CK(hipFree(gave));
    return 0;
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 64;
    Args a; memset(&a, 0, sizeof(a));
    const int max_wg = 512;
    CK(hipMalloc(&a.xg, (size_t)2 * max_wg * TILE_F * 8)); CK(hipMemset(a.xg, 0, (size_t)2 * max_wg * TILE_F * 8));
    CK(hipMalloc(&a.xf, (size_t)2 * max_wg * TILE_F * 4)); CK(hipMemset(a.xf, 0, (size_t)2 * max_wg * TILE_F * 4));
    CK(hipMalloc(&a.flags, (size_t)2 * max_wg * 4)); CK(hipMemset(a.flags, 0, (size_t)2 * max_wg * 4));
    float* w; CK(hipMalloc(&w, (size_t)16 * 8 * 8192 * 4 + (1 << 20))); CK(hipMemset(w, 0, (size_t)16 * 8 * 8192 * 4 + (1 << 20))); a.w = w;
    CK(hipMalloc(&a.bad, 8)); CK(hipMalloc(&a.gaveup, 4)); CK(hipMalloc(&a.cyc, max_wg * 8));
    a.layers = layers;
    for (int nc : {25, 32}) {
        a.nc = nc;
        for (int sx : {0, 1}) {
            if (run<0, false>(a, 0, sx, "hand-off only")) return 1;
            if (run<1, false>(a, 0, sx, "hand-off only")) return 1;
            for (int mpw : {160, 640}) {
                if (run<0, true>(a, mpw, sx, "weight stream + MFMA + hand-off")) return 1;
                if (run<1, true>(a, mpw, sx, "weight stream + MFMA + hand-off")) return 1;
            }
        }
        for (int mpw : {80, 320}) {
            if (run_pp<false>(a, mpw)) return 1;
            if (run_pp<true>(a, mpw)) return 1;
        }
        // the same work as one launch per layer
        for (int mpw : {160, 640}) {
            a.mfma_per_wave = mpw;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float sum = 0.f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0, 0));
                for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(layer_kernel, dim3(nc * 8), dim3(512), (256 + 8 * TILE_F) * sizeof(float), 0, a, l);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) sum += ms;
            }
            printf("%-34s launches                 nc %2d  mfma/wave %4d : %6.2f us per layer\n", "one launch per layer (reference)", nc, mpw, sum / 5 * 1000 / layers);
            if (run_early(a, mpw, 1)) return 1;
            if (run_early(a, mpw, 2)) return 1;
        }
    }
    // MFMA loop alone (no hand-off, no launches): the compute floor of the emulated layer
    return 0;
}
