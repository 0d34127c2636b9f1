// valu_next_to_mfma.hip - how fast does a chain of dependent VALU instructions advance on a SIMD whose OTHER wave streams MFMAs?
// One workgroup of 8 waves per CU (2 per SIMD): waves 0..3 run the VALU chain, waves 4..7 stream MFMAs (or idle).  Dev tool:
//   hipcc --offload-arch=gfx950 -O3 -o valu_next_to_mfma valu_next_to_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE: 0 partner idle, 1 partner streams v_mfma_f32_16x16x4_f32, 2 partner streams v_mfma_f32_32x32x2_f32; PRIO: s_setprio of the VALU wave
// NCH: independent VALU chains interleaved in the VALU wave (1 = every instruction depends on the previous one)
template <int MODE, int PRIO, int NCH>
__global__ __launch_bounds__(512) void probe(long long* out, int n_valu, int n_mfma, float* sink) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(PRIO);
        float v[NCH];
        const float w = 1.0001f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = threadIdx.x * 1e-3f + c;
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u % NCH] = __builtin_fmaf(v[u % NCH], w, 0.25f);
        }
        const long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
        float sv = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) sv += v[c];
        if (sv == 123.f) sink[0] = sv;
    } else {
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        float s = 0.f;
        if (MODE == 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            const float x = threadIdx.x * 1e-3f;
            for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 1.0f, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 1.0f, a1, 0, 0, 0); }
            }
            s = a0[0] + a1[0];
        } else if (MODE >= 3) {   // 16x16x4 stream that steps aside: one `s_nop MODE - 3` (4 cycles per wait state) behind every MFMA (32 cycles)
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            const float x = threadIdx.x * 1e-3f;
            for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 1.0f, a0, 0, 0, 0);
                    asm volatile("s_nop %0" :: "n"(MODE - 3));
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 1.0f, a1, 0, 0, 0);
                    asm volatile("s_nop %0" :: "n"(MODE - 3));
                }
            }
            s = a0[0] + a1[0];
        } else if (MODE == 2) {
            f32x16 a0 = {}, a1 = {};
            const float x = threadIdx.x * 1e-3f;
            for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, a1, 0, 0, 0); }
            }
            s = a0[0] + a1[0];
        }
        const long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
        if (s == 123.f) sink[0] = s;
    }
}
template <int MODE, int PRIO, int NCH = 1> void run(const char* what, long long* d, float* sink) {
    const int n_valu = 2000, n_mfma = 4000;   // 32 000 dependent FMAs; 64 000 x 16x16x4 (or 32 000 x 32x32x2): the partner outlasts the chain
    hipLaunchKernelGGL((probe<MODE, PRIO, NCH>), dim3(256), dim3(512), 0, 0, d, n_valu, n_mfma, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, PRIO, NCH>), dim3(256), dim3(512), 0, 0, d, n_valu, n_mfma, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-46s prio %d chains %d : VALU %6.2f cycles per instruction (wave 0)   partner wave 4: %8lld cycles   kernel %.1f us\n", what, PRIO, NCH, (double)h[0] / (n_valu * 16.0), h[4], ms * 1e3);
}
int main() {
    long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * sizeof(long long)); hipMalloc(&sink, 4);
    run<0, 0>("partner idle", d, sink);
    run<1, 0>("partner streams v_mfma_f32_16x16x4_f32", d, sink);
    run<1, 3>("partner streams v_mfma_f32_16x16x4_f32", d, sink);
    run<2, 0>("partner streams v_mfma_f32_32x32x2_f32", d, sink);
    run<2, 3>("partner streams v_mfma_f32_32x32x2_f32", d, sink);
    run<0, 0, 4>("partner idle", d, sink);
    run<1, 0, 4>("partner streams v_mfma_f32_16x16x4_f32", d, sink);
    run<1, 0, 8>("partner streams v_mfma_f32_16x16x4_f32", d, sink);
    run<4, 0, 1>("partner: 16x16x4, s_nop 1 behind every MFMA", d, sink);
    run<6, 0, 1>("partner: 16x16x4, s_nop 3 behind every MFMA", d, sink);
    run<7, 0, 1>("partner: 16x16x4, s_nop 4 behind every MFMA", d, sink);
    run<8, 0, 1>("partner: 16x16x4, s_nop 5 behind every MFMA", d, sink);
    run<9, 0, 1>("partner: 16x16x4, s_nop 6 behind every MFMA", d, sink);
    run<10, 0, 1>("partner: 16x16x4, s_nop 7 behind every MFMA", d, sink);
    run<8, 0, 4>("partner: 16x16x4, s_nop 5 behind every MFMA", d, sink);
    return 0;
}
