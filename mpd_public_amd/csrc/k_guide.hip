// k_guide.hip - the cost-guidance and trajectory-metrics kernels (guide.hpp) and their C-ABI entry points.
#include "host.hpp"
#include "guide.hpp"

namespace mpdx {

static long long* g_guide_trace = nullptr;  // dev tool (mpdx_guide_trace)

// per-context max|x| (the range test of LimitsNormalizer.unnormalize) for the API path
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, uint32_t* out, size_t per_ctx, int n_ctx) {
    const int ctx = blockIdx.y;
    const float* p = x + (size_t)ctx * per_ctx;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_ctx; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out + ctx, __float_as_uint(m));
}

int launch_guide(const mpdx_guide_params* gp, float* x, float* grad_out, const float* hs, const float* hg,
                        const uint32_t* amax_in, uint32_t* amax_out, int n_per_ctx, int B, int H, int D, hipStream_t st,
                        const float* noise, float noise_scale, float noise_extra, float* chain, float guide_scale, const NoiseRng* rng) {
    if (!gp || !x || !amax_in) return fail(MPDX_E_INVALID, "null argument");
    if (H > 128 || H < 2) return fail(MPDX_E_INVALID, "guide kernel: one support point per lane of one or two waves: H=%d unsupported (max 128)", H);
    if (D != 2 * gp->q_dim || D > 16) return fail(MPDX_E_INVALID, "state dim %d != 2*q_dim (%d)", D, gp->q_dim);
    if (gp->n_fields < 0 || gp->n_fields > MPDX_MAX_FIELDS) return fail(MPDX_E_INVALID, "n_fields %d", gp->n_fields);
    if (gp->interpolate && (gp->n_interp < H || gp->n_interp > 8 * H)) return fail(MPDX_E_INVALID, "n_interp %d unsupported", gp->n_interp);
    if (gp->n_prim_floats > 0 && !gp->prims) return fail(MPDX_E_INVALID, "primitive table missing");
    if (gp->robot == MPDX_ROBOT_PANDA && (((uintptr_t)x & 15) || ((size_t)H * D) % 4))
        return fail(MPDX_E_INVALID, "Panda guide: x must be 16-byte aligned with H * D a multiple of 4 (the trajectory is staged with 16-byte loads)");
    GuideArgs a;
    a.gp = *gp; a.x = x; a.grad_out = grad_out; a.hs = hs; a.hg = hg; a.amax_in = amax_in; a.amax_out = amax_out;
    a.B = B; a.H = H; a.D = D; a.n_per_ctx = n_per_ctx > 0 ? n_per_ctx : B;
    a.noise = noise; a.noise_scale = noise_scale; a.noise_extra = noise_extra; a.chain = chain;
    a.guide_scale = guide_scale;
    memset(&a.rng, 0, sizeof(a.rng));
    if (rng) a.rng = *rng;
    if (gp->clip_grad && gp->clip_rule != 0 && gp->clip_rule != 1) return fail(MPDX_E_INVALID, "clip_rule %d (0 = 'norm', 1 = 'value')", gp->clip_rule);
    a.trace = g_guide_trace;
    // Panda at large batch: the dense variant (no FK table, 128 VGPRs: two workgroups per CU); MPDX_GUIDE_DENSE=0/1 forces it off / on
    static const int dense_env = getenv("MPDX_GUIDE_DENSE") ? atoi(getenv("MPDX_GUIDE_DENSE")) : -1;
    const bool dense = gp->robot == MPDX_ROBOT_PANDA && (dense_env >= 0 ? dense_env != 0 : B >= 512) && guide_lds_bytes(*gp, H, D, true) <= 80 * 1024;
    const size_t lds = guide_lds_bytes(*gp, H, D, dense);
    if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "guide needs %zu B of LDS (n_interp %d too large)", lds, gp->n_interp);
    if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 2 && gp->ws_dim == 2)
        hipLaunchKernelGGL((guide_step_kernel<2, 2, MPDX_ROBOT_POINTMASS, 8>), dim3(B), dim3(512), lds, st, a);
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 3 && gp->ws_dim == 3)
        hipLaunchKernelGGL((guide_step_kernel<3, 3, MPDX_ROBOT_POINTMASS, 8>), dim3(B), dim3(512), lds, st, a);
    else if (gp->robot == MPDX_ROBOT_PANDA && gp->q_dim == 7 && gp->ws_dim == 3) {
        if (dense) {
            if (int rc = raise_lds_limit((const void*)guide_step_panda_kernel<true>)) return rc;
            hipLaunchKernelGGL(guide_step_panda_kernel<true>, dim3(B), dim3(512), lds, st, a);
        } else {
            if (int rc = raise_lds_limit((const void*)guide_step_panda_kernel<false>)) return rc;
            hipLaunchKernelGGL(guide_step_panda_kernel<false>, dim3(B), dim3(512), lds, st, a);
        }
    }
    else
        return fail(MPDX_E_INVALID, "unsupported robot %d / q_dim %d / ws_dim %d", gp->robot, gp->q_dim, gp->ws_dim);
    return 0;
}

}  // namespace mpdx

using namespace mpdx;

extern "C" {

int mpdx_guide_step(const mpdx_guide_params* gp, float* x, float* grad_out, const float* hard_start, const float* hard_goal,
                    const uint32_t* absmax_in, uint32_t* absmax_out, int n_per_ctx, int B, int H, int D, void* stream) {
    if (int rc = launch_guide(gp, x, grad_out, hard_start, hard_goal, absmax_in, absmax_out, n_per_ctx, B, H, D, (hipStream_t)stream)) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_guide_step_scaled(const mpdx_guide_params* gp, float* x, float* grad_out, const float* hard_start, const float* hard_goal,
                           const uint32_t* absmax_in, uint32_t* absmax_out, int n_per_ctx, int B, int H, int D, float guide_scale, void* stream) {
    if (int rc = launch_guide(gp, x, grad_out, hard_start, hard_goal, absmax_in, absmax_out, n_per_ctx, B, H, D, (hipStream_t)stream, nullptr, 0.f,
                              0.f, nullptr, guide_scale))
        return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_traj_metrics(const mpdx_guide_params* gp, const float* x_unnormalised, float* out4, int n_check, int B, int H, int D, void* stream) {
    return mpdx_traj_metrics_mask(gp, x_unnormalised, out4, nullptr, n_check, B, H, D, stream);
}

int mpdx_traj_metrics_mask(const mpdx_guide_params* gp, const float* x_unnormalised, float* out4, uint8_t* mask, int n_check, int B, int H, int D,
                           void* stream) {
    if (!gp || !x_unnormalised || !out4 || B <= 0) return fail(MPDX_E_INVALID, "bad argument");
    if (H > 128 || H < 2) return fail(MPDX_E_INVALID, "H=%d unsupported (max 128)", H);
    if (D != 2 * gp->q_dim || D > 16) return fail(MPDX_E_INVALID, "state dim %d != 2*q_dim (%d)", D, gp->q_dim);
    if (n_check < 2) n_check = H;
    const size_t lds = (size_t)(H * D + gp->n_prim_floats) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 2 && gp->ws_dim == 2)
        hipLaunchKernelGGL((traj_metrics_kernel<2, 2, MPDX_ROBOT_POINTMASS>), dim3(B), dim3(64), lds, st, *gp, x_unnormalised, out4, B, H, n_check, mask);
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 3 && gp->ws_dim == 3)
        hipLaunchKernelGGL((traj_metrics_kernel<3, 3, MPDX_ROBOT_POINTMASS>), dim3(B), dim3(64), lds, st, *gp, x_unnormalised, out4, B, H, n_check, mask);
    else if (gp->robot == MPDX_ROBOT_PANDA && gp->q_dim == 7 && gp->ws_dim == 3)
        hipLaunchKernelGGL((traj_metrics_kernel<7, 3, MPDX_ROBOT_PANDA>), dim3(B), dim3(64), lds, st, *gp, x_unnormalised, out4, B, H, n_check, mask);
    else
        return fail(MPDX_E_INVALID, "unsupported robot %d / q_dim %d / ws_dim %d", gp->robot, gp->q_dim, gp->ws_dim);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_guide_time(const mpdx_guide_params* gp, float* x, float* grad_out, const uint32_t* absmax_in, int n_per_ctx, int B, int H, int D,
                    int reps, void* stream, float* ms_avg) {
    if (!gp || !x || !grad_out || !absmax_in || !ms_avg || reps < 1) return fail(MPDX_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = 0;
    for (int i = 0; i < 3 && !rc; ++i) rc = launch_guide(gp, x, grad_out, nullptr, nullptr, absmax_in, nullptr, n_per_ctx, B, H, D, st);
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < reps && !rc; ++i) rc = launch_guide(gp, x, grad_out, nullptr, nullptr, absmax_in, nullptr, n_per_ctx, B, H, D, st);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    if (!rc) HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_avg = ms / (float)reps;
    return rc;
}

/* dev tool: one guide launch with s_memtime stamps (16 slots per wave, 8 waves -> 128 values) of workgroup 0 */
int mpdx_guide_trace(const mpdx_guide_params* gp, float* x, const uint32_t* absmax_in, int B, int H, int D, void* stream, long long* stamps64) {
#ifndef MPDX_DEV_HOOKS
    return fail(MPDX_E_STATE, "%s needs a development build of libmpdx.so (MPDX_BUILD_DEFS=-DMPDX_DEV_HOOKS MPDX_BUILD_OUT=build_ab/libmpdx_dev.so python -m mpd_public_amd.build, then MPDX_LIB=build_ab/libmpdx_dev.so): "
                "the production kernels carry no trace / ablation hooks", __func__);
#endif
    if (!stamps64) return fail(MPDX_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    long long* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, 128 * sizeof(long long)));
    HIP_TRY(hipMemsetAsync(dev, 0, 128 * sizeof(long long), st));
    static float* scratch = nullptr;
    static size_t scratch_n = 0;
    const size_t need = (size_t)B * H * D;
    if (scratch_n < need) { if (scratch) (void)hipFree(scratch); HIP_TRY(hipMalloc(&scratch, need * sizeof(float))); scratch_n = need; }
    g_guide_trace = dev;
    int rc = launch_guide(gp, x, scratch, nullptr, nullptr, absmax_in, nullptr, B, B, H, D, st);
    g_guide_trace = nullptr;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(stamps64, dev, 128 * sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return rc;
}

int mpdx_absmax(const float* x, uint32_t* absmax_out, int n_per_ctx, int B, int H, int D, void* stream) {
    if (!x || !absmax_out || B <= 0) return fail(MPDX_E_INVALID, "bad argument");
    const int npc = n_per_ctx > 0 ? n_per_ctx : B;
    if (B % npc) return fail(MPDX_E_INVALID, "B=%d is not a multiple of n_per_ctx=%d", B, npc);
    const size_t per = (size_t)npc * H * D;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>((per + 255) / 256, 64), B / npc), dim3(256), 0, (hipStream_t)stream, x,
                       absmax_out, per, B / npc);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
