// loss.hpp - the weighted loss value (helpers.py:71-99) as a device function shared by the stand-alone kernel (mpdx.hip, validation
// p_losses) and the training step's fused loss kernel (train.hpp): ONE workgroup of 1024 threads, fixed summation order.
#pragma once
#include <hip/hip_runtime.h>

namespace mpdx {

// mean over all B*H*D elements of |e| or e^2 (optionally times weights[H*D]) of apply_hard_conditioning(pred) against targ.
// Call with 1024 threads; `part` is 16 doubles of LDS.
__device__ __forceinline__ void weighted_loss_body(const float* pred, const float* targ, const float* weights, const float* hs, const float* hg, int l1,
                                                   float* out, int B, int H, int D, double* part) {
    const size_t n = (size_t)B * H * D;
    double acc = 0.0;
    // A thread's elements i = tid, tid + 1024, ... are added in ascending order (the value of the sum is defined by that order); their loads are
    // issued U at a time, UNCONDITIONALLY (clamped indices, selects afterwards): with the loads inside the element loop and behind the hard-condition
    // branches this one-workgroup reduction was a chain of n / 4096 dependent round trips - 54 us at batch 128 x D = 14 (round 4 profile).
    constexpr unsigned U = 16;
    const unsigned nn = (unsigned)n;   // (32-bit index arithmetic; n = B * H * D is far below 2^32 here)
    for (unsigned i0 = threadIdx.x; i0 < nn; i0 += 1024u * U) {
        float pv[U], tv[U], wv[U], sv[U], gv[U];
        unsigned li[U];
        bool in[U];
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            const unsigned i = i0 + u * 1024u;
            in[u] = i < nn;
            const unsigned ic = in[u] ? i : 0u;
            const unsigned d = ic % (unsigned)D, p = ic / (unsigned)D;
            const unsigned l = p % (unsigned)H, b = p / (unsigned)H;
            li[u] = l;
            pv[u] = pred[ic]; tv[u] = targ[ic];
            wv[u] = weights ? weights[(size_t)l * D + d] : 1.0f;
            sv[u] = hs ? hs[b * D + d] : 0.f;
            gv[u] = hg ? hg[b * D + d] : 0.f;
        }
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            if (!in[u]) continue;
            float v = pv[u];
            if (hs && li[u] == 0) v = sv[u];
            if (hg && li[u] == (unsigned)(H - 1)) v = gv[u];
            const float e = __fsub_rn(v, tv[u]);
            float q = l1 ? fabsf(e) : __fmul_rn(e, e);
            if (weights) q = __fmul_rn(q, wv[u]);
            acc += (double)q;
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < 16; ++k) tot += part[k];
        out[0] = (float)(tot / (double)n);
    }
}

}  // namespace mpdx
