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
#pragma unroll 4
    for (unsigned i = threadIdx.x; i < (unsigned)n; i += 1024u) {   // (32-bit index arithmetic; n = B * H * D is far below 2^32 here)
        const unsigned d = i % (unsigned)D, p = i / (unsigned)D;
        const unsigned l = p % (unsigned)H, b = p / (unsigned)H;
        float v = pred[i];
        if (hs && l == 0) v = hs[b * D + d];
        if (hg && l == (unsigned)(H - 1)) v = hg[b * D + d];
        const float e = __fsub_rn(v, targ[i]);
        float q = l1 ? fabsf(e) : __fmul_rn(e, e);
        if (weights) q = __fmul_rn(q, weights[(size_t)l * D + d]);
        acc += (double)q;
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
