// loss.hpp - the weighted loss value (helpers.py:71-99) as device functions shared by the stand-alone kernel (mpdx.hip, validation p_losses)
// and the training step's fused loss kernel (train.hpp).
//
// The VALUE of the sum is defined by its order: 1024 "threads", thread t adds its elements i = t, t + 1024, ... in ascending order (double), the 64
// threads of a wave are combined by an xor-shuffle tree, the 16 wave sums are added in wave order.  weighted_loss_kernel runs that as ONE workgroup
// of 1024 threads; the training step runs the 16 waves as 16 one-wave workgroups on 16 CUs (same bits, a sixteenth of the time: round 4 found the
// one-workgroup form VALU-bound on its index arithmetic - two 32-bit divisions per element, 48 us per launch at batch 128 x D = 14 - the indices are
// now advanced incrementally).
#pragma once
#include <hip/hip_runtime.h>

namespace mpdx {

// the elements of thread `tid` (0 .. 1023), ascending:  |e| or e^2 (optionally times weights[H*D]) of apply_hard_conditioning(pred) against targ
__device__ __forceinline__ double weighted_loss_thread_sum(const float* pred, const float* targ, const float* weights, const float* hs, const float* hg,
                                                           int l1, int B, int H, int D, unsigned tid) {
    const unsigned nn = (unsigned)((size_t)B * H * D);   // (32-bit index arithmetic; n = B * H * D is far below 2^32 here)
    double acc = 0.0;
    if (tid >= nn) return acc;
    // (d, l, b) of element i, advanced by 1024 elements without a division: i += 1024 -> d += 1024 % D with a carry into the row index p = i / D,
    // p += 1024 / D (+ carry) -> l += that % H with a carry into b
    const unsigned uD = (unsigned)D, uH = (unsigned)H;
    unsigned i = tid, d = i % uD, p = i / uD, l = p % uH, b = p / uH;
    const unsigned sd = 1024u % uD, sp = 1024u / uD, spl = sp % uH, spb = sp / uH;
    constexpr unsigned U = 8;   // loads in flight per thread (issued unconditionally, clamped: a load behind a branch waits for everything before it)
    while (i < nn) {
        float pv[U], tv[U], wv[U], sv[U], gv[U];
        unsigned ll[U];
        bool in[U];
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            in[u] = i < nn;
            const unsigned ic = in[u] ? i : tid, dc = in[u] ? d : 0u, lc = in[u] ? l : 1u, bc = in[u] ? b : 0u;
            ll[u] = lc;
            pv[u] = pred[ic]; tv[u] = targ[ic];
            wv[u] = weights ? weights[(size_t)lc * uD + dc] : 1.0f;
            sv[u] = hs ? hs[bc * uD + dc] : 0.f;
            gv[u] = hg ? hg[bc * uD + dc] : 0.f;
            i += 1024u;
            d += sd;
            const unsigned carry = d >= uD ? 1u : 0u;
            d -= carry * uD;
            l += spl + carry; b += spb;
            if (l >= uH) { l -= uH; ++b; }
        }
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            if (!in[u]) continue;
            float v = pv[u];
            if (hs && ll[u] == 0) v = sv[u];
            if (hg && ll[u] == uH - 1u) v = gv[u];
            const float e = __fsub_rn(v, tv[u]);
            float q = l1 ? fabsf(e) : __fmul_rn(e, e);
            if (weights) q = __fmul_rn(q, wv[u]);
            acc += (double)q;
        }
    }
    return acc;
}

__device__ __forceinline__ double weighted_loss_wave_sum(double acc) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    return acc;
}

// ONE workgroup of 1024 threads; `part`: 16 doubles of LDS
__device__ __forceinline__ void weighted_loss_body(const float* pred, const float* targ, const float* weights, const float* hs, const float* hg, int l1,
                                                   float* out, int B, int H, int D, double* part) {
    const double acc = weighted_loss_wave_sum(weighted_loss_thread_sum(pred, targ, weights, hs, hg, l1, B, H, D, threadIdx.x));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < 16; ++k) tot += part[k];
        out[0] = (float)(tot / (double)((size_t)B * H * D));
    }
}

// wave `w` (0 .. 15) of the same sum as its own workgroup: ALL 1024 threads of the workgroup fetch the wave's elements (coalesced, every load in
// flight at once) and leave each element's term q in LDS; the workgroup's first wave then adds its threads' terms in the sum's order (thread t: elements
// t, t + 1024, ... ascending, in double; xor-shuffle tree).  (A first version let the one wave fetch its own 112 elements per thread, eight at a time:
// fourteen dependent round trips, 20 us at batch 128 x D = 14.)  Partial sums go through `gpart` (16 doubles of global memory, agent-scope atomic
// stores / loads: no L2-flushing fence); the workgroup that takes the last ticket (`ticket`: zero at launch) adds them in wave order.
constexpr int kLossChunk = 128;   // elements per thread and LDS round: 64 x 128 floats = 32 KB
__device__ __forceinline__ void weighted_loss_wave_block(const float* pred, const float* targ, const float* weights, const float* hs, const float* hg, int l1,
                                                         float* out, int B, int H, int D, int w, double* gpart, unsigned* ticket, float* q) {
    const unsigned nn = (unsigned)((size_t)B * H * D), uD = (unsigned)D, uH = (unsigned)H;
    const unsigned K = (nn + 1023u) / 1024u;   // elements per thread of the sum
    const unsigned tid = threadIdx.x;
    double acc = 0.0;
    for (unsigned k0 = 0; k0 < K; k0 += kLossChunk) {
        const unsigned cnt = K - k0 < (unsigned)kLossChunk ? K - k0 : (unsigned)kLossChunk;
        for (unsigned j = tid; j < 64u * cnt; j += 1024u) {
            const unsigned t = j & 63u, k = k0 + (j >> 6);
            const unsigned i = (unsigned)w * 64u + t + 1024u * k;
            float term = 0.f;   // (elements behind the end add +0.0: the double sum is unchanged)
            if (i < nn) {
                const unsigned d = i % uD, p = i / uD, l = p % uH, b = p / uH;
                float v = pred[i];
                if (hs && l == 0) v = hs[b * uD + d];
                if (hg && l == uH - 1u) v = hg[b * uD + d];
                const float e = __fsub_rn(v, targ[i]);
                term = l1 ? fabsf(e) : __fmul_rn(e, e);
                if (weights) term = __fmul_rn(term, weights[(size_t)l * uD + d]);
            }
            q[j] = term;
        }
        __syncthreads();
        if (tid < 64u) {
            // terms kk ascending (the sum's order), their LDS reads eight at a time: one by one, behind the bounds test, every add waited for its own
            // read (~105 cycles per term: 5 of this launch's 15 us at batch 128 x D = 14).  Terms behind the end were stored as +0.0: no test needed.
            unsigned kk = 0;
            for (; kk + 8u <= cnt; kk += 8u) {
                float v[8];
#pragma unroll
                for (unsigned u = 0; u < 8u; ++u) v[u] = q[tid + 64u * (kk + u)];
#pragma unroll
                for (unsigned u = 0; u < 8u; ++u) acc += (double)v[u];
            }
            for (; kk < cnt; ++kk) acc += (double)q[tid + 64u * kk];
        }
        __syncthreads();
    }
    if (tid >= 64u) return;
    acc = weighted_loss_wave_sum(acc);
    if (tid == 0) {
        // last-ticket reduction in the memory model's own terms (MI355X_MICROARCH.md, inter-workgroup visibility): the partial sum is
        // published by the RELEASE half of the ticket's acq_rel read-modify-write, the last arriver's ACQUIRE half makes the other 15
        // visible to its loads (round 4 leaned on relaxed atomics + an inline vmcnt(0): right on gfx950 as compiled, not by contract)
        __hip_atomic_store(gpart + w, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 15u) {
            double tot = 0.0;
            for (int k = 0; k < 16; ++k) tot += __hip_atomic_load(gpart + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[0] = (float)(tot / (double)((size_t)B * H * D));
        }
    }
}

}  // namespace mpdx
