// train_types.hpp - plain tables shared by the training kernels (train.hpp) and the host model (host.hpp).
#pragma once

namespace mpdx {

struct PackDesc {
    unsigned long long src, dst, dstT;   // float offsets (dstT == ~0: no dgrad copy)
    unsigned long long n, pn, pnT;       // floats: reference tensor, forward pack, dgrad pack
    int kind;                            // PK_VEC / PK_CONV / PK_CONVT
    int cout, cin, ks, cin_pad, nslot;   // forward geometry
    int t_cout, t_cin, t_ks, t_cin_pad;  // dgrad geometry (a CONV_S1 convolution t_cin -> t_cout with t_ks taps)
    int t_mode;                          // 0: transpose + flip (Conv1d), 1: ConvTranspose1d k4 -> 5 taps
};

struct PackChunk { int desc; int which; unsigned first; unsigned pad; };   // which: 0 forward pack (vectors too), 1 dgrad pack

// a strided copy inside `packed` (float units): the stream-ordered weight copies and the contiguous parameter block of a fused program
// (mpdx.hip ensure_fused_streams; the training pass runs them as side blocks of its first launch)
struct CopyJobDev { unsigned long long src, dst; int n0, ss0, ds0, n1, ss1, ds1, n_inner; };
__device__ __forceinline__ void restream_job(float* __restrict__ packed, const CopyJobDev j, const unsigned first, const unsigned stride) {
    const unsigned total = (unsigned)j.n0 * (unsigned)j.n1 * (unsigned)j.n_inner, ni = (unsigned)j.n_inner, n1 = (unsigned)j.n1;   // (32-bit index arithmetic)
#pragma unroll 4
    for (unsigned i = first; i < total; i += stride) {
        const unsigned k = i % ni, r = i / ni;
        const unsigned i1 = r % n1, i0 = r / n1;
        packed[j.dst + (size_t)i0 * j.ds0 + (size_t)i1 * j.ds1 + k] = packed[j.src + (size_t)i0 * j.ss0 + (size_t)i1 * j.ss1 + k];
    }
}

}  // namespace mpdx
