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

}  // namespace mpdx
