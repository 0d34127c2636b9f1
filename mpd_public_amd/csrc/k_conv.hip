// k_conv.hip - every instantiation of conv_block_kernel / conv_pair_kernel (conv_block.hpp) behind two plain launchers.
#include "host.hpp"

namespace mpdx {

template <int MODE, int KS, int EPI, int MT, int NT, int WN, int WK>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    const size_t lds = conv_block_lds_bytes<MODE, KS, MT, NT, WK>(a.L_in, a.L_out, a.rs);
    if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "conv tile needs %zu B of LDS", lds);
    auto kern = conv_block_kernel<MODE, KS, EPI, MT, NT, WN, WK>;
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit((const void*)kern)) return rc;
    const int grid = (a.C_out / MT) * a.n_tiles_n;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WN * WK), lds, st, a);
    return 0;
}

template <int MODE, int KS, int EPI>
static int dispatch_tile(const Layer& l, ConvArgs& a, int B, hipStream_t st) {
    int MT, NT;
    choose_tile(l, B, MT, NT);
    if (l.cout % MT) MT = 16;
    if (l.cout % MT || NT % l.L_out) return fail(MPDX_E_INVALID, "layer %s: no tile for C_out=%d L=%d", l.name.c_str(), l.cout, l.L_out);
    a.n_tiles_n = (int)(((long)B * l.L_out + NT - 1) / NT);
    const bool ksplit = layer_ksplit(l);
#define MPDX_TILE(mt, nt)                                                                        \
    if (MT == mt && NT == nt) {                                                                  \
        if (ksplit) return launch_conv<MODE, KS, EPI, mt, nt, 1, 8>(a, st);                      \
        return launch_conv<MODE, KS, EPI, mt, nt, nt / 16, 8 / (nt / 16)>(a, st);               \
    }
    MPDX_TILE(32, 64) MPDX_TILE(32, 32) MPDX_TILE(16, 64) MPDX_TILE(16, 32) MPDX_TILE(32, 16) MPDX_TILE(16, 16)
    if constexpr (EPI != EPI_GN_MISH) { MPDX_TILE(16, 128) MPDX_TILE(32, 128) }   // 128-position levels have regions >= 512: _GEN only
#undef MPDX_TILE
    return fail(MPDX_E_INVALID, "no instantiation for tile %dx%d", MT, NT);
}

template <int MODE, int KS, int EPI>
static int dispatch_tile_ksplit_only(const Layer& l, ConvArgs& a, int B, hipStream_t st) {
    int MT, NT;
    choose_tile(l, B, MT, NT);
    if (l.cout % MT) MT = 16;
    if (l.cout % MT || NT % l.L_out) return fail(MPDX_E_INVALID, "layer %s: no tile for C_out=%d L=%d", l.name.c_str(), l.cout, l.L_out);
    a.n_tiles_n = (int)(((long)B * l.L_out + NT - 1) / NT);
#define MPDX_TILE(mt, nt) \
    if (MT == mt && NT == nt) return launch_conv<MODE, KS, EPI, mt, nt, 1, 8>(a, st);
    MPDX_TILE(32, 64) MPDX_TILE(32, 32) MPDX_TILE(16, 64) MPDX_TILE(16, 32) MPDX_TILE(16, 128) MPDX_TILE(32, 128)
    if constexpr (MODE != CONV_UPT) { MPDX_TILE(32, 16) MPDX_TILE(16, 16) }
#undef MPDX_TILE
    return fail(MPDX_E_INVALID, "no instantiation for tile %dx%d", MT, NT);
}

// one launch of layer l (the tile is chosen for the batch; `a` gets n_tiles_n)
int launch_conv_layer(const Layer& l, ConvArgs& a, int B, hipStream_t st) {
    if (l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_GN_MISH) {
        const int re = l.gs * l.L_out;   // regions other than 128 / 256 elements (horizons other than 64): the general-region instantiations
        if ((re != 128 && re != 256) || (a.Lv_out > 0 && a.Lv_out < l.L_out)) return dispatch_tile<CONV_S1, 5, EPI_GN_MISH_GEN>(l, a, B, st);
        return dispatch_tile<CONV_S1, 5, EPI_GN_MISH>(l, a, B, st);
    }
    if (l.mode == CONV_S1 && l.ks == 1 && l.epi == EPI_BIAS) return dispatch_tile_ksplit_only<CONV_S1, 1, EPI_BIAS>(l, a, B, st);
    if (l.mode == CONV_DOWN && l.ks == 3) return dispatch_tile_ksplit_only<CONV_DOWN, 3, EPI_BIAS>(l, a, B, st);
    if (l.mode == CONV_UPT && l.ks == 4) return dispatch_tile_ksplit_only<CONV_UPT, 4, EPI_BIAS>(l, a, B, st);
    // the input-gradient convolutions of the training step (train_host.hpp)
    if (l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_BIAS) return dispatch_tile_ksplit_only<CONV_S1, 5, EPI_BIAS>(l, a, B, st);
    if (l.mode == CONV_S1 && l.ks == 3 && l.epi == EPI_BIAS) return dispatch_tile_ksplit_only<CONV_S1, 3, EPI_BIAS>(l, a, B, st);
    return fail(MPDX_E_INVALID, "layer %s: unsupported conv (mode %d k %d epi %d)", l.name.c_str(), l.mode, l.ks, l.epi);
}

// blocks[0] + residual 1x1 conv of the same ResidualTemporalBlock in one launch (both read the block's input).
// Returns 1 if the pair was launched, 0 if the shapes do not qualify (caller launches them separately), <0 on error.
template <int MT, int NT>
static int launch_pair_t(const ConvArgs& a1, const ConvArgs& a2, const Layer& l1, const Layer& l2, hipStream_t st) {
    const size_t lds = std::max(conv_block_lds_bytes<CONV_S1, 5, MT, NT, 8>(l1.L_in, l1.L_out, l1.rs),
                                conv_block_lds_bytes<CONV_S1, 1, MT, NT, 8>(l2.L_in, l2.L_out, l2.rs));
    if (lds > 160 * 1024) return 0;
    auto kern = conv_pair_kernel<MT, NT>;
    if (lds > 64 * 1024)
        if (raise_lds_limit((const void*)kern)) return -1;
    const int n1 = (a1.C_out / MT) * a1.n_tiles_n, n2 = (a2.C_out / MT) * a2.n_tiles_n;
    hipLaunchKernelGGL(kern, dim3(n1 + n2), dim3(512), lds, st, a1, a2, n1);
    return 1;
}

int launch_conv_pair(int MT, int NT, const ConvArgs& a1, const ConvArgs& a2, const Layer& l1, const Layer& l2, hipStream_t st) {
#define MPDX_PAIR_TILE(mt, nt) if (MT == mt && NT == nt) return launch_pair_t<mt, nt>(a1, a2, l1, l2, st);
    MPDX_PAIR_TILE(32, 64) MPDX_PAIR_TILE(32, 32) MPDX_PAIR_TILE(16, 64) MPDX_PAIR_TILE(16, 32) MPDX_PAIR_TILE(32, 16) MPDX_PAIR_TILE(16, 16)
#undef MPDX_PAIR_TILE
    return fail(MPDX_E_INVALID, "no pair instantiation for tile %dx%d", MT, NT);
}

}  // namespace mpdx
