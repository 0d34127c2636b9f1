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

// ---- the inner levels' launches with COMPILE-TIME geometry (conv_block.hpp GeoL8): CONV_S1 k5 + GroupNorm + Mish on L = 8, C_in = 16 * NC16
// without padding, row stride C_in + 8, K split over the 8 waves, tiles of (16 | 32) channels x (16 | 32 | 64) positions
static bool geo_l8(const Layer& l, const ConvArgs& a) {
    return l.mode == CONV_S1 && l.L_in == 8 && l.L_out == 8 && l.cin_pad == l.c1 + l.c2 && !(l.c1 & 3) && !(l.c2 & 3) && a.rs == l.cin_pad + 8 &&
           (l.cin_pad == 128 || l.cin_pad == 256 || l.cin_pad == 512) && !a.dbg && !(a.Lv_out > 0 && a.Lv_out < l.L_out) &&
           !(getenv("MPDX_GEO") && atoi(getenv("MPDX_GEO")) == 0);   // MPDX_GEO=0: the runtime-geometry kernels (development A/B)
}
template <int NC16, int MT, int NT, int TBRES>
static int launch_geo_gn(const ConvArgs& a, hipStream_t st) {
    const size_t lds = conv_block_lds_bytes<CONV_S1, 5, MT, NT, 8>(8, 8, NC16 * 16 + 8);
    if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "conv tile needs %zu B of LDS", lds);
    auto kern = conv_block_kernel<CONV_S1, 5, EPI_GN_MISH, MT, NT, 1, 8, GeoL8<NC16>, TBRES>;
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit((const void*)kern)) return rc;
    hipLaunchKernelGGL(kern, dim3((a.C_out / MT) * a.n_tiles_n), dim3(512), lds, st, a);
    return 0;
}
// returns -100 when the (channels, tile) combination has no compile-time-geometry instantiation (the caller takes the generic kernel)
static int dispatch_geo_gn(const Layer& l, ConvArgs& a, int MT, int NT, hipStream_t st) {
    if (a.tbias && a.res) return -100;
    const int tbres = a.tbias ? 1 : (a.res ? 2 : 0);
#define MPDX_GEO_TILE(nc, mt, nt)                                                              \
    if (l.cin_pad == nc * 16 && MT == mt && NT == nt) {                                        \
        if (tbres == 1) return launch_geo_gn<nc, mt, nt, 1>(a, st);                            \
        if (tbres == 2) return launch_geo_gn<nc, mt, nt, 2>(a, st);                            \
        return launch_geo_gn<nc, mt, nt, 0>(a, st);                                            \
    }
    MPDX_GEO_TILE(16, 32, 32) MPDX_GEO_TILE(16, 32, 64) MPDX_GEO_TILE(16, 32, 16)      // 256 -> 256 (downs[3], mid blocks)
    MPDX_GEO_TILE(8, 16, 32) MPDX_GEO_TILE(8, 16, 64) MPDX_GEO_TILE(8, 32, 32) MPDX_GEO_TILE(8, 32, 64)   // 128 -> 128 / 128 -> 256
    MPDX_GEO_TILE(32, 32, 16) MPDX_GEO_TILE(32, 32, 32) MPDX_GEO_TILE(32, 16, 32) MPDX_GEO_TILE(32, 16, 64)   // 512 -> 128 (ups[0])
#undef MPDX_GEO_TILE
    return -100;
}

template <int MODE, int KS, int EPI>
static int dispatch_tile(const Layer& l, ConvArgs& a, int B, hipStream_t st) {
    int MT, NT;
    choose_tile(l, B, MT, NT);
    if (l.cout % MT) MT = 16;
    if (l.cout % MT || NT % l.L_out) return fail(MPDX_E_INVALID, "layer %s: no tile for C_out=%d L=%d", l.name.c_str(), l.cout, l.L_out);
    a.n_tiles_n = (int)(((long)B * l.L_out + NT - 1) / NT);
    const bool ksplit = layer_ksplit(l);
    if constexpr (MODE == CONV_S1 && KS == 5 && EPI == EPI_GN_MISH) {
        if (ksplit && geo_l8(l, a)) {
            const int rc = dispatch_geo_gn(l, a, MT, NT, st);
            if (rc != -100) return rc;
        }
    }
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
    if (l.mode == CONV_UPT && l.ks == 4) {
        // Upsample1d of the innermost up level (C -> C, 8 -> 16 positions) with compile-time geometry, where instantiated
        if (l.L_in == 8 && l.L_out == 16 && l.cin_pad == l.c1 && l.c2 == 0 && !(l.c1 & 3) && a.rs == l.cin_pad + 4 && !a.dbg && !a.pre && !a.accum && !a.dst2 &&
            !(a.Lv_out > 0 && a.Lv_out < l.L_out) && !(getenv("MPDX_GEO") && atoi(getenv("MPDX_GEO")) == 0)) {
            int MT, NT;
            choose_tile(l, B, MT, NT);
            if (l.cout % MT) MT = 16;
            a.n_tiles_n = (int)(((long)B * l.L_out + NT - 1) / NT);
#define MPDX_GEO_UP(nc, mt, nt)                                                                         \
            if (l.cin_pad == nc * 16 && MT == mt && NT == nt) {                                          \
                const size_t lds = conv_block_lds_bytes<CONV_UPT, 4, mt, nt, 8>(8, 16, nc * 16 + 4);     \
                auto kern = conv_block_kernel<CONV_UPT, 4, EPI_BIAS, mt, nt, 1, 8, GeoUp8<nc>>;          \
                if (lds <= 160 * 1024) {                                                                 \
                    if (lds > 64 * 1024)                                                                 \
                        if (int rc = raise_lds_limit((const void*)kern)) return rc;                      \
                    hipLaunchKernelGGL(kern, dim3((a.C_out / mt) * a.n_tiles_n), dim3(512), lds, st, a); \
                    return 0;                                                                            \
                }                                                                                        \
            }
            MPDX_GEO_UP(8, 16, 64) MPDX_GEO_UP(8, 32, 64) MPDX_GEO_UP(8, 16, 32) MPDX_GEO_UP(8, 32, 32)
#undef MPDX_GEO_UP
        }
        return dispatch_tile_ksplit_only<CONV_UPT, 4, EPI_BIAS>(l, a, B, st);
    }
    // the input-gradient convolutions of the training step (train_host.hpp)
    if (l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_BIAS) return dispatch_tile_ksplit_only<CONV_S1, 5, EPI_BIAS>(l, a, B, st);
    if (l.mode == CONV_S1 && l.ks == 3 && l.epi == EPI_BIAS) return dispatch_tile_ksplit_only<CONV_S1, 3, EPI_BIAS>(l, a, B, st);
    return fail(MPDX_E_INVALID, "layer %s: unsupported conv (mode %d k %d epi %d)", l.name.c_str(), l.mode, l.ks, l.epi);
}

// blocks[0] + residual 1x1 conv of the same ResidualTemporalBlock in one launch (both read the block's input).
// Returns 1 if the pair was launched, 0 if the shapes do not qualify (caller launches them separately), <0 on error.
template <int MT, int NT, class GEO = GeoAny>
static int launch_pair_t(const ConvArgs& a1, const ConvArgs& a2, const Layer& l1, const Layer& l2, hipStream_t st) {
    const size_t lds = std::max(conv_block_lds_bytes<CONV_S1, 5, MT, NT, 8>(l1.L_in, l1.L_out, l1.rs),
                                conv_block_lds_bytes<CONV_S1, 1, MT, NT, 8>(l2.L_in, l2.L_out, l2.rs));
    if (lds > 160 * 1024) return 0;
    auto kern = conv_pair_kernel<MT, NT, GEO>;
    if (lds > 64 * 1024)
        if (raise_lds_limit((const void*)kern)) return -1;
    const int n1 = (a1.C_out / MT) * a1.n_tiles_n, n2 = (a2.C_out / MT) * a2.n_tiles_n;
    hipLaunchKernelGGL(kern, dim3(n1 + n2), dim3(512), lds, st, a1, a2, n1);
    return 1;
}

int launch_conv_pair(int MT, int NT, const ConvArgs& a1, const ConvArgs& a2, const Layer& l1, const Layer& l2, hipStream_t st) {
    // blocks[0] (+ time bias, no residual) and the 1x1 residual conv on the same L = 8 input: compile-time geometry where instantiated
    if (geo_l8(l1, a1) && geo_l8(l2, a2) && a1.tbias && !a1.res && l2.rs == l1.rs) {
#define MPDX_GEO_PAIR(nc, mt, nt) if (l1.cin_pad == nc * 16 && MT == mt && NT == nt) return launch_pair_t<mt, nt, GeoL8<nc>>(a1, a2, l1, l2, st);
        MPDX_GEO_PAIR(8, 32, 32) MPDX_GEO_PAIR(8, 32, 64) MPDX_GEO_PAIR(8, 32, 16) MPDX_GEO_PAIR(32, 32, 16) MPDX_GEO_PAIR(32, 32, 32) MPDX_GEO_PAIR(32, 16, 32)
#undef MPDX_GEO_PAIR
    }
#define MPDX_PAIR_TILE(mt, nt) if (MT == mt && NT == nt) return launch_pair_t<mt, nt>(a1, a2, l1, l2, st);
    MPDX_PAIR_TILE(32, 64) MPDX_PAIR_TILE(32, 32) MPDX_PAIR_TILE(16, 64) MPDX_PAIR_TILE(16, 32) MPDX_PAIR_TILE(32, 16) MPDX_PAIR_TILE(16, 16)
#undef MPDX_PAIR_TILE
    return fail(MPDX_E_INVALID, "no pair instantiation for tile %dx%d", MT, NT);
}

}  // namespace mpdx
