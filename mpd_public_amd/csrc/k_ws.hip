// k_ws.hip - the weight-stationary persistent conv kernels (conv_ws.hpp).
#include "host.hpp"
#include "conv_wsn.hpp"

namespace mpdx {

template <int NC16, int MT, bool R1, int NS, int TBRES>
static int launch_ws_t(const Layer& l, ConvArgs& a, const ConvArgs& a2, int B, hipStream_t st) {
    if (l.L_out != kWsL || a.rs != ws_row_stride<NC16>() || l.cout != 8 * MT || l.cin_pad != NC16 * 16 || (R1 && (a2.C_out != l.cout || a2.L_out != kWsL)))
        return fail(MPDX_E_STATE, "layer %s does not have the geometry the weight-stationary kernel is compiled for (L %d, row stride %d, C_out %d)",
                    l.name.c_str(), l.L_out, a.rs, l.cout);
    if ((l.c2 > 0 && (l.c1 % 256 || l.c2 % 4)) || (l.c1 % 4) || (long)B * kWsL * std::max(l.c1, l.c2) * 4 > 0x7fffffffL)
        return fail(MPDX_E_STATE, "layer %s: channel split %d + %d / batch %d outside what the weight-stationary kernel's window loads take", l.name.c_str(), l.c1, l.c2, B);
    a.n_tiles_n = (int)(((long)B * l.L_out + 16 * NS - 1) / (16 * NS));
    const size_t lds = conv_ws_lds_bytes<NC16, MT, R1, NS>(l.L_out, a.rs);
    if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "weight-stationary conv needs %zu B of LDS", lds);
    auto kern = conv_ws_kernel<NC16, MT, R1, NS, TBRES>;
    if (int rc = raise_lds_limit((const void*)kern)) return rc;
    hipLaunchKernelGGL(kern, dim3((l.cout / MT) * kWsGroups), dim3(kWsThreads), lds, st, a, a2);
    return 0;
}
// what the block adds behind Mish: the time-bias row, a residual tensor, or nothing (never both: blocks[0] / blocks[1] of a ResidualTemporalBlock)
template <int NC16, int MT, bool R1, int NS = 1>
static int launch_ws(const Layer& l, ConvArgs& a, const ConvArgs& a2, int B, hipStream_t st) {
    if (a.tbias && a.res) return fail(MPDX_E_STATE, "layer %s: time bias AND residual on one weight-stationary launch", l.name.c_str());
    if (a.tbias) return launch_ws_t<NC16, MT, R1, NS, 1>(l, a, a2, B, st);
    if (a.res) return launch_ws_t<NC16, MT, R1, NS, 2>(l, a, a2, B, st);
    return launch_ws_t<NC16, MT, R1, NS, 0>(l, a, a2, B, st);
}

// the 128-channel layers of the innermost up level without a K split (conv_wsn.hpp): one wave = one (GroupNorm group, trajectory pair) tile
template <int MODE, int TBRES>
static int launch_wsn_t(const Layer& l, ConvArgs& a, int B, hipStream_t st) {
    if (l.L_in != 8 || l.c1 != kWsnC || l.c2 != 0 || l.cin_pad != kWsnC || l.cout != kWsnC || (MODE == CONV_S1 ? (l.L_out != 8 || l.gs != 16) : l.L_out != 16) ||
        a.pre || a.accum || a.dst2 || a.c_split || a.decim || a.stuff || (a.tbias && a.tb_stride) || (a.Lv_out > 0 && a.Lv_out < l.L_out) || (long)B * 8 * kWsnC * 4 > 0x7fffffffL)
        return fail(MPDX_E_STATE, "layer %s does not have the geometry conv_wsn_kernel is compiled for", l.name.c_str());
    const size_t lds = conv_wsn_lds_bytes<MODE>();
    auto kern = conv_wsn_kernel<MODE, TBRES>;
    if (int rc = raise_lds_limit((const void*)kern)) return rc;
    hipLaunchKernelGGL(kern, dim3((kWsnC / 16) * kWsnGroups), dim3(kWsnThreads), lds, st, a);
    return 0;
}

// downs[3]'s first block at large batch: Conv1dBlock 128 -> 256 + the residual 1x1 conv on the same input, one launch (conv_wsp_kernel)
static int launch_wsp(const Layer& l, ConvArgs& a, const ConvArgs& a2, int B, hipStream_t st) {
    if (l.L_in != 8 || l.L_out != 8 || l.c1 != kWsnC || l.c2 != 0 || l.cin_pad != kWsnC || l.cout != 256 || l.gs != 32 || a2.C_out != 256 || a2.L_out != 8 ||
        !a.tbias || a.tb_stride || a.res || a.pre || a2.pre || a2.accum || a2.dst2 || a2.c_split || (a.Lv_out > 0 && a.Lv_out < 8) || (long)B * 8 * kWsnC * 4 > 0x7fffffffL)
        return fail(MPDX_E_STATE, "layer %s does not have the geometry conv_wsp_kernel is compiled for", l.name.c_str());
    if (int rc = raise_lds_limit((const void*)conv_wsp_kernel)) return rc;
    hipLaunchKernelGGL(conv_wsp_kernel, dim3(8 * kWspGroups), dim3(kWspThreads), conv_wsp_lds_bytes(), st, a, a2);
    return 0;
}

int launch_weight_stationary(int variant, const Layer& l, ConvArgs& a, const ConvArgs& a2, int B, hipStream_t st) {
    switch (variant) {
        case 6: return launch_wsp(l, a, a2, B, st);
        case 4:
            if (a.tbias && a.res) return fail(MPDX_E_STATE, "layer %s: time bias AND residual on one weight-stationary launch", l.name.c_str());
            if (a.tbias) return launch_wsn_t<CONV_S1, 1>(l, a, B, st);
            if (a.res) return launch_wsn_t<CONV_S1, 2>(l, a, B, st);
            return launch_wsn_t<CONV_S1, 0>(l, a, B, st);
        case 5: return launch_wsn_t<CONV_UPT, 0>(l, a, B, st);
        case 1: {   // 32-position tiles (one duty wave on every SIMD) from 16 tiles per workgroup on
            static const int ns_env = getenv("MPDX_WS_NS") ? atoi(getenv("MPDX_WS_NS")) : 0;   // dev A/B: 1 / 2 force the tile
            const bool big = ns_env ? ns_env == 2 : (long)B * l.L_out >= 32L * kWsGroups * 16;
            if (big && conv_ws_lds_bytes<16, 32, false, 2>(l.L_out, a.rs) <= 160 * 1024) return launch_ws<16, 32, false, 2>(l, a, a2, B, st);
            return launch_ws<16, 32, false>(l, a, a2, B, st);
        }
        case 3: return launch_ws<32, 16, true>(l, a, a2, B, st);
    }
    return fail(MPDX_E_STATE, "no weight-stationary variant %d", variant);
}

}  // namespace mpdx
