// conv_ws.hpp - WEIGHT-STATIONARY persistent variant of conv_block_kernel for the large-batch inner levels
// (Conv1dBlock = Conv1d(k5, pad 2) -> GroupNorm(8) -> Mish [+ time bias] [+ residual], mpd/models/layers/layers.py:276-293,343-355,
//  at C_in = C_out = 256, L = 8: downs[3] / mid blocks of TemporalUnet, temporal_unet.py:141-157).
//
// Why.  At B = 6400 (cfg 5: one GPU's shard) conv_block_kernel<...,32,32,1,8> runs 12 800 workgroups per layer; every one streams its
// 164 KB weight slice from L2 again and walks staging -> k-loop -> K-reduction -> epilogue with barriers in between: 101 TF/s = 64 % of
// the fp32 MFMA peak (profiles/r02_cfg5_kernel_stats.csv), and a larger tile at one workgroup per CU is slower still (measured:
// MPDX_LDS_CAP_KB=100 -> 638 vs 616 ms per plan) - what the launch lacks is overlap, not reuse.
//
// Design.  256 persistent workgroups (one per CU), workgroup = (channel tile mt of 32 channels) x (position group p of 32); it keeps ITS
// weights in REGISTERS for the whole launch (wave wk owns k-groups wk, wk+8, ...: 10 x 2 A fragments = 80 VGPRs) and loops over the
// position tiles p, p+32, ... of 16 positions (2 trajectories).  Per tile: the k-loop (80 MFMAs per wave out of the LDS window), the
// K-partials into a double-buffered reduction area, ONE barrier; the window of tile i+1 was fetched into registers during tile i's
// k-loop (issue early / write late) and is written to the other window buffer before that barrier; the epilogue of tile i (one wave per
// GroupNorm region, exactly conv_block_kernel's code) is done by TWO ROTATING duty waves while everybody - they too, afterwards - runs
// the k-loop of tile i+1: its VALU work hides under the SIMD partner's MFMAs.  b = mt * 32 + p puts the eight channel tiles of a
// position group on ONE XCD (block b runs on XCD b % 8): an activation window is fetched once per L2.
//
// Numerics.  Same k-group -> wave assignment, same accumulation order inside a wave, same K-partial order, same epilogue code as
// conv_block_kernel<CONV_S1, 5, EPI_GN_MISH, 32, NT, 1, 8>: the outputs are BIT-IDENTICAL (tests/test_gpu_parity.py checks that).
#pragma once
#include "conv_block.hpp"

namespace mpdx {

constexpr int kWsGroups = 32;    // position groups: 8 channel tiles x 32 = 256 workgroups
constexpr int kWsThreads = 512;

template <int NC16>
inline size_t conv_ws_lds_bytes(int L, int rs) {
    const size_t stage = (size_t)(16 / L) * (L + 4) * rs * sizeof(float);
    const size_t red = (size_t)8 * 16 * (32 + 4) * sizeof(float);
    return 2 * stage + 2 * red;
}

template <int NC16>
__global__ __launch_bounds__(kWsThreads) void conv_ws_kernel(const ConvArgs a) {
    constexpr int KS = 5, PAD = 2, MT = 32, MS = 2, NT = 16, WK = 8, NG = NC16 * KS, NIT = NG / WK;
    constexpr int MTP4 = (MT + 4) / 4;
    static_assert(NG % WK == 0, "k-groups split evenly over the 8 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* const smem4 = (f32x4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x / kWsGroups, p = blockIdx.x % kWsGroups;
    const int L = a.L_out, LP = L + 2 * PAD, RS4 = a.rs >> 2;
    const int spt = NT >> a.lg_Lout;                       // trajectories per tile (2 at L = 8)
    const int stage4 = spt * LP * RS4;                     // float4 per window buffer
    const int red4 = WK * NT * MTP4;                       // float4 per reduction buffer
    const int red_off4 = 2 * stage4;
    const int n_tiles = a.n_tiles_n;
    const int c4n = a.cin_pad >> 2;

    // ---- this wave's weights: k-groups wk, wk + 8, ...  (loaded once, kept for the whole launch)
    constexpr int nc16 = NC16;
    const float* wbase = a.wp + (size_t)(mt * MS) * nc16 * KS * 256 + lane * 4;
    f32x4 af[NIT][MS];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int g = wk + it * WK, c16 = g / KS, ts = g - c16 * KS;
#pragma unroll
        for (int m = 0; m < MS; ++m) af[it][m] = *(const f32x4*)(wbase + ((size_t)(m * nc16 + c16) * KS + ts) * 256);
    }
    // ---- halo rows of both window buffers (never overwritten by the window writes)
    for (int idx = tid; idx < 2 * spt * 2 * PAD * c4n; idx += kWsThreads) {
        const int c4 = idx & (c4n - 1), hr = idx >> a.lg_c4n;                 // hr over [buffer][trajectory][4 halo rows]
        const int k = hr & 3, s = (hr >> 2) % spt, buf = (hr >> 2) / spt;
        const int lp = (k < PAD) ? k : (L + k);
        smem4[buf * stage4 + (s * LP + lp) * RS4 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // window of a tile: spt * L * c4n float4 = 1024 at C = 256 -> 2 per thread
    constexpr int SB = (16 * NC16 * 4) / kWsThreads;       // float4 per thread per window (NT positions x cin/4)
    static_assert(SB >= 1 && (16 * NC16 * 4) % kWsThreads == 0, "window divides over the threads");
    int wdst[SB];                                          // LDS index (float4) inside a window buffer
    int wrow[SB], wc[SB], ws_[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int idx = tid + u * kWsThreads;
        const int rowi = idx >> a.lg_c4n, c4 = idx & (c4n - 1);
        const int s = rowi >> a.lg_Lin, li = rowi & (L - 1);
        wdst[u] = (s * LP + li + PAD) * RS4 + c4;
        wrow[u] = li; wc[u] = c4 * 4; ws_[u] = s;
    }
    f32x4 wv[SB];
    auto window_load = [&](int tile) {   // unconditional loads from clamped addresses (a conditional load serialises the queue)
        const int s0 = tile * spt;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            int b = s0 + ws_[u];
            b = b < a.B ? b : a.B - 1;
            wv[u] = *(const f32x4*)(a.src1 + ((size_t)b * L + wrow[u]) * a.c1 + wc[u]);
        }
    };
    auto window_write = [&](int tile, int buf) {
        const int s0 = tile * spt;
#pragma unroll
        for (int u = 0; u < SB; ++u) smem4[buf * stage4 + wdst[u]] = (s0 + ws_[u] < a.B) ? wv[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // lane's B row in a window (tile-local position n = j), and its column in the reduction buffer
    const int j = lane & 15, q = lane >> 4;
    const int boff = ((j >> a.lg_Lout) * LP + (j & (L - 1))) * RS4 + q;

    // ---- epilogue operands of the region this lane would serve (channels are fixed per lane: loaded once)
    const int e0 = lane * 4;
    const int el = e0 >> 5, ec = e0 & 31;                   // region = 32 channels x 8 positions: lane -> (position, 4 channels)
    const int co = mt * MT + ec;
    const f32x4 bi = *(const f32x4*)(a.bias + co), ga = *(const f32x4*)(a.gamma + co), be = *(const f32x4*)(a.beta + co);

    int tile = p;
    if (tile < n_tiles) { window_load(tile); window_write(tile, 0); }
    if (tile + kWsGroups < n_tiles) window_load(tile + kWsGroups);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // dev tool (-DMPDX_DEV_HOOKS, mpdx_layer_trace): stamps of workgroup 0, waves 0 / 1 (slots 0..15 / 16..31), tiles 8 and 9:
    // tile top | k-loop issued | partials + window written | barrier passed | epilogue done
    long long* const trp = (MPDX_TRACE_PTR(a.trace) && blockIdx.x == 0 && lane == 0 && wk < 2) ? a.trace + wk * 16 : nullptr;
#define WS_STAMP(k) do { if (trp && (i == 8 || i == 9)) trp[(i - 8) * 5 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
    for (int i = 0; tile < n_tiles; ++i, tile += kWsGroups) {
        const int cur = i & 1;
        WS_STAMP(0);
        // this wave's epilogue duty for the tile (region r = trajectory r of the tile): its global operands are requested BEFORE the
        // k-loop - a duty wave that waits ~1 us for the residual after the barrier idles its SIMD's matrix pipe once its partner's
        // k-loop is through (measured: 318 us per layer with the loads behind the barrier)
        const int r = (wk - 2 * i) & 7;
        const int b_ep = tile * spt + r;
        const size_t o_ep = ((size_t)(b_ep < a.B ? b_ep : 0) * L + el) * a.C_out + co;
        f32x4 tb = {0.f, 0.f, 0.f, 0.f}, rs4 = {0.f, 0.f, 0.f, 0.f};
        if (r < spt) {
            if (a.tbias) tb = *(const f32x4*)(a.tbias + (size_t)(b_ep < a.B ? b_ep : 0) * a.tb_stride + co);
            if (a.res) rs4 = *(const f32x4*)(a.res + o_ep);
        }
        // ---------------------------------------------------------------- k-loop of this tile (window buffer `cur`)
        f32x4 acc[MS] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const f32x4* win = smem4 + cur * stage4 + boff;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int g = wk + it * WK, c16 = g / KS, ts = g - c16 * KS;
            const f32x4 bf = win[ts * RS4 + c16 * 4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < MS; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[it][m][e], bf[e], acc[m], 0, 0, 0);
        }
        WS_STAMP(1);
        // ---------------------------------------------------------------- K-partials -> reduction buffer `cur`
#pragma unroll
        for (int m = 0; m < MS; ++m) smem4[red_off4 + cur * red4 + (wk * NT + j) * MTP4 + m * 4 + q] = acc[m];
        // ---------------------------------------------------------------- next window -> the other buffer; fetch the one after it
        const int nxt = tile + kWsGroups;
        if (nxt < n_tiles) window_write(nxt, cur ^ 1);
        WS_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        WS_STAMP(3);
        // ---------------------------------------------------------------- epilogue: region r of this tile by duty wave (2 i + r) mod 8
        if (r < spt) {
            const int b = b_ep;
            const int n = r * L + el;
            const size_t o = o_ep;
            const int ri = red_off4 + cur * red4 + n * MTP4 + (ec >> 2);
            f32x4 v = smem4[ri];
#pragma unroll
            for (int k = 1; k < WK; ++k) v += smem4[ri + k * NT * MTP4];
            v += bi;
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
            const f32x4 d = v - mean;
            const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f);
            const float rstd = gn_rstd(var);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = mish(d[e] * rstd * ga[e] + be[e]);
            y += tb;
            y += rs4;
            if (b < a.B) *(f32x4*)(a.dst + o) = y;
        }
        WS_STAMP(4);
        // the window after next: requested AFTER the epilogue (hipcc waits for every outstanding load at the epilogue's first use of
        // the prefetched residual - requested before the barrier, these loads put a ~1.5 us round trip in front of every epilogue:
        // stamps of tools/ws_trace.py, 3.5 k cycles per duty epilogue); they have the whole next k-loop to land
        if (nxt + kWsGroups < n_tiles) window_load(nxt + kWsGroups);
    }
#undef WS_STAMP
}

}  // namespace mpdx
