// conv_wsn.hpp - weight-stationary persistent kernel WITHOUT a K split, for the 128-input-channel layers of the innermost up level at
// large batch (round 5):
//   Conv1dBlock 128 -> 128 on L = 8 (ups[0]: resnet1.blocks[1], resnet2.blocks[0], resnet2.blocks[1]; mpd/models/layers/layers.py:276-293,
//   343-355 via temporal_unet.py:158-160) and Upsample1d(128) = ConvTranspose1d(128, 128, 4, 2, 1), 8 -> 16 positions (layers.py:267-273).
//
// Why not conv_ws_kernel<8, ...>.  conv_ws.hpp splits K over the 8 waves of a workgroup; at C_in = 128 that leaves 5 k-groups = 20-40 MFMAs
// per wave and tile, which do not cover the tile's barrier, K-partial hand-over and window exchange (measured in round 3: 142.7 us against
// the per-layer kernel's 95.9 us per 128 -> 128 launch at B = 6400).  With K = 640 the WHOLE weight slice of a 16-channel GroupNorm group
// fits one wave's registers (40 A fragments = 160 VGPRs at two waves per SIMD: 256 VGPRs each), so here
//   * a wave owns output tile = (16 channels = one GroupNorm group) x (16 positions = 2 trajectories) for the whole K: no K-partials
//     through LDS, GroupNorm regions are wave-local;
//   * the 8 waves of a workgroup work on 8 DIFFERENT position tiles out of wave-private LDS windows: no workgroup barrier anywhere
//     after the prologue - a wave's epilogue (VALU) runs under its SIMD partner's k-loop (MFMA) by plain wave scheduling;
//   * the next tile's window is requested (8 coalesced 1-KiB buffer loads: the two trajectories are contiguous in HBM) at the top of the
//     k-loop and written to the wave's window after it.
// 256 workgroups = 8 channel tiles x 32 position groups, b = mt * 32 + p: the channel tiles of a position group share an XCD (b % 8),
// a window is fetched once per L2.
//
// Numerics: BIT-IDENTICAL to conv_block_kernel<..., 1, 8> (and so to every other path).  That kernel's wave wk accumulates k-groups
// wk, wk + 8, ... in one accumulator and the epilogue adds the 8 partials in the order 0 .. 7; here ONE wave keeps the 8 chains apart
// (4 at a time: independent MFMA chains hide the 40-cycle dependent-accumulator latency) and adds them in the same order; the
// GroupNorm epilogue is the per-layer kernel's code on the same element -> lane mapping (the tile goes through a wave-private LDS
// patch to get there), so the statistics are summed in the same tree.
#pragma once
#include "conv_block.hpp"

namespace mpdx {

constexpr int kWsnGroups = 32;     // position groups per channel tile
constexpr int kWsnThreads = 512;
constexpr int kWsnC = 128;         // input channels (= output channels of the layers this serves)

template <int MODE> constexpr int wsn_rs() { return kWsnC + (MODE == CONV_UPT ? 4 : 8); }   // GeoUp8 / GeoL8 row strides
template <int MODE> constexpr int wsn_lp() { return MODE == CONV_UPT ? 10 : 12; }           // staged rows per trajectory (8 + 2 * pad)
template <int MODE> constexpr size_t conv_wsn_lds_bytes() {
    return (size_t)8 * (2 * wsn_lp<MODE>() * wsn_rs<MODE>() + (MODE == CONV_UPT ? 0 : 16 * 20 + 64)) * sizeof(float);
}

// MODE: CONV_S1 (k5, GroupNorm + Mish; TBRES 1: + time-bias row, 2: + residual tensor, 0: nothing) or CONV_UPT (k4 s2 p1, bias only)
template <int MODE, int TBRES>
__global__ __launch_bounds__(kWsnThreads) void conv_wsn_kernel(const ConvArgs a) {
    constexpr int NC16 = kWsnC / 16, L = 8;
    constexpr int NTAP = MODE == CONV_UPT ? 2 : 5, NSLOT = MODE == CONV_UPT ? 4 : 5, NCLS = MODE == CONV_UPT ? 2 : 1, PAD = MODE == CONV_UPT ? 1 : 2;
    constexpr int NG = NC16 * NTAP, WK = 8, NIT = NG / WK;     // k-groups; the K split this kernel reproduces inside one wave
    constexpr int LP = wsn_lp<MODE>(), RS4 = wsn_rs<MODE>() / 4;
    constexpr int WIN4 = 2 * LP * RS4;                         // float4 per wave window (2 trajectories)
    constexpr int PATCH4 = MODE == CONV_UPT ? 0 : 16 * 5 + 16; // float4 per wave epilogue patch: [16 positions][16 + 4 channels] | bias, gamma, beta rows [16] each
    constexpr int C_OUT = kWsnC, L_OUT = MODE == CONV_UPT ? 16 : 8;
    static_assert(NG % WK == 0, "k-groups split evenly over the 8 emulated K-split chains");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* const smem4 = (f32x4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x / kWsnGroups, p = blockIdx.x % kWsnGroups;
    f32x4* const win = smem4 + wv * WIN4;                                  // this wave's window
    f32x4* const patch = smem4 + 8 * WIN4 + wv * PATCH4;                   // this wave's epilogue patch
    const int n_pairs = (a.B + 1) >> 1;                                    // wave tiles = pairs of trajectories
    // wave tile t of this wave: pairs (p + 32 i) * 8 + wv, i = 0, 1, ...: the 8 waves of a workgroup (and, at the same time, the same
    // waves of the 7 other channel tiles on this XCD) read 16 consecutive trajectories
    const int t_step = kWsnGroups * 8;
    int t = p * 8 + wv;

    // ---- this wave's weights, all k-groups, for the whole launch
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 0x7fffffff, 0x00020000);
    const int wtile = mt * NC16 * NSLOT * 1024;
    f32x4 af[NG][NCLS];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c16 = g / NTAP, ts = g - c16 * NTAP;
#pragma unroll
        for (int pc = 0; pc < NCLS; ++pc) {
            const int slot = MODE == CONV_UPT ? pc * 2 + ts : ts;
            af[g][pc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, wtile + (c16 * NSLOT + slot) * 1024, 0));
        }
    }
    // ---- halo rows of the window (conv padding; never overwritten)
    for (int idx = lane; idx < 2 * 2 * PAD * (kWsnC / 4); idx += 64) {
        const int c4 = idx % (kWsnC / 4), hr = idx / (kWsnC / 4);   // hr over [trajectory][2 * PAD halo rows]
        const int k = hr % (2 * PAD), s = hr / (2 * PAD);
        const int lp = k < PAD ? k : L + k;
        win[(s * LP + lp) * RS4 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // window of a wave tile: 2 trajectories x 8 rows x 128 channels = 2048 contiguous floats of the source tensor: float4 u * 64 + lane
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src1, 0, 0x7fffffff, 0x00020000);
    constexpr int SB = 2 * L * (kWsnC / 4) / 64;   // 8 loads per lane: float4 u * 64 + lane = row 2 u + (lane >> 5), column lane & 31
    // its place in the window: (trajectory u / 4, row 2 (u % 4) + (lane >> 5)) -> ONE lane-dependent base + a compile-time offset per load
    const int wbase = (lane >> 5) * RS4 + (lane & 31);
    f32x4 wreg[SB];
    const int last_b = a.B - 1;
    auto window_load = [&](int tile) {   // (a second trajectory beyond the batch re-reads the last one: its outputs are never stored)
        const int b0 = tile * 2;
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int s = u / (SB / 2);
            const int b = b0 + s <= last_b ? b0 + s : last_b;
            wreg[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, lane * 16 + (u % (SB / 2)) * 1024, b * (L * kWsnC * 4), 0));
        }
    };
    auto window_write = [&]() {
#pragma unroll
        for (int u = 0; u < SB; ++u) win[wbase + ((u / (SB / 2)) * LP + 2 * (u % (SB / 2)) + PAD) * RS4] = wreg[u];
    };

    // B-fragment base of this lane: MFMA column j = lane & 15 -> (trajectory j >> 3, input position j & 7), k sub-row q = lane >> 4
    const int j = lane & 15, q = lane >> 4;
    const int boff = ((j >> 3) * LP + (j & 7) + (MODE == CONV_UPT ? PAD : 0)) * RS4 + q;

    // ---- epilogue constants of this lane
    // CONV_S1: the per-layer kernel's region mapping (region = 16 channels x 8 positions over 64 lanes: lane -> position lane >> 3,
    // channels (lane & 7) * 2, + 1)
    const int el = lane >> 3, ec = (lane & 7) * 2, co = mt * 16 + ec;
    f32x4 bi4 = {0.f, 0.f, 0.f, 0.f};
    float* const prm = (float*)(patch + 16 * 5);   // this wave's copy of bias | gamma | beta of its 16 channels (read per tile: 6 VGPRs the k-loop needs)
    if constexpr (MODE == CONV_UPT) bi4 = *(const f32x4*)(a.bias + mt * 16 + q * 4);
    else {
        if (lane < 48) prm[lane] = (lane < 16 ? a.bias : lane < 32 ? a.gamma : a.beta)[mt * 16 + (lane & 15)];
        if (TBRES == 1 && lane >= 48) prm[lane] = a.tbias[mt * 16 + (lane & 15)];   // the step's time-bias row (ONE row per launch: the launcher checks tb_stride == 0)
    }

    if (t < n_pairs) { window_load(t); window_write(); }
    for (; t < n_pairs; t += t_step) {
        const int b0 = t * 2;
        const bool more = t + t_step < n_pairs;
        if (more) window_load(t + t_step);   // lands during the k-loop
        __builtin_amdgcn_sched_barrier(0);
        // ------------------------------------------------------------------ k-loop: chains wk = 0 .. 7 (k-groups wk, wk + 8, ...), four at a time
        f32x4 v[NCLS];
        constexpr int CH = 2 / NCLS;   // chains in flight x parity classes = 2: consecutive MFMAs alternate between two accumulators (a chain's own
                                       // MFMAs are dependent: 40-cycle accumulator latency against a 32-cycle issue slot); two is what the registers hold
        auto bload = [&](int c0, int it, f32x4 (&bf)[CH][NCLS]) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int g = c0 + c + it * WK, c16 = g / NTAP, ts = g - c16 * NTAP;
#pragma unroll
                for (int pc = 0; pc < NCLS; ++pc) {
                    // conv: staged row (l + tap) holds input l + tap - PAD; convT: even outputs read inputs (m, m - 1), odd ones (m, m + 1)
                    const int roff = MODE == CONV_UPT ? (ts == 0 ? 0 : (pc == 0 ? -1 : 1)) : ts;
                    bf[c][pc] = win[boff + roff * RS4 + c16 * 4];
                }
            }
        };
        f32x4 bfr[2][CH][NCLS];   // B fragments one step ahead (step = the CH x NCLS chains' k-groups of one `it`): their LDS latency hides under the MFMAs
        bload(0, 0, bfr[0]);
#pragma unroll
        for (int c0 = 0; c0 < WK; c0 += CH) {
            f32x4 acc[CH][NCLS];
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int pc = 0; pc < NCLS; ++pc) acc[c][pc] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int step = (c0 / CH) * NIT + it;
                if (it + 1 < NIT) bload(c0, it + 1, bfr[(step + 1) & 1]);
                else if (c0 + CH < WK) bload(c0 + CH, 0, bfr[(step + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int pc = 0; pc < NCLS; ++pc)
                            acc[c][pc] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c0 + c + it * WK][pc][e], bfr[step & 1][c][pc][e], acc[c][pc], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int pc = 0; pc < NCLS; ++pc) {   // the K-partials in the per-layer kernel's order: ((((p0 + p1) + p2) + p3) + p4) ...
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    if (c0 + c == 0) v[pc] = acc[c][pc]; else v[pc] += acc[c][pc];
                }
            }
        }
        // the residual tensor the epilogue adds behind Mish: requested HERE (before the k-loop it would hold 4 more registers through it - the
        // kernel sits at the 256-register limit of two waves per SIMD); its round trip hides under the window write and the statistics
        f32x2 tb[2] = {{0.f, 0.f}, {0.f, 0.f}};
        if constexpr (MODE == CONV_S1 && TBRES == 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int b = b0 + s <= last_b ? b0 + s : last_b;
                tb[s] = *(const f32x2*)(a.res + ((size_t)b * L + el) * C_OUT + co);
            }
        }
        // the next tile's window replaces this one (LDS operations of a wave complete in order: the k-loop's reads are through)
        if (more) window_write();
        if constexpr (MODE == CONV_UPT) {
            // bias, store: lane (j, q) holds channels q * 4 .. + 3 of input position m = j & 7 -> output positions 2 m (class 0), 2 m + 1 (class 1)
            const int b = b0 + (j >> 3);
            if (b <= last_b) {
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    *(f32x4*)(a.dst + ((size_t)b * L_OUT + 2 * (j & 7) + pc) * C_OUT + mt * 16 + q * 4) = v[pc] + bi4;
            }
        } else {
            // tile -> patch [position n = j][16 + 4 floats], then one pass of the per-layer epilogue per trajectory (GroupNorm region)
            patch[j * 5 + q] = v[0];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int b = b0 + s;
                f32x2 x = *(const f32x2*)((const float*)patch + (s * L + el) * 20 + ec);
                const f32x2 bi = *(const f32x2*)(prm + ec), ga = *(const f32x2*)(prm + 16 + ec), be = *(const f32x2*)(prm + 32 + ec);
                x += bi;
                const float mean = wave_sum(x[0] + x[1]) * (1.0f / 128.0f);
                const f32x2 d = x - mean;
                const float var = wave_sum(d[0] * d[0] + d[1] * d[1]) * (1.0f / 128.0f);
                const float rstd = gn_rstd(var);
                f32x2 y;
#pragma unroll
                for (int e = 0; e < 2; ++e) y[e] = mish_nosel(d[e] * rstd * ga[e] + be[e]);
                if constexpr (TBRES == 1) y += *(const f32x2*)(prm + 48 + ec);
                if constexpr (TBRES == 2) y += tb[s];
                if (b <= last_b) *(f32x2*)(a.dst + ((size_t)b * L + el) * C_OUT + co) = y;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// conv_wsp_kernel - the same idea for the first ResidualTemporalBlock of downs[3] at large batch: Conv1dBlock 128 -> 256 (k5, GroupNorm(8):
// groups of 32 channels, + time bias) TOGETHER with the block's residual 1x1 convolution 128 -> 256 on the same input
// (temporal_unet.py:141-150 via layers.py:323-355), which ran as conv_pair_kernel<32, 64, GeoL8<8>> at 59 % of the fp32 MFMA peak.
//
// A GroupNorm group is 32 channels = two MFMA row tiles, and one wave's registers hold the weights of ONE of them (40 A fragments of the k5
// conv + 8 of the 1x1 = 192 VGPRs), so a tile belongs to a PAIR of waves: both read the SAME LDS window (2 trajectories, double-buffered
// per pair), each computes its 16 channels for the whole K (k5: the 8 K-split chains of the per-layer kernel, two at a time, summed in its
// order; 1x1: its 8 single-k-group partials in order), both put their halves into the pair's LDS patch, and after ONE workgroup barrier
// each wave runs the per-layer GroupNorm epilogue for ONE of the two trajectories (region = 32 channels x 8 positions, f32x4 per lane) and
// stores the residual conv's rows of that trajectory.  Workgroup = 2 pairs = 256 threads at <= 256 VGPRs: TWO workgroups per CU, whose
// tile phases drift apart - one workgroup's epilogues run under the other's k-loops.  512 workgroups = 8 channel tiles x 64 position
// groups, b = mt * 64 + p (the channel tiles of a position group share an XCD).  Outputs BIT-IDENTICAL to the per-layer pair.
constexpr int kWspGroups = 64;
constexpr int kWspThreads = 256;
constexpr size_t conv_wsp_lds_bytes() { return ((size_t)2 * (2 * 2 * wsn_lp<CONV_S1>() * wsn_rs<CONV_S1>() + 2 * 16 * 36) + 5 * 32) * sizeof(float); }

__global__ __launch_bounds__(kWspThreads, 2) void conv_wsp_kernel(const ConvArgs a, const ConvArgs a2) {
    constexpr int NC16 = kWsnC / 16, L = 8, NTAP = 5, PAD = 2, NG = NC16 * NTAP, WK = 8, NIT = NG / WK;
    constexpr int LP = wsn_lp<CONV_S1>(), RS4 = wsn_rs<CONV_S1>() / 4, WIN4 = 2 * LP * RS4;
    constexpr int PT4 = 16 * 9;            // float4 per patch: [16 positions][32 + 4 channels]
    constexpr int C_OUT = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* const smem4 = (f32x4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wv >> 1, hf = wv & 1;                                   // pair of the workgroup, channel half / epilogue trajectory of the pair
    const int mt = blockIdx.x / kWspGroups, p = blockIdx.x % kWspGroups;
    f32x4* const win = smem4 + pr * 2 * WIN4;                              // the pair's two window buffers
    f32x4* const patch = smem4 + 2 * 2 * WIN4 + pr * 2 * PT4;              // the pair's patches: k5 tile | 1x1 tile
    const int n_pairs = (a.B + 1) >> 1;
    const int t_step = kWspGroups * 2;
    int t = p * 2 + pr;

    // ---- this wave's weights: its 16-channel half of the group, every k-group of both convolutions
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a2.wp, 0, 0x7fffffff, 0x00020000);
    const int m16 = mt * 2 + hf;
    f32x4 af[NG], af2[NC16];
#pragma unroll
    for (int g = 0; g < NG; ++g) af[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, (m16 * NC16 * NTAP + g) * 1024, 0));
#pragma unroll
    for (int g = 0; g < NC16; ++g) af2[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs2, lane * 16, (m16 * NC16 + g) * 1024, 0));
    // ---- halo rows of both window buffers (this wave: trajectory hf of each buffer)
    for (int idx = lane; idx < 2 * 2 * PAD * (kWsnC / 4); idx += 64) {
        const int c4 = idx % (kWsnC / 4), hr = idx / (kWsnC / 4);   // hr over [buffer][4 halo rows]
        const int k = hr % (2 * PAD), buf = hr / (2 * PAD);
        const int lp = k < PAD ? k : L + k;
        win[buf * WIN4 + (hf * LP + lp) * RS4 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // window of a tile: this wave fetches trajectory hf (8 rows x 128 channels = 1024 contiguous floats: float4 u * 64 + lane = row 2 u + (lane >> 5))
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src1, 0, 0x7fffffff, 0x00020000);
    constexpr int SB = L * (kWsnC / 4) / 64;   // 4 loads per lane
    const int wbase = (hf * LP + (lane >> 5) + PAD) * RS4 + (lane & 31);
    f32x4 wreg[SB];
    const int last_b = a.B - 1;
    auto window_load = [&](int tile) {
        const int b = tile * 2 + hf <= last_b ? tile * 2 + hf : last_b;
#pragma unroll
        for (int u = 0; u < SB; ++u) wreg[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, lane * 16 + u * 1024, b * (L * kWsnC * 4), 0));
    };
    auto window_write = [&](int buf) {
#pragma unroll
        for (int u = 0; u < SB; ++u) win[buf * WIN4 + wbase + 2 * u * RS4] = wreg[u];
    };
    const int j = lane & 15, q = lane >> 4;
    const int boff = ((j >> 3) * LP + (j & 7)) * RS4 + q;
    // ---- epilogue constants: region = 32 channels x 8 positions over 64 lanes: lane -> position lane >> 3, channels (lane & 7) * 4 .. + 3
    const int el = lane >> 3, ec = (lane & 7) * 4, co = mt * 32 + ec;
    // bias | gamma | beta | time-bias row (ONE per launch: the launcher checks tb_stride == 0) | 1x1 bias of the group's 32 channels: staged in
    // LDS once and read per tile (20 VGPRs the k-loop needs: the kernel sits at the 256-register limit of two workgroups per CU)
    float* const prm = smem + (2 * 2 * WIN4 + 2 * 2 * PT4) * 4;
    if (tid < 160) {
        const int row = tid >> 5, c = mt * 32 + (tid & 31);
        prm[tid] = (row == 0 ? a.bias : row == 1 ? a.gamma : row == 2 ? a.beta : row == 3 ? a.tbias : a2.bias)[c];
    }

    if (t < n_pairs) { window_load(t); window_write(0); }
    __syncthreads();
    // (the trip count is that of pair 0 - the two pairs of a workgroup meet at the barriers, and the last tile may exist for pair 0 only)
    for (int i = 0; t - pr < n_pairs; t += t_step, ++i) {
        const int cur = i & 1;
        const bool active = t < n_pairs, more = t + t_step < n_pairs;
        if (more) window_load(t + t_step);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* const w = win + cur * WIN4;
        if (active) {
        // ------------------------------------------------------------------ k5: chains wk = 0 .. 7 (k-groups wk, wk + 8, ...), two at a time
        f32x4 v;
        {
            auto bload = [&](int c0, int it, f32x4 (&bf)[2]) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int g = c0 + c + it * WK, c16 = g / NTAP, ts = g - c16 * NTAP;
                    bf[c] = w[boff + ts * RS4 + c16 * 4];
                }
            };
            f32x4 bfr[2][2];
            bload(0, 0, bfr[0]);
#pragma unroll
            for (int c0 = 0; c0 < WK; c0 += 2) {
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int step = (c0 / 2) * NIT + it;
                    if (it + 1 < NIT) bload(c0, it + 1, bfr[(step + 1) & 1]);
                    else if (c0 + 2 < WK) bload(c0 + 2, 0, bfr[(step + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c0 + c + it * WK][e], bfr[step & 1][c][e], acc[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (c0 == 0) v = acc[0]; else v += acc[0];
                v += acc[1];
            }
        }
        patch[j * 9 + hf * 4 + q] = v;                     // this half's rows of the k5 tile
        // ------------------------------------------------------------------ 1x1: its 8 single-k-group partials (centre row of the same window), in order
        {
            f32x4 v2;
#pragma unroll
            for (int g0 = 0; g0 < NC16; g0 += 2) {
                f32x4 bf[2], acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int c = 0; c < 2; ++c) bf[c] = w[boff + PAD * RS4 + (g0 + c) * 4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af2[g0 + c][e], bf[c][e], acc[c], 0, 0, 0);
                if (g0 == 0) v2 = acc[0]; else v2 += acc[0];
                v2 += acc[1];
            }
            patch[PT4 + j * 9 + hf * 4 + q] = v2;
        }
        }
        if (more) window_write(cur ^ 1);                   // (that buffer was last read before the previous tile's barrier)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ------------------------------------------------------------------ epilogues of trajectory hf of the pair's tile (the per-layer kernels' code)
        if (active) {
            const int b = t * 2 + hf;
            const f32x4 bi = *(const f32x4*)(prm + ec), ga = *(const f32x4*)(prm + 32 + ec), be = *(const f32x4*)(prm + 64 + ec);
            const f32x4 tb = *(const f32x4*)(prm + 96 + ec), bi2 = *(const f32x4*)(prm + 128 + ec);
            f32x4 x = patch[(hf * L + el) * 9 + (lane & 7)];
            x += bi;
            const float mean = wave_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 256.0f);
            const f32x4 d = x - mean;
            const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f);
            const float rstd = gn_rstd(var);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = mish_nosel(d[e] * rstd * ga[e] + be[e]);
            y += tb;
            f32x4 r2 = patch[PT4 + (hf * L + el) * 9 + (lane & 7)];
            r2 += bi2;
            if (b <= last_b) {
                *(f32x4*)(a.dst + ((size_t)b * L + el) * C_OUT + co) = y;
                *(f32x4*)(a2.dst + ((size_t)b * L + el) * C_OUT + co) = r2;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the patches are free for the next tile
    }
}

}  // namespace mpdx
