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
// Variants (template parameters): NC16 = padded input channels / 16, MT = channel tile = GroupNorm group size (32 at C_out = 256,
// 16 at C_out = 128), R1 = the block's residual 1x1 convolution folded in (second weight set, second accumulator over the SAME LDS
// window, bias-only epilogue by a third duty wave): conv_ws_kernel<32, 16, true> replaces the paired launch of ups[0]'s first
// ResidualTemporalBlock (512 -> 128 k5 + 512 -> 128 1x1 on the channel concat, temporal_unet.py:158-160), which ran 16-position tiles
// at 43 TF/s.
//
// Numerics.  Same k-group -> wave assignment, same accumulation order inside a wave, same K-partial order, same epilogue code as
// conv_block_kernel<CONV_S1, 5, EPI_GN_MISH, 32, NT, 1, 8>: the outputs are BIT-IDENTICAL (tests/test_gpu_parity.py checks that).
#pragma once
#include "conv_block.hpp"

namespace mpdx {

constexpr int kWsGroups = 32;    // position groups: (C_out / MT) channel tiles x 32 workgroups
constexpr int kWsThreads = 512;

// NS = position sub-tiles per tile.  NS = 1: 16 positions = two GroupNorm regions = two duty waves, i.e. two of the four SIMDs carry an
// epilogue in a tile and the other two wait for them at the barrier.  NS = 2 (very large batches): 32 positions = FOUR regions, one duty
// wave on EVERY SIMD (waves 0 .. 3), the A fragments feed two B fragments each, half as many barriers and hand-overs per position; the
// K-partials then have ONE buffer (2 x 50 KB of windows + 37 KB) and a second barrier per tile in front of their writes.
// The kernels are written for the inner levels' geometry - L = 8 positions per trajectory, row stride C_in + 8 floats (what
// pick_row_stride finds for them), C_out = 8 GroupNorm groups of MT channels - as COMPILE-TIME constants (round 4: every LDS address of
// the k-loop is `lane base + immediate`, the window addresses advance by one constant per tile; the round-3 form spent 91 VALU
// instructions per wave and tile on runtime strides, 64-bit products and per-element bounds selects - tools/isa_census.py).
constexpr int kWsL = 8;
template <int NC16> constexpr int ws_row_stride() { return NC16 * 16 + 8; }

template <int NC16, int MT, bool R1, int NS = 1>
inline size_t conv_ws_lds_bytes(int L, int rs) {
    const size_t stage = (size_t)(16 * NS / L) * (L + 4) * rs * sizeof(float);
    const size_t red = (size_t)8 * 16 * NS * (MT + 4) * sizeof(float);
    return 2 * stage + (NS == 1 ? 2 : 1) * red * (R1 ? 2 : 1);
}

// TBRES: what the Conv1dBlock's epilogue adds behind Mish - 1: the time-bias row (blocks[0]), 2: a residual tensor (blocks[1]), 0: neither
// (compile-time: the runtime form selected zeros per element, 8 v_cndmask + 2 packed adds on the duty wave's critical path)
template <int NC16, int MT, bool R1, int NS = 1, int TBRES = 0>
__global__ __launch_bounds__(kWsThreads) void conv_ws_kernel(const ConvArgs a, const ConvArgs a2) {
    constexpr int KS = 5, PAD = 2, MS = MT / 16, NT = 16 * NS, WK = 8, NG = NC16 * KS, NIT = NG / WK, NIT2 = NC16 / WK;
    constexpr int NRED = NS == 1 ? 2 : 1;                  // K-partial buffers
    constexpr int MTP4 = (MT + 4) / 4;
    constexpr int EPL = MT / 8;                            // epilogue elements per lane: a region = MT channels x 8 positions over 64 lanes
    static_assert(NG % WK == 0 && NC16 % WK == 0, "k-groups split evenly over the 8 waves");
    static_assert(MT == 16 || MT == 32, "channel tile = GroupNorm group of 16 or 32 channels");
    typedef float fvec __attribute__((ext_vector_type(EPL)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* const smem4 = (f32x4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x / kWsGroups, p = blockIdx.x % kWsGroups;
    constexpr int L = kWsL, LP = L + 2 * PAD, RS4 = ws_row_stride<NC16>() / 4, C_OUT = 8 * MT;   // (the launcher checks a.L_out, a.rs, a.C_out)
    constexpr int spt = NT / L;                            // trajectories per tile (2 or 4)
    constexpr int stage4 = spt * LP * RS4;                 // float4 per window buffer
    constexpr int red4 = WK * NT * MTP4;                   // float4 per reduction buffer
    constexpr int red_off4 = 2 * stage4, red2_off4 = red_off4 + NRED * red4;
    const int n_tiles = a.n_tiles_n;
    constexpr int c4n = NC16 * 4;

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
    f32x4 af2[R1 ? NIT2 : 1][MS];
    if constexpr (R1) {
        const float* wbase2 = a2.wp + (size_t)(mt * MS) * nc16 * 256 + lane * 4;
#pragma unroll
        for (int it = 0; it < NIT2; ++it)
#pragma unroll
            for (int m = 0; m < MS; ++m) af2[it][m] = *(const f32x4*)(wbase2 + (size_t)(m * nc16 + wk + it * WK) * 256);
    }
    // ---- halo rows of both window buffers (never overwritten by the window writes)
    for (int idx = tid; idx < 2 * spt * 2 * PAD * c4n; idx += kWsThreads) {
        const int c4 = idx % c4n, hr = idx / c4n;                              // hr over [buffer][trajectory][4 halo rows]
        const int k = hr & 3, s = (hr >> 2) % spt, buf = (hr >> 2) / spt;
        const int lp = (k < PAD) ? k : (L + k);
        smem4[buf * stage4 + (s * LP + lp) * RS4 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // window of a tile: NT positions x cin/4 float4 (1024 at C_in = 256, 2048 at 512).  A lane's float4 sits at a fixed place of the
    // window: its global address advances by ONE per-lane constant per tile (wstep: spt * L * channels of its source tensor); only a
    // tile that reaches beyond the batch (the last one) takes the clamped / masked path (wave-uniform test).
    constexpr int SB = (NT * NC16 * 4) / kWsThreads;
    static_assert(SB >= 1 && (NT * NC16 * 4) % kWsThreads == 0, "window divides over the threads");
    // The loads are BUFFER loads: resource = the source tensor (SGPR x4; with a channel concat the lane's float4 comes from src1 or src2 -
    // the same for a whole wave, because c1 / 4 is a multiple of 64: the launcher checks it), voffset = the lane's byte offset inside a
    // tile (one VGPR per float4, set up once), soffset = tile * bytes per tile (ONE scalar multiply per tile): no VALU instruction per load
    // (round 3: 64-bit products and clamps per load, 32 VALU per wave and tile).
    int wdst[SB], ws_[SB], wvoff[SB];
    const int first_w = __builtin_amdgcn_readfirstlane((((tid % c4n) * 4) < a.c1) ? 1 : 0);   // wave-uniform (see above)
    const float* const wsrc_t = first_w ? a.src1 : a.src2;   // this wave's source tensor
    const int cs = first_w ? a.c1 : a.c2;                    // its channels
    const int wtraj = L * cs;                                // floats per trajectory
    const int wtile_bytes = spt * wtraj * 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc_t, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int idx = tid + u * kWsThreads;
        const int rowi = idx / c4n, c4 = idx % c4n;
        const int s = rowi / L, li = rowi % L;
        wdst[u] = (s * LP + li + PAD) * RS4 + c4;
        ws_[u] = s;
        const int c = c4 * 4;
        wvoff[u] = ((s * L + li) * cs + (first_w ? c : c - a.c1)) * 4;
    }
    f32x4 wv[SB];
    auto window_load = [&](int tile) {   // unconditional loads from valid addresses (a conditional load serialises the queue)
        if ((tile + 1) * spt <= a.B) {
            const int soff = tile * wtile_bytes;
#pragma unroll
            for (int u = 0; u < SB; ++u) wv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff[u], soff, 0));
        } else {
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int over = tile * spt + ws_[u] - (a.B - 1);   // trajectories beyond the batch re-read the last one (masked at the write)
                wv[u] = *(const f32x4*)((const char*)wsrc_t + (size_t)tile * wtile_bytes + wvoff[u] - (size_t)(over > 0 ? over : 0) * wtraj * 4);
            }
        }
    };
    auto window_write = [&](int tile, int buf) {
        if ((tile + 1) * spt <= a.B) {
#pragma unroll
            for (int u = 0; u < SB; ++u) smem4[buf * stage4 + wdst[u]] = wv[u];
        } else {
#pragma unroll
            for (int u = 0; u < SB; ++u) smem4[buf * stage4 + wdst[u]] = (tile * spt + ws_[u] < a.B) ? wv[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    // lane's B row in a window (tile-local position n = j), and its column in the reduction buffer
    const int j = lane & 15, q = lane >> 4;
    int boff[NS];   // position sub-tile ns: tile-local position n = ns * 16 + j
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) boff[ns] = (((ns * 16 + j) / L) * LP + (j & (L - 1))) * RS4 + q;

    // B-fragment addresses of this wave's k-groups in window buffer 0 (float4 units), computed ONCE: the k-group -> (channel chunk, tap)
    // split depends on the wave, so inside the loop every read cost a VALU add (28 per wave and tile); the tile loop below is unrolled
    // by two so that the window buffer is a compile-time constant and lands in the ds_read's immediate offset.
    int baddr[NIT][NS], baddr2[R1 ? NIT2 : 1][NS];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int g = wk + it * WK, c16 = g / KS, ts = g - c16 * KS;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) baddr[it][ns] = boff[ns] + ts * RS4 + c16 * 4;
    }
    if constexpr (R1) {
#pragma unroll
        for (int it = 0; it < NIT2; ++it)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) baddr2[it][ns] = boff[ns] + PAD * RS4 + (wk + it * WK) * 4;
    }

    // ---- epilogue operands of the region this lane would serve (channels are fixed per lane: loaded once)
    const int e0 = lane * EPL;
    const int el = e0 / MT, ec = e0 % MT;                   // region = MT channels x 8 positions: lane -> (position, EPL channels)
    const int co = mt * MT + ec;
    const fvec bi = *(const fvec*)(a.bias + co), ga = *(const fvec*)(a.gamma + co), be = *(const fvec*)(a.beta + co);
    const float inv_re = 1.0f / (float)(MT * 8);
    f32x4 bias2 = {0.f, 0.f, 0.f, 0.f};   // R1: the residual conv's bias chunk of this lane (loaded once: a load inside the epilogue is a round trip on the tile's critical path)
    if constexpr (R1) bias2 = *(const f32x4*)(a2.bias + mt * MT + (lane % (MT / 4)) * 4);

    int tile = p;
    if (tile < n_tiles) { window_load(tile); window_write(tile, 0); }
    if (tile + kWsGroups < n_tiles) window_load(tile + kWsGroups);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // dev tool (-DMPDX_DEV_HOOKS, mpdx_layer_trace): stamps of workgroup 0, ALL EIGHT waves (4 slots each), kept in registers and written
    // when the kernel ends (a stamp stored on the spot is a VMEM write the next vmcnt(0) waits for - it would time itself):
    // tile 8 top | tile 8 k-loop issued | tile 8 barrier passed | tile 9 top
    long long* const trp = (MPDX_TRACE_PTR(a.trace) && blockIdx.x == 0 && lane == 0) ? a.trace + wk * 4 : nullptr;
    long long st_[4] = {0, 0, 0, 0};
#define WS_STAMP(k) do { if (trp) { if (i == 8 && (k) == 0) st_[0] = (long long)__builtin_readcyclecounter(); if (i == 8 && (k) == 1) st_[1] = (long long)__builtin_readcyclecounter(); \
                                    if (i == 8 && (k) == 3) st_[2] = (long long)__builtin_readcyclecounter(); if (i == 9 && (k) == 0) st_[3] = (long long)__builtin_readcyclecounter(); } } while (0)
    auto tile_body = [&](auto cur_c, const int i, const int tile) {
        constexpr int cur = decltype(cur_c)::value;
        WS_STAMP(0);
        // this wave's epilogue duty for the tile (region r = trajectory r of the tile): its global operands are requested BEFORE the
        // k-loop - a duty wave that waits ~1 us for the residual after the barrier idles its SIMD's matrix pipe once its partner's
        // k-loop is through (measured: 318 us per layer with the loads behind the barrier)
        // duties go to waves 0 .. 3 only: the OLDER wave of each SIMD (see below); two regions per tile rotate over them, four take all
        const int r = NS == 1 ? (wk < 4 ? ((wk - 2 * i) & 3) : 7) : (wk <= 4 ? wk : 7);
        const int b_ep = tile * spt + r;
        const size_t o_ep = ((size_t)(b_ep < a.B ? b_ep : 0) * L + el) * C_OUT + co;
        // UNCONDITIONAL loads (every wave, from valid addresses; zeros are selected in the epilogue): as conditional loads into
        // zero-initialised registers they made hipcc put s_waitcnt vmcnt(0) HERE, at the top of every tile - every wave then sat out the
        // round trip of the window loads it had issued a moment ago, at the bottom of the previous tile, with the matrix pipes idle
        // (~1.5 k of the 7.3 k cycles of a tile: tools/ws_trace.py, the gap between `epilogue done` and the next `tile top`).
        const float* const tb_p = TBRES == 1 ? a.tbias + (size_t)(b_ep < a.B ? b_ep : 0) * a.tb_stride + co : (TBRES == 2 ? a.res + o_ep : a.bias + co);
        fvec tb = *(const fvec*)tb_p;   // the ONE operand added behind Mish (TBRES == 0: a dummy load, never added)
        __builtin_amdgcn_sched_barrier(0);   // requested HERE (hipcc would sink them behind the k-loop, next to the barrier)
        // ---------------------------------------------------------------- k-loop of this tile (window buffer `cur`)
        f32x4 acc[NS][MS], acc2[NS][MS];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int m = 0; m < MS; ++m) acc[ns][m] = acc2[ns][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4* win = smem4 + cur * stage4;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 bf[NS];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) bf[ns] = win[baddr[it][ns]];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                    for (int m = 0; m < MS; ++m) acc[ns][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[it][m][e], bf[ns][e], acc[ns][m], 0, 0, 0);
            // the next tile's window goes to the OTHER buffer (free since the previous tile's closing barrier) in the middle of the
            // k-loop - its loads were issued at the end of the previous tile - instead of between the k-loop and the barrier
            if (it == NIT / 2 && tile + kWsGroups < n_tiles) window_write(tile + kWsGroups, cur ^ 1);
        }
        if constexpr (R1) {   // the residual 1x1 conv reads the centre row of the same window
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                f32x4 bf[NS];
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) bf[ns] = win[baddr2[it][ns]];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                        for (int m = 0; m < MS; ++m) acc2[ns][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af2[it][m][e], bf[ns][e], acc2[ns][m], 0, 0, 0);
            }
        }
        WS_STAMP(1);
        // ---------------------------------------------------------------- K-partials -> reduction buffer(s)
        const int rcur = NRED == 2 ? cur : 0;
        if constexpr (NRED == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the duty waves are through with the previous tile's partials
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                smem4[red_off4 + rcur * red4 + (wk * NT + ns * 16 + j) * MTP4 + m * 4 + q] = acc[ns][m];
                if constexpr (R1) smem4[red2_off4 + rcur * red4 + (wk * NT + ns * 16 + j) * MTP4 + m * 4 + q] = acc2[ns][m];
            }
        // ---------------------------------------------------------------- next window -> the other buffer
        const int nxt = tile + kWsGroups;
        WS_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        WS_STAMP(3);
        // (every wave "uses" the two operands here, where they have long arrived and nothing else is in flight: a load still pending on
        //  their registers at the loop's back edge - the six waves without a duty never read them - costs the same vmcnt(0) at the top
        //  of the next tile; behind the epilogue the wait would include the duty waves' store)
        asm volatile("" :: "v"(tb));
        // ---------------------------------------------------------------- epilogue: region r of this tile by duty wave (2 i + r) mod 8
        if (r < spt) {
            // raised issue priority for the duty: the tile's critical path is THIS wave (epilogue, then its k-loop), while its SIMD
            // partner is already in the next k-loop.  Stamps inside the epilogue (all eight waves, tools/ws_trace.py): 0.5 k cycles for
            // the eight partial reads, 0.8-1.1 k for the statistics, 1.8-2.2 k for Mish + store - a chain of dependent VALU instructions
            // next to a partner that streams MFMAs advances at ~20 cycles per instruction.  With the priority and the duties on the
            // older wave of each SIMD: 302 -> 293 us per 256->256 launch at B = 6400.
            __builtin_amdgcn_s_setprio(3);
            const int n = r * L + el;
            const float* red = smem + (size_t)(red_off4 + rcur * red4) * 4;
            fvec v = *(const fvec*)(red + (size_t)n * (MT + 4) + ec);
#pragma unroll
            for (int k = 1; k < WK; ++k) v += *(const fvec*)(red + (size_t)(k * NT + n) * (MT + 4) + ec);
            v += bi;
            float s1 = 0.f;
            if constexpr (EPL == 4) s1 = (v[0] + v[1]) + (v[2] + v[3]); else s1 = v[0] + v[1];
            const float mean = wave_sum(s1) * inv_re;
            const fvec d = v - mean;
            float s2 = 0.f;
            if constexpr (EPL == 4) s2 = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]); else s2 = d[0] * d[0] + d[1] * d[1];
            const float var = wave_sum(s2) * inv_re;
            const float rstd = gn_rstd(var);
            fvec y;
#pragma unroll
            for (int e = 0; e < EPL; ++e) y[e] = mish_nosel(d[e] * rstd * ga[e] + be[e]);
            if constexpr (TBRES != 0) y += tb;
            if (b_ep < a.B) *(fvec*)(a.dst + o_ep) = y;
            __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (R1) {   // bias-only epilogue of the residual conv: duty wave (2 i + spt) mod 8
            if (r == spt) {
                const float* red = smem + (size_t)(red2_off4 + rcur * red4) * 4;
                static_assert(64 % (MT / 4) == 0, "a lane keeps its channel chunk over the passes");
                for (int idx = lane; idx < NT * (MT / 4); idx += 64) {
                    const int n = idx / (MT / 4), c = (idx - n * (MT / 4)) * 4;
                    const int s = n / L, l = n & (L - 1), b = tile * spt + s;
                    f32x4 v = *(const f32x4*)(red + (size_t)n * (MT + 4) + c);
#pragma unroll
                    for (int k = 1; k < WK; ++k) v += *(const f32x4*)(red + (size_t)(k * NT + n) * (MT + 4) + c);
                    v += bias2;   // (c == (lane % (MT / 4)) * 4 in every pass)
                    if (b < a2.B) *(f32x4*)(a2.dst + ((size_t)b * L + l) * C_OUT + mt * MT + c) = v;
                }
            }
        }
        WS_STAMP(4);
        // the window after next: requested AFTER the epilogue (hipcc waits for every outstanding load at the epilogue's first use of
        // the prefetched residual - requested before the barrier, these loads put a ~1.5 us round trip in front of every epilogue:
        // stamps of tools/ws_trace.py, 3.5 k cycles per duty epilogue); they have the whole next k-loop to land
        if (nxt + kWsGroups < n_tiles) window_load(nxt + kWsGroups);
    };
    for (int i = 0; tile < n_tiles; i += 2) {   // two tiles per trip: window buffers 0 and 1 at compile time
        tile_body(std::integral_constant<int, 0>{}, i, tile);
        tile += kWsGroups;
        if (tile >= n_tiles) break;
        tile_body(std::integral_constant<int, 1>{}, i + 1, tile);
        tile += kWsGroups;
    }
#undef WS_STAMP
    if (trp) { trp[0] = st_[0]; trp[1] = st_[1]; trp[2] = st_[2]; trp[3] = st_[3]; }
}

}  // namespace mpdx
