// fused_bwd.hpp - whole-trajectory BACKWARD programs for the outer levels of TemporalUnet (round 6; training step).
//
// The forward pass of the outer levels runs as two whole-trajectory programs (fused_level.hpp); their backward pass was one launch per
// layer: an input-gradient ("dgrad") convolution with the Mish + GroupNorm backward of the Conv1dBlock below in its epilogue, each a
// latency chain of its own (operands through L2, staging, K-split reduction, epilogue loads - profiles/r06_train_iteration_trace.txt: 5-10 us
// per launch, back to back).  Here the chain of one trajectory runs in ONE workgroup out of LDS, the forward programs' way:
//
//   * one 4-wave workgroup = one trajectory; the gradient tensors (dU of a layer, the residual branch's G) live in zero-haloed
//     channel-last LDS buffers [L + 4 rows][C + pad]; an op = a stride-1 convolution of a dU buffer with the layer's DGRAD pack
//     (packedT: transposed + tap-flipped, train.hpp) - M = the layer's INPUT channels, N = L positions - on v_mfma_f32_16x16x4_f32,
//     tile ownership and accumulation order as fused_conv_op (a wave owns its tiles for the whole K; two accumulator chains);
//   * the dgrad packs are read AS THEY ARE (their [m16][c16][tap] block order is the k-loop's consumption order): a 16-slot register
//     ring per wave that runs on ACROSS the ops (in an op's last 16 steps a consumed slot is refilled with the next op's block of that slot: fused_conv_op's scheme);
//     static programs (BwdSeq): compile-time op sequence, shapes and LDS geometry (BwdGeomOf) - the first, generic op-list version kept every shape's address arithmetic
//     alive across a switch (256 VGPRs + 108 AGPRs, 92 us for the down program; 58 us now);
//   * epilogue = what the per-layer path does between two launches: + the folded residual 1x1's input gradient (second accumulator),
//     + the identity-residual branch's G (LDS), + the other consumers' gradient (global: the skip connection), then - if the tensor
//     below is a Conv1dBlock - its Mish + GroupNorm backward (statistics of the kept GroupNorm input re-derived from global memory,
//     reductions in registers / DPP / one LDS exchange, gn_mish_bwd_kernel's formulas) -> dU of the layer below into LDS (plain,
//     zero-stuffed for a stride-2 layer's dgrad, or decimated for a ConvTranspose's) AND into global memory (the operand of that layer's
//     weight-gradient GEMM, which runs in wgrad_multi_kernel behind the programs); per-trajectory channel sums for gamma / beta / bias /
//     time-bias gradients.
// Summation orders differ from the per-layer kernels' (whole K per wave instead of an 8-way K split): results agree to fp32 rounding, not bit for bit.
#pragma once
#include "fused_level.hpp"

#ifdef MPDX_DEV_HOOKS   // ablation bits of BwdArgs::dbg (MPDX_BWD_DBG): honoured in dev builds only (in the production kernel the tests cost 2.5 us per program)
#define MPDX_BWD_DBGBIT(a, bit) ((a).dbg & (bit))
#else
#define MPDX_BWD_DBGBIT(a, bit) (0)
#endif

namespace mpdx {

// ---- compile-time op shapes: (id, MODE, KS, C_in/16 of the dgrad conv = the layer's C_out/16, folded 1x1 C_in/16 or 0, C_out = the layer's C_in (or the half of a
// channel concat this op produces), L_out, GroupNorm backward of the Conv1dBlock below?).  MODE CONV_DOWN = the input gradient of a ConvTranspose1d(4, 2, 1):
// its 5-tap dgrad pack (a zero tap + the four transposed taps, train.hpp) applied at stride 2 - what the per-layer path computes at full resolution and decimates.
#define MPDX_BWD_SHAPES(X)                                                                                                            \
    X(0, CONV_S1, 3, 8, 0, 128, 16, 1) X(1, CONV_S1, 5, 8, 0, 128, 16, 1) X(2, CONV_S1, 5, 8, 8, 64, 16, 0)                           \
    X(3, CONV_S1, 3, 4, 0, 64, 32, 1) X(4, CONV_S1, 5, 4, 0, 64, 32, 1) X(5, CONV_S1, 5, 4, 4, 32, 32, 0)                             \
    X(6, CONV_S1, 3, 2, 0, 32, 64, 1) X(7, CONV_S1, 5, 2, 0, 32, 64, 1)                                                              \
    X(8, CONV_S1, 5, 2, 0, 32, 64, 0) X(9, CONV_DOWN, 5, 2, 0, 32, 32, 1) X(10, CONV_S1, 5, 2, 0, 32, 32, 1) X(11, CONV_S1, 5, 2, 2, 64, 32, 0) \
    X(12, CONV_DOWN, 5, 4, 0, 64, 16, 1) X(13, CONV_S1, 5, 4, 0, 64, 16, 1) X(14, CONV_S1, 5, 4, 4, 128, 16, 0)              \
    X(15, CONV_S1, 5, 0, 0, 32, 64, 1)   /* no convolution: the result is the global addend (final_conv[0]'s output gradient from the loss kernel) */ \
    X(16, CONV_S1, 5, 0, 0, 128, 16, 1)  /* no convolution: the innermost level of a THREE-level network has no Downsample1d; its blocks.1's output gradient is the addend */
inline int bwd_shape_id(int mode, int ks, int nc16, int rnc16, int cout, int L, int gn) {
#define X(id, M, K, N, R, CO, LO, G) if (mode == M && ks == K && nc16 == N && rnc16 == R && cout == CO && L == LO && gn == G) return id;
    MPDX_BWD_SHAPES(X)
#undef X
    return -1;
}

struct BwdOp {
    int shape;
    int src_off4, src_rs4;      // LDS: the convolution's source (a dU buffer)
    int rsrc_off4, rsrc_rs4;    // LDS: source of the folded 1x1 input gradient (the G buffer of the block's output), shapes with NCR > 0
    int add_off4, add_rs4;      // LDS: G of the identity-residual branch, added to the result (-1: none)
    int gadd;                   // global (float offset in `ws`): [B][L][C] tensor added to the result - the other consumers' gradient (-1: none)
    int gy_off4, gy_rs4;        // LDS destination of the result BEFORE the GroupNorm backward (the G the residual branch needs later; -1: none)
    int gy_g;                   // global copy of the same (the residual 1x1 convolution's dY for its weight gradient; -1: none)
    int dst_off4, dst_rs4;      // LDS destination of dU (-1: none)
    int dst_mode;               // 0: row l; 1: zero-stuffed (row 2 l of a 2 L-row buffer whose odd rows are zeroed: the next op is a stride-2 layer's dgrad)
    int out_g;                  // global destination of dU, dense [B][L][C] (-1: none)
    int wbase, rwbase;          // float offsets in packedT of the dgrad pack (and of the folded 1x1's)
    int pre_g;                  // global: the lower Conv1dBlock's GroupNorm input [B][L][C] (GN shapes)
    int gamma_f, beta_f;        // float offsets in `flat` of its gamma / beta
    int part_g;                 // global: per-trajectory channel sums [3][B][C]: sum(gm vhat) | sum(gm) | sum(dU)   (gamma / beta / conv-bias gradients)
    int dT_g;                   // global: + b * dT_stride + c <- sum over positions of the incoming gradient (the block's time-bias gradient; -1: none)
};
constexpr int kMaxBwdOps = 20;
struct BwdArgs {
    const float* packedT;
    const float* flat;
    float* ws;                  // every global offset above is relative to this
    const float* gin;           // program input: dense [B][Lin][Cin] gradient, staged into the first source buffer
    int in_off4, in_rs4, in_L, in_C, in_stuff;   // in_stuff: staged zero-stuffed (row 2 l), the buffer has 2 in_L + 4 rows
    int B, nops, dT_stride;
    int stat_off;               // LDS exchange area (floats)
    int dbg;                    // dev ablation mask (MPDX_BWD_DBG; results are wrong with any bit set): 1 no epilogue operand loads, 2 no global stores, 4 no GroupNorm backward
    BwdOp ops[kMaxBwdOps];
};

// ---- compile-time LDS geometry of the two static programs (train_host.hpp builds the same layout at run time and compares: a mismatch is an error).
// Five slots of kBwdSlot4 float4 (the largest buffer: 20 rows x (128 + 4) floats): IN, GB, DUA, DUB, GA; a buffer of C channels has rows of C / 4 + 1 float4.
#ifndef MPDX_BWD_PAD4
#define MPDX_BWD_PAD4 1
#endif
struct BwdGeomOp { int src_off4, src_rs4, rsrc_off4, rsrc_rs4, add_off4, add_rs4, gy_off4, gy_rs4, dst_off4, dst_rs4, dst_mode; };
constexpr int kBwdPad4 = MPDX_BWD_PAD4;   // float4 of row padding (dev: -DMPDX_BWD_PAD4=1 / 2)
constexpr int kBwdSlot4 = 68 * (8 + kBwdPad4) > 20 * (32 + kBwdPad4) ? 68 * (8 + kBwdPad4) : 20 * (32 + kBwdPad4);
constexpr int kBwdIN = 0, kBwdGB = 1, kBwdDUA = 2, kBwdDUB = 3, kBwdGA = 4;
constexpr int bwd_slot(int k) { return k * kBwdSlot4; }
constexpr BwdGeomOp bwd_down_geom(int i, bool first_noconv = false) {
    const int k = i < 5 ? 2 : (i < 10 ? 1 : 0), p = i - (i < 5 ? 0 : (i < 10 ? 5 : 10));
    const int C = 32 << k, r4 = C / 4 + kBwdPad4;
    BwdGeomOp g{0, 0, 0, 0, -1, 0, -1, 0, -1, 0, 0};
    if (p == 0) {
        g.src_off4 = bwd_slot(kBwdIN); g.src_rs4 = r4; g.gy_off4 = bwd_slot(kBwdGB); g.gy_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUA); g.dst_rs4 = r4;
        if (i == 0 && first_noconv) { g.src_off4 = 0; g.src_rs4 = 0; }   // (no source: the op's input is its global addend)
    }
    else if (p == 1) { g.src_off4 = bwd_slot(kBwdDUA); g.src_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUB); g.dst_rs4 = r4; }
    else if (p == 2) { g.src_off4 = bwd_slot(kBwdDUB); g.src_rs4 = r4; g.add_off4 = bwd_slot(kBwdGB); g.add_rs4 = r4; g.gy_off4 = bwd_slot(kBwdGA); g.gy_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUA); g.dst_rs4 = r4; }
    else if (p == 3) { g.src_off4 = bwd_slot(kBwdDUA); g.src_rs4 = r4; g.dst_off4 = k > 0 ? bwd_slot(kBwdDUB) : -1; g.dst_rs4 = r4; }
    else { g.src_off4 = bwd_slot(kBwdDUB); g.src_rs4 = r4; g.rsrc_off4 = bwd_slot(kBwdGA); g.rsrc_rs4 = r4; g.dst_off4 = bwd_slot(kBwdIN); g.dst_rs4 = (C / 2) / 4 + kBwdPad4; g.dst_mode = 1; }
    return g;
}
// the three-level network's down program WITH the two middle blocks in front (18 ops): M1 .. M5 = [GroupNorm backward of mid_block2.blocks.1 on the up
// program's gradient | dgrad + GroupNorm backward of mid_block2.blocks.0 | of mid_block1.blocks.1 (+ the identity residual's gradient) | of mid_block1.blocks.0 |
// of downs.2.1.blocks.1 (+ identity residual + the skip connection's gradient from global memory)], then ops 1 .. 13 of the program above
constexpr BwdGeomOp bwd_down_mid_geom(int i) {
    if (i >= 5) return bwd_down_geom(i - 4, true);
    constexpr int r4 = 32 + kBwdPad4;
    BwdGeomOp g{0, 0, 0, 0, -1, 0, -1, 0, -1, 0, 0};
    if (i == 0) { g.gy_off4 = bwd_slot(kBwdGB); g.gy_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUA); g.dst_rs4 = r4; }
    else if (i == 1 || i == 3) { g.src_off4 = bwd_slot(kBwdDUA); g.src_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUB); g.dst_rs4 = r4; }
    else {   // i == 2: + GB -> GA;  i == 4: + GA -> GB
        g.src_off4 = bwd_slot(kBwdDUB); g.src_rs4 = r4; g.add_off4 = bwd_slot(i == 2 ? kBwdGB : kBwdGA); g.add_rs4 = r4;
        g.gy_off4 = bwd_slot(i == 2 ? kBwdGA : kBwdGB); g.gy_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUA); g.dst_rs4 = r4;
    }
    return g;
}
constexpr BwdGeomOp bwd_up_geom(int i) {
    BwdGeomOp g{0, 0, 0, 0, -1, 0, -1, 0, -1, 0, 0};
    if (i == 0) { g.dst_off4 = bwd_slot(kBwdIN); g.dst_rs4 = 8 + kBwdPad4; return g; }   // the GroupNorm backward of final_conv[0] on the loss kernel's gradient -> IN
    if (i == 1) { g.src_off4 = bwd_slot(kBwdIN); g.src_rs4 = 8 + kBwdPad4; g.dst_off4 = bwd_slot(kBwdDUA); g.dst_rs4 = 8 + kBwdPad4; return g; }
    const int k = i < 8 ? 0 : 1, p = i - (k == 0 ? 2 : 8);
    const int C = k == 0 ? 32 : 64, r4 = C / 4 + kBwdPad4;
    if (p == 0) { g.src_off4 = bwd_slot(k == 0 ? kBwdDUA : kBwdIN); g.src_rs4 = r4; g.gy_off4 = bwd_slot(kBwdGB); g.gy_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUB); g.dst_rs4 = r4; }
    else if (p == 1) { g.src_off4 = bwd_slot(kBwdDUB); g.src_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUA); g.dst_rs4 = r4; }
    else if (p == 2) { g.src_off4 = bwd_slot(kBwdDUA); g.src_rs4 = r4; g.add_off4 = bwd_slot(kBwdGB); g.add_rs4 = r4; g.gy_off4 = bwd_slot(kBwdGA); g.gy_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUB); g.dst_rs4 = r4; }
    else if (p == 3) { g.src_off4 = bwd_slot(kBwdDUB); g.src_rs4 = r4; g.dst_off4 = bwd_slot(kBwdDUA); g.dst_rs4 = r4; }
    else {   // the two halves of the concat's gradient
        g.src_off4 = bwd_slot(kBwdDUA); g.src_rs4 = r4; g.rsrc_off4 = bwd_slot(kBwdGA); g.rsrc_rs4 = r4;
        if (p == 4 && k == 0) { g.dst_off4 = bwd_slot(kBwdIN); g.dst_rs4 = (2 * C) / 4 + kBwdPad4; }
    }
    return g;
}
template <int PROG, int I> struct BwdGeomOf { static constexpr bool has = true; static constexpr BwdGeomOp g = PROG == 0 ? bwd_down_geom(I) : (PROG == 2 ? bwd_down_geom(I, true) : (PROG == 3 ? bwd_down_mid_geom(I) : bwd_up_geom(I))); };
struct BwdGeomNone { static constexpr bool has = false; static constexpr BwdGeomOp g{0, 0, 0, 0, -1, 0, -1, 0, -1, 0, 0}; };
// host: does op `o` (as train_host.hpp laid it out) have the table's LDS geometry?
inline bool bwd_geom_matches(const BwdOp& o, const BwdGeomOp& g, bool has_rsrc) {
    return o.src_off4 == g.src_off4 && o.src_rs4 == g.src_rs4 && (!has_rsrc || (o.rsrc_off4 == g.rsrc_off4 && o.rsrc_rs4 == g.rsrc_rs4)) && o.add_off4 == g.add_off4 &&
           (g.add_off4 < 0 || o.add_rs4 == g.add_rs4) && o.gy_off4 == g.gy_off4 && (g.gy_off4 < 0 || o.gy_rs4 == g.gy_rs4) && o.dst_off4 == g.dst_off4 &&
           (g.dst_off4 < 0 || (o.dst_rs4 == g.dst_rs4 && o.dst_mode == g.dst_mode));
}

// request the first min(16, slen) blocks of an op's wave-stream.  Block r of the stream: pass p = r / tot (tile row ms + p * msw), block rr = r % tot of that
// row; rr < nblk: the convolution's pack, else the folded 1x1's.  All wave-uniform.
__device__ __forceinline__ void bwd_ring_request(f32x4 (&ring)[kFusedRing], __amdgpu_buffer_rsrc_t rs, int wbase, int rwbase, int ms, int msw, int nblk, int ncr, int slen,
                                                 unsigned lane_bytes) {
    const int tot = nblk + ncr;
#pragma unroll
    for (int p = 0; p < kFusedRing; ++p) {
        if (p < slen) {
            const int pass = p / tot, rr = p - pass * tot;
            const int row = ms + pass * msw;
            const int off = rr < nblk ? (wbase + (row * nblk + rr) * 256) * 4 : (rwbase + (row * ncr + (rr - nblk)) * 256) * 4;
            ring[p] = fused_ld_block(rs, off, 0, lane_bytes);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

struct BwdNext { int wbase, rwbase, msw, nblk, ncr, slen, msn; };   // what the ring needs of the NEXT op (runtime)

template <int MODE_, int KS_, int NC16_, int NCR_, int COUT_, int LOUT_, int GN_>
using BwdShape = FusedShape<MODE_, KS_, NC16_, NCR_, COUT_, LOUT_, GN_>;

// G: the op's LDS geometry as compile-time constants (static programs: every LDS address folds into `lane base + immediate`) or BwdGeomNone (the descriptor's fields)
template <class S, class G = BwdGeomNone>
__device__ __forceinline__ void fused_bwd_op(const BwdArgs& a, const BwdOp& opd, const BwdNext& nx, f32x4 (&ring)[kFusedRing], float* smem, int wave, int lane, int b) {
    constexpr int P = kFusedRing, DB = MPDX_FUSED_DB, NTW = S::NTW, NJ = S::NJ;
    constexpr int TOTD = S::TOT > 0 ? S::TOT : 1;   // (an op WITHOUT a convolution - S::TOT == 0: the program's first op, a GroupNorm backward on the staged gradient)
    BwdOp op = opd;
    if constexpr (G::has) {
        constexpr BwdGeomOp g = G::g;
        op.src_off4 = g.src_off4; op.src_rs4 = g.src_rs4; op.rsrc_off4 = g.rsrc_off4; op.rsrc_rs4 = g.rsrc_rs4; op.add_off4 = g.add_off4; op.add_rs4 = g.add_rs4;
        op.gy_off4 = g.gy_off4; op.gy_rs4 = g.gy_rs4; op.dst_off4 = g.dst_off4; op.dst_rs4 = g.dst_rs4; op.dst_mode = g.dst_mode;
    }
    static_assert(S::MODE == CONV_S1 || S::MODE == CONV_DOWN, "backward ops: stride-1 convolutions, or a dgrad pack applied at stride 2");
    f32x4* const sm4 = (f32x4*)smem;
    const int j = lane & 15, q = lane >> 4;
    const int ms = wave & (S::MSW - 1);
    const int nsg = wave / S::MSW;
    int ns[NTW], npos[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        ns[t] = (S::MP > 1) ? 0 : nsg * NTW + t;
        npos[t] = ns[t] * 16 + j;
    }
    int c0t[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) c0t[t] = ((S::MP > 1) ? ms + t * S::MSW : ms) * 16 + q * 4;
    // ---- global operands of the epilogue, requested BEFORE the k-loop (their latency hides under it)
    f32x4 uu[NTW], gad[NTW], gam[NTW], bet[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const size_t o = ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0t[t];
        if (MPDX_BWD_DBGBIT(a, 1)) { uu[t] = gad[t] = (f32x4){0.1f * (float)lane, 0.2f, 0.3f, 0.4f}; gam[t] = bet[t] = uu[t]; continue; }
        if (S::GN) {
            uu[t] = *(const f32x4*)(a.ws + op.pre_g + o);
            if (S::MP > 1 || t == 0) { gam[t] = *(const f32x4*)(a.flat + op.gamma_f + c0t[t]); bet[t] = *(const f32x4*)(a.flat + op.beta_f + c0t[t]); }
            else { gam[t] = gam[0]; bet[t] = bet[0]; }
        }
        if (op.gadd >= 0) gad[t] = *(const f32x4*)(a.ws + op.gadd + o);
    }
    const f32x4* brow[NJ];
    const f32x4* rrow[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
        constexpr int pad = S::KS / 2;
        const int l = ns[t] * 16 + j;
        brow[t] = sm4 + op.src_off4 + ((S::MODE == CONV_DOWN ? 2 * l : l) + 2 - pad) * op.src_rs4 + q;
        rrow[t] = sm4 + (S::NCR > 0 ? op.rsrc_off4 + (l + 2) * op.rsrc_rs4 + q : 0);
    }
    const int rs4 = op.src_rs4;
    const __amdgpu_buffer_rsrc_t wrs = fused_weights_rsrc(a.packedT);
    const unsigned lane_bytes = (unsigned)lane * 16u;
    // byte offset of stream block r of THIS wave (compile-time r; ms wave-uniform)
    auto blk_off = [&](int r) -> int {
        const int pass = r / TOTD, rr = r % TOTD;
        const int row = ms + pass * S::MSW;
        return rr < S::NBLK ? (op.wbase + (row * S::NBLK + rr) * 256) * 4 : (op.rwbase + (row * S::NCR + (rr - S::NBLK)) * 256) * 4;
    };
    // byte offset of block k (< 16) of the NEXT op's stream of this wave (runtime shape: wave-uniform scalar arithmetic; clamped to its last block)
    const int nx_tot = nx.nblk + nx.ncr, nx_ms = wave & (nx.msw - 1);
    auto nx_off = [&](int k) -> int {
        const int kk = k < nx.slen ? k : nx.slen - 1;
        const int pass = kk / nx_tot, rr = kk - pass * nx_tot;
        const int row = nx_ms + pass * nx.msw;
        return rr < nx.nblk ? (nx.wbase + (row * nx.nblk + rr) * 256) * 4 : (nx.rwbase + (row * nx.ncr + (rr - nx.nblk)) * 256) * 4;
    };
    auto read_b = [&](int r, int t) -> f32x4 {
        const int rr = r % TOTD;
        if (rr < S::NBLK) {
            const int c16 = rr / S::NTAP, ts = rr % S::NTAP;
            return brow[t][ts * rs4 + c16 * 4];
        }
        return rrow[t][(rr - S::NBLK) * 4];
    };
    f32x4 acc[NTW][2], racc[NTW][2];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t][0] = acc[t][1] = racc[t][0] = racc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bq[DB + 1][NJ];
#pragma unroll
    for (int r = 0; r < DB && r < S::SLEN; ++r)
#pragma unroll
        for (int t = 0; t < NJ; ++t) bq[r][t] = read_b(r, t);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < S::SLEN; ++r) {
        if (r + DB < S::SLEN) {
#pragma unroll
            for (int t = 0; t < NJ; ++t) bq[(r + DB) % (DB + 1)][t] = read_b(r + DB, t);
        }
        const f32x4 af = ring[r % P];
        const bool is_res = (r % TOTD) >= S::NBLK;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int tt = (S::NSEQ > 1) ? r / TOTD : t;
                f32x4& d = is_res ? racc[tt][e & 1] : acc[tt][e & 1];
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bq[r % (DB + 1)][t][e], d, 0, 0, 0);
            }
        // refill slot r % P: the block P further down this op's stream or - in the op's last P steps - the NEXT op's block of that slot: the ring runs on
        // across the ops (fused_conv_op's scheme), so a weight block has 16 blocks of MFMAs plus an epilogue to arrive
        if (r + P < S::SLEN) ring[r % P] = fused_ld_block(wrs, blk_off(r + P), 0, lane_bytes);
        else ring[r % P] = fused_ld_block(wrs, nx_off(r % P), 0, lane_bytes);
        if (S::SLEN < P) {   // slots this op never uses belong to the next op from the start
#pragma unroll
            for (int k = S::SLEN; k < P; ++k)
                if ((k - S::SLEN) % (S::SLEN > 0 ? S::SLEN : 1) == r) ring[k] = fused_ld_block(wrs, nx_off(k), 0, lane_bytes);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (S::SLEN == 0)   // nothing streamed: the next op's first blocks are requested here (they fly under the epilogue)
        bwd_ring_request(ring, wrs, nx.wbase, nx.rwbase, wave & (nx.msw - 1), nx.msw, nx.nblk, nx.ncr, nx.slen, lane_bytes);

    // ------------------------------------------------------------------ epilogue
    f32x4 gy[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        gy[t] = acc[t][0] + acc[t][1];
        if (S::NCR > 0) gy[t] += racc[t][0] + racc[t][1];
        if (op.add_off4 >= 0) gy[t] += sm4[op.add_off4 + (npos[t] + 2) * op.add_rs4 + (c0t[t] >> 2)];
        if (op.gadd >= 0) gy[t] += gad[t];
        if (op.gy_off4 >= 0) sm4[op.gy_off4 + (npos[t] + 2) * op.gy_rs4 + (c0t[t] >> 2)] = gy[t];
        if (op.gy_g >= 0 && !MPDX_BWD_DBGBIT(a, 2)) *(f32x4*)(a.ws + op.gy_g + ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0t[t]) = gy[t];
    }
    f32x4 y[NTW];
    float* const stat = smem + a.stat_off;
    if constexpr (S::GN) {
      if (MPDX_BWD_DBGBIT(a, 4)) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) y[t] = gy[t];
      } else {
        // Mish + GroupNorm backward of the Conv1dBlock below (gn_mish_bwd_kernel's formulas; reciprocals on v_rcp / v_rsq as the forward programs do):
        //   vhat = (u - mean) rstd;  gm = gy mish'(gamma vhat + beta);  dvh = gm gamma;  dU = rstd (dvh - mean(dvh) - vhat mean(dvh vhat))
        // per-trajectory channel sums: sum(gm vhat) [gamma], sum(gm) [beta], sum(dU) [conv bias], sum(gy) [time bias];
        //   sum_l dU[c] = rstd (gamma_c sum_l gm[c] - L s1 - s2 sum_l vhat[c])  - from the sums of gm and vhat, so that everything a partner wave
        //   has to contribute travels in ONE exchange.
        // LOCAL (C_out >= 64: the wave owns every position of its rows): reductions inside the wave (DPP rows, LDS crossbar).  C_out = 32 (two waves share a
        // tile row): two exchange rounds through LDS - (mean, M2) of the halves combined with Chan's formula; then (sum dvh, sum dvh vhat, channel sums).
        constexpr bool LOCAL = (S::MSW == kFusedWaves);
        constexpr int NG = (S::MP > 1) ? NTW : 1;
        constexpr int TPG = NTW / NG;
        auto rows_sum = [&](float x) -> float {
            if constexpr (LOCAL && S::RB >= 2)
                x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));   // lane ^ 16
            if constexpr (LOCAL && S::RB == 4)
                x += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, x)));
            return x;
        };
        auto wave_group_sum = [&](float part) -> float { return rows_sum(row_sum16(part)); };
        constexpr int NLOC = 64 * TPG * (LOCAL ? S::RB : 1);                 // elements one in-wave reduction covers
        constexpr float inv_loc = 1.0f / (float)NLOC;
        constexpr float inv_n = 1.0f / (float)(LOCAL ? NLOC : 2 * NLOC);     // elements of a GroupNorm region
        auto mish_grad_fast = [](float v) -> float {   // d/dv [v tanh(softplus(v))] with v_exp / v_rcp (mish_grad: two IEEE divisions per element)
            const float e = __builtin_amdgcn_exp2f(fminf(v, 20.0f) * 1.4426950408889634f);
            const float p1 = 1.0f + e, n = p1 * p1;
            const float th = (n - 1.0f) * __builtin_amdgcn_rcpf(n + 1.0f);
            const float sg = e * __builtin_amdgcn_rcpf(p1);
            return th + v * (1.0f - th * th) * sg;
        };
        const int pw = (nsg ^ 1) * S::MSW + ms;   // the partner wave (non-LOCAL)
        float mean_g[NG], rstd_g[NG];
        f32x4 d[NTW];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float sl = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) { const f32x4& x = uu[g * TPG + k]; sl += (x[0] + x[1]) + (x[2] + x[3]); }
            mean_g[g] = wave_group_sum(sl) * inv_loc;
            float ql = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int t = g * TPG + k;
                d[t] = uu[t] - mean_g[g];
                ql += (d[t][0] * d[t][0] + d[t][1] * d[t][1]) + (d[t][2] * d[t][2] + d[t][3] * d[t][3]);
            }
            float m2 = wave_group_sum(ql);
            if constexpr (!LOCAL) {   // round A: (mean, M2) of this wave's half <-> the partner's; Chan for two parts of NLOC elements
                if (j == 0) *(f32x2*)(stat + (wave * 4 + q) * 2) = (f32x2){mean_g[g], m2};
                lds_barrier();
                const f32x2 o = *(const f32x2*)(stat + (pw * 4 + q) * 2);
                const float dm = o[0] - mean_g[g];
                const float mean = 0.5f * (mean_g[g] + o[0]);
                m2 = (m2 + o[1]) + (0.5f * (float)NLOC) * (dm * dm);
                const float shift = mean_g[g] - mean;
#pragma unroll
                for (int k = 0; k < TPG; ++k) d[g * TPG + k] = d[g * TPG + k] + shift;
                mean_g[g] = mean;
            }
            rstd_g[g] = gn_rstd(m2 * inv_n);
        }
        f32x4 vh[NTW], gm[NTW], dvh[NTW];
        float p1_g[NG], p2_g[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int t = g * TPG + k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vh[t][e] = d[t][e] * rstd_g[g];
                    gm[t][e] = gy[t][e] * mish_grad_fast(vh[t][e] * gam[t][e] + bet[t][e]);
                    dvh[t][e] = gm[t][e] * gam[t][e];
                    p1 += dvh[t][e];
                    p2 += dvh[t][e] * vh[t][e];
                }
            }
            p1_g[g] = wave_group_sum(p1);
            p2_g[g] = wave_group_sum(p2);
        }
        // per-channel sums over the positions this wave holds: sum(gm vhat), sum(gm), sum(vhat), sum(gy)
        constexpr int NCH = (S::MP > 1) ? NTW : 1;   // distinct channel quads of this lane
        f32x4 cs[NCH][4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) cs[c][k4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NTW / NCH; ++k) {
                const int t = c * (NTW / NCH) + k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[c][0][e] += gm[t][e] * vh[t][e];
                    cs[c][1][e] += gm[t][e];
                    cs[c][2][e] += vh[t][e];
                    cs[c][3][e] += gy[t][e];
                }
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[c][k4][e] = row_sum16(cs[c][k4][e]);
        }
        if constexpr (!LOCAL) {   // round B: (sum dvh, sum dvh vhat) + the channel sums of this wave's positions <-> the partner's
            float* ex = stat + 64;   // [wave][q][2 + 16 floats -> 20]
            if (j == 0) {
                float* e0 = ex + (wave * 4 + q) * 20;
                *(f32x2*)e0 = (f32x2){p1_g[0], p2_g[0]};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) *(f32x4*)(e0 + 4 + 4 * k4) = cs[0][k4];
            }
            lds_barrier();
            const float* o0 = ex + (pw * 4 + q) * 20;
            const f32x2 o = *(const f32x2*)o0;
            p1_g[0] += o[0]; p2_g[0] += o[1];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) cs[0][k4] = cs[0][k4] + *(const f32x4*)(o0 + 4 + 4 * k4);
        }
        float s1_g[NG], s2_g[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { s1_g[g] = p1_g[g] * inv_n; s2_g[g] = p2_g[g] * inv_n; }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int g = (S::MP > 1) ? t : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[t][e] = rstd_g[g] * (dvh[t][e] - s1_g[g] - vh[t][e] * s2_g[g]);
        }
        if (j == 0 && (LOCAL || nsg == 0) && !MPDX_BWD_DBGBIT(a, 2)) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int t0 = c * (NTW / NCH);
                const int g = (S::MP > 1) ? c : 0;
                const int c0 = c0t[t0];
                const size_t po = (size_t)b * S::COUT + c0;
                const size_t BC = (size_t)a.B * S::COUT;
                f32x4 sdu;
#pragma unroll
                for (int e = 0; e < 4; ++e) sdu[e] = rstd_g[g] * (gam[t0][e] * cs[c][1][e] - (float)S::LOUT * s1_g[g] - s2_g[g] * cs[c][2][e]);
                *(f32x4*)(a.ws + op.part_g + po) = cs[c][0];
                *(f32x4*)(a.ws + op.part_g + BC + po) = cs[c][1];
                *(f32x4*)(a.ws + op.part_g + 2 * BC + po) = sdu;
                if (op.dT_g >= 0) *(f32x4*)(a.ws + op.dT_g + (size_t)b * a.dT_stride + c0) = cs[c][3];
            }
        }
      }
    } else {
#pragma unroll
        for (int t = 0; t < NTW; ++t) y[t] = gy[t];
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (op.dst_off4 >= 0) {
            const int row = (op.dst_mode == 1) ? 2 * npos[t] : npos[t];
            sm4[op.dst_off4 + (row + 2) * op.dst_rs4 + (c0t[t] >> 2)] = y[t];
            if (op.dst_mode == 1) sm4[op.dst_off4 + (row + 3) * op.dst_rs4 + (c0t[t] >> 2)] = (f32x4){0.f, 0.f, 0.f, 0.f};   // the stuffed zero row
        }
        if (op.out_g >= 0 && !MPDX_BWD_DBGBIT(a, 2)) *(f32x4*)(a.ws + op.out_g + ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0t[t]) = y[t];
    }
    {   // halo rows of the buffers this op defines (2 above, 2 below the interior rows)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const int tid = wave * 64 + lane;
        if (op.dst_off4 >= 0 && tid < 2 * op.dst_rs4) {
            const int rows = (op.dst_mode == 1) ? 2 * S::LOUT : S::LOUT;
            sm4[op.dst_off4 + tid] = z;
            sm4[op.dst_off4 + (rows + 2) * op.dst_rs4 + tid] = z;
        }
        if (op.gy_off4 >= 0 && tid < 2 * op.gy_rs4) {
            sm4[op.gy_off4 + tid] = z;
            sm4[op.gy_off4 + (S::LOUT + 2) * op.gy_rs4 + tid] = z;
        }
    }
    lds_barrier();
}

template <int ID> struct BwdShapeOf;
#define X(id, M, K, N, R, CO, LO, G) template <> struct BwdShapeOf<id> { using type = BwdShape<M, K, N, R, CO, LO, G>; };
MPDX_BWD_SHAPES(X)
#undef X

__device__ __forceinline__ void fused_bwd_prologue(const BwdArgs& a, const BwdNext& n0, f32x4 (&ring)[kFusedRing], float* smem, int tid, int lane, int wave, int b) {
    f32x4* const sm4 = (f32x4*)smem;
    // the first op's ring, the input gradient into its source buffer (plain or zero-stuffed), zeros everywhere else in that buffer
    bwd_ring_request(ring, fused_weights_rsrc(a.packedT), n0.wbase, n0.rwbase, wave & (n0.msw - 1), n0.msw, n0.nblk, n0.ncr, n0.slen, (unsigned)lane * 16u);
    const int c4n = a.in_C >> 2;
    const int rows = (a.in_stuff ? 2 * a.in_L : a.in_L) + 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < rows * a.in_rs4; i += kFusedThreads) sm4[a.in_off4 + i] = z;
    lds_barrier();
    const int n4 = a.gin ? a.in_L * c4n : 0;   // (gin == null: the first op takes its input as a global addend; only the zeros above)
    for (int i = tid; i < n4; i += kFusedThreads) {
        const int l = i / c4n, c = i - l * c4n;
        sm4[a.in_off4 + ((a.in_stuff ? 2 * l : l) + 2) * a.in_rs4 + c] = *(const f32x4*)(a.gin + ((size_t)b * a.in_L + l) * a.in_C + 4 * c);
    }
    lds_barrier();
}

// ---- STATIC programs: the op sequence as a compile-time list of shape ids - no op loop, no shape switch (the generic kernel below keeps every shape's address
// arithmetic alive across a 8-way switch: 256 VGPRs + 108 AGPRs; fused_level.hpp's round-2 lesson), the next op's ring constants fold
template <int PROG, int... SH>
struct BwdSeq {
    static constexpr int N = sizeof...(SH);
    static constexpr int ids[sizeof...(SH)] = {SH...};
    template <int I>
    __device__ static __forceinline__ BwdNext next_desc(const BwdArgs& a) {
        constexpr int J = I < N ? I : N - 1;   // behind the last op: its own first blocks again (harmless loads, never used)
        using S = typename BwdShapeOf<ids[J]>::type;
        return BwdNext{a.ops[J].wbase, a.ops[J].rwbase, S::MSW, S::NBLK, S::NCR, S::SLEN, S::MSn};
    }
    template <int I>
    __device__ static __forceinline__ void run_from(const BwdArgs& a, f32x4 (&ring)[kFusedRing], float* smem, int wave, int lane, int b) {
        if constexpr (I < N) {
            fused_bwd_op<typename BwdShapeOf<ids[I]>::type, BwdGeomOf<PROG, I>>(a, a.ops[I], next_desc<I + 1>(a), ring, smem, wave, lane, b);
            run_from<I + 1>(a, ring, smem, wave, lane, b);
        }
    }
};
template <class SEQ>
__global__ __launch_bounds__(kFusedThreads) void fused_bwd_program_kernel(const BwdArgs a) {
    warm_kernarg<(int)sizeof(BwdArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    f32x4 ring[kFusedRing];
    fused_bwd_prologue(a, SEQ::template next_desc<0>(a), ring, smem, tid, lane, wave, b);
    SEQ::template run_from<0>(a, ring, smem, wave, lane, b);
}
// the backward pass of downs[0..2] of the standard network (train_host.hpp run_down_program)
using BwdSeqDown3 = BwdSeq<0, 0, 1, 1, 1, 2, 3, 4, 4, 4, 5, 6, 7, 7, 7>;
// the same for the THREE-level network (dim_mults (1, 2, 4)): its innermost level has no Downsample1d - the first op is the GroupNorm backward alone
using BwdSeqDown3Last = BwdSeq<2, 16, 1, 1, 1, 2, 3, 4, 4, 4, 5, 6, 7, 7, 7>;
// ... with mid_block2 and mid_block1 (128 channels on 16 positions, as the innermost level) in front: everything below the up program in ONE launch
using BwdSeqDown3Mid = BwdSeq<3, 16, 1, 1, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 5, 6, 7, 7, 7>;
// the backward pass of final_conv[0] + the two outer up levels (run_up_program; the same shapes in the three- and the four-level network)
using BwdSeqUp2 = BwdSeq<1, 15, 8, 9, 10, 10, 10, 11, 11, 12, 13, 13, 13, 14, 14>;

}  // namespace mpdx
