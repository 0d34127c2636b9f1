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
//     ring per wave, refilled inside the op; the NEXT op's first blocks are requested right behind the k-loop, so they fly under the epilogue;
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

namespace mpdx {

// ---- compile-time op shapes: (id, KS, C_in/16 of the dgrad conv = the layer's C_out/16, folded 1x1 C_in/16 or 0, C_out = the layer's C_in, L, GroupNorm backward?)
#define MPDX_BWD_SHAPES(X)                                                                                                  \
    X(0, 3, 8, 0, 128, 16, 1) X(1, 5, 8, 0, 128, 16, 1) X(2, 5, 8, 8, 64, 16, 0)                                            \
    X(3, 3, 4, 0, 64, 32, 1) X(4, 5, 4, 0, 64, 32, 1) X(5, 5, 4, 4, 32, 32, 0)                                              \
    X(6, 3, 2, 0, 32, 64, 1) X(7, 5, 2, 0, 32, 64, 1)
inline int bwd_shape_id(int ks, int nc16, int rnc16, int cout, int L, int gn) {
#define X(id, K, N, R, CO, LO, G) if (ks == K && nc16 == N && rnc16 == R && cout == CO && L == LO && gn == G) return id;
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
constexpr int kMaxBwdOps = 16;
struct BwdArgs {
    const float* packedT;
    const float* flat;
    float* ws;                  // every global offset above is relative to this
    const float* gin;           // program input: dense [B][Lin][Cin] gradient, staged into the first source buffer
    int in_off4, in_rs4, in_L, in_C, in_stuff;   // in_stuff: staged zero-stuffed (row 2 l), the buffer has 2 in_L + 4 rows
    int B, nops, dT_stride;
    int stat_off;               // LDS exchange area (floats)
    BwdOp ops[kMaxBwdOps];
};

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

template <int KS_, int NC16_, int NCR_, int COUT_, int LOUT_, int GN_>
using BwdShape = FusedShape<CONV_S1, KS_, NC16_, NCR_, COUT_, LOUT_, GN_>;

template <class S>
__device__ __forceinline__ void fused_bwd_op(const BwdArgs& a, const BwdOp& op, const BwdNext& nx, f32x4 (&ring)[kFusedRing], float* smem, int wave, int lane, int b) {
    constexpr int P = kFusedRing, DB = MPDX_FUSED_DB, NTW = S::NTW, NJ = S::NJ;
    static_assert(S::MODE == CONV_S1, "backward ops are stride-1 convolutions");
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
        brow[t] = sm4 + op.src_off4 + (l + 2 - pad) * op.src_rs4 + q;
        rrow[t] = sm4 + (S::NCR > 0 ? op.rsrc_off4 + (l + 2) * op.rsrc_rs4 + q : 0);
    }
    const int rs4 = op.src_rs4;
    const __amdgpu_buffer_rsrc_t wrs = fused_weights_rsrc(a.packedT);
    const unsigned lane_bytes = (unsigned)lane * 16u;
    // byte offset of stream block r of THIS wave (compile-time r; ms wave-uniform)
    auto blk_off = [&](int r) -> int {
        const int pass = r / S::TOT, rr = r % S::TOT;
        const int row = ms + pass * S::MSW;
        return rr < S::NBLK ? (op.wbase + (row * S::NBLK + rr) * 256) * 4 : (op.rwbase + (row * S::NCR + (rr - S::NBLK)) * 256) * 4;
    };
    auto read_b = [&](int r, int t) -> f32x4 {
        const int rr = r % S::TOT;
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
        const bool is_res = (r % S::TOT) >= S::NBLK;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int tt = (S::NSEQ > 1) ? r / S::TOT : t;
                f32x4& d = is_res ? racc[tt][e & 1] : acc[tt][e & 1];
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bq[r % (DB + 1)][t][e], d, 0, 0, 0);
            }
        if (r + P < S::SLEN) ring[r % P] = fused_ld_block(wrs, blk_off(r + P), 0, lane_bytes);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the NEXT op's first blocks: requested here, they fly under this op's epilogue (the ring is free from now on)
    if (nx.slen > 0) bwd_ring_request(ring, wrs, nx.wbase, nx.rwbase, wave & (nx.msw - 1), nx.msw, nx.nblk, nx.ncr, nx.slen, lane_bytes);

    // ------------------------------------------------------------------ epilogue
    f32x4 gy[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        gy[t] = acc[t][0] + acc[t][1];
        if (S::NCR > 0) gy[t] += racc[t][0] + racc[t][1];
        if (op.add_off4 >= 0) gy[t] += sm4[op.add_off4 + (npos[t] + 2) * op.add_rs4 + (c0t[t] >> 2)];
        if (op.gadd >= 0) gy[t] += gad[t];
        if (op.gy_off4 >= 0) sm4[op.gy_off4 + (npos[t] + 2) * op.gy_rs4 + (c0t[t] >> 2)] = gy[t];
        if (op.gy_g >= 0) *(f32x4*)(a.ws + op.gy_g + ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0t[t]) = gy[t];
    }
    f32x4 y[NTW];
    float* const stat = smem + a.stat_off;
    if constexpr (S::GN) {
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
        constexpr int NLOC = 64 * TPG * (LOCAL ? S::RB : 1);
        constexpr float inv_n = 1.0f / (float)(LOCAL ? NLOC : 2 * NLOC);   // elements of a GroupNorm region
        // group sum of a per-lane partial: in-wave (LOCAL) or with the partner wave through LDS (two waves share a tile row: C_out = 32)
        int xr = 0;   // exchange round (alternating halves of the exchange area: a wave may run one round ahead of its partner)
        auto group_sum = [&](float part) -> float {
            float s = rows_sum(row_sum16(part));
            if constexpr (!LOCAL) {
                float* ex = stat + (xr & 1) * 64;
                if (j == 0) ex[(nsg * S::MSW + ms) * 4 + q] = s;
                lds_barrier();
                s += ex[((nsg ^ 1) * S::MSW + ms) * 4 + q];
                ++xr;
            }
            return s;
        };
        float mean_g[NG], rstd_g[NG];
        f32x4 d[NTW];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float sl = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) { const f32x4& x = uu[g * TPG + k]; sl += (x[0] + x[1]) + (x[2] + x[3]); }
            mean_g[g] = group_sum(sl) * inv_n;
            float ql = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int t = g * TPG + k;
                d[t] = uu[t] - mean_g[g];
                ql += (d[t][0] * d[t][0] + d[t][1] * d[t][1]) + (d[t][2] * d[t][2] + d[t][3] * d[t][3]);
            }
            const float var = group_sum(ql) * inv_n;
            rstd_g[g] = 1.0f / sqrtf(var + 1e-5f);   // (gn_mish_bwd_kernel's form)
        }
        f32x4 vh[NTW], gm[NTW], dvh[NTW];
        float s1_g[NG], s2_g[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int t = g * TPG + k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vh[t][e] = d[t][e] * rstd_g[g];
                    gm[t][e] = gy[t][e] * mish_grad(vh[t][e] * gam[t][e] + bet[t][e]);
                    dvh[t][e] = gm[t][e] * gam[t][e];
                    p1 += dvh[t][e];
                    p2 += dvh[t][e] * vh[t][e];
                }
            }
            s1_g[g] = group_sum(p1) * inv_n;
            s2_g[g] = group_sum(p2) * inv_n;
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int g = (S::MP > 1) ? t : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[t][e] = rstd_g[g] * (dvh[t][e] - s1_g[g] - vh[t][e] * s2_g[g]);
        }
        // ---- per-trajectory channel sums over the positions: sum(gm vhat), sum(gm), sum(dU), sum(gy)
        // a lane's four channels x its tiles of the SAME channels (NJ position tiles) -> DPP row sum over the 16 positions
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
                    cs[c][2][e] += y[t][e];
                    cs[c][3][e] += gy[t][e];
                }
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[c][k4][e] = row_sum16(cs[c][k4][e]);
        }
        if constexpr (!LOCAL) {   // the partner wave (other position tiles of the same channels): through the exchange area, [wave][q][4 kinds][4 channels]
            float* ex = stat + 128;
            if (j == 0) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) *(f32x4*)(ex + ((wave * 4 + q) * 4 + k4) * 4) = cs[0][k4];
            }
            lds_barrier();
            const int pw = (nsg ^ 1) * S::MSW + ms;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f32x4 o = *(const f32x4*)(ex + ((pw * 4 + q) * 4 + k4) * 4);
                // fixed order (position-tile group 0 first): both partners compute the same bits
                cs[0][k4] = nsg == 0 ? cs[0][k4] + o : o + cs[0][k4];
            }
        }
        if (j == 0 && (LOCAL || nsg == 0)) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int c0 = c0t[c * (NTW / NCH)];
                const size_t po = (size_t)b * S::COUT + c0;
                const size_t BC = (size_t)a.B * S::COUT;
                *(f32x4*)(a.ws + op.part_g + po) = cs[c][0];
                *(f32x4*)(a.ws + op.part_g + BC + po) = cs[c][1];
                *(f32x4*)(a.ws + op.part_g + 2 * BC + po) = cs[c][2];
                if (op.dT_g >= 0) *(f32x4*)(a.ws + op.dT_g + (size_t)b * a.dT_stride + c0) = cs[c][3];
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
        if (op.out_g >= 0) *(f32x4*)(a.ws + op.out_g + ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0t[t]) = y[t];
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

// the generic op-list kernel: walks the program's ops (runtime shapes)
__global__ __launch_bounds__(kFusedThreads) void fused_bwd_kernel(const BwdArgs a) {
    warm_kernarg<(int)sizeof(BwdArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* const sm4 = (f32x4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    f32x4 ring[kFusedRing];
    auto next_of = [&](int oi) -> BwdNext {
        BwdNext n{0, 0, 1, 1, 0, 0, 1};
        if (oi < a.nops) {
            const BwdOp& o = a.ops[oi];
            switch (o.shape) {
#define X(id, K, N, R, CO, LO, G) case id: { using S = BwdShape<K, N, R, CO, LO, G>; n = BwdNext{o.wbase, o.rwbase, S::MSW, S::NBLK, S::NCR, S::SLEN, S::MSn}; } break;
                MPDX_BWD_SHAPES(X)
#undef X
                default: break;
            }
        }
        return n;
    };
    {   // prologue: the first op's ring, the input gradient into its source buffer (plain or zero-stuffed), zeros everywhere else in that buffer
        const BwdNext n0 = next_of(0);
        bwd_ring_request(ring, fused_weights_rsrc(a.packedT), n0.wbase, n0.rwbase, wave & (n0.msw - 1), n0.msw, n0.nblk, n0.ncr, n0.slen, (unsigned)lane * 16u);
        const int c4n = a.in_C >> 2;
        const int rows = (a.in_stuff ? 2 * a.in_L : a.in_L) + 4;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < rows * a.in_rs4; i += kFusedThreads) sm4[a.in_off4 + i] = z;
        lds_barrier();
        const int n4 = a.in_L * c4n;
        for (int i = tid; i < n4; i += kFusedThreads) {
            const int l = i / c4n, c = i - l * c4n;
            sm4[a.in_off4 + ((a.in_stuff ? 2 * l : l) + 2) * a.in_rs4 + c] = *(const f32x4*)(a.gin + ((size_t)b * a.in_L + l) * a.in_C + 4 * c);
        }
        lds_barrier();
    }
    for (int oi = 0; oi < a.nops; ++oi) {
        const BwdOp op = a.ops[oi];
        const BwdNext nx = next_of(oi + 1);
        switch (op.shape) {
#define X(id, K, N, R, CO, LO, G) case id: fused_bwd_op<BwdShape<K, N, R, CO, LO, G>>(a, op, nx, ring, smem, wave, lane, b); break;
            MPDX_BWD_SHAPES(X)
#undef X
            default: break;
        }
    }
}

}  // namespace mpdx
