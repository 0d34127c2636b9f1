// fused_level.hpp - whole-trajectory fused programs for the outer levels of TemporalUnet (round-2 design).
//
// Why.  At B=100 the per-layer path is bound by launch boundaries and per-kernel fixed phases.  At the
// outer resolutions one workgroup can own ALL channels of a trajectory (C*L = 2048 floats, C <= 64), so GroupNorm statistics
// stay local and a whole run of layers executes out of LDS in ONE launch:
//     downs[i] (i = 0, 1):  ResidualTemporalBlock x2 -> Downsample1d                     (temporal_unet.py:141-150)
//     ups[j]   (last two):  cat(x, skip) -> ResidualTemporalBlock x2 -> Upsample1d       (temporal_unet.py:158-165)
//     and, after the last Upsample1d: final_conv (Conv1dBlock -> Conv1d 1x1) + the DDPM posterior step
//                                                            (temporal_unet.py:167, diffusion_model_base.py:121-155)
// The arithmetic per layer is the same as conv_block.hpp (same packed weights, same fp32 MFMA, GroupNorm, Mish).
//
// Structure.  One workgroup (8 waves) = one trajectory; LDS holds the activation buffers in the zero-haloed channel-last
// layout [L+4 rows][C + pad] (placed by live range on the host).  A small op list (kernel argument) is walked; every conv is
// M = C_out (all), N = L positions, K = C_in*taps as (C_out/16)*(L/16) = 8 or 4 MFMA tiles of 16x16.
//
// What round 2 changed (measured: the round-1 kernel spent 4.3 k of every ~7.7 k cycles per op outside the MFMA loop):
//   * TILE OWNERSHIP: wave w owns tile w for the whole K range (ops with 4 tiles run on waves 0-3).  Accumulators never leave
//     registers: no K-partials through LDS, no partial-sum re-reads.
//   * GroupNorm statistics straight from the accumulators: per 16-lane DPP row (= 4 channels x 16 positions) a local
//     two-pass (mean, M2), exchanged as 8 bytes per row through LDS and combined with Chan's formula (all parts have 64
//     elements) - one barrier; normalise + Mish + time bias + residual in registers; ONE 16-byte LDS store per lane.
//   * WEIGHTS BY REGISTER RING, prefetched ACROSS ops: a wave's A operand stream (its 16 output channels, all K) is a linear
//     run of 1-KiB blocks of the packed weights; a 16-block ring (64 VGPRs) is refilled as blocks are consumed, and the first
//     16 blocks of the NEXT op are requested before the current op's epilogue, so L2/MALL latency hides under the epilogue
//     and the barriers.  No LDS weight window: the k-loop's LDS traffic halves (B fragments only) and LDS use drops to
//     the activations (~40-60 KB).  Barriers are LDS-only (s_waitcnt lgkmcnt(0) + s_barrier): they do not drain the ring.
//   * the k-loop is a compile-time shape (MODE, taps, C_in/16 [, C_in/16 of a folded residual conv]): fully unrolled, every
//     ring slot and LDS offset static, exact s_waitcnt vmcnt counts.
//   * a block's residual 1x1 conv is FOLDED into blocks[1]: its blocks follow blocks[1]'s own in the same ring stream and
//     accumulate the block input into a second accumulator added after Mish (removes an op, a buffer and two barriers).
#pragma once
#include "conv_block.hpp"

namespace mpdx {

enum : int { FOP_CONV_GN = 0, FOP_CONV_BIAS = 1, FOP_FINAL = 2 };

struct FusedBuf {
    int off4;   // offset in LDS, float4 units
    int rs4;    // row stride, float4 units
    int rows;   // L + 4 (two zero halo rows on each side)
    int clear_all;  // 1: the whole buffer must start zeroed (channel padding of the staged input)
};

struct FusedOp {
    int kind;            // FOP_*
    int shape;           // compile-time k-loop shape id (fused_shape_id), -1 for FOP_FINAL
    int mode, ks;        // CONV_S1 / CONV_DOWN / CONV_UPT, taps
    int nc16;            // C_in (padded) / 16
    int src, dst, res;   // LDS buffer ids (-1: none); res = identity residual added after Mish
    int gdst;            // index into FusedArgs.gout (-1: none)
    int cout, L_in, L_out, gs;
    int lg_MSn, T;       // log2(C_out/16); number of 16x16 tiles (4 or 8)
    int NSn, lg_RB;      // tiles along positions; log2(DPP rows per GroupNorm group) = log2(gs/4)
    int w_off;           // float offset of the packed conv weights
    int b_off, ga_off, be_off, tb_off;  // sources of the parameter vectors (packed offsets; tb_off: offset in the time-table row, -1 none)
    int p_off;           // float offset of this op's staged [bias | gamma | beta | tbias | rbias] block (5*cout floats) in LDS
    // folded residual 1x1 conv of the block (blocks[1] only):  out += W_res * block_input + b_res
    int rsrc, rnc16, rw_off, rb_off;    // rsrc = LDS buffer of the block input (-1: none)
};

constexpr int kMaxFusedOps = 16;
constexpr int kMaxFusedBufs = 16;
constexpr int kFusedRing = 16;   // ring depth in 1-KiB A-fragment blocks (4 VGPRs each)

struct FusedArgs {
    const float* packed;
    const float* tt_row;
    const float* gsrc1;  // kernel input [B][L0][gc1]
    const float* gsrc2;  // second half of a channel concat [B][L0][gc2], or null
    float* gout[3];      // global outputs, channel-last [B][L][C]
    int gc1, gc2, L0, in_buf;
    int B, nops, nbufs;
    int stat_off;        // GroupNorm exchange area (floats): [tile 0..7][q 0..3][mean, M2]
    int par_off4;        // staged per-op parameter vectors (float4 units)
    int par_floats;      // their total length
    int lg_c4n;          // log2(float4 per staged input row) or -1 (generic division path)
    int n_runs;          // parameter runs = 5 * (#conv ops): run r = op (r / 5), vector (r % 5)
    FusedOp ops[kMaxFusedOps];
    FusedBuf bufs[kMaxFusedBufs];
    // FOP_FINAL extras (final 1x1 conv + DDPM step), as FinalArgs of mpdx.hip
    const float* x_in; const float* noise; const float* hs; const float* hg;
    float* out; float* chain; uint32_t* absmax;
    int D, Cf, fmode, n_per_ctx, fw_off, fb_off;
    mpdx_step_coefs k;
    long long* trace;    // dev tool: s_memtime stamps of workgroup 0 / wave 0 (null in production)
};

// ---- compile-time k-loop shapes: (MODE, KS, C_in/16, C_in/16 of the folded residual conv or 0) -----------------------------
#define MPDX_FUSED_SHAPES(X)                                                                                      \
    X(0, CONV_S1, 5, 1, 0) X(1, CONV_S1, 5, 2, 0) X(2, CONV_S1, 5, 4, 0) X(3, CONV_S1, 5, 8, 0) X(4, CONV_S1, 5, 16, 0) \
    X(5, CONV_S1, 5, 2, 1) X(6, CONV_S1, 5, 4, 2) X(7, CONV_S1, 5, 4, 16) X(8, CONV_S1, 5, 2, 8) X(9, CONV_S1, 5, 8, 4)   \
    X(10, CONV_S1, 1, 1, 0) X(11, CONV_S1, 1, 2, 0) X(12, CONV_S1, 1, 4, 0) X(13, CONV_S1, 1, 8, 0) X(14, CONV_S1, 1, 16, 0) \
    X(15, CONV_DOWN, 3, 2, 0) X(16, CONV_DOWN, 3, 4, 0) X(17, CONV_DOWN, 3, 8, 0)                                  \
    X(18, CONV_UPT, 4, 2, 0) X(19, CONV_UPT, 4, 4, 0) X(20, CONV_UPT, 4, 8, 0)

inline int fused_shape_id(int mode, int ks, int nc16, int rnc16) {
#define X(id, M, K, N, R) if (mode == M && ks == K && nc16 == N && rnc16 == R) return id;
    MPDX_FUSED_SHAPES(X)
#undef X
    return -1;
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic, NOT for its global loads (the weight ring stays in flight).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// block index (1-KiB units, relative to the wave's stream base) of the r-th block a wave consumes
template <int MODE>
__device__ __forceinline__ constexpr int fused_blk(int r) { return (MODE == CONV_UPT) ? ((r >> 1) * 4 + (r & 1)) : r; }

// Request the first kFusedRing blocks of an op's stream (runtime shape: called one op ahead).  Unconditional loads from
// clamped block indices: short streams just re-request their last block.
__device__ __forceinline__ void fused_prefetch(f32x4 (&ring)[kFusedRing], const float* __restrict__ abase, int mode, int nblk,
                                               const float* __restrict__ rbase, int rnblk) {
#pragma unroll
    for (int p = 0; p < kFusedRing; ++p) {
        const int tot = nblk + rnblk;
        const int r = p < tot ? p : tot - 1;
        const float* src;
        if (r < nblk) src = abase + (size_t)(mode == CONV_UPT ? ((r >> 1) * 4 + (r & 1)) : r) * 256;
        else src = rbase + (size_t)(r - nblk) * 256;
        ring[p] = *(const f32x4*)src;
    }
}

// One wave's whole-K MFMA loop over its 16x16 tile.  ring[] holds blocks 0..kFusedRing-1 of the stream on entry (requested
// one op ahead); every consumed slot is refilled with the block kFusedRing further down the stream.
//   abase: packed weights of this wave's 16 output channels (+ parity slots for CONV_UPT) + lane*4
//   brow : lane's B row in the source buffer (float4 units); tap offsets are multiples of rs4
//   rbase / rrow / racc: the folded residual 1x1 conv's stream, B row and accumulator (NCR > 0)
template <int MODE, int KS, int NC16, int NCR>
__device__ __forceinline__ void fused_kloop(f32x4 (&ring)[kFusedRing], const float* __restrict__ abase, const f32x4* __restrict__ brow, int rs4,
                                            int par, const float* __restrict__ rbase, const f32x4* __restrict__ rrow, f32x4& acc0, f32x4& acc1,
                                            f32x4& racc0, f32x4& racc1) {
    constexpr int NTAP = (MODE == CONV_UPT) ? 2 : KS;
    constexpr int NBLK = NC16 * NTAP, TOT = NBLK + NCR, P = kFusedRing;
    constexpr int DB = 2;   // B fragments are read DB blocks ahead of their MFMAs (LDS latency under two blocks of MFMAs)
    auto read_b = [&](int r) -> f32x4 {
        if (r < NBLK) {
            const int c16 = r / NTAP, ts = r % NTAP;
            const int roff = (MODE == CONV_UPT) ? ((ts == 0) ? 0 : (par == 0 ? -1 : 1)) : ts;
            return brow[roff * rs4 + c16 * 4];
        }
        return rrow[(r - NBLK) * 4];
    };
    f32x4 bq[DB + 1];
#pragma unroll
    for (int r = 0; r < DB && r < TOT; ++r) bq[r] = read_b(r);
    // hipcc's scheduler otherwise sinks every ring refill next to its use (2 loads in flight instead of 16) and issues each
    // ds_read right before its MFMAs: the order below is PINNED block by block with sched_barrier(0); the s_waitcnt counts are
    // still the compiler's (exact: the code is straight-line).
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < TOT; ++r) {
        if (r + DB < TOT) bq[(r + DB) % (DB + 1)] = read_b(r + DB);
        const f32x4 af = ring[r % P];
        const f32x4 bf = bq[r % (DB + 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // two independent accumulator chains (even / odd k)
            if (r >= NBLK) {
                if (e & 1) racc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], racc1, 0, 0, 0);
                else racc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], racc0, 0, 0, 0);
            } else {
                if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], acc0, 0, 0, 0);
            }
        }
        if (r + P < TOT) {
            const int rn = r + P;
            ring[r % P] = (rn < NBLK) ? *(const f32x4*)(abase + (size_t)fused_blk<MODE>(rn) * 256) : *(const f32x4*)(rbase + (size_t)(rn - NBLK) * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// sum over the 16 lanes of a DPP row (every lane of the row gets the sum)
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

struct FusedTile {   // a wave's tile of the current op and where its operands live
    int ms, ns, par, npos;
    const float* abase;
    const float* rbase;
    int nblk, rnblk;
};

__device__ __forceinline__ FusedTile fused_tile(const FusedArgs& a, const FusedOp& op, int wave, int lane) {
    FusedTile t;
    const int MSn = 1 << op.lg_MSn;
    const int sub = wave & (op.T - 1);            // waves >= T alias a valid tile (their loads are harmless, their results unused)
    t.ms = sub & (MSn - 1);
    t.ns = sub >> op.lg_MSn;
    t.par = t.ns & 1;
    const int j = lane & 15;
    const int nslot = (op.mode == CONV_UPT) ? 4 : op.ks;
    t.abase = a.packed + op.w_off + (size_t)t.ms * op.nc16 * nslot * 256 + (op.mode == CONV_UPT ? t.par * 2 * 256 : 0) + lane * 4;
    t.nblk = op.nc16 * ((op.mode == CONV_UPT) ? 2 : op.ks);
    t.rnblk = op.rsrc >= 0 ? op.rnc16 : 0;
    t.rbase = a.packed + (op.rsrc >= 0 ? op.rw_off + (size_t)t.ms * op.rnc16 * 256 : 0) + lane * 4;
    t.npos = (op.mode == CONV_UPT) ? 2 * ((t.ns >> 1) * 16 + j) + t.par : t.ns * 16 + j;
    return t;
}

__global__ __launch_bounds__(512) void fused_level_kernel(const FusedArgs a) {
#ifndef MPDX_NO_WARM_KERNARG   // dev A/B switch
    warm_kernarg<(int)sizeof(FusedArgs)>();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* const sm4 = (f32x4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int j = lane & 15, q = lane >> 4;
    int tr = 0;
#define FUSED_STAMP() do { if (a.trace && b == 0 && tid == 0) a.trace[tr] = (long long)__builtin_readcyclecounter(); ++tr; } while (0)
    FUSED_STAMP();

    // ---- prologue: every global load is issued first (weight ring of op 0, input window, parameter vectors), the halo /
    //      padding zeros are written while they fly, and ONE barrier closes it.
    f32x4 ring[kFusedRing];
    FusedTile tl = fused_tile(a, a.ops[0], wave, lane);
    fused_prefetch(ring, tl.abase, a.ops[0].mode, tl.nblk, tl.rbase, tl.rnblk);
    __builtin_amdgcn_sched_barrier(0);

    const FusedBuf ib = a.bufs[a.in_buf];
    const int cin = a.gc1 + a.gc2;
    const int c4n = (cin + 3) >> 2;
    const int n_in = a.L0 * c4n;
    constexpr int IK = 4;   // input float4 per thread (<= 2048 float4 per trajectory window)
    f32x4 iv[IK];
    int idst[IK], ck[IK];
    const bool vec_ok = ((a.gc1 & 3) == 0) && ((a.gc2 & 3) == 0);
    {
        // Unconditional loads from clamped addresses, zeros selected afterwards: a conditional load into a
        // zero-initialised register makes hipcc wait (vmcnt(0)) for the previous load before issuing the next one.
        int lk[IK];
        bool vk[IK];
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            const int idx = tid + k * 512;
            vk[k] = idx < n_in;
            const int idc = vk[k] ? idx : 0;
            lk[k] = a.lg_c4n >= 0 ? (idc >> a.lg_c4n) : (idc / c4n);
            ck[k] = (idc - lk[k] * c4n) << 2;
            idst[k] = vk[k] ? ib.off4 + (lk[k] + 2) * ib.rs4 + (ck[k] >> 2) : -1;
        }
        if (vec_ok) {
#pragma unroll
            for (int k = 0; k < IK; ++k) {
                const size_t pos = (size_t)b * a.L0 + lk[k];
                const float* src = (ck[k] < a.gc1) ? a.gsrc1 + pos * a.gc1 + ck[k] : a.gsrc2 + pos * a.gc2 + (ck[k] - a.gc1);
                iv[k] = *(const f32x4*)src;
            }
        } else {
#pragma unroll
            for (int k = 0; k < IK; ++k) {   // channel padding (ce >= cin) is masked to zero at the LDS store below
                const size_t pos = (size_t)b * a.L0 + lk[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ce = ck[k] + e;
                    const float* src = (ce < a.gc1) ? a.gsrc1 + pos * a.gc1 + ce : (ce < cin) ? a.gsrc2 + pos * a.gc2 + (ce - a.gc1) : a.gsrc1 + pos * a.gc1;
                    iv[k][e] = *src;
                }
            }
        }
    }
    FUSED_STAMP();   // ring + input loads issued
    // parameter vectors [bias | gamma | beta | tbias | rbias] x C_out of every conv op: run r is handled by wave r % 8, lane = channel
    float* par = smem + (size_t)a.par_off4 * 4;
    constexpr int RK = 10;   // runs per wave (<= 16 ops * 5 / 8)
    float pvv[RK];
    int pdst[RK];   // LDS destination (floats from the start of the parameter area), -1: nothing to store
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const int r = wave + k * 8;
        pvv[k] = 0.f;
        pdst[k] = -1;
        const int oi = r / 5, which = r - oi * 5;
        if (r < a.n_runs && lane < a.ops[oi].cout) {
            const FusedOp& op = a.ops[oi];
            pdst[k] = op.p_off + which * op.cout + lane;
            const float* src = (which == 0) ? a.packed + op.b_off
                               : (which == 4) ? (op.rsrc >= 0 ? a.packed + op.rb_off : nullptr)
                               : (op.kind != FOP_CONV_GN) ? nullptr
                               : (which == 1) ? a.packed + op.ga_off
                               : (which == 2) ? a.packed + op.be_off
                               : (op.tb_off >= 0 ? a.tt_row + op.tb_off : nullptr);
            if (src) pvv[k] = src[lane];
        }
    }
    FUSED_STAMP();   // parameter loads issued
    // zeros: the 2+2 halo rows of the staged input buffer and the channel padding of its rows (disjoint from what the
    // staging writes below).  Every other buffer gets its halo rows zeroed by the op that writes it.
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 2 * ib.rs4; i += 512) {
            sm4[ib.off4 + i] = z;
            sm4[ib.off4 + (ib.rows - 2) * ib.rs4 + i] = z;
        }
        if (ib.clear_all) {
            const int padw = ib.rs4 - c4n;   // float4 columns beyond the staged channels
            for (int i = tid; i < a.L0 * padw; i += 512) {
                const int l = i / padw, cc = i - l * padw;
                sm4[ib.off4 + (l + 2) * ib.rs4 + c4n + cc] = z;
            }
        }
    }
    FUSED_STAMP();
#pragma unroll
    for (int k = 0; k < IK; ++k) {
        if (!vec_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) iv[k][e] = (ck[k] + e < cin) ? iv[k][e] : 0.f;
        }
        if (idst[k] >= 0) sm4[idst[k]] = iv[k];
    }
#pragma unroll
    for (int k = 0; k < RK; ++k)
        if (pdst[k] >= 0) par[pdst[k]] = pvv[k];
    lds_barrier();
    FUSED_STAMP();

    for (int oi = 0; oi < a.nops; ++oi) {
        const FusedOp& op = a.ops[oi];
        if (op.kind == FOP_FINAL) {
            // ---- final_conv[1] (1x1, Cf -> D) + DDPM posterior step + hard conditioning (see final_step_kernel)
            const FusedBuf sb = a.bufs[op.src];
            float vmax = 0.f;
            const int H = op.L_in;
            for (int idx = tid; idx < H * a.D; idx += 512) {
                const int p = idx / a.D, d = idx - p * a.D;
                float s = a.packed[a.fb_off + d];
                const float* wrow = a.packed + a.fw_off + d * a.Cf;
                for (int c = 0; c < a.Cf; c += 4) {
                    const f32x4 hv = sm4[sb.off4 + (p + 2) * sb.rs4 + (c >> 2)];
                    const f32x4 wv = *(const f32x4*)(wrow + c);
                    s = fmaf(hv[0], wv[0], s); s = fmaf(hv[1], wv[1], s);
                    s = fmaf(hv[2], wv[2], s); s = fmaf(hv[3], wv[3], s);
                }
                const size_t o = ((size_t)b * H + p) * a.D + d;
                float r;
                if (a.fmode == 0) {
                    r = s;
                } else {
                    const float xv = a.x_in[o];
                    float x0 = a.k.predict_epsilon
                                   ? __fsub_rn(__fmul_rn(a.k.sqrt_recip_alphas_cumprod, xv), __fmul_rn(a.k.sqrt_recipm1_alphas_cumprod, s))
                                   : s;
                    if (a.fmode == 3) {  // ddim_sample (diffusion_model_base.py:216-237): x_start is not clamped there
                        const float pn = a.k.predict_epsilon
                                             ? s
                                             : __fdiv_rn(__fsub_rn(__fmul_rn(a.k.sqrt_recip_alphas_cumprod, xv), s), a.k.sqrt_recipm1_alphas_cumprod);
                        r = __fadd_rn(__fmul_rn(x0, a.k.ddim_k1), __fmul_rn(a.k.ddim_k2, pn));
                        if (a.hs && p == 0) r = a.hs[(size_t)b * a.D + d];
                        if (a.hg && p == H - 1) r = a.hg[(size_t)b * a.D + d];
                    } else {
                        if (a.k.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                        r = __fadd_rn(__fmul_rn(a.k.posterior_mean_coef1, x0), __fmul_rn(a.k.posterior_mean_coef2, xv));
                        if (a.fmode == 1) {
                            if (a.noise) r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.k.noise_scale, a.noise[o]), a.k.noise_std_extra));
                            if (a.hs && p == 0) r = a.hs[(size_t)b * a.D + d];
                            if (a.hg && p == H - 1) r = a.hg[(size_t)b * a.D + d];
                        }
                    }
                }
                a.out[o] = r;
                if (a.chain) a.chain[o] = r;
                vmax = fmaxf(vmax, fabsf(r));
            }
            if (a.absmax) {
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, s, 64));
                if (lane == 0) atomicMax(a.absmax + b / a.n_per_ctx, __float_as_uint(vmax));
            }
            FUSED_STAMP();
            continue;
        }

        // ------------------------------------------------------------------ conv: this wave's tile, whole K
        const FusedBuf sb = a.bufs[op.src];
        const bool owner = wave < op.T;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, racc0 = {0.f, 0.f, 0.f, 0.f}, racc1 = {0.f, 0.f, 0.f, 0.f};
        if (owner) {
            int boff;
            if (op.mode == CONV_UPT) boff = sb.off4 + ((tl.ns >> 1) * 16 + j + 2) * sb.rs4 + q;
            else {
                const int l = tl.ns * 16 + j;
                const int pad = (op.mode == CONV_S1) ? (op.ks >> 1) : 1;
                boff = sb.off4 + ((op.mode == CONV_DOWN ? 2 * l : l) + 2 - pad) * sb.rs4 + q;
            }
            const f32x4* brow = sm4 + boff;
            const f32x4* rrow = sm4;
            if (op.rsrc >= 0) rrow = sm4 + a.bufs[op.rsrc].off4 + (tl.ns * 16 + j + 2) * a.bufs[op.rsrc].rs4 + q;
            switch (op.shape) {
#define X(id, M, K, N, R) case id: fused_kloop<M, K, N, R>(ring, tl.abase, brow, sb.rs4, tl.par, tl.rbase, rrow, acc0, acc1, racc0, racc1); break;
                MPDX_FUSED_SHAPES(X)
#undef X
                default: break;
            }
        }
        f32x4 acc = acc0 + acc1;
        const f32x4 racc = racc0 + racc1;
        FUSED_STAMP();   // k-loop issued
        const FusedTile cur = tl;
        // request the next conv's first ring blocks now: they arrive under this op's epilogue and barriers
        if (oi + 1 < a.nops && a.ops[oi + 1].kind != FOP_FINAL) {
            tl = fused_tile(a, a.ops[oi + 1], wave, lane);
            fused_prefetch(ring, tl.abase, a.ops[oi + 1].mode, tl.nblk, tl.rbase, tl.rnblk);
            __builtin_amdgcn_sched_barrier(0);   // the requests go out HERE, ahead of the epilogue
        }

        // ------------------------------------------------------------------ epilogue (registers -> destination buffer)
        const float* par_op = smem + (size_t)a.par_off4 * 4 + op.p_off;  // [bias | gamma | beta | tbias | rbias] x cout
        const int c0 = cur.ms * 16 + q * 4;   // this lane's 4 output channels
        const int N = op.L_out;
        f32x4 y;
        if (op.kind == FOP_CONV_GN) {
            const f32x4 bi = *(const f32x4*)(par_op + c0);
            const f32x4 ga = *(const f32x4*)(par_op + op.cout + c0), be = *(const f32x4*)(par_op + 2 * op.cout + c0);
            const f32x4 tb = *(const f32x4*)(par_op + 3 * op.cout + c0);
            f32x4 rsd = {0.f, 0.f, 0.f, 0.f};
            if (owner && op.res >= 0) rsd = sm4[a.bufs[op.res].off4 + (cur.npos + 2) * a.bufs[op.res].rs4 + (c0 >> 2)];
            if (op.rsrc >= 0) rsd = racc + *(const f32x4*)(par_op + 4 * op.cout + c0);
            acc += bi;
            // local two-pass statistics of this DPP row (4 channels x 16 positions = 64 elements)
            const float m_loc = row_sum16((acc[0] + acc[1]) + (acc[2] + acc[3])) * (1.0f / 64.0f);
            const f32x4 dl = acc - m_loc;
            const float m2_loc = row_sum16((dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]));
            float* stat = smem + a.stat_off;
            if (owner && j == 0) *(f32x2*)(stat + ((cur.ms * op.NSn + cur.ns) * 4 + q) * 2) = (f32x2){m_loc, m2_loc};
            lds_barrier();
            FUSED_STAMP();   // statistics exchanged
            // combine the parts of this lane's group: rows q0 .. q0+RB-1 of the tiles (ms, 0..NSn-1); equal counts (64 each)
            const int RB = 1 << op.lg_RB, q0 = q & ~(RB - 1);
            const int nparts = op.NSn << op.lg_RB;   // 2 or 4
            float pm[4], pM2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kk = k < nparts ? k : 0;
                const int ns_k = kk >> op.lg_RB, q_k = q0 + (kk & (RB - 1));
                const f32x2 v = *(const f32x2*)(stat + ((cur.ms * op.NSn + ns_k) * 4 + q_k) * 2);
                pm[k] = v[0]; pM2[k] = v[1];
            }
            float mean, M2;
            if (nparts == 4) {
                mean = ((pm[0] + pm[1]) + (pm[2] + pm[3])) * 0.25f;
                const float d0 = pm[0] - mean, d1 = pm[1] - mean, d2 = pm[2] - mean, d3 = pm[3] - mean;
                M2 = ((pM2[0] + pM2[1]) + (pM2[2] + pM2[3])) + 64.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            } else {
                mean = (pm[0] + pm[1]) * 0.5f;
                const float d0 = pm[0] - mean, d1 = pm[1] - mean;
                M2 = (pM2[0] + pM2[1]) + 64.0f * (d0 * d0 + d1 * d1);
            }
            const float var = M2 * (nparts == 4 ? (1.0f / 256.0f) : (1.0f / 128.0f));
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = mish((acc[e] - mean) * rstd * ga[e] + be[e]);
            y += tb;
            y += rsd;
        } else {  // bias only: Downsample1d / Upsample1d / a stand-alone 1x1 conv
            y = acc + *(const f32x4*)(par_op + c0);
            FUSED_STAMP();   // (keeps four stamps per op)
        }
        if (owner) {
            if (op.dst >= 0) sm4[a.bufs[op.dst].off4 + (cur.npos + 2) * a.bufs[op.dst].rs4 + (c0 >> 2)] = y;
            if (op.gdst >= 0) *(f32x4*)(a.gout[op.gdst] + ((size_t)b * N + cur.npos) * op.cout + c0) = y;
        }
        if (op.dst >= 0) {   // halo rows of the buffer this op defines (2 above, 2 below its L_out interior rows)
            const FusedBuf hb = a.bufs[op.dst];
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < 2 * hb.rs4; i += 512) {
                sm4[hb.off4 + i] = z;
                sm4[hb.off4 + (N + 2) * hb.rs4 + i] = z;
            }
        }
        FUSED_STAMP();   // epilogue done (this wave)
        lds_barrier();
        FUSED_STAMP();
    }
}
#undef FUSED_STAMP

}  // namespace mpdx
