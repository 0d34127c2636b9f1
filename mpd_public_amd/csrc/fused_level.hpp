// fused_level.hpp - whole-trajectory fused programs for the outer levels of TemporalUnet (round-2 design).
//
// Why.  At B=100 the per-layer path is bound by launch boundaries and per-kernel fixed phases.  At the
// outer resolutions one workgroup can own ALL channels of a trajectory (C*L = 2048 floats, C <= 64), so GroupNorm statistics
// stay local and a whole run of layers executes out of LDS in ONE launch:
//     downs[i] (i = 0, 1):  ResidualTemporalBlock x2 -> Downsample1d                     (temporal_unet.py:141-150)
//     ups[j]   (last two):  cat(x, skip) -> ResidualTemporalBlock x2 -> Upsample1d       (temporal_unet.py:158-165)
//     and, after the last Upsample1d: final_conv (Conv1dBlock -> Conv1d 1x1) + the DDPM posterior step
//                                                            (temporal_unet.py:167, diffusion_model_base.py:121-155)
// The arithmetic per layer is the same as conv_block.hpp (same fp32 MFMA, GroupNorm, Mish).
//
// Structure.  One workgroup (8 waves) = one trajectory; LDS holds the activation buffers in the zero-haloed channel-last
// layout [L+4 rows][C + pad] (placed by live range on the host).  A short op list (kernel argument) is walked; every conv is
// M = C_out (all), N = L positions, K = C_in*taps as (C_out/16)*(L/16) = 8 or 4 MFMA tiles of 16x16.
//
// Round-2 design (the round-1 kernel spent 4.3 k of every ~7.7 k cycles per op outside the MFMA loop; the first round-2
// version moved that time into scalar descriptor decoding - both measured with the stamps of tools/fused_trace.py):
//   * STATIC OP SHAPES.  An op's whole geometry (mode, taps, C_in, folded residual C_in, C_out, L_out, GroupNorm or bias) is
//     a compile-time shape picked by ONE switch per op; tile decode, LDS strides of the exchange, the k-loop and the epilogue
//     are specialised per shape.  The runtime descriptor is 13 dwords (buffer offsets, stream base, parameter offsets).
//   * TILE OWNERSHIP, FOUR WAVES: a workgroup is 4 waves (one per SIMD); a wave owns one tile (4-tile ops) or two tiles that
//     share their 16 output channels (8-tile ops: one A fragment feeds both) for the whole K range.  Accumulators never
//     leave registers: no K-partials through LDS.  (Measured with 8 waves x 1 tile: every wave streamed its own copy of A
//     through the CU's 64 B/clk vector-memory path - 128 KB of requests per op, as long as the op's MFMAs.)
//   * GroupNorm statistics straight from the accumulators: per 16-lane DPP row (= 4 channels x 16 positions) a local
//     two-pass (mean, M2), exchanged as 8 bytes per row through LDS and combined with Chan's formula (all parts have 64
//     elements) - one barrier; normalise + Mish + time bias + residual in registers; ONE 16-byte LDS store per lane.
//   * WEIGHT STREAMS.  At load time every fused op gets a stream-ordered copy of its weights: per tile-stream (16 output
//     channels [x parity for ConvTranspose]) the 1-KiB A-fragment blocks in exactly the order the k-loop consumes them, a
//     folded residual conv's blocks appended.  A wave streams its blocks through a 16-slot register ring (64 VGPRs) that
//     runs CONTINUOUSLY ACROSS OPS: every consumed slot is refilled at once - with the block 16 further down the op's
//     stream or, in the op's last 16 steps, with the NEXT op's first blocks - so exactly one 1-KiB request per consumed
//     block is in the memory pipe (no request bursts at op boundaries) and L2/MALL latency hides under 16 blocks of MFMAs plus
//     the epilogue.  No LDS weight window; barriers are LDS-only (s_waitcnt lgkmcnt(0) + s_barrier): they do not drain the ring.
//   * hipcc's scheduler would sink every ring refill next to its use and issue each ds_read right before its MFMAs: the
//     k-loop's order is pinned block by block with sched_barrier(0) (B fragments two blocks ahead, refill after use); the
//     s_waitcnt counts remain the compiler's (steady state: vmcnt(15), lgkmcnt(1)).
//   * a block's residual 1x1 conv is FOLDED into blocks[1]: its blocks follow blocks[1]'s own in the same stream and
//     accumulate the block input into a second accumulator added after Mish (removes an op, a buffer and two barriers).
#pragma once
#include "conv_block.hpp"
#include "fused_geom.hpp"

namespace mpdx {

// ---- compile-time op shapes: (id, MODE, KS, C_in/16, folded-residual C_in/16 or 0, C_out, L_out, GroupNorm+Mish?) -------------
#define MPDX_FUSED_SHAPES(X)                                                                                   \
    X(0, CONV_S1, 5, 1, 0, 32, 64, 1) X(1, CONV_S1, 5, 2, 1, 32, 64, 1) X(2, CONV_S1, 5, 2, 0, 32, 64, 1) X(3, CONV_DOWN, 3, 2, 0, 32, 32, 0)     \
    X(4, CONV_S1, 5, 2, 0, 64, 32, 1) X(5, CONV_S1, 5, 4, 2, 64, 32, 1) X(6, CONV_S1, 5, 4, 0, 64, 32, 1) X(7, CONV_DOWN, 3, 4, 0, 64, 16, 0)     \
    X(8, CONV_S1, 5, 16, 0, 64, 16, 1) X(9, CONV_S1, 5, 4, 16, 64, 16, 1) X(10, CONV_S1, 5, 4, 0, 64, 16, 1) X(11, CONV_UPT, 4, 4, 0, 64, 32, 0)  \
    X(12, CONV_S1, 5, 8, 0, 32, 32, 1) X(13, CONV_S1, 5, 2, 8, 32, 32, 1) X(14, CONV_S1, 5, 2, 0, 32, 32, 1) X(15, CONV_UPT, 4, 2, 0, 32, 64, 0) \
    X(16, CONV_S1, 5, 4, 0, 128, 16, 1) X(17, CONV_S1, 5, 8, 4, 128, 16, 1) X(18, CONV_S1, 5, 8, 0, 128, 16, 1) X(19, CONV_DOWN, 3, 8, 0, 128, 8, 0)
constexpr int kFusedShapeFinal = 63;

inline int fused_shape_id(int mode, int ks, int nc16, int rnc16, int cout, int L_out, int gn) {
#define X(id, M, K, N, R, CO, LO, G) if (mode == M && ks == K && nc16 == N && rnc16 == R && cout == CO && L_out == LO && gn == G) return id;
    MPDX_FUSED_SHAPES(X)
#undef X
    return -1;
}

constexpr int kFusedWaves = 4;   // waves per workgroup (one per SIMD)
constexpr int kFusedThreads = 64 * kFusedWaves;

template <int MODE_, int KS_, int NC16_, int NCR_, int COUT_, int LOUT_, int GN_>
struct FusedShape {
    static constexpr int MODE = MODE_, KS = KS_, NC16 = NC16_, NCR = NCR_, COUT = COUT_, LOUT = LOUT_, GN = GN_;
    static constexpr int NTAP = (MODE == CONV_UPT) ? 2 : KS;
    static constexpr int NBLK = NC16 * NTAP;          // blocks of a tile-stream from the conv itself
    static constexpr int TOT = NBLK + NCR;            // + the folded residual conv's
    static constexpr int MSn = COUT / 16;
    static constexpr int MSW = MSn < kFusedWaves ? MSn : kFusedWaves;   // tile rows worked on at the same time (one per wave)
    static constexpr int MP = MSn / MSW;              // M-PASSES: with 8 tile rows a wave runs rows ms and ms + 4 one after the other
    static constexpr int NSn = LOUT >= 16 ? LOUT / 16 : 1;   // tiles along positions (ConvTranspose: parity sub-tiles included; L_out = 8: a half-used tile)
    static constexpr int T = MSn * NSn;               // 16x16 tiles: 4 or 8
    static constexpr int NTW = T / kFusedWaves;       // tiles per wave
    // Conv / strided conv: the wave's tiles share the output channels -> ONE stream, the tiles advance together (NJ tiles per
    // block).  ConvTranspose: the wave's two tiles are the two output parities, which use different weight slots -> the
    // stream is [parity 0 | parity 1] and the tiles run one after the other (NSEQ = 2).  M-passes: the wave's two tiles are two
    // tile ROWS (different weights, same B fragments) -> the stream is [row ms | row ms + 4], again one after the other.
    static constexpr int NSEQ = (MODE == CONV_UPT) ? NTW : MP;
    static constexpr int NJ = (MODE == CONV_UPT) ? 1 : NTW / MP;
    static constexpr int SLEN = NSEQ * TOT;           // blocks of a wave's stream for this op
    static constexpr int GS = COUT / 8;               // GroupNorm(8 groups): channels per group
    static constexpr int RB = GS / 4;                 // DPP rows (4 channels) per group
    static constexpr int NPARTS = NSn * RB;           // 64-element parts per group: 2 or 4
    static_assert(T == 4 || T == 8, "4 or 8 tiles");
    static_assert(MSn == 2 || MSn == 4 || MSn == 8, "2, 4 or 8 tile rows");
    static_assert(MP == 1 || (MODE != CONV_UPT && NSn == 1), "M-passes: one position tile");
    static_assert(MODE != CONV_UPT || NTW == 2, "ConvTranspose: a wave owns both parities of its tile");
    static_assert(MODE != CONV_UPT || NCR == 0, "no folded residual on resampling ops");
    static_assert(!GN || NPARTS == 2 || NPARTS == 4, "GroupNorm region of 128 or 256 elements");
    static_assert(LOUT >= 16 || !GN, "half-used tiles: resampling ops only");
};

struct FusedOp {          // runtime part of an op: 13 dwords
    int shape;            // MPDX_FUSED_SHAPES id, kFusedShapeFinal for the final 1x1 conv + DDPM step
    int src_off4, src_rs4;
    int rsrc_off4, rsrc_rs4;   // block input read by the folded residual conv (shapes with NCR > 0)
    int res_off4, res_rs4;     // identity residual added after Mish (res_off4 < 0: none)
    int dst_off4, dst_rs4;     // destination buffer (dst_off4 < 0: none)
    int gdst;                  // index into FusedArgs.gout (-1: none)
    int sbase;                 // float offset (in `packed`) of the op's weight streams: [tile-stream][TOT blocks][256]
    int p_off;                 // float offset of the op's [bias | gamma | beta | rbias] x C_out block in the LDS parameter area
    int tb_off;                // float offset of the op's time-bias row in the LDS time-table slice (-1: none)
    // training forward (FusedArgs::save != null): float offsets in `save` of dense [B][L_out][C_out] copies of the op's output and of
    // its GroupNorm input (conv + bias), which the backward pass differentiates through; -1: not kept
    int save_out, save_pre;
};

constexpr int kMaxFusedOps = 16;
#ifndef MPDX_FUSED_RING
#define MPDX_FUSED_RING 16
#endif
#ifndef MPDX_FUSED_DB
#define MPDX_FUSED_DB 2
#endif
constexpr int kFusedRing = MPDX_FUSED_RING;   // ring depth in 1-KiB A-fragment blocks (4 VGPRs each)

struct FusedArgs {
    const float* packed;
    const float* tt_row;
    const float* gsrc1;  // kernel input [B][L0][gc1]
    const float* gsrc2;  // second half of a channel concat [B][L0][gc2], or null
    float* gout[3];      // global outputs, channel-last [B][L][C]
    int gc1, gc2, L0;
    int in_off4, in_rs4, in_rows, in_clear;   // staged input buffer
    // a SECOND staged input: the skip tensor [B][L3][c3] that a later op of the program concatenates behind an LDS-resident
    // tensor (torch.cat((x, h.pop()), dim=1) of the second up level, temporal_unet.py:159): written by the prologue into columns
    // [col3, col3 + c3/4) of that op's (wider) source buffer, whose first columns the producing Upsample1d fills later
    const float* gsrc3;
    float* save;                              // training: base of the kept activations (see FusedOp::save_out); null when planning
    int tt_stride;                            // training: per-trajectory time-table rows (tt_row + b * tt_stride); 0 when planning
    int fpar_off;                             // final_conv[1] weights [D][Cf + 4] + bias [D] inside the staged parameter block (floats)
    int c3, L3, s3_off4, s3_rs4, s3_col4;     // c3 == 0: none
    int B, nops;
    int stat_off;        // GroupNorm exchange area (floats): [tile 0..7][row 0..3][mean, M2]
    int par_off;         // LDS parameter area (floats): copy of packed[gpar_off .. +par_floats)
    int gpar_off, par_floats;
    int tt_off;          // LDS time-table slice (floats): copy of tt_row[tt_lo .. +tt_n)
    int tt_lo, tt_n;
    int lg_c4n;          // log2(float4 per staged input row) or -1 (generic division path)
    FusedOp ops[kMaxFusedOps];
    // runtime copies of what the ring needs ONE OP AHEAD: wave w streams ops[i].sbase + (w & msmask[i]) * slen[i] * 256
    int msmask[kMaxFusedOps];    // (C_out / 16) - 1
    int slen[kMaxFusedOps];      // blocks per wave-stream (FusedShape::SLEN)
    // final 1x1 conv + DDPM step, as FinalArgs of mpdx.hip
    const float* x_in; const float* noise; const float* hs; const float* hg;
    float* out; float* chain; uint32_t* absmax;
    int D, Cf, fmode, n_per_ctx, fw_off, fb_off, H;
    mpdx_step_coefs k;
    NoiseRng rng;        // rng.on: the step's noise is drawn in place
    long long* trace;    // dev tool: s_memtime stamps of workgroup 0, 128 slots per wave (null in production)
};

// The LDS layout scalars of a program as the kernels read them: compile-time constants for a static program with a geometry table
// (fused_geom.hpp; everything folds into immediates), the argument block's fields otherwise.
struct FusedLay {
    int in_off4, in_rs4, in_rows, L0, gc1, gc2, c3, L3, s3_off4, s3_rs4, s3_col4, stat_off, par_off, par_floats, tt_off, tt_n, fpar_off, H, Cf, lg_c4n;
};
template <class GEOM>
__device__ __forceinline__ FusedLay fused_lay(const FusedArgs& a) {
    if constexpr (GEOM::has) {
        constexpr FusedGeom g = GEOM::g;
        constexpr int c4n = (g.gc1 + g.gc2 + 3) >> 2;
        constexpr int lg = g.gc1 < 0 ? -2 : (c4n == 1 ? 0 : c4n == 2 ? 1 : c4n == 4 ? 2 : c4n == 8 ? 3 : c4n == 16 ? 4 : c4n == 32 ? 5 : c4n == 64 ? 6 : -1);
        return FusedLay{g.in_off4, g.in_rs4, g.in_rows, g.L0, g.gc1 < 0 ? a.gc1 : g.gc1, g.gc1 < 0 ? a.gc2 : g.gc2, g.c3, g.L3, g.s3_off4, g.s3_rs4, g.s3_col4,
                        g.stat_off, g.par_off, g.par_floats < 0 ? a.par_floats : g.par_floats, g.tt_off, g.tt_n, g.fpar_off, g.H, g.Cf,
                        lg == -2 ? a.lg_c4n : lg};
    } else {
        return FusedLay{a.in_off4, a.in_rs4, a.in_rows, a.L0, a.gc1, a.gc2, a.c3, a.L3, a.s3_off4, a.s3_rs4, a.s3_col4, a.stat_off, a.par_off, a.par_floats,
                        a.tt_off, a.tt_n, a.fpar_off, a.H, a.Cf, a.lg_c4n};
    }
}

// host: does the geometry the host computed for a segment equal the table of the static program it matched?
inline bool fused_geom_matches(const FusedArgs& a, const FusedGeom& g, int state_dim) {
    if (a.nops != g.nops || a.in_off4 != g.in_off4 || a.in_rs4 != g.in_rs4 || a.in_rows != g.in_rows || a.L0 != g.L0) return false;
    if (g.gc1 < 0) { if (a.gc1 != state_dim || a.gc2 != 0 || state_dim > 16) return false; }
    else if (a.gc1 != g.gc1 || a.gc2 != g.gc2) return false;
    if (a.c3 != g.c3 || (g.c3 && (a.L3 != g.L3 || a.s3_off4 != g.s3_off4 || a.s3_rs4 != g.s3_rs4 || a.s3_col4 != g.s3_col4))) return false;
    if (a.stat_off != g.stat_off || a.par_off != g.par_off || a.tt_off != g.tt_off || a.tt_n != g.tt_n) return false;
    if (g.par_floats >= 0 && a.par_floats != g.par_floats) return false;
    if (g.H && (a.H != g.H || a.Cf != g.Cf || a.fpar_off != g.fpar_off || state_dim > 16)) return false;
    for (int k = 0; k < g.nops; ++k) {
        const FusedOp& o = a.ops[k];
        const FusedGeomOp& t = g.ops[k];
        if (o.shape != t.shape || o.src_off4 != t.src_off4 || o.src_rs4 != t.src_rs4) return false;
        if (o.shape == kFusedShapeFinal) continue;
        if (o.rsrc_off4 != t.rsrc_off4 || o.rsrc_rs4 != t.rsrc_rs4 || o.res_off4 != t.res_off4 || o.res_rs4 != t.res_rs4 || o.dst_off4 != t.dst_off4 ||
            o.dst_rs4 != t.dst_rs4 || o.gdst != t.gdst || o.p_off != t.p_off || o.tb_off != t.tb_off)
            return false;
    }
    return true;
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic, NOT for its global loads (the weight ring stays in flight).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Request the first min(16, slen) blocks of a wave-stream (prologue only).  Slots beyond a short stream stay unset: the first
// op's k-loop fills them with the next op's blocks.  (wave-uniform guards: no request is issued for a block nobody reads -
// every request costs the CU's 64 B/clk vector-memory path 16 cycles.)
// One 1-KiB A-fragment block of a wave-stream, as a BUFFER load: resource = the `packed` allocation (SGPR x4), voffset = lane * 16
// (one VGPR for the whole kernel), soffset = byte offset of the block (wave-uniform: stream base + block * 1024, one SALU add).  No VALU
// instruction per block: as per-lane 64-bit pointers the 472 block loads of the three-level down program cost 81 v_add_co / v_addc pairs
// plus moves (a global load's immediate reaches 4 KB = four blocks) - tools/isa_census.py, round 4.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fused_weights_rsrc(const float* packed) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, 0x7fffffff, 0x00020000);   // raw buffer, no bounds in practice, DATA_FORMAT_32
}
__device__ __forceinline__ f32x4 fused_ld_block(__amdgpu_buffer_rsrc_t rs, int stream_bytes, int blk, unsigned lane_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_bytes, stream_bytes + blk * 1024, 0));
}

__device__ __forceinline__ void fused_ring_request(f32x4 (&ring)[kFusedRing], __amdgpu_buffer_rsrc_t rs, int wbase, int slen, unsigned lane_bytes) {
#pragma unroll
    for (int p = 0; p < kFusedRing; ++p)   // (no zero-init: a load into a pre-initialised register makes hipcc drain vmcnt first)
        if (p < slen) ring[p] = fused_ld_block(rs, wbase, p, lane_bytes);
    __builtin_amdgcn_sched_barrier(0);   // the requests go out HERE
}

// (Tried and rejected, A/B on MI355X: warming the XCD's L2 with the weight stream of the op after next - LDS-DMA loads into a
//  scratch slot, each workgroup of an XCD a 1/12 slice - 25.56 vs 25.20 ms per cfg2 plan: the long k-loops did not speed up (they
//  already run at ~83 % of their MFMA rate: 154 cycles per 128-cycle block), the extra requests only delayed the prologue.)

// sum over the 16 lanes of a DPP row (every lane of the row gets the sum)
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// One conv op of static shape S: k-loop over this wave's tile(s) with the ring running on into the next op's stream, epilogue.
//   nbase: BYTE offset in `packed` of the NEXT conv op's wave-stream (wave-uniform), nmax: its last block index (the ring's cross-over requests are clamped
//   to it; after the last conv op both describe any valid block)
// SAVE: the training forward's variant (every op also stores its output and its GroupNorm input through FusedArgs::save); the planning
// kernels are instantiated without it (measured on one box: 22.54 vs 22.62 ms per cfg-2 plan with the stores merely compiled in)
template <class S, bool SAVE = false>
__device__ __forceinline__ void fused_conv_op(const FusedArgs& a, const FusedLay& lay, const FusedOp& op, f32x4 (&ring)[kFusedRing], float* smem, int wave, int lane, int b,
                                              int nbase, int nmax, long long* tr_base, int& tr) {
    constexpr int P = kFusedRing, DB = MPDX_FUSED_DB, NTW = S::NTW, NJ = S::NJ;
    f32x4* const sm4 = (f32x4*)smem;
    const int j = lane & 15, q = lane >> 4;
    const int ms = wave & (S::MSW - 1);   // this wave's tile row (first of S::MP rows: ms, ms + 4)
    const int nsg = wave / S::MSW;     // which group of position tiles this wave owns
#define FOP_STAMP() do { if (tr_base) tr_base[tr] = (long long)__builtin_readcyclecounter(); ++tr; } while (0)
    // tile t of this wave: position-tile index and output position of this lane's column
    int ns[NTW], npos[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        ns[t] = (S::MP > 1) ? 0 : nsg * NTW + t;                 // ConvTranspose: (input tile nsg, parity t); M-passes: tile row t of position tile 0
        npos[t] = (S::MODE == CONV_UPT) ? 2 * (nsg * 16 + j) + t : ns[t] * 16 + j;
    }
    const bool col_ok = S::LOUT >= 16 || j < S::LOUT;           // half-used tile: columns beyond L_out are computed on clamped rows, never stored
    // lane's B rows in the source buffer (float4 units); taps are row offsets in the zero-haloed buffer
    const f32x4* brow[NJ];
    const f32x4* rrow[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
        int boff;
        if (S::MODE == CONV_UPT) boff = op.src_off4 + (nsg * 16 + j + 2) * op.src_rs4 + q;
        else {
            constexpr int pad = (S::MODE == CONV_S1) ? S::KS / 2 : 1;
            int l = ns[t] * 16 + j;
            if (S::LOUT < 16) l = l < S::LOUT ? l : S::LOUT - 1;
            boff = op.src_off4 + ((S::MODE == CONV_DOWN ? 2 * l : l) + 2 - pad) * op.src_rs4 + q;
        }
        brow[t] = sm4 + boff;
        rrow[t] = sm4 + (S::NCR > 0 ? op.rsrc_off4 + (ns[t] * 16 + j + 2) * op.rsrc_rs4 + q : 0);
    }
    const int rs4 = op.src_rs4;
    const __amdgpu_buffer_rsrc_t wrs = fused_weights_rsrc(a.packed);
    const int sbase = (op.sbase + ms * (S::SLEN * 256)) * 4;   // BYTE offset of this wave's stream in `packed` (wave-uniform; nbase likewise)
    const unsigned lane_bytes = (unsigned)lane * 16u;
    // B fragment of stream block r for joint tile t
    auto read_b = [&](int r, int t) -> f32x4 {
        const int rr = r % S::TOT, par = r / S::TOT;    // (ConvTranspose: second half of the stream = odd outputs)
        if (rr < S::NBLK) {
            const int c16 = rr / S::NTAP, ts = rr % S::NTAP;
            const int roff = (S::MODE == CONV_UPT) ? ((ts == 0) ? 0 : (par == 0 ? -1 : 1)) : ts;
            return brow[t][roff * rs4 + c16 * 4];
        }
        return rrow[t][(rr - S::NBLK) * 4];
    };
    f32x4 acc[NTW][2], racc[NTW][2];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t][0] = acc[t][1] = racc[t][0] = racc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    FOP_STAMP();   // op entered: descriptor decoded, addresses set up
    f32x4 bq[DB + 1][NJ];
#pragma unroll
    for (int r = 0; r < DB && r < S::SLEN; ++r)
#pragma unroll
        for (int t = 0; t < NJ; ++t) bq[r][t] = read_b(r, t);
    // hipcc's scheduler would sink every ring refill next to its use and issue each ds_read right before its MFMAs: the order is
    // PINNED block by block with sched_barrier(0); the s_waitcnt counts are still the compiler's (straight-line code).
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
        for (int e = 0; e < 4; ++e)   // two independent accumulator chains per tile (even / odd k)
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int tt = (S::NSEQ > 1) ? r / S::TOT : t;
                f32x4& d = is_res ? racc[tt][e & 1] : acc[tt][e & 1];
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bq[r % (DB + 1)][t][e], d, 0, 0, 0);
            }
        // refill slot r % P: the block P further down this op's stream, or (no such block) the next op's block of that slot
        if (r + P < S::SLEN) ring[r % P] = fused_ld_block(wrs, sbase, r + P, lane_bytes);
        else { const int k = r % P; ring[k] = fused_ld_block(wrs, nbase, k < nmax ? k : nmax, lane_bytes); }
        if (S::SLEN < P) {   // slots this op never used belong to the next op from the start
#pragma unroll
            for (int k = S::SLEN; k < P; ++k)
                if ((k - S::SLEN) % S::SLEN == r) ring[k] = fused_ld_block(wrs, nbase, k < nmax ? k : nmax, lane_bytes);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    FOP_STAMP();   // k-loop issued

    // ------------------------------------------------------------------ epilogue (registers -> destination buffer)
    // At large batch two or three workgroups share a CU: this wave's statistics + epilogue - chains of dependent VALU instructions -
    // then run next to a SIMD partner that streams MFMAs, at ~20 cycles per instruction (measured in conv_ws.hpp); raised issue
    // priority until the op's closing barrier.  (Measured, same-box A/B: cfg 5 542.45 -> 542.1 ms, cfg 2 21.88 -> 21.82 ms - within
    // noise; kept because it is free.  At B = 100 there is one wave per SIMD and nothing to arbitrate.)
    __builtin_amdgcn_s_setprio(2);
    const float* par_op = smem + lay.par_off + op.p_off;   // [bias | gamma | beta | rbias] x COUT
    // this lane's 4 output channels in tile t (M-passes: tile t is tile row ms + 4 t)
    int c0t[NTW], mst[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) { mst[t] = (S::MP > 1) ? ms + t * S::MSW : ms; c0t[t] = mst[t] * 16 + q * 4; }
    f32x4 y[NTW];
    if (S::GN) {
        // ---- GroupNorm statistics + affine + Mish, round-4 form (instruction diet: fp32 MFMA and VALU share the SIMD's issue port).
        // A reduction runs over ALL of the wave's tiles that belong to one group at once (NJ tiles share their rows: one reduction;
        // M-pass tiles are different tile rows: one each): one row_sum16 per pass instead of one per tile.
        //   LOCAL (C_out >= 64: the wave owns every tile of its rows): plain TWO-PASS statistics of the whole group inside the wave - lane
        //     sums -> DPP row sum -> the other rows of the group through the LDS crossbar (ds_swizzle lane ^ 16; groups of four rows:
        //     ds_bpermute lane ^ 32 on top) -> mean; deviations -> the same for the sum of squares.  No per-row (mean, M2) parts, no
        //     v_readlane -> v_mov -> v_cndmask selection of the parts, no Chan combination: ~60 VALU instructions fewer per op than round 3.
        //   C_out = 32 (two waves share a tile row): the wave's own tiles are reduced two-pass as above (mean, M2 of 64 * NTW elements), ONE
        //     8-byte exchange per (wave, row) through LDS + one barrier, Chan's formula for two equal parts.
        // Normalisation re-uses the deviations: y = mish(d * (rstd * gamma) + beta).
        constexpr bool LOCAL = (S::MSW == kFusedWaves);
        constexpr int NG = (S::MP > 1) ? NTW : 1;   // independent reductions (groups) of this wave
        constexpr int TPG = NTW / NG;               // tiles per reduction
        f32x4 v[NTW], addv[NTW], gat[NTW], bet[NTW];
        bool have_add = false;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int c0 = c0t[t];
            const f32x4 bi = *(const f32x4*)(par_op + c0);
            if (S::MP > 1 || t == 0) { gat[t] = *(const f32x4*)(par_op + S::COUT + c0); bet[t] = *(const f32x4*)(par_op + 2 * S::COUT + c0); }
            else { gat[t] = gat[0]; bet[t] = bet[0]; }
            if (op.tb_off >= 0) { addv[t] = *(const f32x4*)(smem + lay.tt_off + op.tb_off + c0); have_add = true; }
            if (S::NCR > 0) {
                const f32x4 r = (racc[t][0] + racc[t][1]) + *(const f32x4*)(par_op + 3 * S::COUT + c0);
                addv[t] = (op.tb_off >= 0) ? addv[t] + r : r;
                have_add = true;
            } else if (op.res_off4 >= 0) {
                const f32x4 r = sm4[op.res_off4 + (npos[t] + 2) * op.res_rs4 + (c0 >> 2)];
                addv[t] = (op.tb_off >= 0) ? addv[t] + r : r;
                have_add = true;
            }
            v[t] = (acc[t][0] + acc[t][1]) + bi;
            if (SAVE && op.save_pre >= 0) *(f32x4*)(a.save + op.save_pre + ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0) = v[t];
        }
        // sum over the rows of a lane's group that live in other DPP rows of this wave (LOCAL only)
        auto rows_sum = [&](float x) -> float {
            if constexpr (LOCAL && S::RB >= 2)
                x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));   // bit mode: lane ^ 16
            if constexpr (LOCAL && S::RB == 4)
                x += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, x)));
            return x;
        };
        constexpr int NLOC = 64 * TPG * (LOCAL ? S::RB : 1);   // elements one in-wave reduction covers
        float mean_g[NG], rstd_g[NG];
        f32x4 d[NTW];
        float m2_g[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float sl = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const f32x4& x = v[g * TPG + k];
                const float p = (x[0] + x[1]) + (x[2] + x[3]);
                sl = (k == 0) ? p : sl + p;
            }
            mean_g[g] = rows_sum(row_sum16(sl)) * (1.0f / (float)NLOC);
            float ql = 0.f;
#pragma unroll
            for (int k = 0; k < TPG; ++k) {
                const int t = g * TPG + k;
                d[t] = v[t] - mean_g[g];
                const float p = (d[t][0] * d[t][0] + d[t][1] * d[t][1]) + (d[t][2] * d[t][2] + d[t][3] * d[t][3]);
                ql = (k == 0) ? p : ql + p;
            }
            m2_g[g] = rows_sum(row_sum16(ql));
        }
        if constexpr (LOCAL) {
            FOP_STAMP();   // statistics done (no exchange)
#pragma unroll
            for (int g = 0; g < NG; ++g) rstd_g[g] = gn_rstd(m2_g[g] * (1.0f / (float)NLOC));
        } else {
            // two waves (nsg = 0, 1) share tile row ms: exchange (mean, M2) of 64 * NTW elements per DPP row q
            float* stat = smem + lay.stat_off;
            if (j == 0) *(f32x2*)(stat + ((nsg * S::MSW + ms) * 4 + q) * 2) = (f32x2){mean_g[0], m2_g[0]};
            lds_barrier();
            FOP_STAMP();   // statistics exchanged
            const f32x2 o = *(const f32x2*)(stat + (((nsg ^ 1) * S::MSW + ms) * 4 + q) * 2);
            // Chan, two parts of NLOC elements: mean = (m_a + m_b) / 2, M2 = M2_a + M2_b + NLOC / 2 * (m_b - m_a)^2; symmetric in (a, b),
            // so both waves get the same bits
            const float dm = o[0] - mean_g[0];
            const float mean = 0.5f * (mean_g[0] + o[0]);
            const float M2 = (m2_g[0] + o[1]) + (0.5f * (float)NLOC) * (dm * dm);
            rstd_g[0] = gn_rstd(M2 * (1.0f / (float)(2 * NLOC)));
            const float shift = mean_g[0] - mean;   // deviations from the group mean = deviations from the own mean + (own mean - group mean)
#pragma unroll
            for (int t = 0; t < NTW; ++t) d[t] = d[t] + shift;
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int g = (S::MP > 1) ? t : 0;
            const f32x4 sc = gat[t] * rstd_g[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[t][e] = mish_nosel(d[t][e] * sc[e] + bet[t][e]);
            if (have_add) y[t] += addv[t];
        }
    } else {  // bias only: Downsample1d / Upsample1d
#pragma unroll
        for (int t = 0; t < NTW; ++t) y[t] = (acc[t][0] + acc[t][1]) + *(const f32x4*)(par_op + c0t[t]);
        FOP_STAMP();   // (keeps four stamps per op)
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        if (!col_ok) continue;
        if (op.dst_off4 >= 0) sm4[op.dst_off4 + (npos[t] + 2) * op.dst_rs4 + (c0t[t] >> 2)] = y[t];
        if (op.gdst >= 0) *(f32x4*)(a.gout[op.gdst] + ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0t[t]) = y[t];
        if (SAVE && op.save_out >= 0) *(f32x4*)(a.save + op.save_out + ((size_t)b * S::LOUT + npos[t]) * S::COUT + c0t[t]) = y[t];
    }
    if (op.dst_off4 >= 0) {   // halo rows of the buffer this op defines (2 above, 2 below its L_out interior rows)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const int tid = wave * 64 + lane;
        if (tid < 2 * op.dst_rs4) {
            sm4[op.dst_off4 + tid] = z;
            sm4[op.dst_off4 + (S::LOUT + 2) * op.dst_rs4 + tid] = z;
        }
    }
    FOP_STAMP();   // epilogue done (this wave)
    __builtin_amdgcn_s_setprio(0);
    lds_barrier();
    FOP_STAMP();
#undef FOP_STAMP
}

// ---- prologue of a fused program (shared by the generic and the static kernels).  GEOM: the program's geometry table (fused_geom.hpp) -
// the pass counts of the three staging copies, the index decomposition and every LDS address are then compile-time (round 3's generic
// form: 1 177 VALU + 492 SALU + 151 branches per wave, tools/isa_census.py) - or GeomNone (runtime geometry, worst-case pass counts).
template <class GEOM>
__device__ __forceinline__ void fused_prologue(const FusedArgs& a, const FusedLay& lay, float* smem, f32x4 (&ring)[kFusedRing], int tid, int lane, int wave, int b,
                                               long long* tr_base, int& tr) {
    f32x4* const sm4 = (f32x4*)smem;
#define FUSED_STAMP() do { if (tr_base) tr_base[tr] = (long long)__builtin_readcyclecounter(); ++tr; } while (0)
    // ---- prologue: every global load is issued first (weight ring of op 0, input window, parameters), the halo / padding
    //      zeros are written while they fly, and ONE barrier closes it.
    fused_ring_request(ring, fused_weights_rsrc(a.packed), (a.ops[0].sbase + (wave & a.msmask[0]) * a.slen[0] * 256) * 4, a.slen[0], (unsigned)lane * 16u);

    constexpr int NT_ = kFusedThreads;
    // float4 the three copies move at most (-> passes of 256 threads)
    constexpr int kIn4 = !GEOM::has ? 2048 : (GEOM::g.gc1 < 0 ? GEOM::g.L0 * 4 : GEOM::g.L0 * ((GEOM::g.gc1 + GEOM::g.gc2 + 3) / 4));
    constexpr int kS34 = !GEOM::has ? 1024 : GEOM::g.L3 * (GEOM::g.c3 / 4);
    constexpr int kPar4 = !GEOM::has ? 2048
                                     : ((GEOM::g.par_floats >= 0 ? GEOM::g.par_floats : GEOM::g.fpar_off + (16 * (GEOM::g.Cf + 4) + 16 + 3) / 4 * 4) + GEOM::g.tt_n) / 4;
    const int cin = lay.gc1 + lay.gc2;
    const int c4n = (cin + 3) >> 2;
    const int n_in = lay.L0 * c4n;
    constexpr int IK = (kIn4 + NT_ - 1) / NT_;   // input float4 per thread
    f32x4 iv[IK];
    int idst[IK], ck[IK];
    const bool vec_ok = ((lay.gc1 & 3) == 0) && ((lay.gc2 & 3) == 0);
    {
        // Unconditional loads from clamped addresses, zeros selected afterwards: a conditional load into a
        // zero-initialised register makes hipcc wait (vmcnt(0)) for the previous load before issuing the next one.
        int lk[IK];
        bool vk[IK];
#pragma unroll
        for (int k = 0; k < IK; ++k) {
            const int idx = tid + k * NT_;
            vk[k] = idx < n_in;
            const int idc = vk[k] ? idx : 0;
            lk[k] = lay.lg_c4n >= 0 ? (idc >> lay.lg_c4n) : (idc / c4n);
            ck[k] = (idc - lk[k] * c4n) << 2;
            idst[k] = vk[k] ? lay.in_off4 + (lk[k] + 2) * lay.in_rs4 + (ck[k] >> 2) : -1;
        }
        // (pass k is skipped as a whole - wave-uniform - when the window has fewer than k*256 float4: D=4 needs one pass of eight)
        if (vec_ok) {
#pragma unroll
            for (int k = 0; k < IK; ++k) {
                if (k * NT_ >= n_in) continue;   // iv[k] stays unset and is never stored (idst[k] < 0)
                const size_t pos = (size_t)b * lay.L0 + lk[k];
                const float* src = (ck[k] < lay.gc1) ? a.gsrc1 + pos * lay.gc1 + ck[k] : a.gsrc2 + pos * lay.gc2 + (ck[k] - lay.gc1);
                iv[k] = *(const f32x4*)src;
            }
        } else {
#pragma unroll
            for (int k = 0; k < IK; ++k) {   // channel padding (ce >= cin) is masked to zero at the LDS store below
                if (k * NT_ >= n_in) continue;
                const size_t pos = (size_t)b * lay.L0 + lk[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ce = ck[k] + e;
                    const float* src = (ce < lay.gc1) ? a.gsrc1 + pos * lay.gc1 + ce : (ce < cin) ? a.gsrc2 + pos * lay.gc2 + (ce - lay.gc1) : a.gsrc1 + pos * lay.gc1;
                    iv[k][e] = *src;
                }
            }
        }
    }
    // second staged input (skip tensor of a later concat): <= 4 float4 per thread
    constexpr int SK = (kS34 + NT_ - 1) / NT_;
    f32x4 sv[SK > 0 ? SK : 1];
    const int s3c4 = lay.c3 >> 2, n_s3 = lay.L3 * s3c4;
#pragma unroll
    for (int k = 0; k < SK; ++k) {
        const int idx = tid + k * NT_;
        if (k * NT_ >= n_s3) continue;   // wave-uniform (sv[k] stays unset, never stored)
        sv[k] = *(const f32x4*)(a.gsrc3 + ((size_t)b * n_s3 + (idx < n_s3 ? idx : 0)) * 4);
    }
    // parameters of every op ([bias | gamma | beta | rbias] blocks, contiguous in `packed` behind the weight streams) and the
    // slice of this timestep's conditioning row the segment's blocks use: two straight copies
    constexpr int PK = (kPar4 + NT_ - 1) / NT_;   // float4 per thread (<= 2048 float4 = 32 KB of parameters)
    f32x4 pv[PK];
    const int npar4 = lay.par_floats >> 2, ntt4 = lay.tt_n >> 2;
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int idx = tid + k * NT_;
        if (k * NT_ >= npar4 + ntt4) continue;   // wave-uniform: no pass beyond the data (pv[k] stays unset, never stored)
        const float* src = idx < npar4 ? a.packed + a.gpar_off + (size_t)idx * 4 : a.tt_row + (size_t)b * a.tt_stride + a.tt_lo + (size_t)(idx - npar4 < ntt4 ? idx - npar4 : 0) * 4;
        pv[k] = *(const f32x4*)src;
    }
    FUSED_STAMP();   // ring + input + parameter loads issued
    // zeros: the 2+2 halo rows of the staged input buffer and the channel padding of its rows (disjoint from what the
    // staging writes below).  Every other buffer gets its halo rows zeroed by the op that writes it.
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 2 * lay.in_rs4; i += NT_) {
            sm4[lay.in_off4 + i] = z;
            sm4[lay.in_off4 + (lay.in_rows - 2) * lay.in_rs4 + i] = z;
        }
        if (a.in_clear) {
            const int padw = lay.in_rs4 - c4n;   // float4 columns beyond the staged channels
            for (int i = tid; i < lay.L0 * padw; i += NT_) {
                const int l = i / padw, cc = i - l * padw;
                sm4[lay.in_off4 + (l + 2) * lay.in_rs4 + c4n + cc] = z;
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
    for (int k = 0; k < SK; ++k) {
        const int idx = tid + k * NT_;
        if (idx < n_s3) {
            const int l = idx / s3c4, c = idx - l * s3c4;
            sm4[lay.s3_off4 + (l + 2) * lay.s3_rs4 + lay.s3_col4 + c] = sv[k];
        }
    }
#pragma unroll
    for (int k = 0; k < PK; ++k) {
        const int idx = tid + k * NT_;
        if (idx < npar4) sm4[(lay.par_off >> 2) + idx] = pv[k];
        else if (idx - npar4 < ntt4) sm4[(lay.tt_off >> 2) + idx - npar4] = pv[k];
    }
    lds_barrier();
    FUSED_STAMP();

#undef FUSED_STAMP
}

// ---- final_conv[1] (1x1, Cf -> D) + DDPM posterior step + hard conditioning of a fused program (see final_step_kernel)
// Inputs of the DDPM step that do not depend on the network: requested one conv op EARLY (before the k-loop of the last
// Conv1dBlock), so their first-touch latency (~1.4 us from a cold L2) is hidden instead of heading the final op.
constexpr int kFinalPre = 4;   // elements per thread: H * D <= 1024
struct FinalPre { float xv[kFinalPre], nz[kFinalPre], hc[kFinalPre]; };

__device__ __forceinline__ void fused_final_prefetch(const FusedArgs& a, const FusedLay& lay, FinalPre& fp, int tid, int b) {
    if (a.fmode == 0) return;
    const int H = lay.H, n = H * a.D;
#pragma unroll
    for (int k = 0; k < kFinalPre; ++k) {
        const int idx = tid + k * kFusedThreads;
        if (k * kFusedThreads >= n) continue;   // wave-uniform; the registers stay unset and unused
        const int ic = idx < n ? idx : 0;
        const int p = ic / a.D, d = ic - p * a.D;
        const size_t o = (size_t)b * n + ic;
        fp.xv[k] = a.x_in[o];
        if (a.fmode == 1 && !a.rng.on && a.noise) fp.nz[k] = a.noise[o];
        if (a.hs && p == 0) fp.hc[k] = a.hs[(size_t)b * a.D + d];
        if (a.hg && p == H - 1) fp.hc[k] = a.hg[(size_t)b * a.D + d];
    }
}

// CF: the channel count of final_conv[0] when the program fixes it (static programs: the dot product unrolls, its 2 * CF / 4 LDS
// reads are issued together), 0 = read it from the argument block.
template <int CF = 0>
__device__ __forceinline__ void fused_final_op(const FusedArgs& a, const FusedLay& lay, const FusedOp& op, const FinalPre& fp, float* smem, int tid, int lane, int b) {
    f32x4* const sm4 = (f32x4*)smem;
    constexpr int NT_ = kFusedThreads;
    // ---- final_conv[1] (1x1, Cf -> D) + DDPM posterior step + hard conditioning (see final_step_kernel)
    float vmax = 0.f;
    const int H = lay.H, n = H * a.D;
    const int Cf = CF ? CF : lay.Cf;
    const int wrs = Cf + 4;   // LDS row stride of the staged weights: rows d and d + 1 start 4 banks apart
    const float* const fw = smem + lay.par_off + lay.fpar_off;
    // The step's noise drawn in place: a wave's kFinalPre x 64 elements are 64 consecutive-quad groups of the Philox stream (element idx = tid + k * NT_:
    // for each k the wave covers 16 quads), and one counter yields FOUR normals - lane (k, j) = (lane >> 4, lane & 15) evaluates the counter of quad j of
    // pass k ONCE and the lanes pick their component through the LDS crossbar (4 ds_bpermute per pass) instead of evaluating one counter per ELEMENT
    // (3.5 Philox + Box-Muller evaluations per thread at D = 14: ~5 % of the up program at B = 6 400).  Same values, same bits (philox_normal_at).
    float nzr[kFinalPre];
    const bool coop = a.fmode == 1 && a.rng.on && (((a.rng.elem0 + (unsigned long long)b * (unsigned)n) & 3ull) == 0ull) && (n & 3) == 0;   // workgroup-uniform
    if (coop) {
        const int qidx = (tid & ~63) + (lane >> 4) * NT_ + (lane & 15) * 4;   // trajectory-local index of the first element of this lane's quad
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (qidx < n) philox_normal4(a.rng.seed, a.rng.offset + ((a.rng.elem0 + (unsigned long long)b * (unsigned)n + (unsigned)qidx) >> 2), z);
#pragma unroll
        for (int k = 0; k < kFinalPre; ++k) {
            const int src = k * 16 + (lane >> 2);
            const float v0 = __shfl(z[0], src, 64), v1 = __shfl(z[1], src, 64), v2 = __shfl(z[2], src, 64), v3 = __shfl(z[3], src, 64);
            const int c = lane & 3;
            nzr[k] = c == 0 ? v0 : c == 1 ? v1 : c == 2 ? v2 : v3;
        }
    }
#pragma unroll
    for (int k = 0; k < kFinalPre; ++k) {
        const int idx = tid + k * NT_;
        if (idx >= n) continue;
        const int p = idx / a.D, d = idx - p * a.D;
        float s = fw[a.D * wrs + d];
        const float* wrow = fw + d * wrs;
#pragma unroll
        for (int c = 0; c < Cf; c += 4) {
            const f32x4 hv = sm4[op.src_off4 + (p + 2) * op.src_rs4 + (c >> 2)];
            const f32x4 wv = *(const f32x4*)(wrow + c);
            s = fmaf(hv[0], wv[0], s); s = fmaf(hv[1], wv[1], s);
            s = fmaf(hv[2], wv[2], s); s = fmaf(hv[3], wv[3], s);
        }
        const size_t o = (size_t)b * n + idx;
        float r;
        if (a.fmode == 0) {
            r = s;
        } else {
            const float xv = fp.xv[k];
            float x0 = a.k.predict_epsilon
                           ? __fsub_rn(__fmul_rn(a.k.sqrt_recip_alphas_cumprod, xv), __fmul_rn(a.k.sqrt_recipm1_alphas_cumprod, s))
                           : s;
            if (a.fmode == 3) {  // ddim_sample (diffusion_model_base.py:216-237): x_start is not clamped there
                const float pn = a.k.predict_epsilon
                                     ? s
                                     : __fdiv_rn(__fsub_rn(__fmul_rn(a.k.sqrt_recip_alphas_cumprod, xv), s), a.k.sqrt_recipm1_alphas_cumprod);
                r = __fadd_rn(__fmul_rn(x0, a.k.ddim_k1), __fmul_rn(a.k.ddim_k2, pn));
                if (a.hs && p == 0) r = fp.hc[k];
                if (a.hg && p == H - 1) r = fp.hc[k];
            } else {
                if (a.k.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                r = __fadd_rn(__fmul_rn(a.k.posterior_mean_coef1, x0), __fmul_rn(a.k.posterior_mean_coef2, xv));
                if (a.fmode == 1) {
                    if (a.rng.on) r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.k.noise_scale, coop ? nzr[k] : philox_normal_at(a.rng.seed, a.rng.offset, a.rng.elem0 + o)), a.k.noise_std_extra));
                    else if (a.noise) r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.k.noise_scale, fp.nz[k]), a.k.noise_std_extra));
                    if (a.hs && p == 0) r = fp.hc[k];
                    if (a.hg && p == H - 1) r = fp.hc[k];
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
}

// Generic kernel: walks a runtime op list (any sequence of the shapes above).
template <bool SAVE>
__global__ __launch_bounds__(kFusedThreads) void fused_level_kernel(const FusedArgs a) {
#ifndef MPDX_NO_WARM_KERNARG   // dev A/B switch
    warm_kernarg<(int)sizeof(FusedArgs)>();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    int tr = 0;
    long long* const tr_base = (MPDX_TRACE_PTR(a.trace) && b == 0 && lane == 0) ? a.trace + wave * 128 : nullptr;   // 128 slots per wave
    if (tr_base) tr_base[tr] = (long long)__builtin_readcyclecounter();
    ++tr;
    f32x4 ring[kFusedRing];
    FinalPre fp;
    const FusedLay lay = fused_lay<GeomNone>(a);
    fused_prologue<GeomNone>(a, lay, smem, ring, tid, lane, wave, b, tr_base, tr);
    for (int oi = 0; oi < a.nops; ++oi) {
        const FusedOp op = a.ops[oi];
        if (oi + 1 < a.nops && a.ops[oi + 1].shape == kFusedShapeFinal) fused_final_prefetch(a, lay, fp, tid, b);
        if (op.shape == kFusedShapeFinal) {
            fused_final_op(a, lay, op, fp, smem, tid, lane, b);
            if (tr_base) tr_base[tr] = (long long)__builtin_readcyclecounter();
            ++tr;
            continue;
        }
        // the next conv op's wave-stream: this op's ring runs on into it (after the last conv op: any valid block, unused)
        int nbase = op.sbase * 4;
        int nmax = 0;
        if (oi + 1 < a.nops && a.ops[oi + 1].shape != kFusedShapeFinal) {
            nbase = (a.ops[oi + 1].sbase + (wave & a.msmask[oi + 1]) * a.slen[oi + 1] * 256) * 4;
            nmax = a.slen[oi + 1] - 1;
        }
        switch (op.shape) {
#define X(id, M, K, N, R, CO, LO, G) \
    case id: fused_conv_op<FusedShape<M, K, N, R, CO, LO, G>, SAVE>(a, lay, op, ring, smem, wave, lane, b, nbase, nmax, tr_base, tr); break;
            MPDX_FUSED_SHAPES(X)
#undef X
            default: break;
        }
    }
}

// ---- STATIC programs.  The op sequences of the standard networks (dim_mults (1,2,4,8) and (1,2,4), unet_input_dim 32, H 64) are
// compile-time lists of shape ids: the op loop, the shape switch and every descriptor index disappear (descriptor fields are read
// at immediate offsets of the argument block; no per-op dispatch), and per-op address arithmetic is no longer hoisted out of a
// 16-way switch into long-lived registers.  Any other network runs the generic kernel above.
template <int ID> struct FusedShapeOf;
#define X(id, M, K, N, R, CO, LO, G) template <> struct FusedShapeOf<id> { using type = FusedShape<M, K, N, R, CO, LO, G>; };
MPDX_FUSED_SHAPES(X)
#undef X

// op I of a static program: the runtime descriptor (weight-stream base, training offsets) with every LDS geometry field replaced by the
// program's compile-time table entry, when it has one
template <class GEOM, int I, bool SAVE>
__device__ __forceinline__ FusedOp fused_static_desc(const FusedArgs& a) {
    FusedOp op = a.ops[I];
    if constexpr (GEOM::has) {
        constexpr FusedGeomOp g = GEOM::g.ops[I];
        op.shape = g.shape;
        op.src_off4 = g.src_off4; op.src_rs4 = g.src_rs4; op.rsrc_off4 = g.rsrc_off4; op.rsrc_rs4 = g.rsrc_rs4;
        op.res_off4 = g.res_off4; op.res_rs4 = g.res_rs4; op.dst_off4 = g.dst_off4; op.dst_rs4 = g.dst_rs4;
        op.p_off = g.p_off; op.tb_off = g.tb_off;
        if constexpr (!SAVE) op.gdst = g.gdst;   // (the training forward switches the planning path's global outputs off at run time)
    }
    return op;
}

template <class GEOM, int SH, int I, int NEXT_SH, int PREV_COUT, bool SAVE>
__device__ __forceinline__ void fused_static_op(const FusedArgs& a, const FusedLay& lay, f32x4 (&ring)[kFusedRing], FinalPre& fp, float* smem, int tid, int wave,
                                                int lane, int b, long long* tr_base, int& tr) {
    if constexpr (NEXT_SH == kFusedShapeFinal) fused_final_prefetch(a, lay, fp, tid, b);
    const FusedOp op = fused_static_desc<GEOM, I, SAVE>(a);
    if constexpr (SH == kFusedShapeFinal) {
        fused_final_op<PREV_COUT>(a, lay, op, fp, smem, tid, lane, b);
        if (tr_base) tr_base[tr] = (long long)__builtin_readcyclecounter();
        ++tr;
    } else {
        int nbase = a.ops[I].sbase * 4;
        int nmax = 0;
        if constexpr (NEXT_SH >= 0 && NEXT_SH != kFusedShapeFinal) {
            using N = typename FusedShapeOf<NEXT_SH>::type;
            nbase = (a.ops[I + 1].sbase + (wave & (N::MSn < kFusedWaves ? N::MSn - 1 : kFusedWaves - 1)) * (N::SLEN * 256)) * 4;
            nmax = N::SLEN - 1;
        }
        fused_conv_op<typename FusedShapeOf<SH>::type, SAVE>(a, lay, op, ring, smem, wave, lane, b, nbase, nmax, tr_base, tr);
    }
}

template <class GEOM_, int... SH>
struct FusedSeq {
    using GEOM = GEOM_;
    static constexpr int N = sizeof...(SH);
    static constexpr int ids[sizeof...(SH)] = {SH...};
    template <int I>
    static constexpr int prev_cout() {   // C_out of the op before op I (what the final op reads); 0 if there is none
        if constexpr (I > 0 && ids[I > 0 ? I - 1 : 0] != kFusedShapeFinal) return FusedShapeOf<ids[I > 0 ? I - 1 : 0]>::type::COUT;
        else return 0;
    }
    template <int I, bool SAVE>
    __device__ static __forceinline__ void run_from(const FusedArgs& a, const FusedLay& lay, f32x4 (&ring)[kFusedRing], FinalPre& fp, float* smem, int tid, int wave,
                                                    int lane, int b, long long* tr_base, int& tr) {
        if constexpr (I < N) {
            fused_static_op<GEOM, ids[I], I, (I + 1 < N ? ids[I + 1 < N ? I + 1 : I] : -1), prev_cout<I>(), SAVE>(a, lay, ring, fp, smem, tid, wave, lane, b, tr_base, tr);
            run_from<I + 1, SAVE>(a, lay, ring, fp, smem, tid, wave, lane, b, tr_base, tr);
        }
    }
};

template <class SEQ, bool SAVE = false>
__global__ __launch_bounds__(kFusedThreads) void fused_program_kernel(const FusedArgs a) {
#ifndef MPDX_NO_WARM_KERNARG
    warm_kernarg<(int)sizeof(FusedArgs)>();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    int tr = 0;
    long long* const tr_base = (MPDX_TRACE_PTR(a.trace) && b == 0 && lane == 0) ? a.trace + wave * 128 : nullptr;
    if (tr_base) tr_base[tr] = (long long)__builtin_readcyclecounter();
    ++tr;
    f32x4 ring[kFusedRing];
    FinalPre fp;
    const FusedLay lay = fused_lay<typename SEQ::GEOM>(a);
    fused_prologue<typename SEQ::GEOM>(a, lay, smem, ring, tid, lane, wave, b, tr_base, tr);
    SEQ::template run_from<0, SAVE>(a, lay, ring, fp, smem, tid, wave, lane, b, tr_base, tr);
}

// the programs of the standard networks
using FusedSeqDown = FusedSeq<GeomDown, 0, 1, 2, 2, 3, 4, 5, 6, 6, 7>;                       // downs.0 + downs.1
using FusedSeqUpA = FusedSeq<GeomNone, 8, 9, 10, 10, 11>;                                    // the up level at L = 16 (cat 256 -> 64)
using FusedSeqUpB = FusedSeq<GeomNone, 12, 13, 14, 14, 15, 2, kFusedShapeFinal>;             // the up level at L = 32 + final_conv + DDPM step
using FusedSeqUpAB = FusedSeq<GeomUpAB, 8, 9, 10, 10, 11, 12, 13, 14, 14, 15, 2, kFusedShapeFinal>;   // both up levels in one launch
using FusedSeqDown3 = FusedSeq<GeomDown3, 0, 1, 2, 2, 3, 4, 5, 6, 6, 7, 16, 17, 18, 18, 19>;   // downs.0 + downs.1 + downs.2 in one launch
using FusedSeqMid2 = FusedSeq<GeomNone, 16, 17, 18, 18, 19>;                                 // downs.2 (C = 128, L = 16): two tile rows per wave
using FusedSeqMid3 = FusedSeq<GeomMid3, 16, 17, 18, 18, 18, 18, 18, 18>;                     // three-level network: downs.2 (no Downsample1d) + mid_block1 + mid_block2

}  // namespace mpdx
