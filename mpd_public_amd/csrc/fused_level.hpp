// fused_level.hpp - whole-trajectory fused kernels for the outer levels of TemporalUnet.
//
// Why.  At B=100 the per-layer path is bound by launch boundaries and per-kernel fixed phases (DESIGN.md section 3:
// 44 dependent launches per denoising step, ~3 us boundary + ~2 us fixed cost each).  At the two outer resolutions one
// workgroup can own ALL channels of a trajectory (C*L = 2048 floats, C <= 64), so GroupNorm statistics stay local and a
// whole run of layers executes out of LDS in ONE launch:
//     downs[i] (i = 0, 1):  ResidualTemporalBlock x2 -> Downsample1d                     (temporal_unet.py:141-150)
//     ups[j]   (last two):  cat(x, skip) -> ResidualTemporalBlock x2 -> Upsample1d       (temporal_unet.py:158-165)
//     and, after the last Upsample1d: final_conv (Conv1dBlock -> Conv1d 1x1) + the DDPM posterior step
//                                                            (temporal_unet.py:167, diffusion_model_base.py:121-155)
// The arithmetic per layer is the same as conv_block.hpp (same packed weights, same fp32 MFMA, same GroupNorm/Mish);
// only the schedule differs: activations never leave the CU between layers.
//
// Structure.  One workgroup (8 waves) = one trajectory.  LDS holds the activation buffers in the zero-haloed
// channel-last layout [L+4 rows][C + pad] (rows 0,1 and L+2,L+3 are zero: conv padding; written by the op that defines
// the buffer), plus the K-partial buffer.  The host places the buffers by live range (buffers whose lives do not
// intersect share addresses), which lets the two outer down levels run as ONE program of 12 ops within 160 KB.
// A tiny op list (kernel argument) is interpreted: each conv is M = C_out (all), N = L positions, K = C_in*taps;
// the (C_out/16)*(L/16) = 8 (or 4) MFMA sub-tiles map one per wave (K split in two when there are 4), partials meet
// in LDS, then wave g normalises GroupNorm group g of the trajectory.
// Weights reach the MFMAs through LDS-DMA (global_load_lds, 16 B per lane): the packed A-fragment layout
// Wp[m16][c16][slot][lane][4] is already a sequence of lane-linear 1-KiB blocks, which is exactly the DMA's destination
// shape (wave-uniform base + lane*16).  The blocks of op i+1 are DMA'd right after op i's MFMA barrier, so their
// HBM/L2 latency hides under op i's GroupNorm epilogue; the k-loop itself reads A and B fragments from LDS only
// (no vmcnt waits inside it - the register-ring version of this loop was serialised by hipcc to ~2 loads in flight).
// Ops whose weights exceed the LDS weight window (the 128->64 k5 block of ups: 160 KiB) run in c16 chunks.
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
    int kind;          // FOP_*
    int mode, ks;      // CONV_S1 / CONV_DOWN / CONV_UPT, taps
    int src, dst, res; // LDS buffer ids (-1: none)
    int gdst;          // index into FusedArgs.gout (-1: none)
    int cin_pad, cout, L_in, L_out, gs;
    int w_off, b_off, ga_off, be_off, tb_off;  // float offsets into packed weights / the time-table row (-1: none)
    int p_off;         // float offset of this op's staged [bias | gamma | beta | tbias] block (4*cout floats) in LDS
    // host-precomputed so that the kernel needs no integer division (sub-tile counts and group sizes are powers of two)
    int lg_T, lg_MSn, lg_gs, lg_M4, ntap, nslot, nc16, cchunk;
};

constexpr int kMaxFusedOps = 14;
constexpr int kMaxFusedBufs = 16;

struct FusedArgs {
    const float* packed;
    const float* tt_row;
    const float* gsrc1;  // kernel input [B][L0][gc1]
    const float* gsrc2;  // second half of a channel concat [B][L0][gc2], or null
    float* gout[3];      // global outputs, channel-last [B][L][C]
    int gc1, gc2, L0, in_buf;
    int B, nops, nbufs;
    int red_off4;        // K-partial buffer (float4 units)
    int par_off4;        // staged per-op parameter vectors (float4 units)
    int par_floats;      // their total length (<= 8*512)
    int w_off4;          // LDS weight window (float4 units)
    int w_cap_blocks;    // its capacity in 1-KiB A-fragment blocks
    int lg_c4n;          // log2(float4 per staged input row) or -1 (generic division path)
    int n_runs;          // parameter runs = 4 * (#conv ops): run r = op (r>>2), vector (r&3)
    int lds_float4;      // total LDS in float4 units (zeroed at start)
    FusedOp ops[kMaxFusedOps];
    FusedBuf bufs[kMaxFusedBufs];
    // FOP_FINAL extras (final 1x1 conv + DDPM step), as FinalArgs of mpdx.hip
    const float* x_in; const float* noise; const float* hs; const float* hg;
    float* out; float* chain; uint32_t* absmax;
    int D, Cf, fmode, n_per_ctx, fw_off, fb_off;
    mpdx_step_coefs k;
    long long* trace;    // dev tool: s_memtime stamps of workgroup 0 / wave 0 (null in production)
};

struct FusedWork {   // one wave's share of a conv op
    int ms, ns, kpart, ksplit, MSn;
};

__device__ __forceinline__ FusedWork fused_work(const FusedOp& op, int wave) {
    FusedWork w;
    w.MSn = 1 << op.lg_MSn;
    const int sub = wave & ((1 << op.lg_T) - 1);
    w.kpart = wave >> op.lg_T;
    w.ksplit = 8 >> op.lg_T;
    w.ms = sub & (w.MSn - 1);
    w.ns = sub >> op.lg_MSn;
    return w;
}

// LDS-DMA the A-fragment blocks of input-channel chunk [c_lo, c_lo + cn) of `op` into the weight window.
// Window layout [ms][c16_local][slot] x 1 KiB: for a fixed ms both the global source and the window are one contiguous
// run of cn*nslot blocks, so the copy loop needs no index decoding.  Every wave issues its share; completion =
// vmcnt(0) + barrier.
__device__ __forceinline__ void fused_dma_weights(const FusedArgs& a, const FusedOp& op, int c_lo, int cn, int wave, int lane, float* smem) {
    const int run = cn * op.nslot;
    const int MSn = 1 << op.lg_MSn;
    for (int ms = 0; ms < MSn; ++ms) {
        const float* src = a.packed + op.w_off + ((size_t)(ms * op.nc16 + c_lo) * op.nslot) * 256 + lane * 4;
        float* dst = smem + (size_t)a.w_off4 * 4 + (size_t)(ms * run) * 256;
        for (int r = wave; r < run; r += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)r * 256),
                                             (__attribute__((address_space(3))) void*)(dst + (size_t)r * 256), 16, 0, 0);
    }
}

// K-loop of one wave over 16-input-channel chunks [cl_lo, cl_hi) of the weight window: taps are unrolled at compile
// time and there is no per-k-group control flow, so hipcc software-pipelines the ds_reads against the MFMAs.
// wrow: this wave's A blocks in the window (+lane), brow: its B row in the activation buffer (float4 units).
template <int MODE, int KS>
__device__ __forceinline__ void fused_mfma(const f32x4* __restrict__ wrow, const f32x4* __restrict__ brow, int rs4, int c_base, int cl_lo,
                                           int cl_hi, int par, f32x4& acc0, f32x4& acc1) {
    constexpr int NTAP = (MODE == CONV_UPT) ? 2 : KS;
    constexpr int NSLOT = (MODE == CONV_UPT) ? 4 : KS;
    for (int cl = cl_lo; cl < cl_hi; ++cl) {
        f32x4 af[NTAP], bf[NTAP];
#pragma unroll
        for (int ts = 0; ts < NTAP; ++ts) {
            const int roff = (MODE == CONV_UPT) ? ((ts == 0) ? 0 : (par == 0 ? -1 : 1)) : ts;
            const int slot = (MODE == CONV_UPT) ? (par * 2 + ts) : ts;
            af[ts] = wrow[(cl * NSLOT + slot) * 64];
            bf[ts] = brow[roff * rs4 + (c_base + cl) * 4];
        }
#pragma unroll
        for (int ts = 0; ts < NTAP; ++ts)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((ts & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ts][e], bf[ts][e], acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ts][e], bf[ts][e], acc1, 0, 0, 0);
            }
    }
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

    // ---- prologue: every global load is issued first (input window, parameter vectors, weight DMA of the first op), the
    //      halo / padding zeros are written while they fly, and ONE barrier closes it.
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
        // zero-initialised register makes hipcc wait (vmcnt(0)) for the previous load before issuing the next one,
        // which serialised these four round trips (~6 k cycles per launch).
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
    FUSED_STAMP();   // input loads issued
    // parameter vectors [bias | gamma | beta | tbias] x C_out of every op: run r is handled by wave r % 8, lane = channel
    float* par = smem + (size_t)a.par_off4 * 4;
    constexpr int RK = 7;   // runs per wave (<= 14 ops * 4 / 8)
    float pvv[RK];
    int pdst[RK];   // LDS destination (floats from the start of the parameter area), -1: nothing to store
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const int r = wave + k * 8;
        pvv[k] = 0.f;
        pdst[k] = -1;
        if (r < a.n_runs && lane < a.ops[r >> 2].cout) {
            const FusedOp& op = a.ops[r >> 2];
            pdst[k] = op.p_off + (r & 3) * op.cout + lane;
            const int which = r & 3;
            const float* src = (which == 0) ? a.packed + op.b_off
                               : (op.kind != FOP_CONV_GN) ? nullptr
                               : (which == 1) ? a.packed + op.ga_off
                               : (which == 2) ? a.packed + op.be_off
                               : (op.tb_off >= 0 ? a.tt_row + op.tb_off : nullptr);
            if (src) pvv[k] = src[lane];
        }
    }
    FUSED_STAMP();   // parameter loads issued
    // weights of the first op
    FusedWork wk = fused_work(a.ops[0], wave);
    fused_dma_weights(a, a.ops[0], 0, a.ops[0].cchunk, wave, lane, smem);
    FUSED_STAMP();   // first op's weight DMA issued
    // zeros: the 2+2 halo rows of the staged input buffer and the channel padding of its rows (disjoint from what the
    // staging writes below).  Every other buffer gets its halo rows zeroed by the op that writes it (buffers with disjoint
    // live ranges share LDS addresses, so they cannot all be prepared here); interiors are fully overwritten before use.
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
    for (int idx = tid + IK * 512; idx < n_in; idx += 512) {   // (not reached for the supported shapes; kept for safety)
        const int l = idx / c4n, c = (idx - l * c4n) << 2;
        const size_t pos = (size_t)b * a.L0 + l;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ce = c + e;
            if (ce < a.gc1) v[e] = a.gsrc1[pos * a.gc1 + ce];
            else if (ce < cin) v[e] = a.gsrc2[pos * a.gc2 + (ce - a.gc1)];
        }
        sm4[ib.off4 + (l + 2) * ib.rs4 + (c >> 2)] = v;
    }
#pragma unroll
    for (int k = 0; k < RK; ++k)
        if (pdst[k] >= 0) par[pdst[k]] = pvv[k];
    __builtin_amdgcn_s_waitcnt(0);  // this wave's DMA blocks have landed
    __syncthreads();
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

        // ------------------------------------------------------------------ conv: MFMA over this wave's k-groups
        wk = fused_work(op, wave);
        const FusedBuf sb = a.bufs[op.src];
        int boff, npos;
        if (op.mode == CONV_UPT) {
            const int m = (wk.ns >> 1) * 16 + j;
            boff = sb.off4 + (m + 2) * sb.rs4 + q;
            npos = 2 * m + (wk.ns & 1);
        } else {
            const int l = wk.ns * 16 + j;
            const int pad = (op.mode == CONV_S1) ? (op.ks >> 1) : 1;
            boff = sb.off4 + ((op.mode == CONV_DOWN ? 2 * l : l) + 2 - pad) * sb.rs4 + q;
            npos = l;
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const int par = wk.ns & 1;
        const int cchunk = op.cchunk;
        for (int c_lo = 0; c_lo < op.nc16; c_lo += cchunk) {
            const int cn = min(cchunk, op.nc16 - c_lo);
            if (c_lo > 0) {  // next chunk of a large op: the window is free once every wave finished the previous chunk
                __syncthreads();
                fused_dma_weights(a, op, c_lo, cn, wave, lane, smem);
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
            }
            // K split over 16-channel chunks (host guarantees cn % ksplit == 0 or ksplit == 1)
            const int per = (wk.ksplit == 1) ? cn : ((cn + 1) >> 1);   // ksplit is 1 or 2
            const int cl_lo = wk.kpart * per, cl_hi = min(cn, cl_lo + per);
            const f32x4* wrow = sm4 + a.w_off4 + (size_t)(wk.ms * cn) * op.nslot * 64 + lane;
            const f32x4* brow = sm4 + boff;
            if (op.mode == CONV_S1 && op.ks == 5) fused_mfma<CONV_S1, 5>(wrow, brow, sb.rs4, c_lo, cl_lo, cl_hi, par, acc0, acc1);
            else if (op.mode == CONV_S1) fused_mfma<CONV_S1, 1>(wrow, brow, sb.rs4, c_lo, cl_lo, cl_hi, par, acc0, acc1);
            else if (op.mode == CONV_DOWN) fused_mfma<CONV_DOWN, 3>(wrow, brow, sb.rs4, c_lo, cl_lo, cl_hi, par, acc0, acc1);
            else fused_mfma<CONV_UPT, 4>(wrow, brow, sb.rs4, c_lo, cl_lo, cl_hi, par, acc0, acc1);
        }
        const f32x4 acc = acc0 + acc1;
        FUSED_STAMP();   // MFMA loop done
        const int MTP4 = (op.cout + 4) >> 2;
        const int N = op.L_out;
        sm4[a.red_off4 + (wk.kpart * N + npos) * MTP4 + wk.ms * 4 + q] = acc;
        const int ksplit_cur = wk.ksplit;
        __syncthreads();
        FUSED_STAMP();   // partials visible; the weight window is free

        // DMA the next conv's weights now: the transfer overlaps this op's epilogue
        if (oi + 1 < a.nops && a.ops[oi + 1].kind != FOP_FINAL)
            fused_dma_weights(a, a.ops[oi + 1], 0, a.ops[oi + 1].cchunk, wave, lane, smem);

        // ------------------------------------------------------------------ epilogue
        const float* par_op = smem + (size_t)a.par_off4 * 4 + op.p_off;  // [bias | gamma | beta | tbias] x cout
        const float* bias = par_op;
        if (op.kind == FOP_CONV_GN) {
            // wave g normalises GroupNorm group g (8 groups per trajectory)
            const int gs = op.gs, re = gs * N;
            const float inv_re = (re == 256) ? (1.0f / 256.0f) : (1.0f / 128.0f);
            const FusedBuf db = a.bufs[op.dst];
            if (re == 256) {
                const int e0 = lane * 4;
                const int l = e0 >> op.lg_gs, c = wave * gs + (e0 & (gs - 1));
                const f32x4 bi = *(const f32x4*)(bias + c);
                const f32x4 ga = *(const f32x4*)(par_op + op.cout + c), be = *(const f32x4*)(par_op + 2 * op.cout + c);
                const f32x4 tb = *(const f32x4*)(par_op + 3 * op.cout + c);
                f32x4 v = sm4[a.red_off4 + l * MTP4 + (c >> 2)];
                for (int k = 1; k < ksplit_cur; ++k) v += sm4[a.red_off4 + (k * N + l) * MTP4 + (c >> 2)];
                v += bi;
                const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * inv_re;
                const f32x4 d = v - mean;
                const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * inv_re;
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = mish(d[e] * rstd * ga[e] + be[e]);
                y += tb;
                if (op.res >= 0) y += sm4[a.bufs[op.res].off4 + (l + 2) * a.bufs[op.res].rs4 + (c >> 2)];
                sm4[db.off4 + (l + 2) * db.rs4 + (c >> 2)] = y;
                if (op.gdst >= 0) *(f32x4*)(a.gout[op.gdst] + ((size_t)b * N + l) * op.cout + c) = y;
            } else {  // re == 128
                const int e0 = lane * 2;
                const int l = e0 >> op.lg_gs, c = wave * gs + (e0 & (gs - 1));
                const f32x2 bi = *(const f32x2*)(bias + c);
                const f32x2 ga = *(const f32x2*)(par_op + op.cout + c), be = *(const f32x2*)(par_op + 2 * op.cout + c);
                const f32x2 tb = *(const f32x2*)(par_op + 3 * op.cout + c);
                const float* redf = smem + (size_t)a.red_off4 * 4;
                f32x2 v = *(const f32x2*)(redf + (size_t)l * (MTP4 * 4) + c);
                for (int k = 1; k < ksplit_cur; ++k) v += *(const f32x2*)(redf + (size_t)(k * N + l) * (MTP4 * 4) + c);
                v += bi;
                const float mean = wave_sum(v[0] + v[1]) * inv_re;
                const f32x2 d = v - mean;
                const float var = wave_sum(d[0] * d[0] + d[1] * d[1]) * inv_re;
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                f32x2 y;
#pragma unroll
                for (int e = 0; e < 2; ++e) y[e] = mish(d[e] * rstd * ga[e] + be[e]);
                y += tb;
                if (op.res >= 0) y += *(const f32x2*)(smem + ((size_t)a.bufs[op.res].off4 + (size_t)(l + 2) * a.bufs[op.res].rs4) * 4 + c);
                *(f32x2*)(smem + ((size_t)db.off4 + (size_t)(l + 2) * db.rs4) * 4 + c) = y;
                if (op.gdst >= 0) *(f32x2*)(a.gout[op.gdst] + ((size_t)b * N + l) * op.cout + c) = y;
            }
        } else {  // bias only: residual 1x1 conv / Downsample1d / Upsample1d
            const int M4 = op.cout >> 2;
            for (int idx = tid; idx < N * M4; idx += 512) {
                const int l = idx >> op.lg_M4, c4 = idx & (M4 - 1);
                f32x4 v = sm4[a.red_off4 + l * MTP4 + c4];
                for (int k = 1; k < ksplit_cur; ++k) v += sm4[a.red_off4 + (k * N + l) * MTP4 + c4];
                v += *(const f32x4*)(bias + c4 * 4);
                if (op.dst >= 0) sm4[a.bufs[op.dst].off4 + (l + 2) * a.bufs[op.dst].rs4 + c4] = v;
                if (op.gdst >= 0) *(f32x4*)(a.gout[op.gdst] + ((size_t)b * N + l) * op.cout + c4 * 4) = v;
            }
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
        __builtin_amdgcn_s_waitcnt(0);  // the next op's weight blocks issued by this wave have landed
        __syncthreads();
        FUSED_STAMP();
    }
}
#undef FUSED_STAMP

}  // namespace mpdx
