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
// Structure.  One workgroup (8 waves) = one trajectory.  LDS holds a few activation buffers in the zero-haloed
// channel-last layout [L+4 rows][C + pad] (rows 0,1 and L+2,L+3 stay zero: conv padding), plus the K-partial buffer.
// A tiny op list (kernel argument) is interpreted: each conv is M = C_out (all), N = L positions, K = C_in*taps;
// the (C_out/16)*(L/16) = 8 (or 4) MFMA sub-tiles map one per wave (K split in two when there are 4), partials meet
// in LDS, then wave g normalises GroupNorm group g of the trajectory.  The next op's first weight fragments are
// prefetched before the current op's epilogue so that the L2 latency of the weight stream is paid once per launch.
#pragma once
#include "conv_block.hpp"

namespace mpdx {

enum : int { FOP_CONV_GN = 0, FOP_CONV_BIAS = 1, FOP_FINAL = 2 };

struct FusedBuf {
    int off4;   // offset in LDS, float4 units
    int rs4;    // row stride, float4 units
};

struct FusedOp {
    int kind;          // FOP_*
    int mode, ks;      // CONV_S1 / CONV_DOWN / CONV_UPT, taps
    int src, dst, res; // LDS buffer ids (-1: none)
    int gdst;          // index into FusedArgs.gout (-1: none)
    int cin_pad, cout, L_in, L_out, gs;
    int w_off, b_off, ga_off, be_off, tb_off;  // float offsets into packed weights / the time-table row (-1: none)
};

constexpr int kMaxFusedOps = 14;
constexpr int kMaxFusedBufs = 8;

struct FusedArgs {
    const float* packed;
    const float* tt_row;
    const float* gsrc1;  // kernel input [B][L0][gc1]
    const float* gsrc2;  // second half of a channel concat [B][L0][gc2], or null
    float* gout[3];      // global outputs, channel-last [B][L][C]
    int gc1, gc2, L0, in_buf;
    int B, nops, nbufs;
    int red_off4;        // K-partial buffer (float4 units)
    int lds_float4;      // total LDS in float4 units (zeroed at start)
    FusedOp ops[kMaxFusedOps];
    FusedBuf bufs[kMaxFusedBufs];
    // FOP_FINAL extras (final 1x1 conv + DDPM step), as FinalArgs of mpdx.hip
    const float* x_in; const float* noise; const float* hs; const float* hg;
    float* out; float* chain; uint32_t* absmax;
    int D, Cf, fmode, n_per_ctx, fw_off, fb_off;
    mpdx_step_coefs k;
};

constexpr int kFusedPF = 4;

struct FusedWork {   // one wave's share of a conv op
    int ms, ns, kpart, g_lo, g_hi, ntap, nslot, nc16, G;
    const float* wbase;
};

__device__ __forceinline__ FusedWork fused_work(const FusedArgs& a, const FusedOp& op, int wave, int lane) {
    FusedWork w;
    const int MSn = op.cout >> 4;
    const int NSn = (op.mode == CONV_UPT) ? (op.L_in >> 4) * 2 : (op.L_out >> 4);
    const int T = MSn * NSn;            // 4 or 8 sub-tiles (host-verified)
    const int ksplit = 8 / T;
    const int sub = wave % T;
    w.kpart = wave / T;
    w.ms = sub % MSn;
    w.ns = sub / MSn;
    w.ntap = (op.mode == CONV_UPT) ? 2 : op.ks;
    w.nslot = (op.mode == CONV_UPT) ? 4 : op.ks;
    w.nc16 = op.cin_pad >> 4;
    const int G = w.nc16 * w.ntap;
    w.G = G;
    const int per = (G + ksplit - 1) / ksplit;
    w.g_lo = w.kpart * per;
    w.g_hi = min(G, w.g_lo + per);
    w.wbase = a.packed + op.w_off + (size_t)w.ms * w.nc16 * w.nslot * 256 + lane * 4;
    return w;
}

__device__ __forceinline__ f32x4 fused_load_a(const FusedOp& op, const FusedWork& w, int g) {
    g = g < w.G ? g : w.G - 1;  // clamp to a valid k-group: ring loads are unconditional
    const int c16 = g / w.ntap, ts = g - c16 * w.ntap;
    const int slot = (op.mode == CONV_UPT) ? ((w.ns & 1) * 2 + ts) : ts;
    return *(const f32x4*)(w.wbase + ((size_t)c16 * w.nslot + slot) * 256);
}

__global__ __launch_bounds__(512) void fused_level_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* const sm4 = (f32x4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int j = lane & 15, q = lane >> 4;

    // weight ring for the first op (overlaps the LDS clear and the input staging)
    FusedWork wk = fused_work(a, a.ops[0], wave, lane);
    f32x4 ring[kFusedPF];
#pragma unroll
    for (int u = 0; u < kFusedPF; ++u) ring[u] = fused_load_a(a.ops[0], wk, wk.g_lo + u);

    // ---- clear LDS (halo rows / channel padding must be zero), then stage the input trajectory window
    for (int i = tid; i < a.lds_float4; i += 512) sm4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    {
        const FusedBuf ib = a.bufs[a.in_buf];
        const int cin = a.gc1 + a.gc2;
        const int c4n = (cin + 3) >> 2;
        const bool vec_ok = ((a.gc1 & 3) == 0) && ((a.gc2 & 3) == 0);
        for (int idx = tid; idx < a.L0 * c4n; idx += 512) {
            const int l = idx / c4n, c = (idx - l * c4n) << 2;
            const size_t pos = (size_t)b * a.L0 + l;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (vec_ok) {
                v = (c < a.gc1) ? *(const f32x4*)(a.gsrc1 + pos * a.gc1 + c) : *(const f32x4*)(a.gsrc2 + pos * a.gc2 + (c - a.gc1));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ce = c + e;
                    if (ce < a.gc1) v[e] = a.gsrc1[pos * a.gc1 + ce];
                    else if (ce < cin) v[e] = a.gsrc2[pos * a.gc2 + (ce - a.gc1)];
                }
            }
            sm4[ib.off4 + (l + 2) * ib.rs4 + (c >> 2)] = v;
        }
    }
    __syncthreads();

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
                    if (a.k.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                    r = __fadd_rn(__fmul_rn(a.k.posterior_mean_coef1, x0), __fmul_rn(a.k.posterior_mean_coef2, xv));
                    if (a.fmode == 1) {
                        if (a.noise) r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.k.noise_scale, a.noise[o]), a.k.noise_std_extra));
                        if (a.hs && p == 0) r = a.hs[(size_t)b * a.D + d];
                        if (a.hg && p == H - 1) r = a.hg[(size_t)b * a.D + d];
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
            continue;
        }

        // ------------------------------------------------------------------ conv: MFMA over this wave's k-groups
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
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int par = wk.ns & 1;
        for (int g0 = wk.g_lo; g0 < wk.g_hi; g0 += kFusedPF) {
#pragma unroll
            for (int u = 0; u < kFusedPF; ++u) {
                const int g = g0 + u;
                if (g < wk.g_hi) {
                    const int c16 = g / wk.ntap, ts = g - c16 * wk.ntap;
                    const int roff = (op.mode == CONV_UPT) ? ((ts == 0) ? 0 : (par == 0 ? -1 : 1)) : ts;
                    const f32x4 bf = sm4[boff + roff * sb.rs4 + c16 * 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][e], bf[e], acc, 0, 0, 0);
                }
                ring[u] = fused_load_a(op, wk, g + kFusedPF);
            }
        }
        const int MTP4 = (op.cout + 4) >> 2;
        const int N = op.L_out;
        sm4[a.red_off4 + (wk.kpart * N + npos) * MTP4 + wk.ms * 4 + q] = acc;

        // prefetch the next conv's first weight fragments before the epilogue (their latency hides under it)
        const int MSn_cur = op.cout >> 4;
        const int NSn_cur = (op.mode == CONV_UPT) ? (op.L_in >> 4) * 2 : (op.L_out >> 4);
        const int ksplit_cur = 8 / (MSn_cur * NSn_cur);
        if (oi + 1 < a.nops && a.ops[oi + 1].kind != FOP_FINAL) {
            wk = fused_work(a, a.ops[oi + 1], wave, lane);
#pragma unroll
            for (int u = 0; u < kFusedPF; ++u) ring[u] = fused_load_a(a.ops[oi + 1], wk, wk.g_lo + u);
        }
        __syncthreads();

        // ------------------------------------------------------------------ epilogue
        const float* bias = a.packed + op.b_off;
        if (op.kind == FOP_CONV_GN) {
            // wave g normalises GroupNorm group g (8 groups per trajectory)
            const int gs = op.gs, re = gs * N;
            const float inv_re = 1.0f / (float)re;
            const FusedBuf db = a.bufs[op.dst];
            if (re == 256) {
                const int e0 = lane * 4;
                const int l = e0 / gs, c = wave * gs + (e0 - l * gs);
                const f32x4 bi = *(const f32x4*)(bias + c);
                const f32x4 ga = *(const f32x4*)(a.packed + op.ga_off + c), be = *(const f32x4*)(a.packed + op.be_off + c);
                f32x4 tb = {0.f, 0.f, 0.f, 0.f};
                if (op.tb_off >= 0) tb = *(const f32x4*)(a.tt_row + op.tb_off + c);
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
                const int l = e0 / gs, c = wave * gs + (e0 - l * gs);
                const f32x2 bi = *(const f32x2*)(bias + c);
                const f32x2 ga = *(const f32x2*)(a.packed + op.ga_off + c), be = *(const f32x2*)(a.packed + op.be_off + c);
                f32x2 tb = {0.f, 0.f};
                if (op.tb_off >= 0) tb = *(const f32x2*)(a.tt_row + op.tb_off + c);
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
                const int l = idx / M4, c4 = idx - l * M4;
                f32x4 v = sm4[a.red_off4 + l * MTP4 + c4];
                for (int k = 1; k < ksplit_cur; ++k) v += sm4[a.red_off4 + (k * N + l) * MTP4 + c4];
                v += *(const f32x4*)(bias + c4 * 4);
                if (op.dst >= 0) sm4[a.bufs[op.dst].off4 + (l + 2) * a.bufs[op.dst].rs4 + c4] = v;
                if (op.gdst >= 0) *(f32x4*)(a.gout[op.gdst] + ((size_t)b * N + l) * op.cout + c4 * 4) = v;
            }
        }
        __syncthreads();
    }
}

}  // namespace mpdx
