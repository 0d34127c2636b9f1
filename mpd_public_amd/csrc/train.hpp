// train.hpp - device code of the TRAINING step: the backward pass of the TemporalUnet, the loss
// gradient, Adam and the EMA of the reference's trainer.
//
//   reference:  GaussianDiffusionModel.p_losses   mpd/models/diffusion_models/diffusion_model_base.py:331-352
//               WeightedL1 / WeightedL2           mpd/models/diffusion_models/helpers.py:71-99
//               train() / EMA                     mpd/trainer/trainer.py:67-85, 174-300  (torch.optim.Adam, clip_grad_norm_, EMA)
//               the modules differentiated        mpd/models/layers/layers.py:229-355, temporal_unet.py:118-171
//
// The reference gets its gradients from torch autograd over ATen/MIOpen kernels.  Here the backward pass is written out:
//
//   * Conv1d input gradients ("dgrad") are convolutions again, so they run on the forward MFMA kernels (conv_block.hpp) with
//     the weights packed transposed and tap-flipped (pack_train_kernel):
//        Conv1d(k, stride 1, pad k/2)          dX = conv_S1(dU;  Wt[ci][co][k'] = W[co][ci][k-1-k'])
//        Conv1d(3, stride 2, pad 1)            dX = conv_S1_k3(zero-stuffed dU; same transposition)       (Downsample1d)
//        ConvTranspose1d(4, stride 2, pad 1)   dX[m] = f[2m],  f = conv_S1_k5(dY; Wt[ci][co][k'] = k' ? W[ci][co][k'-1] : 0)   (Upsample1d)
//   * weight gradients ("wgrad") are GEMMs with the reduction over batch x horizon: wgrad_kernel (fp32 MFMA 16x16x4, operands
//     staged per trajectory in LDS, all taps of a 16x16 weight tile accumulated from one staged window), split over the batch into
//     partial sums that a second kernel adds in a fixed order (deterministic, no float atomics);
//   * GroupNorm + Mish backward, with the per-channel sums (d gamma, d beta, d bias) and the time-embedding gradient of the block
//     taken from the same registers: gn_mish_bwd_kernel, one wave per GroupNorm region exactly like the forward epilogue;
//   * the time MLP (SinusoidalPosEmb -> Linear -> Mish -> Linear; per block Mish -> Linear) forward with saved activations and
//     backward: time_train_fwd_kernel / time_bwd_* (tiny GEMMs: plain FMA);
//   * Adam (torch.optim.Adam defaults, optional global-norm clipping as torch.nn.utils.clip_grad_norm_) and the EMA in one pass
//     over the flat parameter vector.
//
// Layouts: activations channel-last [B][L][C] (as everywhere); parameters and gradients in ONE flat fp32 vector in reference
// (state-dict) layout, so torch Parameters can alias it.
#pragma once
#include "conv_block.hpp"
#include "train_types.hpp"
#include "loss.hpp"

namespace mpdx {

__device__ __forceinline__ float mish_ref(float v) {   // same formula as the forward kernels' mish()
    return mish(v);
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm(8 groups) + Mish backward for one Conv1dBlock (layers.py:276-293):   y = mish(GN(u)),  u = conv(x) + bias
//   in : gy [B][L][C] (gradient wrt the block output; the +time-bias / +residual of the ResidualTemporalBlock pass it through
//        unchanged), pre = u [B][L][C]
//   out: du [B][L][C];  per-trajectory channel sums pg/pb/pbias [B][C] of (g * vhat), (g), (du)  -> summed over B by colsum_kernel
//        (deterministic);  dT[b][c] = sum_l gy[b][l][c]  when the block adds a time bias (the gradient wrt cond_mlp's output)
// One wave per GroupNorm region (trajectory x group): gs * L = 64 * EPL elements, EPL consecutive channels per lane.
struct GnBwdArgs {
    const float* gy;
    const float* pre;
    const float* gamma;
    const float* beta;
    float* du;
    float* pg;
    float* pb;
    float* pbias;
    float* dT;        // or null
    float* gres;      // or null: gradient buffer of the block's residual branch, += gy (the residual add passes it through)
    int gres_store;   // this launch is the first writer of gres in the pass: = gy
    int dT_stride;
    int B, L, C, gs, lg_gs, n_groups;
    int Lv;           // valid rows of a zero-padded container (horizons that are not powers of two: ConvArgs::Lv_out); 0 or L: all rows
};

// one GroupNorm region (trajectory b, group g) by one wave; du may alias gy (every element is read before it is written, by the same lane)
template <int EPL>
__device__ __forceinline__ void gn_mish_bwd_body(const GnBwdArgs& a, const int region, const int lane) {
    if (region >= a.B * a.n_groups) return;
    const int b = region / a.n_groups, g = region - b * a.n_groups;
    const int e0 = lane * EPL;
    const int l = e0 >> a.lg_gs, c = g * a.gs + (e0 & (a.gs - 1));
    const size_t o = ((size_t)b * a.L + l) * a.C + c;
    float u[EPL], gy[EPL], ga[EPL], be[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { u[e] = a.pre[o + e]; gy[e] = a.gy[o + e]; ga[e] = a.gamma[c + e]; be[e] = a.beta[c + e]; }
    if (a.gres) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) a.gres[o + e] = a.gres_store ? gy[e] : a.gres[o + e] + gy[e];
    }
    const float inv_n = 1.0f / (float)(64 * EPL);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += u[e];
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { u[e] -= mean; q += u[e] * u[e]; }
    const float var = wave_sum(q) * inv_n;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    float vh[EPL], gm[EPL], dvh[EPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        vh[e] = u[e] * rstd;
        gm[e] = gy[e] * mish_grad(vh[e] * ga[e] + be[e]);   // gradient wrt the GroupNorm output
        dvh[e] = gm[e] * ga[e];
        s1 += dvh[e];
        s2 += dvh[e] * vh[e];
    }
    s1 = wave_sum(s1) * inv_n;
    s2 = wave_sum(s2) * inv_n;
    float du[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        du[e] = rstd * (dvh[e] - s1 - vh[e] * s2);
        a.du[o + e] = du[e];
    }
    // per-channel sums over the horizon: lanes with the same channel offset differ in the bits >= log2(gs / EPL)
    float r[4 * EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { r[e] = gm[e] * vh[e]; r[EPL + e] = gm[e]; r[2 * EPL + e] = du[e]; r[3 * EPL + e] = gy[e]; }
    for (int off = a.gs / EPL; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 4 * EPL; ++k) r[k] += __shfl_xor(r[k], off, 64);
    }
    if (l == 0) {
        const size_t po = (size_t)b * a.C + c;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            a.pg[po + e] = r[e];
            a.pb[po + e] = r[EPL + e];
            a.pbias[po + e] = r[2 * EPL + e];
            if (a.dT) a.dT[(size_t)b * a.dT_stride + c + e] = r[3 * EPL + e];
        }
    }
}
template <int EPL>
__global__ __launch_bounds__(256) void gn_mish_bwd_kernel(const GnBwdArgs a) {
    gn_mish_bwd_body<EPL>(a, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

// The same backward for GroupNorm regions of 64, 512, 1024 or 2048 elements (training at horizons other than 64: the reference's trainer is
// horizon-agnostic, trainer.py:186-283).  One wave per region; lane -> NCH chunks of W consecutive channels of one position, chunk k =
// elements (k * 64 + lane) * W ..+W-1 in (position, channel) order - the placement of the forward's EPI_GN_MISH_GEN epilogue (W <= gs).
template <int W, int NCH>
__global__ __launch_bounds__(256) void gn_mish_bwd_gen_kernel(const GnBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int region = blockIdx.x * 4 + wave;
    if (region >= a.B * a.n_groups) return;
    const int b = region / a.n_groups, g = region - b * a.n_groups;
    // a padded container (a.Lv < a.L): the statistics run over the valid rows only, rows behind them carry no gradient (the forward's producers
    // keep them zero whatever the convolution put there: conv_block.hpp EPI_GN_MISH_GEN) - du = 0 there, and nothing of them enters a sum
    const int Lv = (a.Lv > 0 && a.Lv < a.L) ? a.Lv : a.L;
    float u[NCH][W], gy[NCH][W], ga[NCH][W], be[NCH][W];
    size_t o[NCH];
    int cc[NCH];
    bool ok[NCH];
    const float inv_n = 1.0f / (float)(a.gs * Lv);   // (= 1 / (64 W NCH), a power of two, when nothing is masked)
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e0 = (k * 64 + lane) * W;
        const int l = e0 >> a.lg_gs;
        ok[k] = l < Lv;
        cc[k] = g * a.gs + (e0 & (a.gs - 1));
        o[k] = ((size_t)b * a.L + l) * a.C + cc[k];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            u[k][e] = a.pre[o[k] + e]; gy[k][e] = ok[k] ? a.gy[o[k] + e] : 0.f; ga[k][e] = a.gamma[cc[k] + e]; be[k][e] = a.beta[cc[k] + e];
            s += ok[k] ? u[k][e] : 0.f;
        }
        if (a.gres) {
#pragma unroll
            for (int e = 0; e < W; ++e) a.gres[o[k] + e] = a.gres_store ? gy[k][e] : a.gres[o[k] + e] + gy[k][e];
        }
    }
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < W; ++e) { u[k][e] -= mean; q += ok[k] ? u[k][e] * u[k][e] : 0.f; }
    const float var = wave_sum(q) * inv_n;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    float gm[NCH][W], dvh[NCH][W];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < W; ++e) {
            u[k][e] *= rstd;   // vhat
            gm[k][e] = gy[k][e] * mish_grad(u[k][e] * ga[k][e] + be[k][e]);   // (0 on masked rows: gy is)
            dvh[k][e] = gm[k][e] * ga[k][e];
            s1 += dvh[k][e];
            s2 += dvh[k][e] * u[k][e];
        }
    s1 = wave_sum(s1) * inv_n;
    s2 = wave_sum(s2) * inv_n;
    // per-channel sums over the horizon: a lane's chunks hold the same W channels (gs divides 64 W), lanes with the same channel offset
    // differ in the bits >= log2(gs / W)
    float r[4 * W];
#pragma unroll
    for (int e = 0; e < 4 * W; ++e) r[e] = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const float du = ok[k] ? rstd * (dvh[k][e] - s1 - u[k][e] * s2) : 0.f;
            a.du[o[k] + e] = du;
            r[e] += gm[k][e] * u[k][e]; r[W + e] += gm[k][e]; r[2 * W + e] += du; r[3 * W + e] += gy[k][e];
        }
    for (int off = a.gs / W; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 4 * W; ++k) r[k] += __shfl_xor(r[k], off, 64);
    }
    if (lane * W < a.gs) {   // the lanes of position 0 of chunk 0: one per W channels of the group
        const size_t po = (size_t)b * a.C + cc[0];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            a.pg[po + e] = r[e];
            a.pb[po + e] = r[W + e];
            a.pbias[po + e] = r[2 * W + e];
            if (a.dT) a.dT[(size_t)b * a.dT_stride + cc[0] + e] = r[3 * W + e];
        }
    }
}

// out_k[c] = sum_b part_k[b][c]   for up to 4 arrays (k = blockIdx.y); fixed summation order (4 row phases, then ((0+1)+(2+3)))
struct ColsumArgs { const float* part[4]; float* out[4]; int B, C; };
__global__ __launch_bounds__(256) void colsum_kernel(const ColsumArgs a) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx, k = blockIdx.y;
    if (!a.part[k]) return;
    float s = 0.f;
    if (c < a.C) {
        const float* p = a.part[k] + c;
#pragma unroll 8
        for (int b = ry; b < a.B; b += 4) s += p[(size_t)b * a.C];
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < a.C) a.out[k][c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// every column sum of a backward pass in ONE launch (blockIdx.y = entry): out[c] = sum_{r < rows} part[r][c]
struct ColsumAllArgs {
    const float* ws;
    float* grad;
    int n;
    struct E { unsigned long long part, out; int rows, C; } e[120];
};
// (bx of nbx blocks work on entry `ent`: the kernel below, or side blocks of wgrad_reduce_colsum_kernel)
__device__ __forceinline__ void colsum_all_body(const ColsumAllArgs& a, const int ent, const int bx, const int nbx) {
    __shared__ float red[4][64];
    const ColsumAllArgs::E e = a.e[ent];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    for (int c0 = bx * 64; c0 < e.C; c0 += nbx * 64) {
        const int c = c0 + cx;
        float s = 0.f;
        if (c < e.C) {
            const float* p = a.ws + e.part + c;
#pragma unroll 8
            for (int b = ry; b < e.rows; b += 4) s += p[(size_t)b * e.C];
        }
        __syncthreads();
        red[ry][cx] = s;
        __syncthreads();
        if (ry == 0 && c < e.C) a.grad[e.out + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
    }
}
__global__ __launch_bounds__(256) void colsum_all_kernel(const ColsumAllArgs a) { colsum_all_body(a, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x); }

// channel sums of a dense [n_rows][C] tensor (bias gradient of the convolutions without GroupNorm): two passes through `part`
__global__ __launch_bounds__(256) void rowsum_part_kernel(const float* __restrict__ x, float* __restrict__ part, int n_rows, int C, int rows_per_block) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int r = r0; r < r1; ++r) s += x[(size_t)r * C + c];
        part[(size_t)blockIdx.x * C + c] = s;
    }
}

// dst[b][l][c] += src[b][l * step][c_off + c]      dst dense [B][L][Cd];  src [B][Ls][Cs]
__global__ __launch_bounds__(256) void acc_slice_kernel(float* __restrict__ dst, const float* __restrict__ src, int B, int L, int Cd, int Ls, int Cs,
                                                        int c_off, int step, int store) {
    const size_t total = (size_t)B * L * Cd;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cd);
        const size_t r = i / Cd;
        const int l = (int)(r % L), b = (int)(r / L);
        const float v = src[((size_t)b * Ls + (size_t)l * step) * Cs + c_off + c];
        dst[i] = store ? v : dst[i] + v;
    }
}

// z[b][2j][c] = x[b][j][c], z[b][2j+1][c] = 0   (input gradient of a stride-2 convolution = stride-1 convolution of this)
__global__ __launch_bounds__(256) void zero_stuff_kernel(const float* __restrict__ x, float* __restrict__ z, int B, int L, int C) {
    const size_t total = (size_t)B * 2 * L * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        const int l2 = (int)(r % (2 * L)), b = (int)(r / (2 * L));
        z[i] = (l2 & 1) ? 0.f : x[((size_t)b * L + (l2 >> 1)) * C + c];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 1-D convolution:   G[m][n][k] = sum_{b, t < LA}  A[b][t][m] * Bm[b][sb * t + k + ob][n]     (zero outside [0, LB))
//   Conv1d(k5 / k1, stride 1):   A = dU (m = c_out), Bm = x (n = c_in), sb = 1, ob = -pad           -> G = dW[c_out][c_in][k]
//   Conv1d(3, stride 2, pad 1):  A = dU,             Bm = x,            sb = 2, ob = -1
//   ConvTranspose1d(4, 2, 1):    A = x (m = c_in),   Bm = dY (n = c_out), sb = 2, ob = -1             -> G = dW[c_in][c_out][k]
// Workgroup = 4 waves = a 32 x 32 tile of (m, n), every tap; per trajectory the two operand windows are staged in LDS and each wave
// runs  LA / 4 * KS  v_mfma_f32_16x16x4_f32 on its 16 x 16 sub-tile.  gridDim.z splits the batch; partial sums go to
// part[z][m][n][k] and are added in order by wgrad_reduce_kernel.
struct WgradArgs {
    const float* A;
    const float* Bm;
    float* part;
    int LA, lda, a_off, M;   // A row stride (floats), first channel, channels
    int LB, ldb, b_off, N;
    int sb, ob;
    int B, b_per_split;
    // bias gradient of a convolution without GroupNorm = column sums of its dY, which is this GEMM's A operand (Conv1d) or B operand
    // (ConvTranspose1d): the first tile column / row of blocks adds them up on the side -> bias_part[split][channel]
    float* bias_part;    // or null
    int bias_from_b;
};

constexpr int kWgRS = 48;   // LDS row stride: 4 consecutive rows start 16 banks apart

// The trajectory loop of wgrad_body for LA = 8 NR rows of A and LB = LA (Conv1d k5 / k1) or 2 LA (stride-2 layers: KS = 3 / 4) rows of B, round 4:
// the NEXT trajectory's operand rows are fetched into registers (unconditional loads, clamped columns, zeros selected at the LDS store) while the
// current one's MFMAs run; the barriers wait for LDS traffic only (lds_barrier; hipcc's __syncthreads() does the same here: checked in the ISA).  The
// generic loop below issues its loads inside column-guard branches - one dependent round trip per row - and a block walks 8-32 trajectories: that
// chain, not the MFMAs, was the length of every backward launch.  Same MFMA order per trajectory, trajectories ascending: same bits.
template <int KS, int NR>
// (bs, be): this wave group's trajectories; `trips` >= be - bs: the trip count every wave group of the workgroup runs (the barriers are workgroup-wide);
// (lo, hi): the batch split's whole range, for the clamped prefetch of a trip without a trajectory
__device__ __forceinline__ void wgrad_loop_prefetch(const WgradArgs& a, float* As, float* Bs, const int bs, const int be, const int trips, const int lo, const int hi,
                                                    const int m0, const int n0, const int mi,
                                                    const int ni, const int i16, const int kq, const int col, const int row0, const bool do_bias, float& bsum,
                                                    f32x4 (&acc)[KS]) {
    constexpr int NRB = (KS == 3 || KS == 4) ? 2 * NR : NR, LA = 8 * NR, LB = 8 * NRB;
    const bool okA = m0 + col < a.M, okB = n0 + col < a.N;
    const float* pA = a.A + a.a_off + (okA ? m0 + col : a.M - 1) + (size_t)row0 * a.lda;
    const float* pB = a.Bm + a.b_off + (okB ? n0 + col : a.N - 1) + (size_t)row0 * a.ldb;
    float ra[NR], rb[NRB];
    auto fetch = [&](int b) {
        const float* qa = pA + (size_t)b * LA * a.lda;
        const float* qb = pB + (size_t)b * LB * a.ldb;
#pragma unroll
        for (int u = 0; u < NR; ++u) ra[u] = qa[(size_t)(8 * u) * a.lda];
#pragma unroll
        for (int u = 0; u < NRB; ++u) rb[u] = qb[(size_t)(8 * u) * a.ldb];
    };
    auto clampb = [&](int b) { return b < lo ? lo : (b > hi - 1 ? hi - 1 : b); };
    fetch(clampb(bs));
    for (int it = 0; it < trips; ++it) {
        const int b = bs + it;
        const bool live = b < be;
        lds_barrier();   // the previous trajectory's fragments are read
#pragma unroll
        for (int u = 0; u < NR; ++u) As[(size_t)(row0 + 8 * u) * kWgRS + col] = (okA && live) ? ra[u] : 0.f;
#pragma unroll
        for (int u = 0; u < NRB; ++u) Bs[(size_t)(row0 + 8 * u + 2) * kWgRS + col] = (okB && live) ? rb[u] : 0.f;
        fetch(clampb(b + 1));   // (a trip without a successor re-reads a valid trajectory: no branch around the loads)
        lds_barrier();
        if (!live) continue;
        if (do_bias) {   // fixed order: this thread's rows ascending, trajectories ascending; the 8 row phases are combined at the end
            if (a.bias_from_b) {
#pragma unroll
                for (int u = 0; u < NRB; ++u) bsum += Bs[(size_t)(row0 + 8 * u + 2) * kWgRS + col];
            } else {
#pragma unroll
                for (int u = 0; u < NR; ++u) bsum += As[(size_t)(row0 + 8 * u) * kWgRS + col];
            }
        }
#pragma unroll
        for (int t0 = 0; t0 < LA; t0 += 4) {
            const float av = As[(size_t)(t0 + kq) * kWgRS + mi + i16];
            const int pr = a.sb * (t0 + kq) + a.ob + 2;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const float bv = Bs[(size_t)(pr + k) * kWgRS + ni + i16];
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
            }
        }
    }
}

// ngrp = 2 (inside bwd_pair_kernel's 512-thread workgroups, prefetching loop only): the block's trajectories are walked by TWO groups of four waves,
// each on its own LDS slot and half of the range (a block's launch-long chain of trajectories halves); group 1's accumulators are added to group 0's
// through LDS at the end (sum = first half + second half: fixed order, deterministic).
template <int KS>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int bx, const int by, const int bz, const int ngrp = 1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = ngrp == 2 ? (int)(threadIdx.x >> 8) : 0;
    float* As = smem + (size_t)grp * (a.LA + a.LB + 4) * kWgRS;   // [LA][48]
    float* Bs = As + (size_t)a.LA * kWgRS;                        // [LB + 4][48], row r holds position r - 2
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int n0 = bx * 32, m0 = by * 32;
    const int b0 = bz * a.b_per_split, b1 = min(a.B, b0 + a.b_per_split);
    const int mi = (wave & 1) * 16, ni = (wave >> 1) * 16;
    const int i16 = lane & 15, kq = lane >> 4;
    f32x4 acc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // halo rows of Bs stay zero
    for (int i = tid; i < 4 * kWgRS; i += 256) {
        const int r = i / kWgRS, c = i - r * kWgRS;
        Bs[(size_t)(r < 2 ? r : a.LB + r) * kWgRS + c] = 0.f;
    }
    const int col = tid & 31, row0 = tid >> 5;   // 8 rows per pass
    const bool do_bias = a.bias_part && (a.bias_from_b ? by == 0 : bx == 0);   // all 256 threads: (column, row phase of 8)
    float bsum = 0.f;
    const int nr = a.LA >> 3;
    const bool fast = (a.LA & 7) == 0 && a.LB == ((KS == 3 || KS == 4) ? 2 * a.LA : a.LA) && (nr == 1 || nr == 2 || nr == 4 || nr == 8) && b1 > b0;
    if (ngrp == 2 && !fast) {   // (the host pairs wave groups only with shapes the prefetching loop takes; an empty split: nothing to add)
        if (grp == 1) return;
    }
    const int nb_all = b1 - b0, trips = (ngrp == 2 && fast) ? (nb_all + 1) / 2 : nb_all;
    const int bs = b0 + grp * trips, be = min(b1, bs + trips);
    if (fast) {
        __syncthreads();   // the halo rows
        if (nr == 1) wgrad_loop_prefetch<KS, 1>(a, As, Bs, bs, be, trips, b0, b1, m0, n0, mi, ni, i16, kq, col, row0, do_bias, bsum, acc);
        else if (nr == 2) wgrad_loop_prefetch<KS, 2>(a, As, Bs, bs, be, trips, b0, b1, m0, n0, mi, ni, i16, kq, col, row0, do_bias, bsum, acc);
        else if (nr == 4) wgrad_loop_prefetch<KS, 4>(a, As, Bs, bs, be, trips, b0, b1, m0, n0, mi, ni, i16, kq, col, row0, do_bias, bsum, acc);
        else wgrad_loop_prefetch<KS, 8>(a, As, Bs, bs, be, trips, b0, b1, m0, n0, mi, ni, i16, kq, col, row0, do_bias, bsum, acc);
    } else
    for (int b = b0; b < b1; ++b) {
        __syncthreads();   // the previous trajectory's fragments are read
        for (int r = row0; r < a.LA; r += 8)
            As[(size_t)r * kWgRS + col] = (m0 + col < a.M) ? a.A[((size_t)b * a.LA + r) * a.lda + a.a_off + m0 + col] : 0.f;
        for (int r = row0; r < a.LB; r += 8)
            Bs[(size_t)(r + 2) * kWgRS + col] = (n0 + col < a.N) ? a.Bm[((size_t)b * a.LB + r) * a.ldb + a.b_off + n0 + col] : 0.f;
        __syncthreads();
        if (do_bias) {   // fixed order: this thread's rows ascending, trajectories ascending; the 8 row phases are combined at the end
            if (a.bias_from_b) for (int r = row0; r < a.LB; r += 8) bsum += Bs[(size_t)(r + 2) * kWgRS + col];
            else for (int r = row0; r < a.LA; r += 8) bsum += As[(size_t)r * kWgRS + col];
        }
        for (int t0 = 0; t0 < a.LA; t0 += 4) {
            const float av = As[(size_t)(t0 + kq) * kWgRS + mi + i16];
            const int pr = a.sb * (t0 + kq) + a.ob + 2;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const float bv = Bs[(size_t)(pr + k) * kWgRS + ni + i16];
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
            }
        }
    }
    if (ngrp == 2 && fast) {   // group 1 -> LDS -> group 0
        __syncthreads();   // every fragment is read: the slots are free
        float* X = smem;   // [256 lanes][KS][4] accumulators | [256] bias sums
        if (grp == 1) {
#pragma unroll
            for (int k = 0; k < KS; ++k) *(f32x4*)(X + ((size_t)tid * KS + k) * 4) = acc[k];
            X[256 * KS * 4 + tid] = bsum;
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int k = 0; k < KS; ++k) acc[k] += *(const f32x4*)(X + ((size_t)tid * KS + k) * 4);
        bsum += X[256 * KS * 4 + tid];
        As = smem;   // (group 0's slot: the bias reduction below)
    }
    if (do_bias) {
        __syncthreads();   // the last trajectory's fragments are read: As is free
        As[row0 * 32 + col] = bsum;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) t += As[p * 32 + tid];
            const int c = (a.bias_from_b ? n0 : m0) + tid, C = a.bias_from_b ? a.N : a.M;
            if (c < C) a.bias_part[(size_t)bz * C + c] = t;
        }
    }
    // D fragment: lane holds rows 4 * kq + r (r = 0..3), column i16
    const int n = n0 + ni + i16;
    if (n < a.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + mi + 4 * kq + r;
            if (m >= a.M) continue;
            float* dst = a.part + (((size_t)bz * a.M + m) * a.N + n) * KS;
#pragma unroll
            for (int k = 0; k < KS; ++k) dst[k] = acc[k][r];
        }
    }
}

template <int KS>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    wgrad_body<KS>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// One launch for everything that depends only on a layer's dU: the input-gradient convolution (blocks [0, n_dgrad): the forward
// kernel's body on the dgrad pack, 512 threads) and the weight-gradient GEMM(s) against the layer's one or two sources (the blocks
// behind them: 4 of the 8 waves work, the others leave at once - a finished wave does not take part in the workgroup's barriers).
struct BwdPairArgs {
    ConvArgs cd;
    WgradArgs w[3];
    int n_dgrad;         // 0: weight-gradient GEMMs only (the layers whose input needs no gradient + final_conv[1], collected into ONE launch)
    int nw[3];           // blocks of each weight-gradient GEMM (0: unused slot)
    int gx[3], gy[3];    // their grids: x = N tiles, y = M tiles, z = batch splits
    int ks_w[3];         // taps of each weight gradient (1, 3, 4 or 5)
    int two[3];          // the job's blocks run two wave groups (wgrad_body ngrp = 2: all 512 threads work)
    // a SECOND input-gradient convolution behind the first one's blocks (round 6): the 1x1 dgrad of a ResidualTemporalBlock's residual_conv riding on the
    // launch of the block's blocks[1] dgrad - they read different gradients (G of the block output / dU of blocks[1]) and write different tensors
    ConvArgs cd2;
    int n_dgrad2;        // 0: none
};
// EPI_D = EPI_GN_BWD: the dgrad blocks also take their result through the Mish + GroupNorm backward of the Conv1dBlock below (conv_block.hpp)
template <int KS_D, int MT, int NT, int EPI_D = EPI_BIAS>
__global__ __launch_bounds__(512) void bwd_pair_kernel(const BwdPairArgs a) {
    if ((int)blockIdx.x < a.n_dgrad) {
        conv_block_body<CONV_S1, KS_D, EPI_D, MT, NT, 1, 8>(a.cd, blockIdx.x);
        return;
    }
    if ((int)blockIdx.x < a.n_dgrad + a.n_dgrad2) {
        conv_block_body<CONV_S1, 1, EPI_BIAS, MT, NT, 1, 8>(a.cd2, (int)blockIdx.x - a.n_dgrad);
        return;
    }
    int idx = (int)blockIdx.x - a.n_dgrad - a.n_dgrad2;
    int which = 0;
    while (which < 2 && idx >= a.nw[which]) { idx -= a.nw[which]; ++which; }
    const int ngrp = a.two[which] ? 2 : 1;
    if (threadIdx.x >= 256 && ngrp == 1) return;
    const WgradArgs& w = a.w[which];
    const int bx = idx % a.gx[which], r = idx / a.gx[which];
    const int by = r % a.gy[which], bz = r / a.gy[which];
    switch (a.ks_w[which]) {
        case 1: wgrad_body<1>(w, bx, by, bz, ngrp); break;
        case 3: wgrad_body<3>(w, bx, by, bz, ngrp); break;
        case 4: wgrad_body<4>(w, bx, by, bz, ngrp); break;
        default: wgrad_body<5>(w, bx, by, bz, ngrp); break;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The backward CHAIN of the outer levels as ONE launch (round 6).  At the levels whose trajectories own whole tiles (L = 64 / 32 / 16) the
// input-gradient convolutions - with the Mish + GroupNorm backward of the Conv1dBlock below in their epilogue - and the stand-alone GroupNorm
// backwards form a chain in which every step needs only what the SAME trajectory's earlier steps produced.  One 512-thread workgroup per
// trajectory walks the steps the per-layer path would have launched one by one - same bodies (conv_block_body on the dgrad pack with
// NT = L positions: one trajectory per tile, the channel tiles one after the other; gn_mish_bwd_body), same operands through the same gradient
// buffers in global memory (L2), same order - separated by workgroup barriers instead of launch boundaries (a barrier orders the workgroup's
// global stores before its later loads).  The layers' weight gradients do not ride: they run in wgrad_multi_kernel behind the chains.
constexpr int kChainMaxConv = 22, kChainMaxGn = 6, kChainMaxSteps = kChainMaxConv + kChainMaxGn;
struct ChainStep { short kind, sel, n_mt, idx; };   // kind 0: dgrad conv (sel = selector below, idx into cd), 1: GroupNorm backward (sel = EPL, idx into gn)
struct ChainArgs {
    int n;
    ChainStep st[kChainMaxSteps];
    GnBwdArgs gn[kChainMaxGn];
    ConvArgs cd[kChainMaxConv];
};
// selector of a dgrad body: taps (5 / 3 / 1), positions per trajectory (64 / 32 / 16), epilogue (plain / GroupNorm backward); MT = 32
__host__ __device__ constexpr int chain_sel(int ks, int nt, int gnbwd) { return ((ks == 5 ? 0 : ks == 3 ? 1 : 2) * 3 + (nt == 64 ? 0 : nt == 32 ? 1 : 2)) * 2 + (gnbwd ? 1 : 0); }
__global__ __launch_bounds__(512) void bwd_chain_kernel(const ChainArgs a) {
    warm_kernarg<(int)sizeof(ChainArgs)>();
    const int b = (int)blockIdx.x;
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    for (int k = 0; k < a.n; ++k) {
        const ChainStep s = a.st[k];
        if (s.kind == 1) {
            const GnBwdArgs& g = a.gn[s.idx];
            if (wave < g.n_groups) {
                if (s.sel == 4) gn_mish_bwd_body<4>(g, b * g.n_groups + wave, lane);
                else gn_mish_bwd_body<2>(g, b * g.n_groups + wave, lane);
            }
            __syncthreads();
            continue;
        }
        const ConvArgs& c = a.cd[s.idx];
        for (int mt = 0; mt < s.n_mt; ++mt) {
            const int tile = mt + s.n_mt * b;   // conv_block_body: mt = tile % n_mt, position tile = tile / n_mt = this trajectory
            switch (s.sel) {
#define MPDX_CHAIN_CASE(KS, NT, GN) case chain_sel(KS, NT, GN): conv_block_body<CONV_S1, KS, GN ? EPI_GN_BWD : EPI_BIAS, 32, NT, 1, 8>(c, tile); break;
                MPDX_CHAIN_CASE(5, 64, 0) MPDX_CHAIN_CASE(5, 64, 1) MPDX_CHAIN_CASE(5, 32, 0) MPDX_CHAIN_CASE(5, 32, 1) MPDX_CHAIN_CASE(5, 16, 0) MPDX_CHAIN_CASE(5, 16, 1)
                MPDX_CHAIN_CASE(3, 64, 0) MPDX_CHAIN_CASE(3, 64, 1) MPDX_CHAIN_CASE(3, 32, 0) MPDX_CHAIN_CASE(3, 32, 1) MPDX_CHAIN_CASE(3, 16, 0) MPDX_CHAIN_CASE(3, 16, 1)
                MPDX_CHAIN_CASE(1, 64, 0) MPDX_CHAIN_CASE(1, 32, 0) MPDX_CHAIN_CASE(1, 16, 0)
#undef MPDX_CHAIN_CASE
                default: break;
            }
            __syncthreads();   // the tile's epilogue has stored (workgroup-visible) before anything of this workgroup reads it / restages LDS
        }
    }
}

// MANY weight-gradient GEMMs in ONE launch (round 6): every job keeps its own grid (x = N tiles, y = M tiles, z = batch splits); the blocks of job k
// are [start[k], start[k + 1]) - found by bisection, as wgrad_reduce_all_body does.  The weight gradients of a layer depend only on the layer's dU and on
// its (kept) input: nothing in the backward chain waits for them, so they need not ride on the chain's launches (where a dgrad launch lasts as long as
// its longest weight-gradient block) - collected here they run over all 256 CUs at once, behind the chain.
constexpr int kWgradMultiMax = 64;
struct WgradMultiArgs {
    int n;
    int start[kWgradMultiMax + 1];
    short gx[kWgradMultiMax], gy[kWgradMultiMax];
    signed char ks[kWgradMultiMax], two[kWgradMultiMax];
    WgradArgs w[kWgradMultiMax];
};
__global__ __launch_bounds__(512) void wgrad_multi_kernel(const WgradMultiArgs a) {
    int lo = 0, hi = a.n;
    const int bid = (int)blockIdx.x;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bid >= a.start[mid]) lo = mid; else hi = mid; }
    const int which = lo, idx = bid - a.start[which];
    const int ngrp = a.two[which] ? 2 : 1;
    if (threadIdx.x >= 256 && ngrp == 1) return;
    const WgradArgs& w = a.w[which];
    const int gx = a.gx[which], gy = a.gy[which];
    const int bx = idx % gx, r = idx / gx;
    const int by = r % gy, bz = r / gy;
    switch (a.ks[which]) {
        case 1: wgrad_body<1>(w, bx, by, bz, ngrp); break;
        case 3: wgrad_body<3>(w, bx, by, bz, ngrp); break;
        case 4: wgrad_body<4>(w, bx, by, bz, ngrp); break;
        default: wgrad_body<5>(w, bx, by, bz, ngrp); break;
    }
}

// g[(m * n_tot + n_off + n) * KS + k] = sum_s part[s][m][n][k]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ g, int S, int M, int N, int KS, int n_tot,
                                                           int n_off) {
    const size_t per = (size_t)M * N * KS;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
#pragma unroll 8
        for (int z = 0; z < S; ++z) s += part[(size_t)z * per + i];
        const int k = (int)(i % KS);
        const size_t mn = i / KS;
        const int n = (int)(mn % N), m = (int)(mn / N);
        g[((size_t)m * n_tot + n_off + n) * KS + k] = s;
    }
}

// every split-batch weight-gradient reduction of a backward pass in ONE launch (blockIdx.y = entry)
struct ReduceAllArgs {
    const float* ws;
    float* grad;
    int n;
    struct E { unsigned long long part, g; int S, M, N, KS, n_tot, n_off, zsl, pad_; } e[96];   // zsl: reduce_zsl() of the entry (0: scalar form)
    int cstart[97];   // first block of entry j (1024 outputs per block, 1024 / zsl when zsl > 1); cstart[n] = blocks of the launch
};
// The 16-byte form of wgrad_reduce_all_body needs every index involved to be a multiple of four floats (always, for the networks the library
// builds: a factor 32 in every M N KS); then 1 / 4 / 16 z-slices per block for S <= 8 / <= 32 / more.  0: the scalar form.
static inline int reduce_zsl(const ReduceAllArgs::E& e) {
    const unsigned per = (unsigned)e.M * e.N * e.KS, NK = (unsigned)e.N * e.KS;
    const bool vec = (per & 3u) == 0u && (e.part & 3ull) == 0ull && (e.g & 3ull) == 0ull &&
                     (e.n_tot == e.N || ((NK & 3u) == 0u && (((unsigned)e.n_tot * e.KS) & 3u) == 0u && (((unsigned)e.n_off * e.KS) & 3u) == 0u));
    if (!vec) return 0;
    return e.S <= 8 ? 1 : (e.S <= 32 ? 4 : 16);
}
// one block = 1024 outputs of one entry (found by bisection of cstart): every block has work, a thread's loads fly together
__device__ __forceinline__ void wgrad_reduce_all_body(const ReduceAllArgs& a, const int block_id, const int tid) {
    int lo = 0, hi = a.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (block_id >= a.cstart[mid]) lo = mid; else hi = mid; }
    const ReduceAllArgs::E e = a.e[lo];
    const float* part = a.ws + e.part;
    float* g = a.grad + e.g;
    const unsigned per = (unsigned)e.M * e.N * e.KS, KS = (unsigned)e.KS, N = (unsigned)e.N;   // (32-bit index arithmetic, as pack_train_kernel)
    const unsigned NK0 = N * KS;
    // Round 5: the 16-byte form - a thread sums the four CONSECUTIVE outputs i .. i + 3 with one dwordx4 load per split, exactly its splits' loads
    // (the scalar form below issues eight clamped loads per output whatever S is).  What the launch actually waited for were the SMALL layers: a
    // 32 x 32 layer has one tile, so the rule "about one block per CU" splits its batch into min(256, B) partial sums, and a thread walked them
    // eight at a time - 16 dependent round trips at batch 128 (47.8 us for 79 MB).  With zsl > 1 (host: reduce_zsl) a block takes 1024 / zsl outputs
    // and its threads split the z range into zsl contiguous slices (each ascending), combined through LDS in slice order: fixed order, deterministic.
    if (e.zsl > 0) {
        __shared__ f32x4 comb[256];
        const unsigned zsl = (unsigned)e.zsl, qpb = 256u / zsl;
        const unsigned quad = (unsigned)tid % qpb, slice = (unsigned)tid / qpb;
        const unsigned i = (unsigned)(block_id - a.cstart[lo]) * (1024u / zsl) + 4u * quad;
        const bool live = i < per;
        const f32x4* p4 = (const f32x4*)(part + (live ? i : 0u));
        const size_t zs = (size_t)(per >> 2);   // split stride in f32x4 units
        const int zper = (e.S + (int)zsl - 1) / (int)zsl;
        int z = (int)slice * zper;
        const int ze = min(e.S, z + zper);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (; z + 8 <= ze; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int zz = 0; zz < 8; ++zz) v[zz] = p4[(size_t)(z + zz) * zs];
#pragma unroll
            for (int zz = 0; zz < 8; ++zz) acc += v[zz];
        }
        if (ze - z >= 4) {
            f32x4 v[4];
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) v[zz] = p4[(size_t)(z + zz) * zs];
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) acc += v[zz];
            z += 4;
        }
        if (ze - z >= 2) {
            const f32x4 v0 = p4[(size_t)z * zs], v1 = p4[(size_t)(z + 1) * zs];
            acc += v0; acc += v1;
            z += 2;
        }
        if (ze - z >= 1) acc += p4[(size_t)z * zs];
        if (zsl > 1) {   // (block-uniform)
            comb[tid] = acc;
            __syncthreads();
            if (slice != 0) return;
            for (unsigned sl = 1; sl < zsl; ++sl) acc += comb[sl * qpb + quad];
        }
        if (!live) return;
        size_t o = i;
        if ((unsigned)e.n_tot != N) { const unsigned m = i / NK0, r = i - m * NK0; o = ((size_t)m * e.n_tot + e.n_off) * KS + r; }
        *(f32x4*)(g + o) = acc;
        return;
    }
    const unsigned base = (unsigned)(block_id - a.cstart[lo]) * 1024u + (unsigned)tid;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned ic[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned i = base + k * 256u; ic[k] = i < per ? i : 0u; }
    // splits z ascending (the sum's order); eight splits' loads (32 per thread) are issued together
    for (int z0 = 0; z0 < e.S; z0 += 8) {
        float v[8][4];
#pragma unroll
        for (int zz = 0; zz < 8; ++zz) {
            const int z = z0 + zz < e.S ? z0 + zz : e.S - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[zz][k] = part[(size_t)z * per + ic[k]];
        }
#pragma unroll
        for (int zz = 0; zz < 8; ++zz)
            if (z0 + zz < e.S) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += v[zz][k];
            }
    }
    // g[(m * n_tot + n_off + n) * KS + kk] for i = (m * N + n) * KS + kk: the same index when the layer has one source (n_tot == N), else one division
    const unsigned NK = N * KS;
    const bool dense = (unsigned)e.n_tot == N;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = base + k * 256u;
        if (i >= per) continue;
        size_t o = i;
        if (!dense) { const unsigned m = i / NK, r = i - m * NK; o = ((size_t)m * e.n_tot + e.n_off) * KS + r; }
        g[o] = s[k];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(const ReduceAllArgs a) { wgrad_reduce_all_body(a, (int)blockIdx.x, (int)threadIdx.x); }
// ... with every column sum of the pass (colsum_all_body) as side blocks: the two reductions read different partial sums and write different
// gradients - one launch (round 5; colsum_all_kernel was 4.9 / 7.7 us at batch 32 / 128 behind this one)
struct ReduceColsumArgs { ReduceAllArgs red; ColsumAllArgs col; int n_red_blocks; };
__global__ __launch_bounds__(256) void wgrad_reduce_colsum_kernel(const ReduceColsumArgs a) {
    const int blk = (int)blockIdx.x;
    if (blk < a.n_red_blocks) { wgrad_reduce_all_body(a.red, blk, (int)threadIdx.x); return; }
    const int id = blk - a.n_red_blocks;
    colsum_all_body(a.col, id >> 1, id & 1, 2);
}

// ------------------------------------------------------------------------------------------------------------------
// Packing for a training step, ONE launch for every parameter (blockIdx.y = parameter): flat reference layout ->
//   packed  : the forward layout of conv_block.hpp (pack_conv_weights_kernel) / plain copies for vectors
//   packedT : the dgrad layout (see the file header); only for convolutions whose input gradient is needed
// struct PackDesc: train_types.hpp (the host model keeps a table of them)

// One block = one CHUNK of 1024 outputs of one pack of one parameter (table built once on the host): every block of the launch has work,
// a thread's four loads fly together (unconditional, from clamped addresses; zeros selected afterwards), index arithmetic in 32 bits.
// (First version: grid (64, parameters), each thread looping over its parameter with one dependent load per trip and four 64-bit
//  divisions per element - 42 us per call for 8 M outputs.)
// struct PackChunk: train_types.hpp

// Round 6: a thread owns one lane of one GROUP = the nslot consecutive 256-float fragment blocks that share (m16, c16); a chunk (block) = kPackGroups groups,
// one per wave.  In the forward layout of a Conv1d weight the 4 x nslot floats a lane needs are CONTIGUOUS in the reference tensor
// (w[co][ci0 .. ci0 + 3][0 .. ks - 1]): nslot 16-byte loads instead of 4 nslot dword loads.  Measured (tools/pack_probe.py): the forward pack of the
// four-level network alone took 18.3 us of the launch's 22.0 - 4.65 M outputs whose dword loads each touch ~40 different cache lines per wave
// instruction (16 rows x 4 segments 80 B apart) at ~4 clocks a line in the texture cache; the dgrad pack (12 lines per instruction) 3.6 us.
constexpr unsigned kPackGroups = 4;
template <int KS>
__device__ __forceinline__ void pack_group_wide(const float* __restrict__ src16, float* __restrict__ dst16) {
    // src16: the lane's 4 KS contiguous floats w[e][k] = src16[e * KS + k] (16-byte aligned); dst16: its float4 in the group's first block, blocks 256 floats apart
    f32x4 v[KS];
#pragma unroll
    for (int x = 0; x < KS; ++x) v[x] = *(const f32x4*)(src16 + 4 * x);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int x = e * KS + k; o[e] = v[x >> 2][x & 3]; }
        *(f32x4*)(dst16 + k * 256) = o;
    }
}
template <int KS>
__device__ __forceinline__ void pack_group_flip(const float* __restrict__ src, unsigned sa, unsigned estride, bool oko, int nvalid, float* __restrict__ dst16) {
    // rows e < nvalid (of 4) exist; out[slot][e] = row e's tap KS - 1 - slot
    float w[4][KS];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < KS; ++k) w[e][k] = src[(oko && e < nvalid) ? sa + (unsigned)e * estride + (unsigned)k : 0u];
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (oko && e < nvalid) ? w[e][KS - 1 - sl] : 0.f;
        *(f32x4*)(dst16 + sl * 256) = o;
    }
}
__global__ __launch_bounds__(256) void pack_train_kernel(const PackDesc* __restrict__ descs, const PackChunk* __restrict__ chunks, const float* __restrict__ flat,
                                                         float* __restrict__ packed, float* __restrict__ packedT) {
    const PackChunk c = chunks[blockIdx.x];
    const PackDesc d = descs[c.desc];
    const float* src = flat + d.src;
    const unsigned n = (unsigned)d.n;
    const unsigned cin = (unsigned)d.cin, cout = (unsigned)d.cout, ks = (unsigned)d.ks;
    if (c.which != 0 && !packedT) return;
    float* const dst = c.which == 0 ? packed + d.dst : packedT + d.dstT;
    const unsigned total = c.which == 0 ? (unsigned)d.pn : (unsigned)d.pnT;
    const unsigned ns = c.which == 0 ? (d.kind == 0 ? 1u : (unsigned)d.nslot) : (unsigned)d.t_ks;   // blocks per group
    const unsigned lane = threadIdx.x & 63, g = c.first + (threadIdx.x >> 6);
    const unsigned gfirst = g * ns * 256u + 4u * lane;   // the lane's first output (slot 0)
    if (g * ns * 256u >= total) return;
    if (c.which == 0 && d.kind == 1) {   // Conv1d weight, forward layout: the wide path when the lane's 4 ks floats are all there and 16-byte aligned
        const unsigned nc16 = (unsigned)d.cin_pad >> 4;
        const unsigned c16 = g % nc16, m16 = g / nc16;
        const unsigned co = m16 * 16 + (lane & 15), ci0 = c16 * 16 + (lane >> 4) * 4;
        const unsigned long long sa = ((unsigned long long)co * cin + ci0) * ks;
        if (ci0 + 3 < cin && (g + 1) * ns * 256u <= total && (((size_t)(src + sa)) & 15) == 0 && (((size_t)dst) & 15) == 0 && ns == ks) {
            if (ks == 5) { pack_group_wide<5>(src + sa, dst + gfirst); return; }
            if (ks == 3) { pack_group_wide<3>(src + sa, dst + gfirst); return; }
            if (ks == 1) { pack_group_wide<1>(src + sa, dst + gfirst); return; }
        }
    }
    if (c.which != 0 && d.t_mode == 0 && ns == ks && (g + 1) * ns * 256u <= total && (((size_t)dst) & 15) == 0) {
        // dgrad layout of a Conv1d weight: the lane's rows W[ci0 + e][o][0 .. ks - 1] (ks contiguous floats each, taps flipped on the way out), float4 stores
        const unsigned tnc16 = (unsigned)d.t_cin_pad >> 4;
        const unsigned c16 = g % tnc16, m16 = g / tnc16;
        const unsigned o = m16 * 16 + (lane & 15), ci0 = c16 * 16 + (lane >> 4) * 4;
        const unsigned sa = (ci0 * cin + o) * ks, estride = cin * ks;
        const bool oko = o < (unsigned)d.t_cout;
        if (ks == 5) { pack_group_flip<5>(src, sa, estride, oko, (unsigned)d.t_cin - ci0, dst + gfirst); return; }
        if (ks == 3) { pack_group_flip<3>(src, sa, estride, oko, (unsigned)d.t_cin - ci0, dst + gfirst); return; }
        if (ks == 1) { pack_group_flip<1>(src, sa, estride, oko, (unsigned)d.t_cin - ci0, dst + gfirst); return; }
    }
    // the general path: per slot the lane's four outputs i0 .. i0 + 3, sources a constant stride apart (the loads of every slot issued before the first store)
    constexpr int NSMAX = 5;
    bool ok[NSMAX][4];
    float v[NSMAX][4];
#pragma unroll
    for (int sl = 0; sl < NSMAX; ++sl) {
        const unsigned slot = (unsigned)sl < ns ? (unsigned)sl : 0u;
        const unsigned i0 = gfirst + slot * 256u;
        unsigned sa0 = i0, sstep = 1, lim_ci = 0, ci0 = 0;   // source of element e: sa0 + e * sstep, valid while ci0 + e < lim_ci (vectors: i0 + e < n)
        bool okq = (unsigned)sl < ns;
        if (c.which == 0) {
            if (d.kind == 0) { ci0 = i0; lim_ci = n; }   // PK_VEC: a copy
            else {   // forward layout [m16][c16][slot][lane][4]
                const unsigned nc16 = (unsigned)d.cin_pad >> 4;
                const unsigned c16 = g % nc16, m16 = g / nc16;
                const unsigned co = m16 * 16 + (lane & 15);
                ci0 = c16 * 16 + (lane >> 4) * 4; lim_ci = cin;
                if (d.kind == 2) { sa0 = (ci0 * cout + co) * ks + (unsigned)upt_slot_to_k((int)slot); sstep = cout * ks; }
                else { sa0 = (co * cin + ci0) * ks + slot; sstep = ks; }
            }
        } else {   // dgrad layout: a CONV_S1 weight [t_cout][t_cin][t_ks]
            const unsigned tnc16 = (unsigned)d.t_cin_pad >> 4, tcin = (unsigned)d.t_cin, tcout = (unsigned)d.t_cout;
            const unsigned c16 = g % tnc16, m16 = g / tnc16;
            const unsigned o = m16 * 16 + (lane & 15);     // output channel of the dgrad conv = input channel of the layer
            ci0 = c16 * 16 + (lane >> 4) * 4;               // input channel of the dgrad conv = output channel of the layer
            lim_ci = tcin;
            okq = okq && o < tcout && (d.t_mode == 0 || slot > 0);
            if (d.t_mode == 0) { sa0 = (ci0 * cin + o) * ks + (ks - 1 - slot); sstep = cin * ks; }   // W[co = ii][ci = o][k - 1 - k']
            else { sa0 = (o * cout + ci0) * ks + (slot - 1); sstep = ks; }                             // W[ci = o][co = ii][k' - 1]
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ok[sl][e] = okq && i0 + e < total && ci0 + e < lim_ci;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[sl][e] = src[ok[sl][e] ? sa0 + e * sstep : 0u];   // (unconditional loads from clamped addresses; zeros selected afterwards)
    }
#pragma unroll
    for (int sl = 0; sl < NSMAX; ++sl) {
        if ((unsigned)sl >= ns) break;
        const unsigned i0 = gfirst + (unsigned)sl * 256u;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i0 + e < total) dst[i0 + e] = ok[sl][e] ? v[sl][e] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Time conditioning for per-sample timesteps, with everything the backward pass needs (layers.py:229-255, 336-340):
//   emb[b] = [sin(t f) | cos(t f)] (32)   h1 = W1 emb + b1 (128)   temb = W3 mish(h1) + b3 (32)   tb[b][row] = Wj mish(temb) + bj
struct TimeTrainArgs {
    const float* flat;          // parameters, reference layout
    const long long* t;         // [B]
    const float* freqs;         // [16]
    float* emb;                 // [B][32]
    float* h1;                  // [B][128]  (pre-Mish)
    float* temb;                // [B][32]   (pre-Mish)
    float* tm;                  // [B][32]   mish(temb): the input of every cond_mlp (kept: its backward needs it once per table row)
    float* h1m;                 // [B][128]  mish(h1)
    float* tb;                  // [B][row]
    unsigned long long w1, b1, w3, b3;
    int row, nblk;
    unsigned long long woff[40], boff[40];
    int cout[40], toff[40];
    // q_sample (diffusion_model_base.py:320-330) + apply_hard_conditioning (:335) of the same sample, and the pass's zero words: this is the
    // first launch of a training pass, and both are per-sample work with no dependence on the time MLP (round 4: were a memset + a launch)
    const float* x0; const float* noise; const float* sqrt_ac; const float* sqrt_1mac; const float* hs; const float* hg;
    float* xn;                  // [B][H][D]
    float* zero_words;          // block 0 clears n_zero floats (the dgrad convolutions' zero bias + the time backward's ticket)
    int H, D, T, n_zero;
    // draw mode (mpdx_train_draw: an iteration replayed as a hipGraph): the timestep and the noise of sample b are DRAWN here - Philox4x32-10 keyed
    // by rng_seed, stream position (*rng_counter) * B + b: the device-resident optimiser step count, so every replay draws afresh - and written to
    // t_out / noise_out (what torch.randint / torch.randn_like were two launches for)
    unsigned long long rng_seed;
    const int* rng_counter;     // null: t and noise are inputs
    long long* t_out;
    float* noise_out;
    // side blocks [B, B + kRestreamBlocksPerJob * n_jobs): the fused programs' weight-stream copies inside `packed` (train_types.hpp restream_job) - they
    // depend on the pack launch before this one only, and were a launch of their own (restream_all_kernel, 5.8 us) between this kernel and the programs
    int B;
    float* packed;
    const CopyJobDev* jobs;
    int n_jobs;
    int Hc;                     // rows per trajectory of xn: the power-of-two container of a horizon that is not one (rows [H, Hc) written as zeros); 0: H
};
constexpr int kRestreamBlocksPerJob = 32;

// uniform integer in [0, T) from one Philox counter (multiply-shift of 32 random bits)
__device__ __forceinline__ int philox_randint(uint64_t seed, uint64_t ctr, int T) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (int)(((uint64_t)c0 * (uint64_t)(uint32_t)T) >> 32);
}

// one block of 512 threads per sample; every stage spreads its dot products over all threads it can use and keeps its loads in flight
// together (the first version ran encoder.3 on 32 threads x 32 dependent loads and the table on 128 threads: 17 us per call)
__global__ __launch_bounds__(512) void time_train_fwd_kernel(const TimeTrainArgs a) {
    __shared__ float emb[32], h1m[128], tm[32];
    __shared__ int s_toff[41];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= 2 * a.B) {   // a side block: its share of one copy job
        const int rb = (int)blockIdx.x - 2 * a.B, job = rb / kRestreamBlocksPerJob, part = rb - job * kRestreamBlocksPerJob;
        restream_job(a.packed, a.jobs[job], (unsigned)part * 512u + (unsigned)tid, (unsigned)kRestreamBlocksPerJob * 512u);
        return;
    }
    // blocks [0, B): the time MLP of sample b; blocks [B, 2 B): its q_sample (+ the draw of t and the noise) - the two halves share nothing but t_b
    // (round 5: one block ran them one after the other, the noise draw in front of the MLP's chain of four dependent stages)
    const bool mlp_half = (int)blockIdx.x < a.B;
    const int b = mlp_half ? (int)blockIdx.x : (int)blockIdx.x - a.B;
    if (tid < a.nblk) s_toff[tid] = a.toff[tid];
    if (tid == 0) s_toff[a.nblk] = a.row;
    const bool draw = a.rng_counter != nullptr;
    const unsigned long long rng_pos = draw ? (unsigned long long)(unsigned)(*a.rng_counter) * (unsigned)a.B + (unsigned)b : 0ull;
    const long long t_b = draw ? (long long)philox_randint(a.rng_seed ^ 0x74696D6573746570ull, rng_pos, a.T) : a.t[b];
    if (draw && tid == 0 && mlp_half) a.t_out[b] = t_b;
    const size_t xn_stride = (size_t)(a.Hc > a.H ? a.Hc : a.H) * a.D;   // floats per trajectory of xn
    if (mlp_half) {
    } else
    if (a.xn && draw) {   // the sample's noise drawn in place (HD % 4 == 0: checked on the host), then q_sample's arithmetic
        const long long tb = t_b < 0 ? 0 : (t_b >= a.T ? a.T - 1 : t_b);
        const float ca = a.sqrt_ac[tb], cb = a.sqrt_1mac[tb];
        const int HD = a.H * a.D, nq = HD >> 2;
        for (int q4 = tid; q4 < nq; q4 += 512) {
            float z[4];
            // key domain-separated from mpdx_randn / fill_randn (seed, offset + q) and from the planning loop's in-kernel draws, which use the
            // model's seed as is: the eager steps between replays (summary / warm-up steps) advance THAT stream's offset and must not meet
            // noise a replayed step already used (the timestep stream above has its own key for the same reason)
            philox_normal4(a.rng_seed ^ 0x747261696E6E6F69ull, rng_pos * (unsigned long long)nq + (unsigned)q4, z);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q4 + e, l = i / a.D, d = i - l * a.D;
                const size_t g = (size_t)b * HD + i;
                a.noise_out[g] = z[e];
                float r = __fadd_rn(__fmul_rn(ca, a.x0[g]), __fmul_rn(cb, z[e]));
                if (a.hs && l == 0) r = a.hs[b * a.D + d];
                if (a.hg && l == a.H - 1) r = a.hg[b * a.D + d];
                a.xn[(size_t)b * xn_stride + i] = r;
            }
        }
    } else if (a.xn) {   // x_t = sqrt(acp[t_b]) x0 + sqrt(1 - acp[t_b]) noise, hard conditions: q_sample_kernel's arithmetic (mpdx.hip)
        long long tb = t_b;
        tb = tb < 0 ? 0 : (tb >= a.T ? a.T - 1 : tb);
        const float ca = a.sqrt_ac[tb], cb = a.sqrt_1mac[tb];
        const int HD = a.H * a.D;
        for (int i = tid; i < HD; i += 512) {
            const int l = i / a.D, d = i - l * a.D;
            const size_t g = (size_t)b * HD + i;
            float r = __fadd_rn(__fmul_rn(ca, a.x0[g]), __fmul_rn(cb, a.noise[g]));
            if (a.hs && l == 0) r = a.hs[b * a.D + d];
            if (a.hg && l == a.H - 1) r = a.hg[b * a.D + d];
            a.xn[(size_t)b * xn_stride + i] = r;
        }
    }
    if (!mlp_half) {
        if (a.xn && a.Hc > a.H)   // the container's rows behind the horizon
            for (int i = a.H * a.D + tid; i < a.Hc * a.D; i += 512) a.xn[(size_t)b * xn_stride + i] = 0.f;
        return;
    }
    // encoder.1 / encoder.3 weights of this thread: requested HERE, in front of the chain of stages that use them (behind a barrier each: hipcc does
    // not move a load across s_barrier, and every stage then began with a round trip to HBM - the optimiser rewrote `flat` two launches ago)
    f32x4 wv1[8], w3a, w3b;
    float b1v, b3v;
    {
        const int t1 = tid < 128 ? tid : 0;
        const f32x4* w = (const f32x4*)(a.flat + a.w1 + t1 * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) wv1[k] = w[k];
        b1v = a.flat[a.b1 + t1];
        const f32x4* w3p = (const f32x4*)(a.flat + a.w3 + (tid >> 4) * 128);
        w3a = w3p[(tid & 15) * 2]; w3b = w3p[(tid & 15) * 2 + 1];
        b3v = a.flat[a.b3 + (tid >> 4)];
    }
    if (b == 0 && a.zero_words)
        for (int i = tid; i < a.n_zero; i += 512) a.zero_words[i] = 0.f;
    if (tid < 16) {
        const float arg = (float)t_b * a.freqs[tid];
        emb[tid] = sinf(arg);
        emb[tid + 16] = cosf(arg);
        a.emb[(size_t)b * 32 + tid] = emb[tid];
        a.emb[(size_t)b * 32 + tid + 16] = emb[tid + 16];
    }
    __syncthreads();
    if (tid < 128) {
        float s = b1v;   // (parameter offsets are multiples of 4 floats)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s = fmaf(wv1[k][0], emb[4 * k], s); s = fmaf(wv1[k][1], emb[4 * k + 1], s); s = fmaf(wv1[k][2], emb[4 * k + 2], s); s = fmaf(wv1[k][3], emb[4 * k + 3], s);
        }
        a.h1[(size_t)b * 128 + tid] = s;
        h1m[tid] = mish(s);
        a.h1m[(size_t)b * 128 + tid] = h1m[tid];
    }
    __syncthreads();
    {   // encoder.3: row r = tid >> 4 (32 rows), 16 lanes per row take two float4 each; the sum runs k ascending inside a lane, lanes by DPP row sum
        const int r = tid >> 4, l16 = tid & 15;
        const f32x4 w0 = w3a, w1 = w3b;
        const float* h = h1m + l16 * 8;
        float s = w0[0] * h[0];
        s = fmaf(w0[1], h[1], s); s = fmaf(w0[2], h[2], s); s = fmaf(w0[3], h[3], s);
        s = fmaf(w1[0], h[4], s); s = fmaf(w1[1], h[5], s); s = fmaf(w1[2], h[6], s); s = fmaf(w1[3], h[7], s);
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));
        s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, true));
        if (l16 == 0) {
            s += b3v;
            a.temb[(size_t)b * 32 + r] = s;
            tm[r] = mish(s);
            a.tm[(size_t)b * 32 + r] = tm[r];
        }
    }
    __syncthreads();
    for (int rr = tid; rr < a.row; rr += 512) {   // the table row of this sample: tb[row] = W_blk[c] . mish(temb) + b_blk[c]
        int blk = 0;
        while (blk + 1 < a.nblk && rr >= s_toff[blk + 1]) ++blk;
        const int c = rr - s_toff[blk];
        const f32x4* w = (const f32x4*)(a.flat + a.woff[blk] + c * 32);
        float s = a.flat[a.boff[blk] + c];
        f32x4 wv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wv[k] = w[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s = fmaf(wv[k][0], tm[4 * k], s); s = fmaf(wv[k][1], tm[4 * k + 1], s); s = fmaf(wv[k][2], tm[4 * k + 2], s); s = fmaf(wv[k][3], tm[4 * k + 3], s);
        }
        a.tb[(size_t)b * a.row + rr] = s;
    }
}

// cond_mlp weight / bias gradients: for row index c (a channel of some block):  dW[c][e] = sum_b dT[b][c] * mish(temb[b][e]),
// db[c] = sum_b dT[b][c].  One thread per (c, e).
struct TimeBwdArgs {
    const float* flat;
    float* grad;                // flat gradient vector
    const float* dT;            // [B][row]
    const float* emb;
    const float* h1;
    const float* temb;
    const float* tm;            // mish(temb), mish(h1) as the forward kernel stored them
    const float* h1m;
    float* dtm;                 // [B][32]  gradient wrt mish(temb), then wrt temb
    float* dh1;                 // [B][128] gradient wrt h1 (kept in LDS by time_bwd_all_kernel; unused)
    unsigned* ticket;           // zero at launch: the sample blocks of time_bwd_all_kernel count themselves here
    unsigned long long w1, b1, w3, b3;
    int B, row, nblk;
    int split_tail;             // the encoder tail runs in its own launch (time_tail_kernel): the sample blocks take no ticket
    unsigned long long woff[40], boff[40];
    int cout[40], toff[40];
};

// The whole backward pass of the time conditioning in ONE launch of 1024-thread blocks (it used to be four dependent launches of
// latency-bound scalar loops, 52 us per iteration at batch 32):
//   blocks [0, B)        sample b:  dtm[b][e] = mish'(temb[b][e]) * sum_rows dT[b][row] * W_row[e]   (rows over 32 parts, dT row and the
//                        rows' weight offsets staged in LDS, loads eight deep), then a ticket; the LAST sample block to finish runs the
//                        tail: time_mlp.encoder.3 (dW3, db3, dh1 = mish'(h1) * W3^T dtemb) and encoder.1 (dW1, db1), 32 samples at a
//                        time out of LDS - no block ever waits for another
//   blocks [B, B + ...)  cond_mlp weight / bias gradients, 32 table rows per block:  dW[c][e] = sum_b dT[b][c] * mish(temb[b][e]),
//                        db[c] = sum_b dT[b][c]
// Every sum runs in the order of the four kernels it replaces (b ascending; table rows by part, parts ascending).
constexpr int kTimeBwdMaxRow = 2560;
__device__ __forceinline__ void time_bwd_all_body(const TimeBwdArgs& a, const int block_id) {
    __shared__ __attribute__((aligned(16))) float sh[10240];   // 40 KB: stage 1 [dT row | row offsets | partial sums], tail [dtm | h1m | dh1 | emb]
    __shared__ int s_toff[41];
    __shared__ unsigned s_woff[40];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x;
#ifdef MPDX_TB_STAMPS   // dev probe: s_memtime differences printed by thread 0 of a few blocks
    long long tbs[8]; int tbn = 0;
#define TB_STAMP() do { tbs[tbn++] = __builtin_amdgcn_s_memtime(); } while (0)
#define TB_DUMP(tag) do { if (tid == 0) printf("time_bwd %s blk %d: %lld %lld %lld %lld %lld %lld\n", tag, block_id, tbs[1]-tbs[0], tbs[2]-tbs[0], tbs[3]-tbs[0], tbs[4]-tbs[0], tbs[5]-tbs[0], tbs[6]-tbs[0]); } while (0)
    for (int i = 0; i < 8; ++i) tbs[i] = 0;
    TB_STAMP();
#else
#define TB_STAMP()
#define TB_DUMP(tag)
#endif
    if (tid < a.nblk) { s_toff[tid] = a.toff[tid]; s_woff[tid] = (unsigned)a.woff[tid]; }
    if (tid == 0) s_toff[a.nblk] = a.row;
    __syncthreads();
    auto block_of = [&](int r) { int blk = 0; while (blk + 1 < a.nblk && r >= s_toff[blk + 1]) ++blk; return blk; };   // toff ascending
    if (block_id >= a.B) {   // ---------------------------------------------- cond_mlp gradients of 32 table rows
        // dT[b][rr0 .. rr0 + 32) and mish(temb)[b][0 .. 32) of 128 samples at a time through LDS: ONE global round trip per chunk (round 3 read
        // both straight from global memory inside the b loop, eight deep)
        const int rr0 = (block_id - a.B) * 32, rloc = tid >> 5, e = tid & 31, rr = rr0 + rloc;
        constexpr int CH = 128;
        float* const dS = sh;              // [CH][32]
        float* const tS = sh + CH * 32;    // [CH][32]
        float sw = 0.f, sb = 0.f;
        for (int b0 = 0; b0 < a.B; b0 += CH) {
            const int nb = a.B - b0 < CH ? a.B - b0 : CH;
            float vd[4], vt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // (unconditional loads, clamped indices)
                const int idx = tid + k * 1024, bb = idx >> 5, c = idx & 31;
                const int bs = b0 + (bb < nb ? bb : nb - 1), rc = rr0 + c < a.row ? rr0 + c : a.row - 1;
                vd[k] = a.dT[(size_t)bs * a.row + rc];
                vt[k] = a.tm[(size_t)bs * 32 + c];
            }
            __syncthreads();   // the previous chunk's reads are done
#pragma unroll
            for (int k = 0; k < 4; ++k) { dS[tid + k * 1024] = vd[k]; tS[tid + k * 1024] = vt[k]; }
            __syncthreads();
#pragma unroll 8
            for (int b = 0; b < nb; ++b) {
                const float d = dS[b * 32 + rloc];
                sw = fmaf(d, tS[b * 32 + e], sw);
                sb += d;
            }
        }
        TB_STAMP();
        if (rr < a.row) {
            const int blk = block_of(rr);
            const int c = rr - s_toff[blk];
            a.grad[a.woff[blk] + (size_t)c * 32 + e] = sw;
            if (e == 0) a.grad[a.boff[blk] + c] = sb;
        }
        TB_STAMP();
        if (block_id == a.B || block_id == a.B + 30) TB_DUMP("cond");
        return;
    }
    // ------------------------------------------------------------------------------- sample b: gradient wrt temb
    const int b = block_id, e = tid & 31, part = tid >> 5;
    float* const dTs = sh;
    unsigned* const roff = (unsigned*)(sh + kTimeBwdMaxRow);
    float* const red = sh + 2 * kTimeBwdMaxRow;   // [32][32]
    for (int r = tid; r < a.row; r += 1024) {
        dTs[r] = a.dT[(size_t)b * a.row + r];
        const int blk = block_of(r);
        roff[r] = s_woff[blk] + (unsigned)(r - s_toff[blk]) * 32u;
    }
    __syncthreads();
    TB_STAMP();   // 1: staged
    {
        // rows part, part + 32, ... in ascending order; their weight loads are issued sixteen at a time, unconditionally (clamped row): as a plain
        // loop (even with an unroll pragma) hipcc issued one load per trip and waited for it - 75-85 k cycles for the 60 trips of a 1 920-row table
        float s = 0.f;
        for (int r0 = part; r0 < a.row; r0 += 32 * 16) {
            float wv[16], dv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = r0 + 32 * u, rc = r < a.row ? r : part;
                wv[u] = a.flat[roff[rc] + e];
                dv[u] = dTs[rc];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (r0 + 32 * u < a.row) s = fmaf(dv[u], wv[u], s);
        }
        red[part * 32 + e] = s;
    }
    __syncthreads();
    if (part == 0) {
        float t = 0.f;
        for (int p = 0; p < 32; ++p) t += red[p * 32 + e];
        // WRITE-THROUGH store (agent-scope atomic = sc1): the row is at the device's coherence point once vmcnt drains.  Round 3 made the rows visible
        // with __threadfence() - a release fence at agent scope writes back the WHOLE L2 of the XCD (buffer_wbl2), dirty with the megabytes of
        // gradient partials the reductions before this launch left there: 70 k cycles per sample block, 35 k more for the acquire side of the
        // last block (s_memtime stamps, -DMPDX_TB_STAMPS): two thirds of the launch.
        __hip_atomic_store(a.dtm + (size_t)b * 32 + e, t * mish_grad(a.temb[(size_t)b * 32 + e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TB_STAMP();
    if (a.split_tail) return;   // (the tail is the next launch: the kernel boundary orders the rows)
    __syncthreads();   // every store of the row has drained (above) before the ticket is taken
    if (tid == 0) s_ticket = atomicAdd(a.ticket, 1u);
    __syncthreads();
    TB_STAMP();   // 3: ticket
    if (s_ticket != (unsigned)a.B - 1u) { if (block_id == 0 || block_id == 77) TB_DUMP("sample"); return; }
    // ------------------------------------------------------------------------------- tail (one block): encoder.3 and encoder.1
    // (the other samples' dtm rows are read with agent-scope atomic loads below: they bypass this XCD's L2, no acquire fence)
    float* const dtmS = sh;            // [32][32]
    float* const h1mS = sh + 1024;     // [32][128]
    float* const dh1S = sh + 5120;     // [32][128]
    float* const embS = sh + 9216;     // [32][32]
    // Output mapping (round 4): a thread's four outputs are CONSECUTIVE rows, so that one 16-byte LDS read feeds four FMAs and the operand shared by
    // the four is read once per sample - 2 LDS reads per 4 FMAs instead of 8 (the loops were LDS-issue bound: 25 k cycles per chunk of 32 samples).
    // Every output still adds its samples in ascending order: same bits.
    const int k = tid & 127, q8 = tid >> 7;     // encoder.3: column k; rows e = 4 q8 + i (dW3), samples bb = q8 + 8 i (dh1)
    const int j = tid & 31, k0 = tid >> 5;      // encoder.1: column j; rows kk = 4 k0 + i
    float w3g[4] = {0.f, 0.f, 0.f, 0.f}, w1g[4] = {0.f, 0.f, 0.f, 0.f};
    float b3s = 0.f, b1s = 0.f;   // bias gradients: thread e < 32 sums dtemb[.][e], thread kk < 128 sums dh1[.][kk] (samples ascending) - not every thread of the GEMM loops
    float w3c[32];   // W3[e][k], e = 0..31 (this thread's column)
#pragma unroll
    for (int ee = 0; ee < 32; ++ee) w3c[ee] = a.flat[a.w3 + (size_t)ee * 128 + k];
    TB_STAMP();   // 4: tail entered
    // a chunk's operands are fetched while the chunk before it is worked on; samples beyond the batch are staged as ZEROS: fmaf(0, x, acc) == acc
    // and acc + 0 == acc exactly, so the loops below run full 32 trips
    float z0, z1, z2[4], zh[4];
    auto fetch = [&](int b0) {
        const int nb = a.B - b0 < 32 ? a.B - b0 : 32;   // (<= 0 behind the last chunk: everything zero, nothing loaded)
        z0 = tid < nb * 32 ? __hip_atomic_load(a.dtm + (size_t)b0 * 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        z1 = tid < nb * 32 ? a.emb[(size_t)b0 * 32 + tid] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) z2[q] = tid + q * 1024 < nb * 128 ? a.h1m[(size_t)b0 * 128 + tid + q * 1024] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) zh[i] = q8 + 8 * i < nb ? a.h1[(size_t)(b0 + q8 + 8 * i) * 128 + k] : 0.f;
    };
    fetch(0);
    for (int b0 = 0; b0 < a.B; b0 += 32) {
        const int nb = a.B - b0 < 32 ? a.B - b0 : 32;
        float hcur[4];
        __syncthreads();
        dtmS[tid] = z0; embS[tid] = z1;
#pragma unroll
        for (int q = 0; q < 4; ++q) h1mS[tid + q * 1024] = z2[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) hcur[i] = zh[i];
        __syncthreads();
        fetch(b0 + 32);
#pragma unroll 8
        for (int bb = 0; bb < 32; ++bb) {   // dW3[e][k] += sum_b dtemb[b][e] mish(h1[b][k]);  db3[e] += sum_b dtemb[b][e]
            const f32x4 d = *(const f32x4*)(dtmS + bb * 32 + 4 * q8);
            const float h = h1mS[bb * 128 + k];
#pragma unroll
            for (int i = 0; i < 4; ++i) w3g[i] = fmaf(d[i], h, w3g[i]);
        }
        if (tid < 32) {
#pragma unroll 8
            for (int bb = 0; bb < 32; ++bb) b3s += dtmS[bb * 32 + tid];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // dh1[b][k] = mish'(h1[b][k]) sum_e dtemb[b][e] W3[e][k]
            const int bb = q8 + 8 * i;
            const float hv = hcur[i];
            float s = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 8; ++e4) {
                const f32x4 d = *(const f32x4*)(dtmS + bb * 32 + 4 * e4);
#pragma unroll
                for (int u = 0; u < 4; ++u) s = fmaf(d[u], w3c[4 * e4 + u], s);
            }
            dh1S[bb * 128 + k] = bb < nb ? s * mish_grad(hv) : 0.f;   // (zero rows behind the batch: see above)
        }
        __syncthreads();
#pragma unroll 8
        for (int bb = 0; bb < 32; ++bb) {   // dW1[k][j] += sum_b dh1[b][k] emb[b][j];  db1[k] += sum_b dh1[b][k]
            const f32x4 d = *(const f32x4*)(dh1S + bb * 128 + 4 * k0);
            const float em = embS[bb * 32 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i) w1g[i] = fmaf(d[i], em, w1g[i]);
        }
        if (tid < 128) {
#pragma unroll 8
            for (int bb = 0; bb < 32; ++bb) b1s += dh1S[bb * 128 + tid];
        }
    }
    if (tid < 32) a.grad[a.b3 + tid] = b3s;
    if (tid < 128) a.grad[a.b1 + tid] = b1s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ee = 4 * q8 + i, kk = 4 * k0 + i;
        a.grad[a.w3 + (size_t)ee * 128 + k] = w3g[i];
        a.grad[a.w1 + (size_t)kk * 32 + j] = w1g[i];
        if (i == 3) { TB_STAMP(); TB_STAMP(); TB_DUMP("tail"); }
    }
}

// (Measured and rejected: these blocks and the weight-gradient reductions in ONE launch - 59 us against 32 + 27 us apart: under the
//  reductions' memory traffic every dependent load of this chain takes several times longer.)
__global__ __launch_bounds__(1024) void time_bwd_all_kernel(const TimeBwdArgs a) { time_bwd_all_body(a, (int)blockIdx.x); }

// The encoder tail of the time backward (time_mlp.encoder.3 / encoder.1: dW3, db3, dh1 = mish'(h1) W3^T dtemb, dW1, db1) as its OWN launch of 8 blocks
// (round 5).  Inside time_bwd_all_kernel the tail is the work of ONE block - the last sample block to finish - and walks the batch 32 samples at a
// time: ~6.6 us per chunk, 26 of the launch's 37 us at batch 128, 105 us at batch 512.  Every quantity of the tail is independent per hidden unit k
// (128 of them): block q takes k in [16 q, 16 q + 16).  Same sums in the same order (samples ascending, fmaf chains over e ascending): same bits.
constexpr int kTimeTailBlocks = 8;
__global__ __launch_bounds__(512) void time_tail_kernel(const TimeBwdArgs a) {
    __shared__ float dtmS[32 * 32], embS[32 * 32], h1mS[32 * 16], dh1S[32 * 16];
    const int tid = threadIdx.x, q = blockIdx.x, k0 = 16 * q;
    const int eA = tid >> 4, kA = tid & 15;       // dW3[eA][k0 + kA];  phase B: sample bb = eA, hidden unit kA
    const int kC = tid >> 5, jC = tid & 31;       // dW1[k0 + kC][jC]
    float w3c[32];                                // W3[e][k0 + kA], e = 0..31
#pragma unroll
    for (int e = 0; e < 32; ++e) w3c[e] = a.flat[a.w3 + (size_t)e * 128 + k0 + kA];
    float w3g = 0.f, w1g = 0.f, b3s = 0.f, b1s = 0.f;
    // a chunk's operands are fetched while the chunk before it is worked on; samples beyond the batch are staged as ZEROS (fmaf(0, x, acc) == acc)
    float z0[2], z1[2], z2, zh;
    auto fetch = [&](int b0) {
        const int nb = a.B - b0 < 32 ? a.B - b0 : 32;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * 512;
            z0[u] = idx < nb * 32 ? a.dtm[(size_t)b0 * 32 + idx] : 0.f;
            z1[u] = idx < nb * 32 ? a.emb[(size_t)b0 * 32 + idx] : 0.f;
        }
        z2 = eA < nb ? a.h1m[(size_t)(b0 + eA) * 128 + k0 + kA] : 0.f;
        zh = eA < nb ? a.h1[(size_t)(b0 + eA) * 128 + k0 + kA] : 0.f;
    };
    fetch(0);
    for (int b0 = 0; b0 < a.B; b0 += 32) {
        const int nb = a.B - b0 < 32 ? a.B - b0 : 32;
        __syncthreads();   // the previous chunk's reads are done
#pragma unroll
        for (int u = 0; u < 2; ++u) { dtmS[tid + u * 512] = z0[u]; embS[tid + u * 512] = z1[u]; }
        h1mS[tid] = z2;
        const float hv = zh;
        __syncthreads();
        fetch(b0 + 32);
#pragma unroll 8
        for (int bb = 0; bb < 32; ++bb) w3g = fmaf(dtmS[bb * 32 + eA], h1mS[bb * 16 + kA], w3g);
        if (q == 0 && tid < 32) {
#pragma unroll 8
            for (int bb = 0; bb < 32; ++bb) b3s += dtmS[bb * 32 + tid];
        }
        {
            float sacc = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) sacc = fmaf(dtmS[eA * 32 + e], w3c[e], sacc);
            dh1S[tid] = eA < nb ? sacc * mish_grad(hv) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int bb = 0; bb < 32; ++bb) w1g = fmaf(dh1S[bb * 16 + kC], embS[bb * 32 + jC], w1g);
        if (tid < 16) {
#pragma unroll 8
            for (int bb = 0; bb < 32; ++bb) b1s += dh1S[bb * 16 + tid];
        }
    }
    a.grad[a.w3 + (size_t)eA * 128 + k0 + kA] = w3g;
    a.grad[a.w1 + (size_t)(k0 + kC) * 32 + jC] = w1g;
    if (q == 0 && tid < 32) a.grad[a.b3 + tid] = b3s;
    if (tid < 16) a.grad[a.b1 + k0 + tid] = b1s;
}

// ------------------------------------------------------------------------------------------------------------------
// Loss gradient (helpers.py:71-99; mean over B*H*D of |e| or e^2 [* weights]):  dE = s * w * d|e|^p / de / (B H D), zero where
// apply_hard_conditioning overwrote the prediction with a constant (diffusion_model_base.py:343).
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ targ, const float* __restrict__ weights_hd,
                                                        int has_hs, int has_hg, int l1, float scale, float* __restrict__ dE, int B, int H, int D) {
    const size_t total = (size_t)B * H * D;
    const float inv = scale / (float)total;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int hd = (int)(i % ((size_t)H * D)), h = hd / D;
        float g = 0.f;
        if (!((has_hs && h == 0) || (has_hg && h == H - 1))) {
            const float e = pred[i] - targ[i];
            g = l1 ? (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 2.0f * e;
            if (weights_hd) g *= weights_hd[hd];
            g *= inv;
        }
        dE[i] = g;
    }
}

// final_conv[1] (Conv1d 1x1, C -> D) input gradient:  gH[b][l][c] = sum_d dE[b][l][d] W[d][c]
__global__ __launch_bounds__(256) void final_dgrad_kernel(const float* __restrict__ dE, const float* __restrict__ w, float* __restrict__ gH, size_t n_rows, int D,
                                                          int C) {
    const size_t total = n_rows * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(dE[r * D + d], w[(size_t)d * C + c], s);
        gH[i] = s;
    }
}

// The loss value, its gradient and the gradient's way back through final_conv[1] in ONE launch (round 4: were three).  Blocks [0, gridDim.x - 1):
// gH[r][c] = sum_d dE[r][d] W[d][c] with dE computed on the fly (loss_grad_kernel's arithmetic; the lane with c == d also stores dE[r][d], which
// the weight / bias gradients of final_conv[1] read) - needs C >= D.  The LAST 16 blocks: the loss value, weighted_loss_kernel's summation order.
__global__ __launch_bounds__(1024) void train_loss_kernel(const float* __restrict__ pred, const float* __restrict__ targ, const float* __restrict__ weights_hd,
                                                          const float* __restrict__ hs, const float* __restrict__ hg, int l1, float scale, float* __restrict__ dE,
                                                          const float* __restrict__ w, float* __restrict__ gH, int B, int H, int D, int C, float* __restrict__ loss_out,
                                                          double* __restrict__ loss_part, unsigned* __restrict__ loss_ticket, int Hc) {
    // Hc > H: dE and gH are written in the network's container layout [B][Hc][.] (rows [H, Hc) zero) - a horizon that is not a power of two
    const unsigned nb = gridDim.x - 16;
    if (blockIdx.x >= nb) {   // the loss value: wave w of weighted_loss_kernel's sum as its own one-wave workgroup (loss.hpp)
        __shared__ float lq[64 * kLossChunk];
        weighted_loss_wave_block(pred, targ, weights_hd, hs, hg, l1, loss_out, B, H, D, (int)(blockIdx.x - nb), loss_part, loss_ticket, lq);
        return;
    }
    const size_t rows = (size_t)B * H, total = rows * C;
    const float inv = scale / (float)(rows * D);
    if (Hc > H) {   // (rare: the plain form, rows of the container; pred / targ are [B][H][D])
        const size_t crow = (size_t)B * Hc;
        for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < crow * C; i += (size_t)nb * 1024) {
            const int c = (int)(i % C);
            const size_t rc = i / C;
            const int h = (int)(rc % Hc);
            const size_t b = rc / Hc, r = b * H + h;
            float sacc = 0.f;
            for (int d = 0; d < D; ++d) {
                float g = 0.f;
                if (h < H && !((hs && h == 0) || (hg && h == H - 1))) {
                    const float e = pred[r * D + d] - targ[r * D + d];
                    g = l1 ? (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 2.0f * e;
                    if (weights_hd) g *= weights_hd[h * D + d];
                    g *= inv;
                }
                if (c == d) dE[rc * D + d] = g;
                sacc = fmaf(g, w[(size_t)d * C + c], sacc);
            }
            gH[i] = sacc;
        }
        return;
    }
    if (1024 % C == 0 && C <= 1024) {
        // Round 5: a block takes 1024 / C rows at a time - the rows' dE terms are computed ONCE (D per row, by the first rows x D threads) and staged in
        // LDS with final_conv[1]'s weights, then thread (row, c) runs its D FMAs out of LDS.  (Below: every (row, c) thread loaded all of its row's D
        // predictions, targets and weights itself - 64 loads per output, 16.6 us for the launch at batch 128 x D = 14.)  Same g, same fmaf chain over d.
        __shared__ float gS[64 * 32], wS[1024];   // [rows per block <= 64][32] | [D][C] (D <= 32, D * C <= 1024, rows per block x D <= 1024: the guard below)
        const int rpb = 1024 / C;
        if (rpb <= 64 && D <= 32 && D * C <= 1024 && rpb * D <= 1024) {
            const int tid = threadIdx.x, rl = tid / C, c = tid - rl * C;
            for (int k = tid; k < D * C; k += 1024) wS[k] = w[k];
            for (size_t r0 = (size_t)blockIdx.x * rpb; r0 < rows; r0 += (size_t)nb * rpb) {
                __syncthreads();   // the previous trip's reads of gS are done (and wS is staged)
                if (tid < rpb * D) {
                    const int rr = tid / D, d = tid - rr * D;
                    const size_t r = r0 + rr;
                    float g = 0.f;
                    if (r < rows) {
                        const int h = (int)(r % H);
                        const bool hard = (hs && h == 0) || (hg && h == H - 1);
                        if (!hard) {
                            const float e = pred[r * D + d] - targ[r * D + d];
                            g = l1 ? (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 2.0f * e;
                            if (weights_hd) g *= weights_hd[h * D + d];
                            g *= inv;
                        }
                        dE[r * D + d] = g;
                    }
                    gS[rr * 32 + d] = g;
                }
                __syncthreads();
                const size_t r = r0 + rl;
                if (r < rows) {
                    float sacc = 0.f;
                    for (int d = 0; d < D; ++d) sacc = fmaf(gS[rl * 32 + d], wS[d * C + c], sacc);
                    gH[r * C + c] = sacc;
                }
            }
            return;
        }
    }
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < total; i += (size_t)nb * 1024) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        const int h = (int)(r % H);
        const bool hard = (hs && h == 0) || (hg && h == H - 1);
        // D <= 16 (checked on the host): the row's operands are loaded up front, unconditionally - with the loads inside a loop of runtime length
        // every element was a dependent round trip (48 us per launch at batch 128 x D = 14)
        float pv[16], tv[16], wv[16], ww[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const int dc = d < D ? d : D - 1;
            pv[d] = pred[r * D + dc]; tv[d] = targ[r * D + dc];
            wv[d] = weights_hd ? weights_hd[h * D + dc] : 1.0f;
            ww[d] = w[(size_t)dc * C + c];
        }
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            if (d >= D) break;
            float g = 0.f;
            if (!hard) {
                const float e = pv[d] - tv[d];
                g = l1 ? (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 2.0f * e;
                if (weights_hd) g *= wv[d];
                g *= inv;
            }
            if (c == d) dE[r * D + d] = g;
            s = fmaf(g, ww[d], s);
        }
        gH[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Optimiser.  sumsq_kernel + adam_kernel = torch.nn.utils.clip_grad_norm_(params, max_norm) (trainer.py:268-272) followed by
// torch.optim.Adam.step() with the defaults the reference uses (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// `cnt` (or null): the device-resident optimiser step counter of a step replayed as a hipGraph (mpdx_adam_step with step < 0): block 0 advances it
// here, the Adam kernel behind this launch reads the new value - the step-dependent bias corrections cannot be kernel arguments of a replayed graph.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part, int* __restrict__ cnt) {
    __shared__ float red[4];
    if (cnt && blockIdx.x == 0 && threadIdx.x == 0) cnt[0] += 1;
    float s = 0.f;
    // a thread's elements in ascending order (the sum's order), their loads eight at a time (unconditional, clamped: fmaf(0, 0, s) == s)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const size_t k = i + u * stride; v[u] = g[k < n ? k : i]; if (k >= n) v[u] = 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(v[u], v[u], s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// norm[0] = sqrt(sum part), norm[1] = clip coefficient min(1, max_norm / (norm + 1e-6))   (one block)
__global__ __launch_bounds__(256) void norm_finish_kernel(const float* __restrict__ part, int n_part, float max_norm, float* __restrict__ norm) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        norm[0] = nrm;
        norm[1] = max_norm > 0.f ? fminf(1.0f, max_norm / (nrm + 1e-6f)) : 1.0f;
    }
}
// `part` (n_part sumsq_kernel partial sums; null: no clipping): every block adds them up in norm_finish_kernel's order (same bits) instead of
// waiting for a one-block launch in between; block 0 publishes norm[0] = |g|, norm[1] = the clip coefficient.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, const float* __restrict__ part, int n_part,
                                                   float max_norm, float* __restrict__ norm, const int* __restrict__ cnt) {
    float coef = 1.0f;
    __shared__ float red[4];
    if (cnt) {   // graph replay: 1-based step count from device memory; the corrections as mpdx_adam_step computes them on the host (double pow, float result)
        __shared__ float s_bc[2];
        if (threadIdx.x == 0) {
            const double st = (double)cnt[0];
            s_bc[0] = 1.0f - (float)pow((double)b1, st);
            s_bc[1] = sqrtf(1.0f - (float)pow((double)b2, st));
        }
        __syncthreads();
        bc1 = s_bc[0]; bc2_sqrt = s_bc[1];
        if (lr < 0.f) lr = norm[5];   // the learning rate from device memory as well (a schedule must not re-capture the graph): scratch[5]
    }
    if (part) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n_part; i += 256) s += part[i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        coef = max_norm > 0.f ? fminf(1.0f, max_norm / (nrm + 1e-6f)) : 1.0f;
        if (blockIdx.x == 0 && threadIdx.x == 0) { norm[0] = nrm; norm[1] = coef; }
    }
    const float step = lr / bc1;
    size_t first = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (((((size_t)p) | ((size_t)g) | ((size_t)m) | ((size_t)v)) & 15) == 0) {   // four elements a thread, 16-byte accesses (the same arithmetic per element)
        const size_t n4 = n >> 2;
        for (size_t i = first; i < n4; i += (size_t)gridDim.x * 256) {
            const f32x4 g4 = ((const f32x4*)g)[i];
            f32x4 m4 = ((const f32x4*)m)[i], v4 = ((const f32x4*)v)[i], p4 = ((const f32x4*)p)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gi = g4[e] * coef;
                const float mi = b1 * m4[e] + (1.0f - b1) * gi;
                const float vi = b2 * v4[e] + (1.0f - b2) * gi * gi;
                m4[e] = mi; v4[e] = vi;
                p4[e] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
            }
            ((f32x4*)m)[i] = m4; ((f32x4*)v)[i] = v4; ((f32x4*)p)[i] = p4;
        }
        first += n4 << 2;   // the tail (n % 4 elements)
    }
    for (size_t i = first; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}
// EMA.update_model_average (trainer.py:67-85): ema = beta * ema + (1 - beta) * p
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, size_t n, float beta) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) ema[i] = ema[i] * beta + (1.0f - beta) * p[i];
}

}  // namespace mpdx
