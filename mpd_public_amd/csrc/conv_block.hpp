// conv_block.hpp - the temporal-conv building block of TemporalUnet as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces (one launch each) the ATen chains behind
//   Conv1dBlock            Conv1d(k=5,pad=2) -> GroupNorm(8) -> Mish      mpd/models/layers/layers.py:276-293
//   ResidualTemporalBlock  "+ cond_mlp(t)" and "+ residual"               mpd/models/layers/layers.py:343-355
//   Downsample1d           Conv1d(k=3,stride=2,pad=1)                     mpd/models/layers/layers.py:258-264
//   Upsample1d             ConvTranspose1d(k=4,stride=2,pad=1)            mpd/models/layers/layers.py:267-273
//   residual_conv          Conv1d(k=1)                                    mpd/models/layers/layers.py:340-341
//
// Formulation.  out[co, (b,l)] = sum_{ci,tap} W[co,ci,tap] * X[b, ci, l*stride + tap - pad]
//   GEMM M = C_out, N = batch*positions, K = C_in*taps, exact fp32 on v_mfma_f32_16x16x4_f32
//   (bitwise an fmaf chain; 157 TFLOP/s peak = the fp32 vector peak, but reachable from one wave per SIMD).
//
// Data layout.  Activations live in HBM channel-last [B][L][C] (the model's own [B,H,D] is already that), so a
//   trajectory's horizon window is staged into LDS with straight 16-B copies and an MFMA B-fragment
//   (4 consecutive input channels at one horizon position) is ONE ds_read_b128.  The horizon halo (conv padding)
//   is materialised as zero rows in LDS, so taps are just row offsets into the staged window.
//   Weights are pre-packed ONCE (pack_conv_weights) in MFMA A-fragment order
//   Wp[m16][c16][slot][lane][4]: a wave's A operand for (16 out-channels, 16 in-channels, tap) is one coalesced 1-KiB load.
//
// Work split.  A workgroup owns MT output channels x NT positions (whole trajectories, so GroupNorm statistics are
//   tile-local: MT is a multiple of the group size, NT a multiple of L).  Its waves split N (WN) and K (WK);
//   K-partials are reduced through LDS in a fixed order (deterministic).  Then one wave per GroupNorm region
//   (group x trajectory) does bias -> mean/var (wave shuffles) -> affine -> Mish -> (+time bias | +residual) -> store.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// Development hooks (cycle stamps, phase-ablation masks) exist only in builds with -DMPDX_DEV_HOOKS
// (MPDX_BUILD_DEFS=-DMPDX_DEV_HOOKS MPDX_BUILD_OUT=build_ab/libmpdx_dev.so python -m mpd_public_amd.build; -DMPDX_LOOP_ABLATION implies it): in the production
// library the trace pointer is the constant nullptr and the ablation mask the constant 0, so every hook folds away at compile time.
#if defined(MPDX_LOOP_ABLATION) && !defined(MPDX_DEV_HOOKS)
#define MPDX_DEV_HOOKS 1
#endif
#ifdef MPDX_DEV_HOOKS
#define MPDX_TRACE_PTR(p) (p)
#define MPDX_DBG(a, bits) ((a).dbg & (bits))
#else
#define MPDX_TRACE_PTR(p) ((long long*)nullptr)
#define MPDX_DBG(a, bits) (0)
#endif

namespace mpdx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum : int { CONV_S1 = 0, CONV_DOWN = 1, CONV_UPT = 2 };
enum : int { EPI_BIAS = 0, EPI_GN_MISH = 1, EPI_GN_MISH_GEN = 2, EPI_GN_BWD = 3 };   // _GEN: GroupNorm regions other than 128 / 256 elements (horizons other than 64)

struct ConvArgs {
    const float* src1;   // [B][L_in][c1]
    const float* src2;   // [B][L_in][c2] second half of a channel concat (torch.cat((x, h.pop()), dim=1)), or null
    const float* wp;     // packed weights
    const float* bias;   // [C_out]
    const float* gamma;  // GroupNorm weight [C_out]
    const float* beta;   // GroupNorm bias   [C_out]
    const float* tbias;  // per-timestep cond_mlp row [C_out] added after Mish, or null
    const float* res;    // residual [B][L_out][C_out] added after Mish, or null
    float* dst;          // [B][L_out][C_out]
    int c1, c2;
    int B, L_in, L_out, C_out;
    int cin_pad;         // C_in rounded up to 16
    int rs;              // LDS row stride in floats (cin_pad + bank pad)
    int gs;              // GroupNorm channels per group
    int n_tiles_n;       // ceil(B*L_out / NT)
    int lg_c4n, lg_Lin, lg_Lout, lg_gs;  // log2 of cin_pad/4, L_in, L_out, gs (all powers of two): no integer division on device
    int dbg;             // ablation mask for mpdx_bench_layer, honoured only in -DMPDX_DEV_HOOKS builds: 1 skip staging, 2 skip MFMA loop,
                         // 4 skip epilogue; only with -DMPDX_LOOP_ABLATION: 8 no weight-ring refills in the loop, 32 no B-fragment LDS reads
    long long* trace;    // s_memtime stamps of workgroups 0 and last / wave 0; read only in -DMPDX_DEV_HOOKS builds
    // training (train.hpp): per-trajectory time-bias rows (tbias + b * tb_stride; 0 on the planning path: one row for the batch)
    // and an optional second destination for the GroupNorm INPUT (conv + bias), which the backward pass differentiates through
    int tb_stride;
    float* pre;          // [B][L_out][C_out] or null
    // EPI_BIAS only (the input-gradient convolutions of train_host.hpp): channels >= c_split go to dst2 (a channel concat is split
    // back into its two sources; either destination may be null = not needed), and with `accum` the result is ADDED to the destination
    // (bit 0: dst, bit 1: dst2 - the first writer of a gradient buffer in a backward pass stores, the others add: no memset)
    float* dst2;
    int c_split, accum;
    // EPI_GN_BWD only (training): the convolution's result is the gradient wrt the OUTPUT of the Conv1dBlock that produced this
    // convolution's input; the epilogue takes it through that block's Mish + GroupNorm backward (gn_mish_bwd_kernel's arithmetic) and
    // stores the gradient wrt the block's convolution output in dst.  `res` = that block's GroupNorm input (the forward's `pre`),
    // gamma / beta = its GroupNorm parameters, bias = zeros; bw_pg / bw_pb / bw_pbias [B][C_out]: per-trajectory channel sums of
    // (g * vhat), (g), (du);  bw_dT (or null): per-trajectory channel sums of the incoming gradient (the block's time-bias gradient)
    // With `accum` the convolution's result is ADDED to what dst already holds (the other consumers of that block's output have put
    // their gradients there) before the backward; bw_gres (or null): the gradient buffer of the block's identity-residual branch, += the
    // incoming gradient.
    float* bw_pg; float* bw_pb; float* bw_pbias; float* bw_dT; float* bw_gres;
    int bw_dT_stride;
    int bw_gres_store;   // this launch is the FIRST writer of bw_gres in the pass: store instead of +=
    // Horizons that are not powers of two (H % 8 == 0: 24, 40, 48, 96 ...; temporal_unet.py:24,80-103 accepts them) run in a power-of-two
    // CONTAINER [B][L_out][C] whose rows l >= Lv_out are kept ZERO by every producer - the zero rows double as the convolution's own
    // padding, so the k-loop and the staging are untouched: only the epilogues mask (GroupNorm statistics over the valid rows,
    // zeros stored beyond them).  Lv_out == L_out (or 0): nothing masked.  Honoured by EPI_GN_MISH_GEN and EPI_BIAS.
    int Lv_out;
    // training, EPI_BIAS input-gradient convolutions of the resampling layers (round 4: were a zero_stuff_kernel / acc_slice_kernel launch each):
    //   stuff: src1 is [B][L_in / 2][c1] and is read ZERO-STUFFED (row li of the window = source row li / 2 for even li, zero for odd li): the
    //          input gradient of a stride-2 convolution is a stride-1 convolution of the stuffed output gradient;
    //   decim: only the even output rows are stored, at row l / 2 of a [B][L_out / 2][.] destination: the input gradient of ConvTranspose1d(4, 2, 1)
    //          is every second output of a stride-1 5-tap convolution of the output gradient.
    int stuff, decim;
};

// wave64 all-reduce (sum) with DPP row operations + 4 readlanes instead of a 6-step ds_bpermute butterfly:
// quad_perm xor1, quad_perm xor2, row_half_mirror, row_mirror give every lane its 16-lane row sum in registers
// (no LDS crossbar round trips), then the four row sums are combined through SGPRs.
// Kernel arguments are read with scalar loads, and hipcc sinks each s_load next to its first use: a kernel with a large
// argument block (or one it indexes dynamically) then pays one cold scalar-cache miss (~1 k cycles to HBM) after another,
// in series.  warm_kernarg<BYTES>() touches every 64-byte line of the argument block with a dummy s_load at kernel entry
// and waits once: one round trip, after which every argument read is a scalar-cache hit.  Used by fused_level_kernel
// (1.9 KB of arguments, indexed per op); measured no effect on kernels whose arguments are read at static offsets.
template <int OFF, int END>
__device__ __forceinline__ void warm_kernarg_lines(unsigned long long kp, int& t) {
    if constexpr (OFF < END) {
        asm volatile("s_load_dword %0, %1, %2" : "+s"(t) : "s"(kp), "n"(OFF));
        warm_kernarg_lines<OFF + 64, END>(kp, t);
    }
}
template <int BYTES>
__device__ __forceinline__ void warm_kernarg() {
    const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    int t = 0;
    warm_kernarg_lines<0, BYTES>(kp, t);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    // the four row sums r0 .. r3 -> (r3 + r2) + (r1 + r0) in row 3, by two row broadcasts (rows 1 and 3 add lane 15 of the row before
    // them; rows 2 and 3 add lane 31) and ONE readlane - round 3 read all four rows out (4 v_readlane + 3 adds through SGPRs + moves, and
    // the s_nops a VALU write -> v_readlane needs); the pairing of the additions is the same, so the sum has the same bits.
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));  // row_bcast:31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Mish(x) = x*tanh(softplus(x)) (torch: softplus threshold 20).  tanh(log1p(e^x)) == n/(n+2), n = e^x(e^x+2).
// Branch-free on the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each): the result is within a few ulp of the libm
// form, far inside the 2e-5 tolerance the U-Net parity tests state.  (Round 1 used __frcp_rn - a correctly rounded reciprocal,
// 11 dependent instructions - and an `if (x > 20) return x` that compiled to one exec-masked branch PER ELEMENT: the four
// elements of a lane ran as four serial dependent chains.  ~2 k cycles per fused-level epilogue went there.)
__device__ __forceinline__ float mish(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    const float n = e * (e + 2.0f);
    const float r = x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
    return x > 20.0f ? x : r;   // (x > 44: n overflows, r is NaN, the select takes x)
}

// The same without the select (fused programs, round 4: one v_min instead of v_cmp + v_cndmask per element): the exponent is clamped at
// 2^30, where n / (n + 2) is 1 to the last bit (from x = 8.7 on), so x > 20.8 gives x * (n * rcp(n)) = x * (1 +- 1 ulp) instead of
// exactly x - and never overflows.  Bit-identical to mish() for x <= 20; GroupNorm outputs (|v - mean| * rstd <= 16) do not reach 20.
__device__ __forceinline__ float mish_nosel(float x) {
    const float e = __builtin_amdgcn_exp2f(fminf(x * 1.4426950408889634f, 30.0f));
    const float n = e * (e + 2.0f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
}

// d/dv [ v * tanh(softplus(v)) ]   (training: GroupNorm + Mish backward)
__device__ __forceinline__ float mish_grad(float v) {
    const float e = __expf(fminf(v, 20.0f));
    const float n = (1.0f + e) * (1.0f + e);
    const float th = (n - 1.0f) / (n + 1.0f);          // tanh(softplus(v))
    const float sg = e / (1.0f + e);                   // sigmoid(v)
    return th + v * (1.0f - th * th) * sg;
}

// 1/sqrt(var + eps) of GroupNorm on v_rsq_f32 (~1 ulp) instead of the IEEE sqrt + divide sequences (~40 dependent instructions)
__device__ __forceinline__ float gn_rstd(float var) { return __builtin_amdgcn_rsqf(var + 1e-5f); }

// ---- Philox4x32-10 counter-based generator + Box-Muller (4 normals per 128-bit counter).  ONE definition serves the stand-alone
// generator (mpdx_randn) and the kernels that draw their noise in place (mpdx_plan with noise == NULL): element i of a stream that
// starts at counter `offset` is component i & 3 of counter offset + (i >> 2), so both routes produce the same bits.
struct NoiseRng {        // in-kernel noise source of one reverse step
    unsigned long long seed;
    unsigned long long offset;   // counter (quad) offset of the plan's stream
    unsigned long long elem0;    // element index of this step's first value in the stream
    int on;                      // 0: read the noise pointer instead
};
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t ctr, float (&z)[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, cs0, s1, cs1;
    sincosf(6.28318530717958647692f * u1, &s0, &cs0);
    sincosf(6.28318530717958647692f * u3, &s1, &cs1);
    z[0] = r0 * cs0; z[1] = r0 * s0; z[2] = r1 * cs1; z[3] = r1 * s1;
}
// the value mpdx_randn(out, n, seed, offset) writes to out[i]
__device__ __forceinline__ float philox_normal_at(uint64_t seed, uint64_t offset, uint64_t i) {
    float z[4];
    philox_normal4(seed, offset + (i >> 2), z);
    const int c = (int)(i & 3);
    return c == 0 ? z[0] : c == 1 ? z[1] : c == 2 ? z[2] : z[3];
}

template <int MODE, int KS>
struct ConvGeom {
    static constexpr int PAD = (MODE == CONV_S1) ? KS / 2 : 1;
    static constexpr int NTAP = (MODE == CONV_UPT) ? 2 : KS;   // k-groups per 16-channel chunk (per parity class)
    static constexpr int NSLOT = (MODE == CONV_UPT) ? 4 : KS;  // packed weight slots per chunk
};

// weight slot -> tap index k of the reference weight tensor.
//   CONV_S1/CONV_DOWN: slot == k.
//   CONV_UPT (ConvTranspose1d k=4,s=2,p=1: out[o] += x[i]*w[k] with o = 2i - 1 + k):
//     even o=2m : k=1 (i=m),   k=3 (i=m-1)   -> slots 0,1
//     odd  o=2m+1: k=2 (i=m),  k=0 (i=m+1)   -> slots 2,3
__host__ __device__ inline int upt_slot_to_k(int slot) { return slot == 0 ? 1 : slot == 1 ? 3 : slot == 2 ? 2 : 0; }

// Compile-time geometry of a launch (round 4).  GeoAny: every length / stride / channel count is read from the argument block (any
// layer).  GeoL8<NC16>: the inner U-Net levels - CONV_S1 on L_in = L_out = 8 positions per trajectory, C_in = 16 * NC16 input channels
// without padding (c1, c2 multiples of 4), LDS row stride C_in + 8 floats (what pick_row_stride finds for them): the twelve conv
// launches of a B = 100 step that are not whole-trajectory programs.  With the geometry known the staging indices are constants and
// shifts, the k-loop is straight-line code (round 3: ~40 SALU instructions of divisions / clamps per k-group between the MFMAs) and
// every LDS address is `lane base + immediate`; tools/isa_census.py: 1 406 VALU + 574 SALU in the generic staging prologue.
struct GeoAny { static constexpr int L = 0, NC16 = 0, RSPAD = 0; };
template <int NC16_> struct GeoL8 { static constexpr int L = 8, NC16 = NC16_, RSPAD = 8; };
// the ConvTranspose1d(k4, s2, p1) of the innermost up level: 8 -> 16 positions (10 staged rows per trajectory: row stride C_in + 4)
template <int NC16_> struct GeoUp8 { static constexpr int L = 8, NC16 = NC16_, RSPAD = 4; };
constexpr int geo_ilog2(int v) { return v <= 1 ? 0 : 1 + geo_ilog2(v >> 1); }

// TBRES: what a GroupNorm epilogue adds behind Mish, when known at compile time (-1: read the pointers; 0 nothing, 1 time bias, 2 residual)
template <int MODE, int KS, int EPI, int MT, int NT, int WN, int WK, class GEO = GeoAny, int TBRES = -1>
__device__ __forceinline__ void conv_block_body(const ConvArgs& a, const int block_id) {
    using G = ConvGeom<MODE, KS>;
    constexpr bool GK = GEO::L > 0;   // geometry known at compile time
    static_assert(!GK || MODE == CONV_S1 || MODE == CONV_UPT, "compile-time geometry: stride-1 and transposed convolutions on 8 input positions");
    constexpr int NWAVE = WN * WK, NTHR = 64 * NWAVE;
    constexpr int MS = MT / 16, NSUB = NT / 16, NSW = NSUB / WN;
    constexpr int PAD = G::PAD, NTAP = G::NTAP, NSLOT = G::NSLOT;
    constexpr int MTP = MT + 4;  // padded row of the reduction buffer (conflict-free ds_write_b128)
    constexpr int MTP4 = MTP / 4;
    static_assert(NSUB % WN == 0, "N split");
    static_assert(MODE != CONV_UPT || (NSW % 2 == 0), "transposed conv pairs even/odd sub-tiles");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wk = wave / WN;
    int tr_i = 0;
#define CB_STAMP() do { if (MPDX_TRACE_PTR(a.trace) && tid == 0 && (block_id == 0 || block_id == (int)gridDim.x - 1)) a.trace[(block_id ? 16 : 0) + tr_i] = (long long)__builtin_readcyclecounter(); ++tr_i; } while (0)
    const int n_mt = a.C_out / MT;   // MT is a compile-time power of two; one uniform division per workgroup
    const int mt = block_id % n_mt, nt = block_id / n_mt;
    constexpr int GLO = (MODE == CONV_UPT) ? 2 * GEO::L : GEO::L;   // output positions per trajectory when the geometry is known
    const int L_in = GK ? GEO::L : a.L_in, L_out = GK ? GLO : a.L_out;
    const int lg_Lin = GK ? geo_ilog2(GEO::L) : a.lg_Lin, lg_Lout = GK ? geo_ilog2(GLO) : a.lg_Lout, lg_c4n = GK ? geo_ilog2(GEO::NC16 * 4) : a.lg_c4n;
    const int spt = NT >> lg_Lout;  // trajectories per tile
    const int s0 = nt * spt;
    const int LP = L_in + 2 * PAD;
    const int RS4 = GK ? (GEO::NC16 * 16 + GEO::RSPAD) / 4 : a.rs >> 2;  // LDS row stride in float4 units (all LDS indexing is in 16-B units: provably aligned)
    f32x4* const smem4 = (f32x4*)smem;
    const int cin = GK ? GEO::NC16 * 16 : a.c1 + a.c2;
    const int c4n = GK ? GEO::NC16 * 4 : a.cin_pad >> 2;

    CB_STAMP();  // 0: kernel entry
    // ------------------------------------------------------------------ weight prefetch (independent of LDS)
    // Each wave owns k-groups wk, wk+WK, ...; their A fragments are streamed from L2/HBM through a PF-deep register
    // ring so that the ~1-2 us load latency is paid once per kernel, under the staging phase, not once per iteration.
    constexpr int NCLS = (MODE == CONV_UPT) ? 2 : 1;
#ifndef MPDX_PF
#define MPDX_PF 2
#endif
    constexpr int PF = MPDX_PF;   // ring depth in k-groups.  Measured (cfg2 / cfg5 plan, ms): PF 1: 27.90 / 739, 2: 27.70 / 745, 3: 27.95 / 760,
                            // 4: 28.22 / 797, 6: 28.65 / 856 - a deeper ring only adds unrolled code and registers.  Re-measured in
                            // round 2 with the order PINNED by sched_barrier(0) and B fragments read one k-group ahead (so that the
                            // depth is real, not re-serialised by the scheduler): 4 / 8 / 16 / 32 blocks per wave -> cfg2 26.36 /
                            // 27.32 / 30.09 / 35.85 ms, cfg5 738 / - / 1012 / 1383 ms (unpinned depth 4: 25.95 / 733).  Deeper is
                            // monotonically WORSE: the 8 waves' up-front requests queue in the CU's 64 B/clk vector-memory path ahead
                            // of the activation staging loads (in-order return), and the clamped refills at the tail re-request the
                            // last k-group.  The mid-level launches are not bound by weight latency.
    const int nc16 = GK ? GEO::NC16 : a.cin_pad >> 4;
    const int ngroups = nc16 * NTAP;
    // A fragments as BUFFER loads: resource = the layer's packed weights, voffset = lane * 16 (one VGPR), soffset = byte offset of the
    // 1-KiB block (wave-uniform: SALU) - no VALU address arithmetic per load (round 3: a 64-bit add per block)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 0x7fffffff, 0x00020000);
    const int wlane = lane * 16;
    const int wtile = (mt * MS) * nc16 * NSLOT * 1024;
    f32x4 af[PF][MS][NCLS];
    auto load_a = [&](int g, f32x4 (&dst)[MS][NCLS]) {
        const int c16 = g / NTAP, ts = g - c16 * NTAP;
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int p = 0; p < NCLS; ++p) {
                const int slot = (MODE == CONV_UPT) ? (p * 2 + ts) : ts;
                dst[m][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, wtile + ((m * nc16 + c16) * NSLOT + slot) * 1024, 0));
            }
    };
    const int niter = (ngroups + WK - 1) / WK;  // uniform over the workgroup's waves
    // ring loads are UNCONDITIONAL (index clamped to a valid k-group) so that the compiler can count them:
    // the wait in front of ring slot u is then a counted vmcnt((PF-1)*MS*NCLS), never vmcnt(0).
    auto ring_g = [&](int it) { const int g = wk + it * WK; return g < ngroups ? g : ngroups - 1; };
    // The first ring fill is issued AFTER the first pass of staging loads (loads return in order: the activation window is needed
    // first - the workgroup barrier waits for it - and the weights stream in behind it).  A/B on MI355X (cfg2 / cfg5 plan):
    // ring first 25.22 / 738 ms, staging first 24.99 / 730 ms; with staging first, depth 3 / 4 / 6 k-groups: 25.30 / 25.52 / 25.99.
    auto ring_init = [&]() {
#pragma unroll
        for (int u = 0; u < PF; ++u) load_a(ring_g(u), af[u]);
    };
#ifdef MPDX_RING_FIRST
    ring_init();
#endif

    // ------------------------------------------------------------------ stage the horizon windows (+halo) into LDS
    if (!MPDX_DBG(a, 1)) {
        // interior rows: idx -> (trajectory s, input position li, float4 column) by shifts; halo rows are zeroed separately
        const int total = (spt << lg_Lin) << lg_c4n;
        const bool vec_ok = GK ? true : ((a.c1 & 3) == 0) && ((a.c2 & 3) == 0);
        constexpr int SB = 4;  // loads in flight per thread
        // The loads are UNCONDITIONAL (addresses clamped to something valid, zeros selected at the LDS store): a load inside
        // a branch into a zero-initialised register makes hipcc wait for ALL outstanding loads (vmcnt(0)) before it issues
        // it, which put the weight-ring round trip and the SB staging round trips in series (5 x ~1.5 k cycles per launch).
        // The first pass is peeled out of the loop (for the B <= 512 shapes it is the only one): inside a loop the
        // compiler's wait insertion is conservative across the back edge and makes the first pass wait for most of the
        // weight ring before it issues its own loads.
        // The vector and the scalar variant are separate code paths with their own registers: sharing them makes the wait
        // counts of one path include the (never issued) loads of the other.
        auto stage_pass = [&](int base, auto vec, auto first) {
            constexpr bool VEC = decltype(vec)::value;
            constexpr bool FIRST = decltype(first)::value;
            f32x4 v[SB];
            int dsto[SB], cc[SB];
            bool ok[SB];
            size_t pos[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * NTHR;
                const bool in = idx < total;
                const int idc = in ? idx : 0;
                const int rowi = idc >> lg_c4n, c = (idc & (c4n - 1)) << 2;
                const int s = rowi >> lg_Lin, li = rowi & (L_in - 1);
                const int b = s0 + s;
                dsto[u] = in ? (s * LP + li + PAD) * RS4 + (c >> 2) : -1;
                ok[u] = in && b < a.B && c < cin;
                cc[u] = c < cin ? c : 0;
                pos[u] = (size_t)(b < a.B ? b : a.B - 1) * L_in + li;
                if (a.stuff) { ok[u] = ok[u] && !(li & 1); pos[u] = (size_t)(b < a.B ? b : a.B - 1) * (L_in >> 1) + (li >> 1); }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if constexpr (VEC) {
                    const float* src = (cc[u] < a.c1) ? a.src1 + pos[u] * a.c1 + cc[u] : a.src2 + pos[u] * a.c2 + (cc[u] - a.c1);
                    v[u] = *(const f32x4*)src;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ce = cc[u] + e;
                        const float* src = (ce < a.c1) ? a.src1 + pos[u] * a.c1 + ce : (ce < cin) ? a.src2 + pos[u] * a.c2 + (ce - a.c1) : a.src1 + pos[u] * a.c1;
                        v[u][e] = *src;
                    }
                }
            }
#ifndef MPDX_RING_FIRST
            if constexpr (FIRST) ring_init();
#endif
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if constexpr (!VEC) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = (cc[u] + e < cin) ? v[u][e] : 0.f;
                }
                if (dsto[u] >= 0) smem4[dsto[u]] = ok[u] ? v[u] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        };
        if (vec_ok) {
            stage_pass(tid, std::true_type{}, std::true_type{});
            for (int base = tid + NTHR * SB; base < total; base += NTHR * SB) stage_pass(base, std::true_type{}, std::false_type{});
        } else {
            stage_pass(tid, std::false_type{}, std::true_type{});
            for (int base = tid + NTHR * SB; base < total; base += NTHR * SB) stage_pass(base, std::false_type{}, std::false_type{});
        }
        // zero halo rows (conv padding): 2*PAD rows per trajectory
        if (PAD > 0) {
            constexpr int P2 = PAD > 0 ? 2 * PAD : 1;
            const int htot = (spt * P2) << lg_c4n;
            for (int idx = tid; idx < htot; idx += NTHR) {
                const int hr = idx >> lg_c4n, c4 = idx & (c4n - 1);
                const int s = hr / P2, k = hr - s * P2;                         // compile-time divisor
                const int lp = (k < PAD) ? k : (L_in + k);                      // rows 0..PAD-1 and L_in+PAD..L_in+2PAD-1
                smem4[(s * LP + lp) * RS4 + c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
#ifndef MPDX_RING_FIRST
    if (MPDX_DBG(a, 1)) ring_init();
#endif
    CB_STAMP();  // 1: own staging loads issued/written
    __syncthreads();
    CB_STAMP();  // 2: window staged (all waves)

    // ------------------------------------------------------------------ MFMA main loop (K split over wk)
    const int j = lane & 15, q = lane >> 4;
    int boff[NSW];   // lane's base offset (floats) into the staged window for each of its N sub-tiles
    int npos[NSW];   // tile-local output position of the lane's MFMA column
#pragma unroll
    for (int i = 0; i < NSW; ++i) {
        const int ns = wn * NSW + i;
        if (MODE == CONV_UPT) {
            const int gm = (ns >> 1) * 16 + j;          // global input index within the tile
            const int s = gm >> lg_Lin, m = gm & (L_in - 1);
            boff[i] = (s * LP + m + PAD) * RS4 + q;  // row of input m (tap row offsets added below), float4 units
            npos[i] = s * L_out + 2 * m + (ns & 1);
        } else {
            const int n = ns * 16 + j;
            const int s = n >> lg_Lout, l = n & (L_out - 1);
            const int r0 = (MODE == CONV_DOWN) ? 2 * l : l;
            boff[i] = (s * LP + r0) * RS4 + q;
            npos[i] = n;
        }
    }

    f32x4 acc[MS][NSW];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int i = 0; i < NSW; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (An explicit one-k-group-ahead register pipeline of the B fragments was measured 2.7 % SLOWER than letting hipcc
    //  interleave the ds_reads of this unrolled body - interleaved A/B, cfg 2: 30.2 vs 29.4 ms per plan.)
    auto kgroup = [&](const int g, f32x4 (&afu)[MS][NCLS]) {   // the MFMAs of k-group g (wave-uniform) out of the staged window
        const int c16 = g / NTAP, ts = g - c16 * NTAP;
#pragma unroll
        for (int i = 0; i < NSW; ++i) {
            const int p = (MODE == CONV_UPT) ? (i & 1) : 0;
            // row offset of this tap in the staged (zero-haloed) window:
            //   conv/down: staged row (l*stride + tap) holds true index l*stride + tap - PAD;
            //   convT:     boff already points at input m; even outputs use (m, m-1), odd outputs (m, m+1).
            const int roff = (MODE == CONV_UPT) ? ((ts == 0) ? 0 : (p == 0 ? -1 : 1)) : ts;
#ifdef MPDX_LOOP_ABLATION
            const f32x4 bf = (a.dbg & 32) ? (f32x4){1.f, 2.f, 3.f, 4.f} : smem4[boff[i] + roff * RS4 + c16 * 4];
#else
            const f32x4 bf = smem4[boff[i] + roff * RS4 + c16 * 4];
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < MS; ++m)
                    acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(afu[m][p][e], bf[e], acc[m][i], 0, 0, 0);
        }
    };
    if constexpr (GK) {
        // geometry known: the wave's k-groups wk, wk + WK, ... as straight-line code (their number is a constant; the (chunk, tap)
        // of a k-group still depends on the wave: scalar arithmetic, once per k-group, nothing between the MFMAs of a k-group)
        constexpr int NGR = GEO::NC16 * NTAP;
        constexpr int NIT = (NGR + WK - 1) / WK;
        static_assert(NGR % WK == 0, "GeoL8: the k-groups split evenly over the K-split waves");
        if (!MPDX_DBG(a, 2)) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                kgroup(wk + it * WK, af[it % PF]);
                if (it + PF < NIT) load_a(wk + (it + PF) * WK, af[it % PF]);   // refill this ring slot
            }
        }
    } else
    for (int it0 = 0; it0 < niter && !MPDX_DBG(a, 2); it0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int g = wk + (it0 + u) * WK;
            if (g < ngroups) kgroup(g, af[u]);   // wave-uniform; only LDS reads + MFMAs are conditional
#ifdef MPDX_LOOP_ABLATION   // a CONDITIONAL refill serialises the ring (hipcc waits vmcnt(0)): never in the production build
            if (!(a.dbg & 8)) load_a(ring_g(it0 + u + PF), af[u]);
#else
            load_a(ring_g(it0 + u + PF), af[u]);  // refill this ring slot (unconditional, clamped)
#endif
        }
    }

    // ------------------------------------------------------------------ K-partials -> LDS (fixed-order reduction)
    CB_STAMP();  // 3: this wave's MFMA loop done
    __syncthreads();
    CB_STAMP();  // 4: all waves done  // every wave is done reading the staged window; reuse the memory
    float* red = smem;
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int i = 0; i < NSW; ++i)
            smem4[(wk * NT + npos[i]) * MTP4 + m * 4 + q] = acc[m][i];
    __syncthreads();
    CB_STAMP();  // 5: partials visible

    // ------------------------------------------------------------------ epilogue
    if (MPDX_DBG(a, 4)) {
        if (tid == 0 && red[0] == 123.456f) a.dst[0] = red[1];
        return;
    }
    if constexpr (EPI == EPI_GN_MISH_GEN) {
        // GroupNorm regions (group x horizon) of 64, 512, 1024 or 2048 elements - horizons other than the shipped 64 (n_support_points
        // 16 ... 128).  Kept OUT of the EPI_GN_MISH instantiations: compiled into the hot kernels, the extra variants cost 3.5 % of the
        // cfg-2 plan (22.55 vs 21.80 ms, A/B on one box).  One wave per region; lane -> elements in (position, channel) order,
        // channels fastest: re = 256 * NCH: NCH float4 chunks per lane, chunk k = elements (k * 64 + lane) * 4 ..+3; re = 64: one per lane.
        const int gs = a.gs;
        const int lg_gpt = (MT == 32 ? 5 : 4) - a.lg_gs;   // log2(groups per tile)
        const int gpt = 1 << lg_gpt;
        const int nreg = spt << lg_gpt;
        const int re = gs << a.lg_Lout;
        const int Lv = (a.Lv_out > 0 && a.Lv_out < L_out) ? a.Lv_out : L_out;   // valid rows of the container (see ConvArgs::Lv_out)
        const float inv_re = 1.0f / (float)(gs * Lv);    // (a power of two - exact - when nothing is masked)
        // a region of re = 64 * W * NCH elements: NCH chunks of W consecutive channels of one position per lane (W = 4: re = 256 NCH;
        // W = 2: re = 128; W = 1: re = 64)
        auto region = [&](auto w_, auto nch_, int s, int gl, int b) {
            constexpr int W = decltype(w_)::value, NCH = decltype(nch_)::value;
            float v[NCH][W], tb[NCH][W], rsd[NCH][W], ga[NCH][W], be[NCH][W];
            size_t o[NCH];
            bool ok[NCH];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int e0 = (k * 64 + lane) * W;
                const int l = e0 >> a.lg_gs, c = gl * gs + (e0 & (gs - 1));
                const int n = s * L_out + l, co = mt * MT + c;
                ok[k] = l < Lv;
                o[k] = ((size_t)(b < a.B ? b : 0) * L_out + l) * a.C_out + co;
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    ga[k][e] = a.gamma[co + e]; be[k][e] = a.beta[co + e];
                    tb[k][e] = a.tbias ? a.tbias[(size_t)(b < a.B ? b : 0) * a.tb_stride + co + e] : 0.f;
                    rsd[k][e] = a.res ? a.res[o[k] + e] : 0.f;
                    float x = red[(size_t)n * MTP + c + e];
#pragma unroll
                    for (int kk = 1; kk < WK; ++kk) x += red[((size_t)(kk * NT + n)) * MTP + c + e];
                    x += a.bias[co + e];
                    v[k][e] = x;
                    if (a.pre && b < a.B) a.pre[o[k] + e] = x;
                }
                float p = v[k][0];
#pragma unroll
                for (int e = 1; e < W; ++e) p += v[k][e];   // (W == 4: ((v0 + v1) + v2) + v3)
                sum += ok[k] ? p : 0.f;
            }
            const float mean = wave_sum(sum) * inv_re;
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                float p = 0.f;
#pragma unroll
                for (int e = 0; e < W; ++e) { v[k][e] -= mean; p += v[k][e] * v[k][e]; }
                sq += ok[k] ? p : 0.f;
            }
            const float var = wave_sum(sq) * inv_re;
            const float rstd = gn_rstd(var);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    float y = mish_nosel(v[k][e] * rstd * ga[k][e] + be[k][e]);
                    y += tb[k][e];
                    y += rsd[k][e];
                    if (b < a.B) a.dst[o[k] + e] = ok[k] ? y : 0.f;
                }
            }
        };
        for (int r = wave; r < nreg; r += NWAVE) {
            const int s = r >> lg_gpt, gl = r & (gpt - 1);
            const int b = s0 + s;
            using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
            using I8 = std::integral_constant<int, 8>;
            switch (re) {
                case 64: region(I1{}, I1{}, s, gl, b); break;
                case 128: region(I2{}, I1{}, s, gl, b); break;
                case 256: region(I4{}, I1{}, s, gl, b); break;
                case 512: region(I4{}, I2{}, s, gl, b); break;
                case 1024: region(I4{}, I4{}, s, gl, b); break;
                default: region(I4{}, I8{}, s, gl, b); break;
            }
        }
    } else
    if constexpr (EPI == EPI_GN_BWD) {
        // Mish + GroupNorm BACKWARD of the Conv1dBlock below this input-gradient convolution, one wave per GroupNorm region exactly as
        // EPI_GN_MISH places them (lane -> EPL consecutive channels of one position): the arithmetic and the summation orders of
        // gn_mish_bwd_kernel (train.hpp), which this epilogue replaces together with its launch.
        const int gs = a.gs;
        const int lg_gpt = (MT == 32 ? 5 : 4) - a.lg_gs;
        const int gpt = 1 << lg_gpt;
        const int nreg = spt << lg_gpt;
        const int re = gs << a.lg_Lout;    // 256 or 128 (checked on the host)
        auto region = [&](auto epl_, int s, int gl, int b) {
            constexpr int EPL = decltype(epl_)::value;
            const float inv_n = 1.0f / (float)(64 * EPL);
            const int e0 = lane * EPL;
            const int l = e0 >> a.lg_gs, c = gl * gs + (e0 & (gs - 1));
            const int n = s * L_out + l, co = mt * MT + c;
            const int bb = b < a.B ? b : 0;
            const size_t o = ((size_t)bb * L_out + l) * a.C_out + co;
            float u[EPL], gy[EPL], ga[EPL], be[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) { u[e] = a.res[o + e]; ga[e] = a.gamma[co + e]; be[e] = a.beta[co + e]; }
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float v = red[(size_t)n * MTP + c + e];
#pragma unroll
                for (int k = 1; k < WK; ++k) v += red[((size_t)(k * NT + n)) * MTP + c + e];
                gy[e] = v + a.bias[co + e];
            }
            if (a.accum & 1) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) gy[e] += a.dst[o + e];
            }
            if (a.bw_gres && b < a.B) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) a.bw_gres[o + e] = a.bw_gres_store ? gy[e] : a.bw_gres[o + e] + gy[e];
            }
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) sm += u[e];
            const float mean = wave_sum(sm) * inv_n;
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
                gm[e] = gy[e] * mish_grad(vh[e] * ga[e] + be[e]);
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
                if (b < a.B) a.dst[o + e] = du[e];
            }
            float r[4 * EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) { r[e] = gm[e] * vh[e]; r[EPL + e] = gm[e]; r[2 * EPL + e] = du[e]; r[3 * EPL + e] = gy[e]; }
            for (int off = gs / EPL; off < 64; off <<= 1) {
#pragma unroll
                for (int k = 0; k < 4 * EPL; ++k) r[k] += __shfl_xor(r[k], off, 64);
            }
            if (l == 0 && b < a.B) {
                const size_t po = (size_t)b * a.C_out + co;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    a.bw_pg[po + e] = r[e];
                    a.bw_pb[po + e] = r[EPL + e];
                    a.bw_pbias[po + e] = r[2 * EPL + e];
                    if (a.bw_dT) a.bw_dT[(size_t)b * a.bw_dT_stride + co + e] = r[3 * EPL + e];
                }
            }
        };
        for (int r = wave; r < nreg; r += NWAVE) {
            const int s = r >> lg_gpt, gl = r & (gpt - 1);
            if (re == 256) region(std::integral_constant<int, 4>{}, s, gl, s0 + s);
            else region(std::integral_constant<int, 2>{}, s, gl, s0 + s);
        }
    } else
    if (EPI == EPI_GN_MISH) {
        const int gs = a.gs;
        const int lg_gpt = (MT == 32 ? 5 : 4) - a.lg_gs;   // log2(groups per tile)
        const int gpt = 1 << lg_gpt;
        const int nreg = spt << lg_gpt;    // GroupNorm regions in the tile
        const int re = gs << lg_Lout;      // elements per region: 256 (down/mid/final) or 128 (up path)
        const bool has_tb = TBRES < 0 ? a.tbias != nullptr : TBRES == 1, has_res = TBRES < 0 ? a.res != nullptr : TBRES == 2;
        const float inv_re = (re == 256) ? (1.0f / 256.0f) : (1.0f / 128.0f);
        for (int r = wave; r < nreg; r += NWAVE) {
            const int s = r >> lg_gpt, gl = r & (gpt - 1);
            const int b = s0 + s;
            if (re == 256) {
                const int e0 = lane * 4;
                const int l = e0 >> a.lg_gs, c = gl * gs + (e0 & (gs - 1));
                const int n = s * L_out + l, co = mt * MT + c;
                // issue the epilogue's global operands first: their latency hides under the LDS reduction + statistics
                const size_t o = ((size_t)(b < a.B ? b : 0) * L_out + l) * a.C_out + co;
                const f32x4 bi = *(const f32x4*)(a.bias + co);
                const f32x4 ga = *(const f32x4*)(a.gamma + co), be = *(const f32x4*)(a.beta + co);
                f32x4 tb = {0.f, 0.f, 0.f, 0.f}, rs4 = {0.f, 0.f, 0.f, 0.f};
                if (has_tb) tb = *(const f32x4*)(a.tbias + (size_t)(b < a.B ? b : 0) * a.tb_stride + co);
                if (has_res) rs4 = *(const f32x4*)(a.res + o);
                const int ri = n * MTP4 + (c >> 2);  // gs % 4 == 0 -> c % 4 == 0
                f32x4 v = smem4[ri];
#pragma unroll
                for (int k = 1; k < WK; ++k) v += smem4[ri + k * NT * MTP4];
                v += bi;
                if (a.pre && b < a.B) *(f32x4*)(a.pre + o) = v;
                const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * inv_re;
                const f32x4 d = v - mean;
                const float var = wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * inv_re;
                const float rstd = gn_rstd(var);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = mish_nosel(d[e] * rstd * ga[e] + be[e]);
                if (TBRES < 0 || TBRES == 1) y += tb;
                if (TBRES < 0 || TBRES == 2) y += rs4;
                if (b < a.B) *(f32x4*)(a.dst + o) = y;
            } else {  // re == 128
                const int e0 = lane * 2;
                const int l = e0 >> a.lg_gs, c = gl * gs + (e0 & (gs - 1));
                const int n = s * L_out + l, co = mt * MT + c;
                const size_t o = ((size_t)(b < a.B ? b : 0) * L_out + l) * a.C_out + co;
                const f32x2 bi = *(const f32x2*)(a.bias + co);
                const f32x2 ga = *(const f32x2*)(a.gamma + co), be = *(const f32x2*)(a.beta + co);
                f32x2 tb = {0.f, 0.f}, rs2 = {0.f, 0.f};
                if (has_tb) tb = *(const f32x2*)(a.tbias + (size_t)(b < a.B ? b : 0) * a.tb_stride + co);
                if (has_res) rs2 = *(const f32x2*)(a.res + o);
                f32x2 v = *(const f32x2*)(red + (size_t)n * MTP + c);
#pragma unroll
                for (int k = 1; k < WK; ++k) v += *(const f32x2*)(red + ((size_t)(k * NT + n)) * MTP + c);
                v += bi;
                if (a.pre && b < a.B) *(f32x2*)(a.pre + o) = v;
                const float mean = wave_sum(v[0] + v[1]) * inv_re;
                const f32x2 d = v - mean;
                const float var = wave_sum(d[0] * d[0] + d[1] * d[1]) * inv_re;
                const float rstd = gn_rstd(var);
                f32x2 y;
#pragma unroll
                for (int e = 0; e < 2; ++e) y[e] = mish_nosel(d[e] * rstd * ga[e] + be[e]);
                if (TBRES < 0 || TBRES == 1) y += tb;
                if (TBRES < 0 || TBRES == 2) y += rs2;
                if (b < a.B) *(f32x2*)(a.dst + o) = y;
            }
        }
    } else {
        constexpr int M4 = MT / 4;
        for (int idx = tid; idx < NT * M4; idx += NTHR) {
            const int n = idx / M4, c = (idx - n * M4) * 4;   // M4 is a compile-time power of two
            const int s = n >> lg_Lout, l = n & (L_out - 1), b = s0 + s;
            const int co = mt * MT + c;
            const int ri = n * MTP4 + (c >> 2);
            f32x4 v = smem4[ri];
#pragma unroll
            for (int k = 1; k < WK; ++k) v += smem4[ri + k * NT * MTP4];
            v += *(const f32x4*)(a.bias + co);
            float* d = a.dst;
            int ld = a.C_out, cc = co;
            if (a.c_split > 0) {
                if (co >= a.c_split) { d = a.dst2; cc = co - a.c_split; ld = a.C_out - a.c_split; }
                else ld = a.c_split;
            }
            if (a.decim && (l & 1)) continue;
            if (b < a.B && d) {
                f32x4* q = a.decim ? (f32x4*)(d + ((size_t)b * (L_out >> 1) + (l >> 1)) * ld + cc) : (f32x4*)(d + ((size_t)b * L_out + l) * ld + cc);
                if (a.accum & (d == a.dst2 ? 2 : 1)) v += *q;
                if (a.Lv_out > 0 && l >= a.Lv_out) v = (f32x4){0.f, 0.f, 0.f, 0.f};   // rows beyond the valid horizon of the container stay zero
                *q = v;
            }
        }
    }
    CB_STAMP();  // 6: epilogue done (wave 0)
#undef CB_STAMP
}

template <int MODE, int KS, int EPI, int MT, int NT, int WN, int WK, class GEO = GeoAny, int TBRES = -1>
__global__ __launch_bounds__(64 * WN * WK) void conv_block_kernel(const ConvArgs a) {
    conv_block_body<MODE, KS, EPI, MT, NT, WN, WK, GEO, TBRES>(a, blockIdx.x);
}

// blocks[0] of a ResidualTemporalBlock (k5 conv + GroupNorm + Mish + time bias) and the block's residual 1x1 conv read the
// same input and are independent: ONE launch, the first n_first workgroups run the former, the rest the latter.
template <int MT, int NT, class GEO = GeoAny>
__global__ __launch_bounds__(512) void conv_pair_kernel(const ConvArgs a1, const ConvArgs a2, const int n_first) {
    if ((int)blockIdx.x < n_first) conv_block_body<CONV_S1, 5, EPI_GN_MISH, MT, NT, 1, 8, GEO, (GEO::L > 0 ? 1 : -1)>(a1, blockIdx.x);   // blocks[0]: + time bias
    else conv_block_body<CONV_S1, 1, EPI_BIAS, MT, NT, 1, 8, GEO>(a2, blockIdx.x - n_first);
}

// LDS bytes a launch needs: max(staged windows, K-partial buffer)
template <int MODE, int KS, int MT, int NT, int WK>
inline size_t conv_block_lds_bytes(int L_in, int L_out, int rs) {
    using G = ConvGeom<MODE, KS>;
    const size_t stage = (size_t)(NT / L_out) * (L_in + 2 * G::PAD) * rs * sizeof(float);
    const size_t red = (size_t)WK * NT * (MT + 4) * sizeof(float);
    return stage > red ? stage : red;
}

}  // namespace mpdx
