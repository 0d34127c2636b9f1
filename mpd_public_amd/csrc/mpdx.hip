// mpdx.hip - libmpdx.so: host-side layer plan + C ABI (include/mpdx.h) + the small streaming kernels.
//
// gfx950 only.  No CUDA shims, no dual paths.  All device memory is caller-owned; nothing here synchronises.
#include "host.hpp"
#include "loss.hpp"

namespace mpdx {

// ------------------------------------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) must be applied once per (device, kernel): guarded by a mutex and keyed by
// the current device, so that several host threads / several GPUs in one process are safe.
int raise_lds_limit(const void* kern) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kern})) return 0;
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({dev, kern});
    return 0;
}

// MPDX_DEBUG=1: check hipGetLastError() after every launch group of mpdx_plan (2: also synchronise the stream there, so
// that an asynchronous fault is attributed to the step that caused it).  Off by default: the plan never synchronises.
int debug_level() {
    static const int lv = getenv("MPDX_DEBUG") ? atoi(getenv("MPDX_DEBUG")) : 0;
    return lv;
}

static long long* g_conv_trace = nullptr;   // dev tool (mpdx_layer_trace)
static thread_local int g_plan_chains = 1;  // > 1 while mpdx_plan enqueues a plan as that many concurrent sub-batch chains (tile choice)

// ------------------------------------------------------------------------------------------------ small kernels

// TimeEncoder (layers.py:229-255) and every ResidualTemporalBlock.cond_mlp (layers.py:336-340) depend only on the
// integer timestep: tabulate row t = [ cond_mlp_0(temb_t) | cond_mlp_1(temb_t) | ... ] once per model.
struct TimeTabArgs {
    const float* packed;
    const float* freqs;  // [16]
    float* tab;          // [T][row]
    int w1, b1, w2, b2;  // offsets of time_mlp.encoder.{1,3}.{weight,bias}
    int row;             // sum of C_out over blocks
    int nblk;
    int woff[40], boff[40], cout[40], toff[40];
};

__global__ __launch_bounds__(128) void timetab_kernel(const TimeTabArgs a) {
    __shared__ float emb[32], h1[128], te[32];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (tid < 16) {
        const float arg = (float)t * a.freqs[tid];  // x[:, None] * emb[None, :]  layers.py:252
        emb[tid] = sinf(arg);
        emb[tid + 16] = cosf(arg);
    }
    __syncthreads();
    {   // Linear(32,128) + Mish
        const float* w = a.packed + a.w1 + tid * 32;
        float s = a.packed[a.b1 + tid];
        for (int k = 0; k < 32; ++k) s = fmaf(w[k], emb[k], s);
        h1[tid] = mish(s);
    }
    __syncthreads();
    if (tid < 32) {  // Linear(128,32), then the Mish that opens every cond_mlp
        const float* w = a.packed + a.w2 + tid * 128;
        float s = a.packed[a.b2 + tid];
        for (int k = 0; k < 128; ++k) s = fmaf(w[k], h1[k], s);
        te[tid] = mish(s);
    }
    __syncthreads();
    for (int blk = 0; blk < a.nblk; ++blk) {
        for (int c = tid; c < a.cout[blk]; c += 128) {
            const float* w = a.packed + a.woff[blk] + c * 32;
            float s = a.packed[a.boff[blk] + c];
            for (int k = 0; k < 32; ++k) s = fmaf(w[k], te[k], s);
            a.tab[(size_t)t * a.row + a.toff[blk] + c] = s;
        }
    }
}


__global__ __launch_bounds__(256) void final_step_kernel(const FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* w = sm;                 // [D][C]
    float* bs = sm + a.D * a.C;    // [D]
    for (int i = threadIdx.x; i < a.D * a.C; i += blockDim.x) w[i] = a.w[i];
    for (int i = threadIdx.x; i < a.D; i += blockDim.x) bs[i] = a.bias[i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;  // b*H + l
    const bool live = p < a.B * a.H;
    const int b = live ? p / a.H : 0, l = live ? p - b * a.H : 0;
    float vmax = 0.f;
    if (live) {
        const float* hp = a.h + ((size_t)b * (a.Hc > 0 ? a.Hc : a.H) + l) * a.C;
        for (int d = 0; d < a.D; ++d) {
            float s = bs[d];
            for (int c = 0; c < a.C; c += 4) {
                const f32x4 hv = *(const f32x4*)(hp + c);
                const f32x4 wv = *(const f32x4*)(w + d * a.C + c);
                s = fmaf(hv[0], wv[0], s); s = fmaf(hv[1], wv[1], s);
                s = fmaf(hv[2], wv[2], s); s = fmaf(hv[3], wv[3], s);
            }
            const size_t o = (size_t)p * a.D + d;
            float r;
            if (a.mode == 0) {
                r = s;
            } else {
                const float xv = a.x_in[o];
                float x0;
                if (a.k.predict_epsilon)
                    x0 = __fsub_rn(__fmul_rn(a.k.sqrt_recip_alphas_cumprod, xv), __fmul_rn(a.k.sqrt_recipm1_alphas_cumprod, s));
                else
                    x0 = s;
                if (a.mode == 3) {  // ddim_sample (diffusion_model_base.py:216-237): x_start is not clamped there
                    const float pn = a.k.predict_epsilon
                                         ? s
                                         : __fdiv_rn(__fsub_rn(__fmul_rn(a.k.sqrt_recip_alphas_cumprod, xv), s), a.k.sqrt_recipm1_alphas_cumprod);
                    r = __fadd_rn(__fmul_rn(x0, a.k.ddim_k1), __fmul_rn(a.k.ddim_k2, pn));
                    if (a.hs && l == 0) r = a.hs[(size_t)b * a.D + d];
                    if (a.hg && l == a.H - 1) r = a.hg[(size_t)b * a.D + d];
                } else {
                    if (a.k.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                    r = __fadd_rn(__fmul_rn(a.k.posterior_mean_coef1, x0), __fmul_rn(a.k.posterior_mean_coef2, xv));
                    if (a.mode == 1) {
                        if (a.rng.on) r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.k.noise_scale, philox_normal_at(a.rng.seed, a.rng.offset, a.rng.elem0 + o)), a.k.noise_std_extra));
                        else if (a.noise) r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.k.noise_scale, a.noise[o]), a.k.noise_std_extra));
                        if (a.hs && l == 0) r = a.hs[(size_t)b * a.D + d];
                        if (a.hg && l == a.H - 1) r = a.hg[(size_t)b * a.D + d];
                    }
                }
            }
            a.out[o] = r;
            if (a.chain) a.chain[o] = r;
            vmax = fmaxf(vmax, fabsf(r));
        }
    }
    if (a.absmax) {
        // one wave == one trajectory when H == 64: reduce in-wave, one atomic per wave
        const int ctx = b / a.n_per_ctx;
        const int ctx0 = __builtin_amdgcn_readfirstlane(ctx);
        if (__all(ctx == ctx0)) {
            float m = vmax;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
            if ((threadIdx.x & 63) == 0) atomicMax(a.absmax + ctx0, __float_as_uint(m));
        } else if (live) {
            atomicMax(a.absmax + ctx, __float_as_uint(vmax));
        }
    }
}

__global__ __launch_bounds__(256) void add_noise_kernel(float* x, const float* noise, const float* hs, const float* hg,
                                                         float scale, float extra, float* chain, int B, int H, int D) {
    const size_t n = (size_t)B * H * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = i % D;
        const size_t p = i / D;
        const int l = p % H;
        const size_t b = p / H;
        float r = x[i];
        if (noise) r = __fadd_rn(r, __fmul_rn(__fmul_rn(scale, noise[i]), extra));
        if (hs && l == 0) r = hs[b * D + d];
        if (hg && l == H - 1) r = hg[b * D + d];
        x[i] = r;
        if (chain) chain[i] = r;
    }
}

// apply_hard_conditioning (sample_functions.py:5-8) for ARBITRARY horizon indices: x[:, idx[k], :] = vals[k][:, :] for k < n, in the dict's
// order (a later entry wins on a repeated index, as successive indexed writes do).  One thread per (entry, trajectory, dimension).
constexpr int kMaxHardConds = 16;
struct HardCondArgs { int idx[kMaxHardConds]; const float* vals[kMaxHardConds]; int n; };
__global__ __launch_bounds__(256) void hard_conds_kernel(float* x, float* chain, const HardCondArgs a, int B, int H, int D) {
    const size_t per = (size_t)B * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / D;
        const int d = (int)(i - b * D);
        for (int k = 0; k < a.n; ++k) {   // in order: every thread owns its (b, d) column of all entries, so a repeated index resolves as in the reference
            const float v = a.vals[k][i];
            const size_t o = (b * H + a.idx[k]) * D + d;
            x[o] = v;
            if (chain) chain[o] = v;
        }
    }
}

// q_sample (diffusion_model_base.py:320-330) + apply_hard_conditioning (:335): per-trajectory timestep, schedule rows
// looked up on the device.  x_t = sqrt(acp[t_b]) * x0 + sqrt(1 - acp[t_b]) * noise
__global__ __launch_bounds__(256) void q_sample_kernel(const float* x0, const float* noise, const long long* t, const float* sqrt_ac,
                                                        const float* sqrt_1mac, const float* hs, const float* hg, float* out, int B, int H,
                                                        int D, int T) {
    const size_t n = (size_t)B * H * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = i % D;
        const size_t p = i / D;
        const int l = p % H;
        const size_t b = p / H;
        long long tb = t[b];
        tb = tb < 0 ? 0 : (tb >= T ? T - 1 : tb);
        float r = __fadd_rn(__fmul_rn(sqrt_ac[tb], x0[i]), __fmul_rn(sqrt_1mac[tb], noise[i]));
        if (hs && l == 0) r = hs[b * D + d];
        if (hg && l == H - 1) r = hg[b * D + d];
        out[i] = r;
    }
}

// WeightedL1 / WeightedL2 (helpers.py:71-99) of apply_hard_conditioning(pred) against targ: mean over all B*H*D elements
// of |.| or (.)^2, optionally times weights[H*D].  One workgroup, fixed summation order (deterministic); validation-sized
// inputs (B*H*D ~ 1e5 - 1e7).
__global__ __launch_bounds__(1024) void weighted_loss_kernel(const float* pred, const float* targ, const float* weights, const float* hs,
                                                             const float* hg, int l1, float* out, int B, int H, int D) {
    __shared__ double part[16];
    weighted_loss_body(pred, targ, weights, hs, hg, l1, out, B, H, D, part);
}

// standard-normal generator (Philox4x32-10 + Box-Muller, conv_block.hpp): 4 normals per counter.
__global__ __launch_bounds__(256) void randn_kernel(float* out, size_t n, uint64_t seed, uint64_t offset) {
    const size_t nquad = (n + 3) / 4;
    for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < nquad; qd += (size_t)gridDim.x * blockDim.x) {
        float z[4];
        philox_normal4(seed, qd + offset, z);
        const size_t base = qd * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (base + e < n) out[base + e] = z[e];
    }
}

// network input [B][H][D] -> container [B][Hc][D] with zero rows behind the H real ones (horizons that are not powers of two)
__global__ __launch_bounds__(256) void pad_input_kernel(const float* __restrict__ x, float* __restrict__ xc, int B, int H, int Hc, int D) {
    const size_t n = (size_t)B * Hc * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const size_t p = i / D;
        const int l = (int)(p % Hc);
        const size_t b = p / Hc;
        xc[i] = l < H ? x[(b * H + l) * D + d] : 0.f;
    }
}

__global__ void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------------------------
// weight repacking: reference layout -> MFMA A-fragment order  Wp[m16][c16][slot][lane][4]
//   conv:   src [C_out][C_in][k]        (nn.Conv1d)
//   convT:  src [C_in][C_out][k]        (nn.ConvTranspose1d)
__global__ void pack_conv_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int C_out, int C_in,
                                         int ksz, int cin_pad, int nslot, int transposed) {
    const int nc16 = cin_pad >> 4;
    const size_t total = (size_t)(C_out >> 4) * nc16 * nslot * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3, lane = (i >> 2) & 63;
        size_t r = i >> 8;
        const int slot = r % nslot; r /= nslot;
        const int c16 = r % nc16; const int m16 = r / nc16;
        const int co = m16 * 16 + (lane & 15);
        const int ci = c16 * 16 + (lane >> 4) * 4 + e;
        float v = 0.f;
        if (ci < C_in) {
            if (transposed) v = src[((size_t)ci * C_out + co) * ksz + upt_slot_to_k(slot)];
            else v = src[((size_t)co * C_in + ci) * ksz + slot];
        }
        dst[i] = v;
    }
}


// ------------------------------------------------------------------------------------------------ host-side model

}  // namespace mpdx


namespace mpdx {

static int gn_groups(int c) {  // layers.py:389-395
    if (c < 8) return 1;
    for (int g = 8; g < 18; ++g)
        if (c % g == 0) return g;
    return 1;
}

// input channels as the kernels see them: a multiple of 16 (one MFMA k-group) that is also a power of two (channel indices are shifts) - 33 ... 48
// real channels run in a 64-channel container whose extra channels are zero in the staged input AND in the packed weights
static int pad_cin(int c) {
    int p = (c + 15) / 16 * 16;
    while (p & (p - 1)) p += 16;
    return p;
}

static int add_param(mpdx_unet* u, const std::string& name, std::initializer_list<int> shape, int kind = PK_VEC) {
    Param p;
    p.name = name;
    p.ndim = (int)shape.size();
    p.n = 1;
    int i = 0;
    for (int s : shape) { p.shape[i++] = s; p.n *= (size_t)s; }
    p.kind = kind;
    if (kind == PK_CONV) {
        p.cout = p.shape[0]; p.cin = p.shape[1]; p.ksz = p.shape[2];
        p.nslot = p.ksz;
    } else if (kind == PK_CONVT) {
        p.cin = p.shape[0]; p.cout = p.shape[1]; p.ksz = p.shape[2];
        p.nslot = 4;
    }
    if (kind != PK_VEC) {
        p.cin_pad = pad_cin(p.cin);
        p.pn = (size_t)(p.cout / 16) * (p.cin_pad / 16) * p.nslot * 256;
    } else {
        p.pn = (p.n + 3) / 4 * 4;
    }
    p.off = u->packed_floats;
    u->packed_floats += p.pn;
    u->pidx[name] = (int)u->params.size();
    u->params.push_back(p);
    return (int)u->params.size() - 1;
}

// LDS row stride (floats) for the staged window: smallest pad that minimises ds_read_b128 bank conflicts of the
// B-fragment gather (lane (j,q) reads 16 B at row(j)*rs + 4q; ds_read_b128 is served in the four 16-lane groups
// listed in MI355X_MICROARCH.md section LDS; bank = dword address mod 64).
int pick_row_stride(int cin_pad, int mode, int L_in, int L_out, int LP) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int best_rs = cin_pad, best = 1 << 30;
    for (int pad = 0; pad <= 64; pad += 4) {
        const int rs = cin_pad + pad;
        int score = 0;
        for (int g = 0; g < 4; ++g) {
            int cnt[16] = {0};
            int worst = 0;
            for (int i = 0; i < 16; ++i) {
                const int lane = groups[g][i], j = lane & 15, q = lane >> 4;
                int row;
                if (mode == CONV_UPT) { const int s = j / L_in, m = j % L_in; row = s * LP + m; }
                else { const int s = j / L_out, l = j % L_out; row = s * LP + (mode == CONV_DOWN ? 2 * l : l); }
                const int slot16 = ((row * rs + q * 4) & 63) >> 2;
                worst = std::max(worst, ++cnt[slot16]);
            }
            score += worst;
        }
        if (score < best) { best = score; best_rs = rs; }
        if (best == 4) break;
    }
    return best_rs;
}

static void build_model(mpdx_unet* u) {
    const mpdx_unet_cfg& c = u->cfg;
    const int nl = c.n_levels, D = c.state_dim, te = c.time_emb_dim;
    const int Hv = c.n_support_points;   // the horizon; H below = its power-of-two container (== Hv for 16 / 32 / 64 / 128)
    int H = 1;
    while (H < Hv) H <<= 1;
    u->Hc = H;
    std::vector<int> dims(nl + 1);
    dims[0] = D;
    for (int i = 0; i < nl; ++i) dims[i + 1] = c.unet_input_dim * c.dim_mults[i];

    add_param(u, "time_mlp.encoder.1.weight", {128, 32});
    add_param(u, "time_mlp.encoder.1.bias", {128});
    add_param(u, "time_mlp.encoder.3.weight", {te, 128});
    add_param(u, "time_mlp.encoder.3.bias", {te});

    const int P0 = 0, P1 = 1, HB = 2, RB = 3, S0 = 4;
    u->n_slots = S0 + nl;
    auto other = [&](int s) { return s == P0 ? P1 : P0; };
    size_t slot = 0;

    auto conv_layer = [&](const std::string& wname, const std::string& bname, int mode, int ks, int epi, int src1, int c1,
                          int src2, int c2, int cout, int L_in, int L_out, int dst) -> Layer& {
        Layer l;
        l.name = wname; l.mode = mode; l.ks = ks; l.epi = epi;
        l.src1 = src1; l.c1 = c1; l.src2 = src2; l.c2 = c2; l.cout = cout; l.L_in = L_in; l.L_out = L_out; l.dst = dst;
        l.Lv_out = (H != Hv) ? (int)((long)L_out * Hv / H) : 0;   // (levels halve container and horizon alike: Hv % 2^(levels-1) == 0)
        if (mode == CONV_UPT) l.w = add_param(u, wname, {c1 + c2, cout, ks}, PK_CONVT);
        else l.w = add_param(u, wname, {cout, c1 + c2, ks}, PK_CONV);
        l.b = add_param(u, bname, {cout});
        l.cin_pad = pad_cin(c1 + c2);
        const int pad = (mode == CONV_S1) ? ks / 2 : 1;
        l.rs = pick_row_stride(l.cin_pad, mode, L_in, L_out, L_in + 2 * pad);
        slot = std::max(slot, (size_t)cout * L_out);
        u->layers.push_back(l);
        return u->layers.back();
    };
    auto cblock = [&](const std::string& p, int src1, int c1, int src2, int c2, int cout, int L, int dst) -> Layer& {
        Layer& l = conv_layer(p + ".block.0.weight", p + ".block.0.bias", CONV_S1, 5, EPI_GN_MISH, src1, c1, src2, c2, cout, L, L, dst);
        const int idx = (int)u->layers.size() - 1;
        const int ga = add_param(u, p + ".block.2.weight", {cout});
        const int be = add_param(u, p + ".block.2.bias", {cout});
        Layer& ll = u->layers[idx];
        ll.gamma = ga; ll.beta = be;
        ll.gs = cout / gn_groups(cout);
        (void)l;
        return ll;
    };
    auto rtb = [&](const std::string& p, int src1, int c1, int src2, int c2, int cout, int L, int dst) {
        {
            Layer& b0 = cblock(p + ".blocks.0", src1, c1, src2, c2, cout, L, HB);
            b0.tb_off = u->tt_row;
        }
        const int i1 = (int)u->layers.size();
        cblock(p + ".blocks.1", HB, cout, SRC_NONE, 0, cout, L, dst);
        const int tw = add_param(u, p + ".cond_mlp.1.weight", {cout, te});
        const int tbp = add_param(u, p + ".cond_mlp.1.bias", {cout});
        u->tt_w.push_back(tw); u->tt_b.push_back(tbp); u->tt_cout.push_back(cout); u->tt_off.push_back(u->tt_row);
        u->tt_row += cout;
        int res = src1;
        if (c1 + c2 != cout) {
            // residual 1x1 conv runs BEFORE blocks.1 in launch order: insert it ahead of that layer
            Layer keep = u->layers[i1];
            u->layers.pop_back();
            conv_layer(p + ".residual_conv.weight", p + ".residual_conv.bias", CONV_S1, 1, EPI_BIAS, src1, c1, src2, c2, cout, L, L, RB);
            u->layers.push_back(keep);
            res = RB;
        }
        u->layers.back().res = res;
    };

    int L = H, cur = SRC_X, curC = D;
    for (int i = 0; i < nl; ++i) {
        const int co = dims[i + 1];
        const std::string p = "downs." + std::to_string(i);
        const int a = other(cur);
        rtb(p + ".0", cur, curC, SRC_NONE, 0, co, L, a);
        rtb(p + ".1", a, co, SRC_NONE, 0, co, L, S0 + i);
        cur = S0 + i; curC = co;
        if (i < nl - 1) {
            conv_layer(p + ".4.conv.weight", p + ".4.conv.bias", CONV_DOWN, 3, EPI_BIAS, cur, co, SRC_NONE, 0, co, L, L / 2, P0);
            cur = P0; L /= 2;
        }
    }
    rtb("mid_block1", cur, curC, SRC_NONE, 0, curC, L, P0);
    rtb("mid_block2", P0, curC, SRC_NONE, 0, curC, L, P1);
    cur = P1;
    for (int j = 0; j < nl - 1; ++j) {
        const int lv = nl - 1 - j;            // level whose skip is popped
        const int dout = dims[lv + 1], din = dims[lv];
        const std::string p = "ups." + std::to_string(j);
        const int a = other(cur);
        rtb(p + ".0", cur, dout, S0 + lv, dout, din, L, a);
        const int b = other(a);
        rtb(p + ".1", a, din, SRC_NONE, 0, din, L, b);
        const int d = other(b);
        conv_layer(p + ".4.conv.weight", p + ".4.conv.bias", CONV_UPT, 4, EPI_BIAS, b, din, SRC_NONE, 0, din, L, 2 * L, d);
        cur = d; L *= 2; curC = din;
    }
    cblock("final_conv.0", cur, curC, SRC_NONE, 0, c.unet_input_dim, L, HB);
    u->final_slot = HB;
    add_param(u, "final_conv.1.weight", {D, c.unet_input_dim, 1});
    add_param(u, "final_conv.1.bias", {D});
    u->slot_floats = std::max(slot, (size_t)c.unet_input_dim * H);
    if (H != Hv) u->xpad_slot = u->n_slots++;   // one more workspace slot: the padded copy of the network input
}

// ------------------------------------------------------------------------------------------------ fused segments
// Try to turn layers [i0, i1) (an outer U-Net level: 2 residual blocks + resample [+ final_conv[0]]) into one
// fused_level_kernel program.  Returns false (and leaves the per-layer path) if any shape constraint fails.
// MPDX_DEBUG_FUSE=1 prints which shape constraint rejected a fused segment (dev aid)
static bool fuse_reject(int line) {
    if (getenv("MPDX_DEBUG_FUSE")) fprintf(stderr, "[mpdx] fused segment rejected at mpdx.hip:%d\n", line);
    return false;
}

struct HostBuf { int off4 = -1, rs4 = 0, rows = 0; size_t size4 = 0; int def = 1 << 30, last = -1; };

static bool build_fused_segment(mpdx_unet* u, int i0, int i1, bool with_final) {
    mpdx_unet::Fused f;
    f.first = i0; f.count = i1 - i0; f.has_final = with_final;
    FusedArgs& a = f.tmpl;
    memset(&a, 0, sizeof(a));
    const Layer& l0 = u->layers[i0];
    f.in1 = l0.src1; f.in2 = l0.src2;
    a.gc1 = l0.c1; a.gc2 = l0.c2; a.L0 = l0.L_in;
    std::vector<HostBuf> bufs;
    std::unordered_map<long, int> bufmap;  // (slot, L) -> LDS buffer
    // LDS activation buffers are placed AFTER the op list is known, by live range [first write, last read] in op indices
    // (-1 = staged by the prologue): buffers whose ranges do not intersect share addresses.
    auto new_buf = [&](int cpad, int L) {
        HostBuf hb;
        const int rs = pick_row_stride(cpad, CONV_S1, L, L, L + 4);
        hb.rs4 = rs / 4; hb.rows = L + 4; hb.size4 = (size_t)(L + 4) * (rs / 4);
        bufs.push_back(hb);
        return (int)bufs.size() - 1;
    };
    auto touch = [&](int id, int opi, bool write) {
        if (id < 0) return;
        if (write) bufs[id].def = std::min(bufs[id].def, opi);
        bufs[id].last = std::max(bufs[id].last, opi);
    };
    int in_buf = -1;
    bool in_slot_rewritten = false;   // a layer of the segment has written the workspace slot the segment's input came in
    // the LDS buffer a layer writes its output (workspace slot `slot`, L positions) to.  A slot written again with the same shape re-uses its buffer (the
    // blocks' HB / RB temporaries, an up level's second block writing the slot the level's input came in) - except a buffer too narrow for it (three-level
    // network, round 6: mid_block1's 128 channels go to the slot downs.1's Downsample1d output - the segment input, 64 channels - came in): a buffer of its own
    auto buf_for = [&](int slot, int L, int cpad) {
        const long key = (long)(slot + 8) * 4096 + L;
        auto it = bufmap.find(key);
        if (it != bufmap.end()) {
            const int rs = pick_row_stride(cpad, CONV_S1, L, L, L + 4);
            if (it->second == in_buf) in_slot_rewritten = true;   // (src_buf: from here on that slot is a tensor of the segment, no longer its input)
            if (bufs[it->second].rs4 >= rs / 4) return it->second;
        }
        const int id = new_buf(cpad, L);
        bufmap[key] = id;
        return id;
    };
    in_buf = new_buf(l0.cin_pad, l0.L_in);
    a.in_clear = (l0.cin_pad != l0.c1 + l0.c2) ? 1 : 0;  // channel padding of the staged input
    touch(in_buf, -1, true);
    bufmap[(long)(l0.src1 + 8) * 4096 + l0.L_in] = in_buf;
    int cat_buf = -1;    // buffer whose tail columns hold a skip tensor staged by the prologue (concat inside the program)
    auto src_buf = [&](const Layer& l, int i) -> int {   // LDS buffer a layer reads (-1: not available inside the segment)
        if (i == i0 || (!in_slot_rewritten && l.src1 == l0.src1 && l.src2 == l0.src2 && l.L_in == l0.L_in)) return in_buf;
        const long key = (long)(l.src1 + 8) * 4096 + l.L_in;
        if (!bufmap.count(key)) return -1;
        if (l.src2 != SRC_NONE) {   // cat(x produced in LDS, skip from global): the producer's buffer was made wide enough (below)
            if (bufmap[key] != cat_buf || f.in3 != l.src2) return -1;
        }
        return bufmap[key];
    };
    // a layer of the segment (not the first) that concatenates a global skip tensor behind a tensor produced inside
    auto cat_consumer = [&](int from, int slot, int L) -> const Layer* {
        for (int k = from; k < i1; ++k) {
            const Layer& n = u->layers[k];
            if (n.src1 == slot && n.L_in == L && n.src2 != SRC_NONE && !(n.src1 == l0.src1 && n.src2 == l0.src2)) return &n;
            if (n.dst == slot) break;   // overwritten: later readers see another tensor
        }
        return nullptr;
    };
    struct HostOp { int src = -1, rsrc = -1, res = -1, dst = -1; const Layer* l = nullptr; const Layer* r = nullptr; int nblk = 0, ncr = 0, tot = 0, nstream = 0; };
    std::vector<HostOp> hops;
    int ng = 0;
    int pending_res = -1;   // index of a residual 1x1 conv waiting to be folded into the block's blocks[1]
    for (int i = i0; i < i1; ++i) {
        const Layer& l = u->layers[i];
        // a block's residual 1x1 conv is folded into blocks[1] (the next layer, which adds its output after Mish)
        if (l.mode == CONV_S1 && l.ks == 1 && l.epi == EPI_BIAS && i + 1 < i1 && u->layers[i + 1].res == l.dst &&
            u->layers[i + 1].epi == EPI_GN_MISH && u->layers[i + 1].L_out == l.L_out && u->layers[i + 1].cout == l.cout) {
            pending_res = i;
            continue;
        }
        if (a.nops >= kMaxFusedOps - (with_final ? 1 : 0)) return fuse_reject(__LINE__);
        FusedOp& op = a.ops[a.nops];
        memset(&op, 0, sizeof(op));
        HostOp ho;
        ho.l = &l;
        const int gn = l.epi == EPI_GN_MISH ? 1 : 0;
        if (gn && (l.gs * 8 != l.cout || l.mode != CONV_S1)) return fuse_reject(__LINE__);   // the shapes assume GroupNorm(8 groups)
        ho.src = src_buf(l, i);
        if (ho.src < 0) return fuse_reject(__LINE__);
        if (pending_res >= 0) {
            ho.r = &u->layers[pending_res];
            ho.rsrc = src_buf(*ho.r, pending_res);
            if (ho.rsrc < 0) return fuse_reject(__LINE__);
            pending_res = -1;
        } else if (l.res != SRC_NONE) {
            const long key = (long)(l.res + 8) * 4096 + l.L_out;
            if (!bufmap.count(key)) return fuse_reject(__LINE__);
            ho.res = bufmap[key];
        }
        const int nc16 = l.cin_pad / 16, rnc16 = ho.r ? ho.r->cin_pad / 16 : 0;
        op.shape = fused_shape_id(l.mode, l.ks, nc16, rnc16, l.cout, l.L_out, gn);
        if (op.shape < 0) return fuse_reject(__LINE__);
        ho.nblk = nc16 * (l.mode == CONV_UPT ? 2 : l.ks); ho.ncr = rnc16; ho.tot = ho.nblk + ho.ncr;
        ho.nstream = (l.cout / 16) * (l.mode == CONV_UPT ? 2 : 1);
        const int msn = l.cout / 16, msw = std::min(msn, kFusedWaves), mp = msn / msw;   // tile rows, rows in flight, M-passes (FusedShape)
        a.msmask[a.nops] = msw - 1;
        a.slen[a.nops] = ho.tot * (l.mode == CONV_UPT ? 2 : mp);
        // destination: LDS if a later layer of the segment (or the final op) reads it; global if someone outside does
        bool read_inside = with_final && i == i1 - 1;
        for (int k = i + 1; k < i1; ++k) {
            const Layer& n = u->layers[k];
            if ((n.src1 == l.dst && n.L_in == l.L_out) || (n.res == l.dst && n.L_out == l.L_out)) read_inside = true;
            if (n.dst == l.dst) break;  // overwritten
        }
        bool read_outside = false;
        bool overwritten_inside = false;  // the slot is re-used by a later layer of this segment: this value never leaves
        for (int k = i + 1; k < i1; ++k)
            if (u->layers[k].dst == l.dst) { overwritten_inside = true; break; }
        for (size_t k = i1; k < u->layers.size() && !overwritten_inside; ++k) {
            const Layer& n = u->layers[k];
            if (n.src1 == l.dst || n.src2 == l.dst || n.res == l.dst) { read_outside = true; break; }
            if (n.dst == l.dst) break;
        }
        if (i == i1 - 1 && !with_final) read_outside = true;
        if (read_inside) {
            const Layer* cc = cat_consumer(i + 1, l.dst, l.L_out);
            if (cc) {   // this op's output is the head of a concat: make the buffer wide enough for the skip tensor behind it
                if (cat_buf >= 0 || cc->c1 != l.cout || (cc->c2 & 3) || (l.cout & 3) || (size_t)cc->L_in * (cc->c2 / 4) > 1024) {
                    if (getenv("MPDX_DEBUG_FUSE")) fprintf(stderr, "[mpdx] cat: layer %s -> %s cat_buf %d c1 %d c2 %d cout %d L %d\n", l.name.c_str(), cc->name.c_str(), cat_buf, cc->c1, cc->c2, l.cout, cc->L_in);
                    return fuse_reject(__LINE__);
                }
                ho.dst = buf_for(l.dst, l.L_out, cc->c1 + cc->c2);
                cat_buf = ho.dst;
                f.in3 = cc->src2;
                f.in3_consumer = (int)(cc - &u->layers[0]);
                a.c3 = cc->c2; a.L3 = cc->L_in; a.s3_col4 = cc->c1 / 4;
                touch(cat_buf, -1, true);   // its skip columns are written by the prologue: live from the start
            } else ho.dst = buf_for(l.dst, l.L_out, l.cout);
        } else ho.dst = -1;
        if (ho.dst >= 0 && (ho.dst == ho.src || ho.dst == ho.res || ho.dst == ho.rsrc)) return fuse_reject(__LINE__);
        op.gdst = -1;
        if (read_outside) {
            if (ng >= 3) return fuse_reject(__LINE__);
            f.gout_slot[ng] = l.dst;
            op.gdst = ng++;
        }
        touch(ho.src, a.nops, false); touch(ho.res, a.nops, false); touch(ho.rsrc, a.nops, false); touch(ho.dst, a.nops, true);
        hops.push_back(ho);
        f.op_layer.push_back(i);
        a.nops++;
    }
    if (pending_res >= 0) return fuse_reject(__LINE__);
    int final_src = -1;
    if (with_final) {
        const Layer& lf = u->layers[i1 - 1];
        FusedOp& op = a.ops[a.nops++];
        memset(&op, 0, sizeof(op));
        op.shape = kFusedShapeFinal;
        final_src = bufmap[(long)(lf.dst + 8) * 4096 + lf.L_out];
        touch(final_src, a.nops - 1, false);
        a.H = lf.L_out;
        a.Cf = u->cfg.unet_input_dim; a.D = u->cfg.state_dim;
        a.fw_off = (int)u->params[u->pidx.at("final_conv.1.weight")].off;
        a.fb_off = (int)u->params[u->pidx.at("final_conv.1.bias")].off;
    }
    size_t off4 = 0;
    {   // first-fit placement in order of definition; two buffers may share addresses iff one is dead strictly before the
        // op that first writes the other
        const int nbuf = (int)bufs.size();
        std::vector<int> order(nbuf);
        for (int i = 0; i < nbuf; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int x, int y) { return bufs[x].def < bufs[y].def; });
        for (int oi = 0; oi < nbuf; ++oi) {
            HostBuf& bi = bufs[order[oi]];
            if (bi.last < bi.def) bi.last = bi.def;
            size_t cand = 0;
            for (bool moved = true; moved;) {
                moved = false;
                for (int oj = 0; oj < oi; ++oj) {
                    const HostBuf& bj = bufs[order[oj]];
                    const bool live_overlap = !(bj.last < bi.def || bi.last < bj.def);
                    const size_t lo = (size_t)bj.off4, hi = lo + bj.size4;
                    if (live_overlap && cand < hi && lo < cand + bi.size4) { cand = hi; moved = true; }
                }
            }
            bi.off4 = (int)cand;
            off4 = std::max(off4, cand + bi.size4);
        }
    }
    a.in_off4 = bufs[in_buf].off4; a.in_rs4 = bufs[in_buf].rs4; a.in_rows = bufs[in_buf].rows;
    if (cat_buf >= 0) { a.s3_off4 = bufs[cat_buf].off4; a.s3_rs4 = bufs[cat_buf].rs4; }
    // weight streams + parameter block of the segment: a dedicated area at the end of `packed`
    size_t area = u->packed_floats;
    int poff = 0, tt_lo = 1 << 30, tt_hi = 0;
    for (size_t k = 0; k < hops.size(); ++k) {
        const HostOp& ho = hops[k];
        const Layer& l = *ho.l;
        FusedOp& op = a.ops[k];
        op.src_off4 = bufs[ho.src].off4; op.src_rs4 = bufs[ho.src].rs4;
        op.rsrc_off4 = ho.rsrc >= 0 ? bufs[ho.rsrc].off4 : 0; op.rsrc_rs4 = ho.rsrc >= 0 ? bufs[ho.rsrc].rs4 : 0;
        op.res_off4 = ho.res >= 0 ? bufs[ho.res].off4 : -1; op.res_rs4 = ho.res >= 0 ? bufs[ho.res].rs4 : 0;
        op.dst_off4 = ho.dst >= 0 ? bufs[ho.dst].off4 : -1; op.dst_rs4 = ho.dst >= 0 ? bufs[ho.dst].rs4 : 0;
        op.sbase = (int)area;
        const int MSn = l.cout / 16, nc16 = l.cin_pad / 16;
        const size_t woff = u->params[l.w].off;
        if (l.mode == CONV_UPT) {   // streams (ms, parity): slots {2 par, 2 par + 1} of every 16-channel chunk
            for (int par = 0; par < 2; ++par)
                f.jobs.push_back({woff + (size_t)par * 2 * 256, area + (size_t)par * ho.tot * 256, MSn, nc16 * 4 * 256, 2 * ho.tot * 256, nc16, 4 * 256, 2 * 256, 512});
        } else if (MSn > kFusedWaves) {   // M-passes: wave-stream s = [tile row s | tile row s + 4], each [conv blocks | folded residual blocks]
            const int mp = MSn / kFusedWaves;
            f.jobs.push_back({woff, area, mp, kFusedWaves * ho.nblk * 256, ho.tot * 256, kFusedWaves, ho.nblk * 256, mp * ho.tot * 256, ho.nblk * 256});
            if (ho.r)
                f.jobs.push_back({u->params[ho.r->w].off, area + (size_t)ho.nblk * 256, mp, kFusedWaves * ho.ncr * 256, ho.tot * 256, kFusedWaves, ho.ncr * 256,
                                  mp * ho.tot * 256, ho.ncr * 256});
        } else {
            f.jobs.push_back({woff, area, MSn, ho.nblk * 256, ho.tot * 256, 1, 0, 0, ho.nblk * 256});
            if (ho.r) f.jobs.push_back({u->params[ho.r->w].off, area + (size_t)ho.nblk * 256, MSn, ho.ncr * 256, ho.tot * 256, 1, 0, 0, ho.ncr * 256});
        }
        area += (size_t)ho.nstream * ho.tot * 256;
        op.p_off = poff;
        poff += 4 * l.cout;
        op.tb_off = l.tb_off;   // made relative to the staged slice below
        if (l.tb_off >= 0) { tt_lo = std::min(tt_lo, l.tb_off); tt_hi = std::max(tt_hi, l.tb_off + l.cout); }
    }
    if (with_final) { a.ops[a.nops - 1].src_off4 = bufs[final_src].off4; a.ops[a.nops - 1].src_rs4 = bufs[final_src].rs4; }
    area += (size_t)kFusedRing * 256;   // the ring request of the last stream may read up to 16 blocks past its end
    if (with_final) {   // final_conv[1]: rows padded to Cf + 4 floats (bank spread), the bias behind them
        if (a.H * a.D > kFinalPre * kFusedThreads || (a.Cf & 3)) return fuse_reject(__LINE__);
        a.fpar_off = poff;
        poff += (a.D * (a.Cf + 4) + a.D + 3) / 4 * 4;
    }
    a.gpar_off = (int)area; a.par_floats = poff;
    for (size_t k = 0; k < hops.size(); ++k) {
        const HostOp& ho = hops[k];
        const Layer& l = *ho.l;
        const size_t pb = area + a.ops[k].p_off;
        f.jobs.push_back({u->params[l.b].off, pb, 1, 0, 0, 1, 0, 0, l.cout});
        if (l.gamma >= 0) f.jobs.push_back({u->params[l.gamma].off, pb + l.cout, 1, 0, 0, 1, 0, 0, l.cout});
        if (l.beta >= 0) f.jobs.push_back({u->params[l.beta].off, pb + 2 * (size_t)l.cout, 1, 0, 0, 1, 0, 0, l.cout});
        if (ho.r) f.jobs.push_back({u->params[ho.r->b].off, pb + 3 * (size_t)l.cout, 1, 0, 0, 1, 0, 0, l.cout});
    }
    if (with_final) {
        f.jobs.push_back({(size_t)a.fw_off, area + a.fpar_off, a.D, a.Cf, a.Cf + 4, 1, 0, 0, a.Cf});
        f.jobs.push_back({(size_t)a.fb_off, area + a.fpar_off + (size_t)a.D * (a.Cf + 4), 1, 0, 0, 1, 0, 0, a.D});
    }
    area += poff;
    if (tt_hi > 0) {
        a.tt_lo = tt_lo; a.tt_n = (tt_hi - tt_lo + 3) / 4 * 4;
        for (int k = 0; k < (int)hops.size(); ++k)
            if (a.ops[k].tb_off >= 0) a.ops[k].tb_off -= tt_lo;
    }
    // behind the activation buffers: GroupNorm exchange | time-table slice | parameter block.  The block whose size depends on the state
    // dimension (final_conv[1]'s weights) comes LAST, so that every other LDS offset of a program is the same for every state_dim
    // (fused_geom.hpp holds them as compile-time constants)
    a.stat_off = (int)off4 * 4;     // GroupNorm exchange: 8 tiles x 4 rows x (mean, M2)
    off4 += 16;
    a.tt_off = (int)off4 * 4;
    off4 += (size_t)a.tt_n / 4;
    a.par_off = (int)off4 * 4;
    off4 += (size_t)(poff + 3) / 4;

    if ((size_t)(poff / 4 + a.tt_n / 4) > 2048) return fuse_reject(__LINE__);   // prologue: 2048 float4 of parameters per workgroup
    {
        const int c4n = (a.gc1 + a.gc2 + 3) / 4;
        int l4 = 0;
        while ((1 << l4) < c4n) ++l4;
        a.lg_c4n = ((1 << l4) == c4n) ? l4 : -1;
        if ((size_t)a.L0 * c4n > 2048) return fuse_reject(__LINE__);   // prologue holds the input window in registers (2048 float4)
    }
    f.lds_bytes = off4 * 16;
    if (f.lds_bytes > 160 * 1024) return fuse_reject(__LINE__);
    u->packed_floats = area;
    {   // a known op sequence runs as a static program
        auto matches = [&](const int* ids, int n) {
            if (n != a.nops) return false;
            for (int k = 0; k < n; ++k) if (a.ops[k].shape != ids[k]) return false;
            return true;
        };
        static const bool off = getenv("MPDX_STATIC_PROGRAMS") && atoi(getenv("MPDX_STATIC_PROGRAMS")) == 0;
        if (!off) {
            if (matches(FusedSeqDown::ids, FusedSeqDown::N)) f.program = 0;
            else if (matches(FusedSeqUpA::ids, FusedSeqUpA::N)) f.program = 1;
            else if (matches(FusedSeqUpB::ids, FusedSeqUpB::N)) f.program = 2;
            else if (matches(FusedSeqUpAB::ids, FusedSeqUpAB::N)) f.program = 3;
            else if (matches(FusedSeqMid2::ids, FusedSeqMid2::N)) f.program = 4;
            else if (matches(FusedSeqDown3::ids, FusedSeqDown3::N)) f.program = 5;
            else if (matches(FusedSeqMid3::ids, FusedSeqMid3::N)) f.program = 6;
            // the static programs with a geometry table read their LDS layout as compile-time constants (fused_geom.hpp): the layout computed
            // above must BE that table, otherwise the segment runs on the generic op-list kernel (runtime descriptors)
            const int sdim = u->cfg.state_dim;
            if ((f.program == 0 && !fused_geom_matches(a, GeomDown::g, sdim)) || (f.program == 3 && !fused_geom_matches(a, GeomUpAB::g, sdim)) ||
                (f.program == 5 && !fused_geom_matches(a, GeomDown3::g, sdim)) || (f.program == 6 && !fused_geom_matches(a, GeomMid3::g, sdim))) {
                if (getenv("MPDX_DEBUG_FUSE")) fprintf(stderr, "[mpdx] fused segment: geometry differs from the table of program %d -> generic kernel\n", f.program);
                f.program = -1;
            }
        }
    }
    if (getenv("MPDX_DEBUG_FUSE") && atoi(getenv("MPDX_DEBUG_FUSE")) >= 2) {   // dev: the segment's LDS geometry as a fused_geom.hpp initialiser
        fprintf(stderr, "// program %d: layers [%d,%d) %s..%s, LDS %zu B\n{ %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, {\n", f.program, i0, i1,
                u->layers[i0].name.c_str(), u->layers[i1 - 1].name.c_str(), f.lds_bytes, a.nops, a.in_off4, a.in_rs4, a.in_rows, a.L0,
                (a.gc1 == u->cfg.state_dim && a.gc2 == 0) ? -1 : a.gc1, a.gc2, a.c3, a.L3, a.s3_off4, a.s3_rs4, a.s3_col4, a.stat_off, a.par_off, with_final ? -1 : a.par_floats, a.tt_off,
                a.tt_n, a.fpar_off, with_final ? a.H : 0, with_final ? a.Cf : 0);
        for (int k = 0; k < a.nops; ++k) {
            const FusedOp& o = a.ops[k];
            fprintf(stderr, "    {%d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d},\n", o.shape, o.src_off4, o.src_rs4, o.rsrc_off4, o.rsrc_rs4, o.res_off4, o.res_rs4,
                    o.dst_off4, o.dst_rs4, o.gdst, o.p_off, o.tb_off);
        }
        fprintf(stderr, "}},\n");
    }
    u->fused.push_back(f);
    if (getenv("MPDX_DEBUG_FUSE"))
        fprintf(stderr, "[mpdx] fused segment %zu: layers [%d,%d) %s..%s  %d ops  %zu buffers  LDS %zu B  streams+params %zu floats  program %d\n",
                u->fused.size() - 1, i0, i1, u->layers[i0].name.c_str(), u->layers[i1 - 1].name.c_str(), a.nops, bufs.size(), f.lds_bytes,
                area - (size_t)a.ops[0].sbase, u->fused.back().program);
    return true;
}

static void build_units(mpdx_unet* u) {
    const int nl = u->cfg.n_levels;
    const int n = (int)u->layers.size();
    auto range_of = [&](const std::string& prefix, int& i0, int& i1) {
        i0 = -1; i1 = -1;
        for (int i = 0; i < n; ++i)
            if (u->layers[i].name.compare(0, prefix.size(), prefix) == 0) { if (i0 < 0) i0 = i; i1 = i + 1; }
        return i0 >= 0;
    };
    std::vector<int> owner(n, -1);
    if (u->masked()) { u->owner = owner; return; }   // a horizon in a zero-padded container: every layer as its own (masking) launch
    auto try_seg = [&](const std::string& prefix, bool with_final) {
        int i0, i1;
        if (!range_of(prefix, i0, i1)) return;
        if (with_final) {
            if (i1 != n - 1 || u->layers[n - 1].name.compare(0, 12, "final_conv.0") != 0) return;
            i1 = n;
        }
        for (int i = i0; i < i1; ++i) if (owner[i] >= 0) return;
        if (build_fused_segment(u, i0, i1, with_final))
            for (int i = i0; i < i1; ++i) owner[i] = (int)u->fused.size() - 1;
    };
    // the outer down levels as ONE program if it fits (every launch boundary + prologue removed is ~5 us per step): with four
    // levels downs.0 + downs.1 + downs.2 (15 ops; measured cfg 2 23.10 -> 22.47 ms, cfg 5 shard 624 -> 617 ms against two programs;
    // MPDX_MERGE_DOWN3=0 keeps them apart), else downs.0 + downs.1
    bool merged_down = false;
    if (nl >= 4 && !getenv("MPDX_NO_MERGE") && !(getenv("MPDX_MERGE_DOWN3") && atoi(getenv("MPDX_MERGE_DOWN3")) == 0)) {
        int a0, a1, b0, b1, c0, c1;
        if (range_of("downs.0.", a0, a1) && range_of("downs.1.", b0, b1) && range_of("downs.2.", c0, c1) && a1 == b0 && b1 == c0 &&
            build_fused_segment(u, a0, c1, false)) {
            for (int i = a0; i < c1; ++i) owner[i] = (int)u->fused.size() - 1;
            merged_down = true;
        }
    }
    if (!merged_down && nl >= 3 && !getenv("MPDX_NO_MERGE")) {
        int a0, a1, b0, b1;
        if (range_of("downs.0.", a0, a1) && range_of("downs.1.", b0, b1) && a1 == b0 && build_fused_segment(u, a0, b1, false)) {
            for (int i = a0; i < b1; ++i) owner[i] = (int)u->fused.size() - 1;
            merged_down = true;
        }
    }
    if (!merged_down) {
        try_seg("downs.0.", false);
        if (nl >= 3) try_seg("downs.1.", false);
    }
    // the third down level (C = 128, L = 16: two tile rows per wave) as its own program
    if (nl >= 4 && !getenv("MPDX_NO_MID2")) try_seg("downs.2.", false);
    // three levels: the innermost level (no Downsample1d) and the two middle blocks - eight Conv1dBlocks of 128 channels on L / 4 positions - as ONE
    // program (round 6; they were nine launches of ~4.8 us: a training iteration at batch 32 spent 43 us there).  MPDX_NO_MID3=1: per layer as before
    if (nl == 3 && !getenv("MPDX_NO_MID3")) {
        int a0, a1, b0, b1, c0, c1;
        bool free_ = range_of("downs.2.", a0, a1) && range_of("mid_block1.", b0, b1) && range_of("mid_block2.", c0, c1) && a1 == b0 && b1 == c0;
        for (int i = a0; free_ && i < c1; ++i) free_ = owner[i] < 0;
        if (free_ && build_fused_segment(u, a0, c1, false))
            for (int i = a0; i < c1; ++i) owner[i] = (int)u->fused.size() - 1;
    }
    // the two outer up levels + final_conv + DDPM step as ONE program (the second level's skip tensor is staged by the prologue)
    bool merged_up = false;
    if (nl >= 3 && !getenv("MPDX_NO_MERGE_UP") && !getenv("MPDX_NO_MERGE")) {
        int a0, a1, b0, b1;
        if (range_of("ups." + std::to_string(nl - 3) + ".", a0, a1) && range_of("ups." + std::to_string(nl - 2) + ".", b0, b1) && a1 == b0 &&
            b1 == n - 1 && u->layers[n - 1].name.compare(0, 12, "final_conv.0") == 0 && build_fused_segment(u, a0, n, true)) {
            for (int i = a0; i < n; ++i) owner[i] = (int)u->fused.size() - 1;
            merged_up = true;
        }
    }
    if (!merged_up) {
        if (nl >= 3) try_seg("ups." + std::to_string(nl - 3) + ".", false);
        try_seg("ups." + std::to_string(nl - 2) + ".", true);
    }
    u->owner = owner;
}


// tile choice.  Measured on MI355X at B=100 (tools/ablate_layers.py): per-launch time is dominated by fixed costs
// (launch boundary ~3 us, epilogue ~1.9 us), halving the tile to co-schedule two workgroups per CU does NOT pay
// (the MFMA phase gets slower: every output column re-streams the weights), so: the largest tile that still
// gives >= `target` workgroups (default 160 of the 256 CUs), growing with the batch for weight reuse.
// MPDX_TILE=MTxNT / MPDX_TARGET_WGS override (development).
void choose_tile(const Layer& l, int B, int& MT, int& NT) {
    const int min_mt = (l.epi == EPI_GN_MISH && l.gs > 16) ? 32 : 16;
    const int min_nt = std::max(l.mode == CONV_UPT ? 32 : 16, l.L_out);
    const long npos = (long)B * l.L_out;
    static const char* ov = getenv("MPDX_TILE");
    if (ov) {
        int mt = 0, nt = 0;
        if (sscanf(ov, "%dx%d", &mt, &nt) == 2 && mt >= min_mt && nt >= min_nt && l.cout % mt == 0 && nt % l.L_out == 0) { MT = mt; NT = nt; return; }
    }
    if (min_nt > 64) {   // a level with more than 64 positions (n_support_points = 128): one trajectory per tile
        NT = min_nt;
        MT = (min_mt <= 16 && l.cout % 16 == 0) ? 16 : 32;
        return;
    }
    static const int target_env = getenv("MPDX_TARGET_WGS") ? atoi(getenv("MPDX_TARGET_WGS")) : 160;
    const int target = std::max(1, target_env / g_plan_chains);   // concurrent sub-batch chains of a plan share the CUs
    auto wgs = [&](int mt, int nt) { return (long)(l.cout / mt) * ((npos + nt - 1) / nt); };
    const int pad = (l.mode == CONV_S1) ? l.ks / 2 : 1;
    auto lds = [&](int mt, int nt) {  // max(staged windows, K-partial buffer), as conv_block_lds_bytes
        const size_t stage = (size_t)(nt / l.L_out) * (l.L_in + 2 * pad) * l.rs * sizeof(float);
        const size_t red = (size_t)8 * nt * (mt + 4) * sizeof(float);
        return std::max(stage, red);
    };
    const int mts[2] = {32, 16}, nts[3] = {64, 32, 16};
    // largest tile that fits LDS (<= 96 KiB so that a second workgroup can co-reside; MPDX_LDS_CAP_KB overrides) and still
    // yields >= target workgroups; else the smallest legal tile
    static const size_t cap = (size_t)(getenv("MPDX_LDS_CAP_KB") ? atoi(getenv("MPDX_LDS_CAP_KB")) : 96) * 1024;
    for (int nt : nts)
        for (int mt : mts) {
            if (mt < min_mt || nt < min_nt || l.cout % mt || lds(mt, nt) > cap) continue;
            if (wgs(mt, nt) >= target) { MT = mt; NT = nt; return; }
        }
    MT = (min_mt <= 16 && l.cout % 16 == 0) ? 16 : 32;
    NT = min_nt;
}

static int layer_ntap(const Layer& l) { return l.mode == CONV_UPT ? 2 : l.ks; }
bool layer_ksplit(const Layer& l) {
    if (!(l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_GN_MISH)) return true;
    static const int forced = getenv("MPDX_KSPLIT") ? atoi(getenv("MPDX_KSPLIT")) : -1;   // dev: 0 = (NT/16) x (8/(NT/16)) waves, 1 = 1 x 8
    if (forced >= 0) return forced != 0;
    return (l.cin_pad / 16) * layer_ntap(l) >= 16;  // enough K to feed 8 K-split waves
}
static double layer_flops(const Layer& l, int B) {
    return 2.0 * l.cout * (double)B * l.L_out * (l.c1 + l.c2) * layer_ntap(l);
}


static int make_conv_args(const mpdx_unet* u, const Layer& l, const float* packed, const float* tt_row, const float* x, float* ws, int B,
                          int dbg, ConvArgs& a) {
    const size_t slot = u->slot_floats * (size_t)B;
    auto src = [&](int s) -> const float* { return s == SRC_X ? x : (s == SRC_NONE ? nullptr : ws + slot * s); };
    memset(&a, 0, sizeof(a));
    a.src1 = src(l.src1); a.src2 = src(l.src2);
    a.c1 = l.c1; a.c2 = l.c2;
    a.wp = packed + u->params[l.w].off;
    a.bias = packed + u->params[l.b].off;
    a.gamma = l.gamma >= 0 ? packed + u->params[l.gamma].off : nullptr;
    a.beta = l.beta >= 0 ? packed + u->params[l.beta].off : nullptr;
    a.tbias = (l.tb_off >= 0 && tt_row) ? tt_row + l.tb_off : nullptr;
    a.res = src(l.res);
    a.dst = ws + slot * l.dst;
    a.B = B; a.L_in = l.L_in; a.L_out = l.L_out; a.C_out = l.cout;
    a.cin_pad = l.cin_pad; a.rs = l.rs; a.gs = l.gs; a.dbg = dbg;
    a.Lv_out = l.Lv_out;
    a.trace = g_conv_trace;
    auto lg2 = [](int v) { int k = 0; while ((1 << k) < v) ++k; return k; };
    a.lg_c4n = lg2(l.cin_pad / 4); a.lg_Lin = lg2(l.L_in); a.lg_Lout = lg2(l.L_out); a.lg_gs = l.gs > 0 ? lg2(l.gs) : 0;
    if ((1 << a.lg_c4n) != l.cin_pad / 4 || (1 << a.lg_Lin) != l.L_in || (1 << a.lg_Lout) != l.L_out || (l.gs > 0 && (1 << a.lg_gs) != l.gs))
        return fail(MPDX_E_INVALID, "layer %s: channel/length/group sizes must be powers of two", l.name.c_str());
    return 0;
}

// Large batches: the Conv1dBlocks of the inner levels (L = 8) run on the weight-stationary persistent kernels (conv_ws.hpp;
// bit-identical outputs).  From 8 position tiles per workgroup on (B >= 512 at L = 8); MPDX_WS=0 switches them off (read per call: A/B
// runs and the bit-identity test flip it inside one process).  Returns the variant: 0 none, 1 <16,32> (256 -> 256); with a paired
// residual 1x1 conv l2: 3 <32,16,R1> (512 -> 128 on a channel concat).  Measured and NOT used (rocprofv3, B = 6400, us per launch,
// weight-stationary vs per-layer kernels): 128 -> 128 <8,16>: 142.7 vs 95.9; 128 -> 256 + 1x1 <8,32,R1>: 268 vs 234 - with 5 k-groups per
// wave a tile's 20-40 MFMAs per wave do not cover its barrier and window hand-over; kept: 256 -> 256: 319 vs 332, 512 -> 128 + 1x1: 387 vs 468.
constexpr int kWsnMinB = 512, kWspMinB = 512;   // (set from the sweep)
static int weight_stationary_variant(const Layer& l, const Layer* l2, const ConvArgs& a, int B, int dbg) {
    const char* wsn = getenv("MPDX_WSN");   // dev A/B: 0 = the 128-channel layers stay on the per-layer kernels
    // conv_wsn / conv_wsp load a wave's WHOLE weight slice in their prologue (40-48 KB per wave, ~10 us per launch): they pay from a few tiles per
    // wave on - batch thresholds from tools/wsn_threshold_sweep.py (profiles/r05_wsn_threshold_sweep.txt); MPDX_WSN_MIN_B / MPDX_WSP_MIN_B override
    static const int wsn_min_b = getenv("MPDX_WSN_MIN_B") ? atoi(getenv("MPDX_WSN_MIN_B")) : kWsnMinB;
    static const int wsp_min_b = getenv("MPDX_WSP_MIN_B") ? atoi(getenv("MPDX_WSP_MIN_B")) : kWspMinB;
    const bool ws_env_on = !(getenv("MPDX_WS") && atoi(getenv("MPDX_WS")) == 0);
    const bool wsn_on = !(wsn && atoi(wsn) == 0) && ws_env_on && B >= wsn_min_b;
    const bool wsp_on = ws_env_on && B >= wsp_min_b && !(getenv("MPDX_WSP") && atoi(getenv("MPDX_WSP")) == 0);
    // Upsample1d(128) of the innermost up level, 8 -> 16 positions: conv_wsn_kernel<CONV_UPT> (round 5)
    if (l.mode == CONV_UPT && l.ks == 4 && l.epi == EPI_BIAS && !l2 && l.L_in == 8 && l.L_out == 16 && !l.Lv_out && l.c1 == 128 && l.c2 == 0 &&
        l.cin_pad == 128 && l.cout == 128 && !dbg && !a.pre && !a.accum && !a.dst2 && wsn_on && (long)B * 8 >= 16L * kWsGroups * 8)
        return 5;
    if (!(l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_GN_MISH && l.L_out == 8 && l.L_in == 8) || l.Lv_out) return 0;
    if (dbg || a.pre || (l.c1 & 3) || (l.c2 & 3) || l.cin_pad != l.c1 + l.c2) return 0;
    if ((long)B * l.L_out < 16L * kWsGroups * 8) return 0;
    const char* e = getenv("MPDX_WS");
    if (e && atoi(e) == 0) return 0;
    if (l2 && e && atoi(e) == 2) return 0;   // dev A/B: 2 = single layers only
    if (l2) {
        if (!(l2->mode == CONV_S1 && l2->ks == 1 && l2->epi == EPI_BIAS && l2->cout == l.cout && l2->L_out == 8 && l2->c1 == l.c1 && l2->c2 == l.c2)) return 0;
        if (l.cout == 128 && l.gs == 16 && l.cin_pad == 512) return 3;
        if (l.cout == 256 && l.gs == 32 && l.cin_pad == 128 && l.c2 == 0 && wsp_on && !a.res)
            return 6;   // conv_wsp_kernel: 128 -> 256 k5 + 1x1, a pair of waves per tile, whole K per wave (round 5)
        return 0;
    }
    if (l.cout == 256 && l.gs == 32 && l.cin_pad == 256) return 1;
    if (l.cout == 128 && l.gs == 16 && l.cin_pad == 128 && l.c2 == 0 && wsn_on) return 4;   // conv_wsn_kernel<CONV_S1>: no K split (round 5)
    return 0;
}


static int run_layer(const mpdx_unet* u, const Layer& l, const float* packed, const float* tt_row, const float* x,
                     float* ws, int B, hipStream_t st, int dbg = 0) {
    ConvArgs a;
    if (int rc = make_conv_args(u, l, packed, tt_row, x, ws, B, dbg, a)) return rc;
    if (const int v = weight_stationary_variant(l, nullptr, a, B, dbg)) return launch_weight_stationary(v, l, a, a, B, st);
    return launch_conv_layer(l, a, B, st);
}

// blocks[0] + residual 1x1 conv of one ResidualTemporalBlock qualify for ONE launch (conv_pair_kernel)?  On success the tile.
bool pair_tile(const Layer& l1, const Layer& l2, int B, int& MT, int& NT) {
    static const bool off = getenv("MPDX_PAIR") && atoi(getenv("MPDX_PAIR")) == 0;
    if (off) return false;
    if (!(l1.mode == CONV_S1 && l1.ks == 5 && l1.epi == EPI_GN_MISH && l2.mode == CONV_S1 && l2.ks == 1 && l2.epi == EPI_BIAS)) return false;
    if ((l1.gs * l1.L_out != 128 && l1.gs * l1.L_out != 256) || l1.Lv_out) return false;   // general / masked GroupNorm regions: no paired instantiations
    if (l1.src1 != l2.src1 || l1.src2 != l2.src2 || l1.cout != l2.cout || l1.L_out != l2.L_out || !layer_ksplit(l1)) return false;
    choose_tile(l1, B, MT, NT);   // the k5 block decides the tile; the 1x1 conv has no constraint beyond it
    if (l1.cout % MT) MT = 16;
    if (l1.cout % MT || NT % l1.L_out) return false;
    const size_t lds = std::max({(size_t)(NT / l1.L_out) * (l1.L_in + 4) * l1.rs * sizeof(float), (size_t)(NT / l2.L_out) * l2.L_in * l2.rs * sizeof(float),
                                 (size_t)8 * NT * (MT + 4) * sizeof(float)});   // as launch_pair computes it
    return lds <= 160 * 1024;
}

static int run_pair(const mpdx_unet* u, const Layer& l1, const Layer& l2, const float* packed, const float* tt_row, const float* x, float* ws,
                    int B, hipStream_t st) {
    int MT, NT;
    if (!pair_tile(l1, l2, B, MT, NT)) return fail(MPDX_E_STATE, "layers %s / %s do not pair", l1.name.c_str(), l2.name.c_str());
    ConvArgs a1, a2;
    if (int rc = make_conv_args(u, l1, packed, tt_row, x, ws, B, 0, a1)) return rc;
    if (int rc = make_conv_args(u, l2, packed, tt_row, x, ws, B, 0, a2)) return rc;
    if (const int v = weight_stationary_variant(l1, &l2, a1, B, 0)) return launch_weight_stationary(v, l1, a1, a2, B, st);
    a1.n_tiles_n = a2.n_tiles_n = (int)(((long)B * l1.L_out + NT - 1) / NT);
    return launch_conv_pair(MT, NT, a1, a2, l1, l2, st) == 1 ? 0 : fail(MPDX_E_INVALID, "pair launch failed (tile %dx%d)", MT, NT);
}

int check_ready(const mpdx_unet* u) {
    if (u->n_done != (int)u->params.size())
        return fail(MPDX_E_STATE, "%d of %zu parameters packed; call mpdx_unet_pack_param for every state-dict tensor first",
                    u->n_done, u->params.size());
    return 0;
}

// bit k enables fused segment k.  Default: all segments at every batch size.  A fused program streams the segment's weights once
// per TRAJECTORY (from the L2, warm within a launch) where the per-layer kernels stream them once per tile of 4-8 trajectories and
// round-trip every activation through HBM; with the static programs (136-170 VGPRs: two workgroups per CU) the fused path wins
// everywhere.  Measured on MI355X: U-Net pass D=14, round-2 generic kernel: B=800 1.305 vs 1.337 ms (fused vs per-layer), 1600: 2.04 vs
// 2.16, 3200: 3.52 vs 3.57, 6400 equal; static programs at B=6400 (cfg5 plan): 644 vs 726 ms.  (Round 1's kernel crossed over at
// B~600.)  MPDX_FUSED=0/1 forces none/all, MPDX_FUSED_MASK=<bits> selects segments.
unsigned fused_mask(int B) {
    (void)B;
    // read on every call (two getenv per pass): tests and A/B runs switch the path inside one process
    const char* f = getenv("MPDX_FUSED");
    const char* m = getenv("MPDX_FUSED_MASK");
    if (f && atoi(f) == 0) return 0u;
    if (m) return (unsigned)strtoul(m, nullptr, 0);
    return ~0u;
}
// launch units for batch B
static std::vector<mpdx_unet::Unit> current_units(const mpdx_unet* u, int B, bool* final_in_fused) {
    std::vector<mpdx_unet::Unit> out;
    const unsigned m = fused_mask(B);
    bool fin = false;
    const int nlay = (int)u->layers.size();
    auto fused_on = [&](int i) { const int o = u->owner[i]; return o >= 0 && ((m >> o) & 1u); };
    for (int i = 0; i < nlay; ++i) {
        const int o = u->owner[i];
        if (fused_on(i)) {
            if (i == u->fused[o].first) { out.push_back({o, i, false}); fin |= u->fused[o].has_final; }
            continue;
        }
        int MT, NT;
        if (i + 1 < nlay && !fused_on(i + 1) && pair_tile(u->layers[i], u->layers[i + 1], B, MT, NT)) {
            out.push_back({-1, i, true});   // blocks[0] followed by the same block's residual 1x1 conv: one launch
            ++i;
            continue;
        }
        out.push_back({-1, i, false});
    }
    if (final_in_fused) *final_in_fused = fin;
    return out;
}

static int run_final(mpdx_unet* u, const float* packed, FinalArgs& fa, int B, float* ws, hipStream_t st) {
    const mpdx_unet_cfg& c = u->cfg;
    fa.h = ws + u->slot_floats * (size_t)B * u->final_slot;
    fa.w = packed + u->params[u->pidx.at("final_conv.1.weight")].off;
    fa.bias = packed + u->params[u->pidx.at("final_conv.1.bias")].off;
    fa.B = B; fa.H = c.n_support_points; fa.D = c.state_dim; fa.C = c.unet_input_dim;
    fa.Hc = u->Hc;
    return launch_final_step(fa, st);
}
int launch_final_step(const FinalArgs& fa, hipStream_t st) {
    const int n = fa.B * fa.H;
    const size_t lds = (size_t)(fa.D * fa.C + fa.D) * sizeof(float);
    hipLaunchKernelGGL(final_step_kernel, dim3((n + 255) / 256), dim3(256), lds, st, fa);
    return 0;
}

static long long* g_fused_trace = nullptr;  // dev tool (mpdx_fused_trace)
static int g_fused_trace_seg = -1;

// strided copy inside `packed`: dst[i0*ds0 + i1*ds1 + k] = src[i0*ss0 + i1*ss1 + k]
__global__ void restream_kernel(float* __restrict__ packed, size_t src, size_t dst, int n0, int ss0, int ds0, int n1, int ss1, int ds1, int n_inner) {
    const size_t total = (size_t)n0 * n1 * n_inner;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % n_inner);
        const size_t r = i / n_inner;
        const int i1 = (int)(r % n1), i0 = (int)(r / n1);
        packed[dst + (size_t)i0 * ds0 + (size_t)i1 * ds1 + k] = packed[src + (size_t)i0 * ss0 + (size_t)i1 * ss1 + k];
    }
}

// all strided copies of every fused segment in ONE launch (blockIdx.y = job; the table lives in device memory; CopyJobDev, restream_job: train_types.hpp)
__global__ __launch_bounds__(256) void restream_all_kernel(float* __restrict__ packed, const CopyJobDev* __restrict__ jobs) {
    restream_job(packed, jobs[blockIdx.y], blockIdx.x * 256u + threadIdx.x, gridDim.x * 256u);
}

static int ensure_stream_jobs(mpdx_unet* u) {
    if (!u->jobs_dev) {   // the job table never changes after build_units
        std::vector<CopyJobDev> all;
        for (const auto& f : u->fused)
            for (const auto& j : f.jobs) all.push_back({(unsigned long long)j.src, (unsigned long long)j.dst, j.n0, j.ss0, j.ds0, j.n1, j.ss1, j.ds1, j.n_inner});
        u->n_jobs = (int)all.size();
        if (u->n_jobs) {
            HIP_TRY(hipMalloc(&u->jobs_dev, all.size() * sizeof(CopyJobDev)));
            HIP_TRY(hipMemcpy(u->jobs_dev, all.data(), all.size() * sizeof(CopyJobDev), hipMemcpyHostToDevice));
        }
    }
    return 0;
}
// The fused segments read stream-ordered copies of their weights and one contiguous parameter block (fused_level.hpp);
// (re)assemble them in `packed` after the state dict was (re)packed.  Enqueues copies on `st`; no synchronisation.
int ensure_fused_streams(mpdx_unet* u, const float* packed, hipStream_t st) {
    if (u->streams_for == packed && u->streams_version == u->pack_version) return 0;
    if (int rc = ensure_stream_jobs(u)) return rc;
    if (u->n_jobs)
        hipLaunchKernelGGL(restream_all_kernel, dim3(64, (unsigned)u->n_jobs), dim3(256), 0, st, const_cast<float*>(packed), (const CopyJobDev*)u->jobs_dev);
    HIP_TRY(hipGetLastError());
    u->streams_for = packed; u->streams_version = u->pack_version;
    return 0;
}
// training: the caller runs the copies itself, as side blocks of its next launch (time_train_fwd_kernel) - hands out the device job table (null / 0:
// the streams are current) and marks the streams of `packed` current
int claim_fused_stream_jobs(mpdx_unet* u, const float* packed, const void** jobs, int* n) {
    *jobs = nullptr; *n = 0;
    if (u->streams_for == packed && u->streams_version == u->pack_version) return 0;
    if (int rc = ensure_stream_jobs(u)) return rc;
    *jobs = u->jobs_dev; *n = u->n_jobs;
    u->streams_for = packed; u->streams_version = u->pack_version;
    return 0;
}

// programs that exist in the training-forward variant (the two of the standard 4-level network, and the generic op-list kernel)
bool fused_save_variant(const mpdx_unet::Fused& f) { return f.program == 0 || f.program == 3 || f.program == 5 || f.program == 6 || f.program < 0; }
static int run_fused(mpdx_unet* u, const mpdx_unet::Fused& f, const float* packed, const float* tt_row, const float* x, float* ws,
                     int B, const FinalArgs* fa, hipStream_t st) {
    const size_t slot = u->slot_floats * (size_t)B;
    auto src = [&](int s) -> const float* { return s == SRC_X ? x : (s == SRC_NONE ? nullptr : ws + slot * s); };
    if (int rc = ensure_fused_streams(u, packed, st)) return rc;
    FusedArgs a = f.tmpl;
    a.packed = packed; a.tt_row = tt_row;
    a.gsrc1 = src(f.in1); a.gsrc2 = src(f.in2);
    a.gsrc3 = f.in3 != SRC_NONE ? src(f.in3) : a.gsrc1;
    for (int k = 0; k < 3; ++k) a.gout[k] = f.gout_slot[k] >= 0 ? ws + slot * f.gout_slot[k] : nullptr;
    a.B = B;
    a.trace = (g_fused_trace && (g_fused_trace_seg < 0 || g_fused_trace_seg == (int)(&f - &u->fused[0]))) ? g_fused_trace : nullptr;
    if (f.has_final) {
        if (!fa) return fail(MPDX_E_STATE, "fused final segment needs the step arguments");
        a.x_in = fa->x_in; a.noise = fa->noise; a.hs = fa->hs; a.hg = fa->hg; a.out = fa->out; a.chain = fa->chain;
        a.absmax = fa->absmax; a.fmode = fa->mode; a.n_per_ctx = fa->n_per_ctx > 0 ? fa->n_per_ctx : B; a.k = fa->k;
        a.rng = fa->rng;
    }
    return launch_fused_args(f, a, B, st);
}


// one launch unit of the pass (the SAME function serves the planning path, the profiler and the in-situ timer)
static int run_unit(mpdx_unet* u, const mpdx_unet::Unit& un, const float* packed, const float* row, const float* x, float* ws, int B,
                    const FinalArgs* fa, hipStream_t st) {
    if (un.fused >= 0) return run_fused(u, u->fused[un.fused], packed, row, x, ws, B, fa, st);
    if (un.pair) return run_pair(u, u->layers[un.layer], u->layers[un.layer + 1], packed, row, x, ws, B, st);
    return run_layer(u, u->layers[un.layer], packed, row, x, ws, B, st);
}
static double unit_flops(const mpdx_unet* u, const mpdx_unet::Unit& un, int B) {
    if (un.fused >= 0) {
        const auto& f = u->fused[un.fused];
        double fl = 0.0;
        for (int k = f.first; k < f.first + f.count; ++k) fl += layer_flops(u->layers[k], B);
        return fl;
    }
    return layer_flops(u->layers[un.layer], B) + (un.pair ? layer_flops(u->layers[un.layer + 1], B) : 0.0);
}

// ALGORITHMIC bytes of a launch unit: every weight / parameter it needs once + the activations that cross its boundary once (inputs,
// residual, outputs; what stays in LDS inside a fused program does not count) - the denominator of bench.py's traffic_over_algorithmic
static double layer_param_bytes(const mpdx_unet* u, const Layer& l) {
    double n = (double)u->params[l.w].n + l.cout;
    if (l.gamma >= 0) n += 2.0 * l.cout;
    if (l.tb_off >= 0) n += l.cout;
    return 4.0 * n;
}
static double unit_bytes(const mpdx_unet* u, const mpdx_unet::Unit& un, int B) {
    auto act = [&](int L, int C) { return 4.0 * B * (double)L * C; };
    if (un.fused >= 0) {
        const auto& f = u->fused[un.fused];
        const Layer& l0 = u->layers[f.first];
        double b = act(l0.L_in, l0.c1 + l0.c2);
        if (f.in3_consumer >= 0) b += act(u->layers[f.in3_consumer].L_in, u->layers[f.in3_consumer].c2);
        for (int k = f.first; k < f.first + f.count; ++k) b += layer_param_bytes(u, u->layers[k]);
        for (int k = 0; k < f.tmpl.nops; ++k)
            if (f.tmpl.ops[k].shape != kFusedShapeFinal && f.tmpl.ops[k].gdst >= 0) {
                const Layer& l = u->layers[f.op_layer[k]];
                b += act(l.L_out, l.cout);
            }
        if (f.has_final) b += 3.0 * act(u->cfg.n_support_points, u->cfg.state_dim);   // x_t in, noise in, x_{t-1} out
        return b;
    }
    double b = 0.0;
    for (int k = un.layer; k < un.layer + (un.pair ? 2 : 1); ++k) {
        const Layer& l = u->layers[k];
        b += layer_param_bytes(u, l) + act(l.L_out, l.cout) + (l.res != SRC_NONE ? act(l.L_out, l.cout) : 0.0);
        if (k == un.layer) b += act(l.L_in, l.c1 + l.c2);   // a paired launch reads its input once
    }
    return b;
}

// one U-Net pass + the final 1x1 conv / DDPM step described by `fa` (fa.mode 0: eps only)
static int run_unet_and_final(mpdx_unet* u, const float* packed, const float* timetab, int T, const float* x, int t, int B,
                              float* ws, FinalArgs& fa, hipStream_t st) {
    if (int rc = check_ready(u)) return rc;
    if (t < 0 || t >= T) return fail(MPDX_E_INVALID, "timestep %d outside [0,%d)", t, T);
    if (B <= 0) return fail(MPDX_E_INVALID, "B must be positive");
    const float* row = timetab + (size_t)t * u->tt_row;
    const auto units = current_units(u, B, nullptr);
    bool final_done = false;
    if (u->masked()) {   // the network reads its input from the zero-padded container copy
        float* xc = ws + u->slot_floats * (size_t)B * u->xpad_slot;
        const size_t nx = (size_t)B * u->Hc * u->cfg.state_dim;
        hipLaunchKernelGGL(pad_input_kernel, dim3((unsigned)std::min<size_t>((nx + 255) / 256, 2048)), dim3(256), 0, st, x, xc, B, u->cfg.n_support_points, u->Hc,
                           u->cfg.state_dim);
        x = xc;
    }
    for (const auto& un : units) {
        if (int rc = run_unit(u, un, packed, row, x, ws, B, &fa, st)) return rc;
        if (un.fused >= 0) final_done |= u->fused[un.fused].has_final;
    }
    if (!final_done)
        if (int rc = run_final(u, packed, fa, B, ws, st)) return rc;
    return 0;
}


}  // namespace mpdx

using namespace mpdx;

extern "C" {

const char* mpdx_last_error(void) { return g_err; }
int mpdx_version(void) { return 1; }

int mpdx_unet_create(const mpdx_unet_cfg* cfg, mpdx_unet** out) {
    if (!cfg || !out) return fail(MPDX_E_INVALID, "null argument");
    if (cfg->n_levels < 2 || cfg->n_levels > MPDX_MAX_LEVELS) return fail(MPDX_E_INVALID, "n_levels %d unsupported", cfg->n_levels);
    if (cfg->state_dim < 1 || cfg->state_dim > 64) return fail(MPDX_E_INVALID, "state_dim %d unsupported", cfg->state_dim);
    if (cfg->time_emb_dim != 32) return fail(MPDX_E_INVALID, "time_emb_dim must be 32 (TimeEncoder(32, .), temporal_unet.py:66)");
    if (cfg->unet_input_dim % 16) return fail(MPDX_E_INVALID, "unet_input_dim must be a multiple of 16");
    // final_conv is Conv1dBlock(unet_input_dim, unet_input_dim) on the output of the last up level, which has unet_input_dim * dim_mults[0]
    // channels (temporal_unet.py:98-116): the reference's own forward fails for dim_mults[0] != 1
    if (cfg->dim_mults[0] != 1) return fail(MPDX_E_INVALID, "dim_mults[0] must be 1: final_conv takes unet_input_dim channels (temporal_unet.py:113-116)");
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->dim_mults[i] < 1) return fail(MPDX_E_INVALID, "dim_mults[%d] = %d", i, cfg->dim_mults[i]);
    const int H = cfg->n_support_points;
    // the reference's U-Net takes every horizon its stride-2 / transposed convolutions map back onto itself: H % 2^(levels - 1) == 0
    // (temporal_unet.py:24,80-103).  Powers of two run natively; the others (24, 40, 48, 96 ...) in the next power-of-two container
    // with zeroed, masked rows (ConvArgs::Lv_out), one launch per layer
    if (H < 16 || H > 128 || (H % (1 << (cfg->n_levels - 1))) || (H >> (cfg->n_levels - 1)) < 2)
        return fail(MPDX_E_INVALID, "n_support_points %d must be a multiple of 2^(levels-1) = %d in [16, 128] with >= 2 points at the coarsest level", H,
                    1 << (cfg->n_levels - 1));
    mpdx_unet* u = new mpdx_unet();
    u->cfg = *cfg;
    build_model(u);
    // GroupNorm regions (group x horizon) of 64 ... 2048 elements: one wave per region, 1, 2 or 4 x {1, 2, 4, 8} elements per lane
    // (conv_block.hpp).  The whole-trajectory fused programs exist for the shapes of H = 64 only; other horizons run one launch per layer.
    for (const Layer& l : u->layers)
        if (l.epi == EPI_GN_MISH) {
            const int re = l.gs * l.L_out;
            if (re < 64 || re > 2048 || (re & (re - 1)) || l.gs < 4 || 32 % l.gs) {
                std::string nm = l.name;
                delete u;
                return fail(MPDX_E_INVALID, "layer %s: GroupNorm region of %d elements (group of %d) unsupported", nm.c_str(), re, l.gs);
            }
        }
    if ((int)u->tt_w.size() > 40) { delete u; return fail(MPDX_E_INVALID, "too many residual blocks"); }
    build_units(u);
    u->packed_floats += 64;   // tail padding
    *out = u;
    return 0;
}

void mpdx_unet_destroy(mpdx_unet* u) {
    if (u && u->pack_descs_dev) (void)hipFree(u->pack_descs_dev);
    if (u && u->pack_chunks_dev) (void)hipFree(u->pack_chunks_dev);
    if (u && u->jobs_dev) (void)hipFree(u->jobs_dev);
    if (u && u->side.stream) { (void)hipStreamDestroy(u->side.stream); (void)hipEventDestroy(u->side.fork); (void)hipEventDestroy(u->side.join); }
    delete u;
}

int mpdx_unet_num_params(const mpdx_unet* u) { return u ? (int)u->params.size() : 0; }

int mpdx_unet_param_info(const mpdx_unet* u, int idx, const char** name, int32_t shape[3], int32_t* ndim) {
    if (!u || idx < 0 || idx >= (int)u->params.size()) return fail(MPDX_E_INVALID, "bad parameter index %d", idx);
    const Param& p = u->params[idx];
    if (name) *name = p.name.c_str();
    if (shape) { shape[0] = p.shape[0]; shape[1] = p.shape[1]; shape[2] = p.shape[2]; }
    if (ndim) *ndim = p.ndim;
    return 0;
}

size_t mpdx_unet_packed_floats(const mpdx_unet* u) { return u ? u->packed_floats : 0; }
size_t mpdx_unet_timetab_floats(const mpdx_unet* u, int T) { return u ? (size_t)T * u->tt_row : 0; }
size_t mpdx_unet_workspace_floats(const mpdx_unet* u, int B) { return u ? u->slot_floats * (size_t)B * u->n_slots : 0; }

int mpdx_unet_pack_param(mpdx_unet* u, const char* name, const float* src, size_t n, float* packed, void* stream) {
    if (!u || !name || !src || !packed) return fail(MPDX_E_INVALID, "null argument");
    auto it = u->pidx.find(name);
    if (it == u->pidx.end()) return fail(MPDX_E_NOTFOUND, "unexpected state-dict key '%s'", name);
    Param& p = u->params[it->second];
    if (n != p.n) return fail(MPDX_E_INVALID, "'%s': got %zu floats, expected %zu", name, n, p.n);
    hipStream_t st = (hipStream_t)stream;
    if (p.kind == PK_VEC) {
        hipLaunchKernelGGL(copy_kernel, dim3((unsigned)std::min<size_t>((p.n + 255) / 256, 1024)), dim3(256), 0, st, src, packed + p.off, p.n);
    } else {
        hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)std::min<size_t>((p.pn + 255) / 256, 2048)), dim3(256), 0, st, src,
                           packed + p.off, p.cout, p.cin, p.ksz, p.cin_pad, p.nslot, p.kind == PK_CONVT ? 1 : 0);
    }
    HIP_TRY(hipGetLastError());
    if (!p.done) { p.done = true; u->n_done++; }
    u->pack_version++;   // the stream-ordered copies of the fused segments are stale now
    return 0;
}

int mpdx_unet_build_timetab(mpdx_unet* u, const float* packed, const float* freqs16, int T, float* timetab, void* stream) {
    if (!u || !packed || !freqs16 || !timetab || T <= 0) return fail(MPDX_E_INVALID, "bad argument");
    if (int rc = check_ready(u)) return rc;
    TimeTabArgs a;
    memset(&a, 0, sizeof(a));
    a.packed = packed; a.freqs = freqs16; a.tab = timetab;
    a.w1 = (int)u->params[u->pidx.at("time_mlp.encoder.1.weight")].off;
    a.b1 = (int)u->params[u->pidx.at("time_mlp.encoder.1.bias")].off;
    a.w2 = (int)u->params[u->pidx.at("time_mlp.encoder.3.weight")].off;
    a.b2 = (int)u->params[u->pidx.at("time_mlp.encoder.3.bias")].off;
    a.row = u->tt_row;
    a.nblk = (int)u->tt_w.size();
    for (int i = 0; i < a.nblk; ++i) {
        a.woff[i] = (int)u->params[u->tt_w[i]].off;
        a.boff[i] = (int)u->params[u->tt_b[i]].off;
        a.cout[i] = u->tt_cout[i];
        a.toff[i] = u->tt_off[i];
    }
    hipLaunchKernelGGL(timetab_kernel, dim3(T), dim3(128), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_unet_forward(mpdx_unet* u, const float* packed, const float* timetab, int T, const float* x, int t, float* eps,
                      int B, float* ws, void* stream) {
    if (!u || !packed || !timetab || !x || !eps || !ws) return fail(MPDX_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.out = eps; fa.mode = 0; fa.n_per_ctx = 1;
    if (int rc = run_unet_and_final(u, packed, timetab, T, x, t, B, ws, fa, st)) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_ddpm_step(mpdx_unet* u, const float* packed, const float* timetab, int T, float* x_io, const float* noise,
                   const float* hard_start, const float* hard_goal, const mpdx_step_coefs* coefs, int t, int mean_only,
                   float* chain_out, uint32_t* absmax_out, int n_per_ctx, int B, float* ws, void* stream) {
    if (!u || !packed || !timetab || !x_io || !coefs || !ws) return fail(MPDX_E_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.x_in = x_io; fa.out = x_io; fa.noise = noise; fa.hs = hard_start; fa.hg = hard_goal;
    fa.chain = chain_out; fa.absmax = absmax_out; fa.n_per_ctx = n_per_ctx > 0 ? n_per_ctx : B;
    fa.mode = mean_only == 2 ? 3 : (mean_only ? 2 : 1);
    fa.k = *coefs;
    if (int rc = run_unet_and_final(u, packed, timetab, T, x_io, t, B, ws, fa, st)) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_add_noise(float* x_io, const float* noise, const float* hard_start, const float* hard_goal, float noise_scale,
                   float noise_std_extra, float* chain_out, int B, int H, int D, void* stream) {
    if (!x_io || B <= 0) return fail(MPDX_E_INVALID, "bad argument");
    const size_t n = (size_t)B * H * D;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                       x_io, noise, hard_start, hard_goal, noise_scale, noise_std_extra, chain_out, B, H, D);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_hard_conds(float* x_io, float* chain_out, int n, const int32_t* horizon_idx, const float* const* values, int B, int H, int D, void* stream) {
    if (!x_io || B <= 0 || H <= 0 || D <= 0 || n < 0 || (n && (!horizon_idx || !values))) return fail(MPDX_E_INVALID, "bad argument");
    if (n > kMaxHardConds) return fail(MPDX_E_INVALID, "at most 16 hard conditions per call");
    if (n == 0) return 0;
    HardCondArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    for (int k = 0; k < n; ++k) {
        int t = horizon_idx[k];
        if (t < 0) t += H;   // python indexing: x[:, -1, :]
        if (t < 0 || t >= H) return fail(MPDX_E_INVALID, "hard condition index out of range for this horizon");
        if (!values[k]) return fail(MPDX_E_INVALID, "null hard condition table");
        a.idx[k] = t; a.vals[k] = values[k];
    }
    const size_t per = (size_t)B * D;
    hipLaunchKernelGGL(hard_conds_kernel, dim3((unsigned)std::min<size_t>((per + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, x_io, chain_out, a, B, H, D);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_q_sample(const float* x_start, const float* noise, const long long* t_dev, const float* sqrt_alphas_cumprod_dev,
                  const float* sqrt_one_minus_alphas_cumprod_dev, const float* hard_start, const float* hard_goal, float* out, int B, int H,
                  int D, int T, void* stream) {
    if (!x_start || !noise || !t_dev || !sqrt_alphas_cumprod_dev || !sqrt_one_minus_alphas_cumprod_dev || !out || B <= 0 || T <= 0)
        return fail(MPDX_E_INVALID, "bad argument");
    const size_t n = (size_t)B * H * D;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, x_start, noise,
                       t_dev, sqrt_alphas_cumprod_dev, sqrt_one_minus_alphas_cumprod_dev, hard_start, hard_goal, out, B, H, D, T);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_weighted_loss(const float* pred, const float* targ, const float* weights_hd, const float* hard_start, const float* hard_goal,
                       int l1, float* out1, int B, int H, int D, void* stream) {
    if (!pred || !targ || !out1 || B <= 0) return fail(MPDX_E_INVALID, "bad argument");
    hipLaunchKernelGGL(weighted_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, targ, weights_hd, hard_start, hard_goal, l1, out1,
                       B, H, D);
    HIP_TRY(hipGetLastError());
    return 0;
}


// ---- sub-batch chains (MEASURED AND REJECTED, round 4; kept behind MPDX_PLAN_CHAINS=2 so that the measurement can be repeated).
// Idea: a plan of a small batch (B = 100) is a chain of ~1 600 dependent launches whose kernels cannot fill 256 CUs and whose boundaries
// (~2.3 us of every launch's rocprof duration lie outside the workgroups' own lifetimes) add up to a sixth of the step; trajectories are
// independent, so the batch is cut into two halves that run as two independent chains on two HIP streams (side stream forked from / joined
// into the caller's stream with events, no host synchronisation; the halves meet only at a guided step's guide iterations, whose
// whole-tensor range test couples them).  Results are bit-identical to the single chain (tests/test_gpu_parity.py, test_gpu_guide.py pass
// with the split on).  Measured on MI355X (profiles/r04_plan_chains.txt): cfg 2 24.8 vs 20.3 ms, cfg 3 27.3 vs 21.9, cfg 4 29.6 vs 24.1 -
// SLOWER, and not because of the host: replayed as ONE hipGraph (tools/graph_probe_plan.py) the two-chain plan takes 25.2 ms against 20.15
// for the single chain's graph.  Two half-batch kernels do not share the chip the way one full-batch kernel uses it: every launch pays its
// fixed phases (staging, K-reduction, epilogue, boundary) for half the work, and the dispatcher does not interleave the two queues finely
// enough to hide one chain's boundaries under the other's kernels.
namespace mpdx {
static int plan_side(mpdx_unet* u, mpdx_unet::PlanSide** out) {   // the handle's side stream + fork / join events (created on first use; one handle = one device)
    mpdx_unet::PlanSide& s = u->side;
    if (!s.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
    }
    *out = &s;
    return 0;
}
// trajectories of the first chain (0: one chain - the default).  MPDX_PLAN_CHAINS=2 switches the split on (development A/B).
static int plan_split_point(int B, int npc, int n_ctx) {
    static const int forced = getenv("MPDX_PLAN_CHAINS") ? atoi(getenv("MPDX_PLAN_CHAINS")) : 0;
    if (forced != 2 || B < 16) return 0;
    if (n_ctx >= 2) return (n_ctx / 2) * npc;     // whole contexts per chain: the range-test flags stay per chain
    return std::min(B - 4, ((B / 2) + 3) & ~3);   // one context: both chains publish into its one flag (atomicMax)
}
}  // namespace mpdx

int mpdx_plan(mpdx_unet* u, const float* packed, const float* timetab, int T, const mpdx_step_coefs* coefs, int n_without_noise,
              float* x, const float* noise, const float* hard_start, const float* hard_goal, float* chain, int B, float* ws,
              const mpdx_guide_params* guide, int n_guide_steps, int t_start_guide, uint32_t* guide_flags, int n_per_ctx,
              uint64_t rng_seed, uint64_t rng_offset, void* stream) {
    if (!u || !packed || !timetab || !coefs || !x || !ws || T <= 0 || n_without_noise < 0 || B <= 0)
        return fail(MPDX_E_INVALID, "bad argument");
    if (int rc = check_ready(u)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int H = u->cfg.n_support_points, D = u->cfg.state_dim;
    const size_t n = (size_t)B * H * D;
    const int npc = n_per_ctx > 0 ? n_per_ctx : B;
    const int n_ctx = (B + npc - 1) / npc;
    const int steps = T + n_without_noise;
    if (n_guide_steps < 0) return fail(MPDX_E_INVALID, "n_guide_steps %d", n_guide_steps);
    if (n_guide_steps == 0) guide = nullptr;   // range(0): the reference runs no guide iteration (sample_functions.py:74)
    const int dbg = debug_level();
    if (guide) {
        if (!guide_flags) return fail(MPDX_E_INVALID, "guide needs guide_flags");
        if (B % npc) return fail(MPDX_E_INVALID, "B=%d is not a multiple of n_per_ctx=%d", B, npc);
        HIP_TRY(hipMemsetAsync(guide_flags, 0, (size_t)steps * (n_guide_steps + 1) * n_ctx * sizeof(uint32_t), st));
    }
    // x_T with hard conditioning; chain[0]
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, st, x, (const float*)nullptr,
                       hard_start, hard_goal, 0.f, 0.f, chain, B, H, D);
    // two concurrent sub-batch chains (see plan_split_point): chain c covers trajectories [b0[c], b0[c] + nb[c])
    const int B0 = plan_split_point(B, npc, n_ctx);
    const int nch = B0 > 0 ? 2 : 1;
    mpdx_unet::PlanSide* side = nullptr;
    if (nch == 2) {
        if (int rc = plan_side(u, &side)) return rc;
        if (int rc = ensure_fused_streams(u, packed, st)) return rc;   // (a one-off re-assembly of the weight streams stays ahead of the fork)
        HIP_TRY(hipEventRecord(side->fork, st));
        HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
    }
    const int b0[2] = {0, B0}, nb[2] = {nch == 2 ? B0 : B, B - B0};
    hipStream_t sts[2] = {st, side ? side->stream : st};
    float* wss[2] = {ws, ws + u->slot_floats * (size_t)nb[0] * u->n_slots};   // two disjoint workspaces inside the caller's one
    struct ChainsGuard { int* p; ~ChainsGuard() { *p = 1; } } guard{&g_plan_chains};
    g_plan_chains = nch;
    bool forked = nch == 2;
    int k = 0;
    for (int i = T - 1; i >= -n_without_noise; --i, ++k) {
        const int t = i < 0 ? 0 : i;
        const bool guided = guide && i < t_start_guide;  // sample_functions.py:39 compares the un-clamped index
        const float* nz = (t == 0 || !noise) ? nullptr : noise + (size_t)k * n;  // noise[t == 0] = 0  (sample_functions.py:52)
        // noise == NULL: the step's draw is generated in place; iteration k uses elements [(k+1) n, (k+2) n) of the stream whose
        // first n elements are x_T (what mpdx_randn(x, n, seed, offset) wrote): the same bits a pre-generated tensor would hold
        NoiseRng rng;
        memset(&rng, 0, sizeof(rng));
        if (!noise && t != 0) { rng.on = 1; rng.seed = rng_seed; rng.offset = rng_offset; rng.elem0 = (unsigned long long)(k + 1) * n; }
        float* ch = chain ? chain + (size_t)(k + 1) * n : nullptr;
        uint32_t* fl = guided ? guide_flags + (size_t)k * (n_guide_steps + 1) * n_ctx : nullptr;
        if (nch == 2 && !forked) {   // behind a guided step's joined guide iterations: fork again
            HIP_TRY(hipEventRecord(side->fork, st));
            HIP_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
            forked = true;
        }
        for (int c = 0; c < nch; ++c) {
            const size_t eo = (size_t)b0[c] * H * D;   // element offset of the chain's first trajectory
            FinalArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.x_in = x + eo; fa.out = x + eo;
            fa.k = coefs[t];
            fa.n_per_ctx = npc;
            if (!guided) {
                fa.rng = rng;
                fa.rng.elem0 += eo;
                fa.noise = nz ? nz + eo : nullptr;
                fa.hs = hard_start ? hard_start + (size_t)b0[c] * D : nullptr; fa.hg = hard_goal ? hard_goal + (size_t)b0[c] * D : nullptr;
                fa.chain = ch ? ch + eo : nullptr; fa.mode = 1;
            } else {
                fa.mode = 2; fa.absmax = fl + b0[c] / npc;  // posterior mean + its max|.| per context
            }
            if (int rc = run_unet_and_final(u, packed, timetab, T, x + eo, t, nb[c], wss[c], fa, sts[c])) return rc;
        }
        if (guided) {
            if (nch == 2) {   // the guide iterations couple the chains (whole-tensor range test of a context): joined stream, whole batch
                HIP_TRY(hipEventRecord(side->join, side->stream));
                HIP_TRY(hipStreamWaitEvent(st, side->join, 0));
                forked = false;
            }
            for (int j = 0; j < n_guide_steps; ++j) {
                const bool last = j == n_guide_steps - 1;  // the last iteration also adds the noise term and appends to the chain
                if (int rc = launch_guide(guide, x, nullptr, hard_start, hard_goal, fl + (size_t)j * n_ctx, fl + (size_t)(j + 1) * n_ctx, npc, B, H,
                                          D, st, last ? nz : nullptr, coefs[t].noise_scale, coefs[t].noise_std_extra, last ? ch : nullptr,
                                          coefs[t].guide_scale, last ? &rng : nullptr))
                    return rc;
            }
        }
        if (dbg) {   // MPDX_DEBUG: attribute launch / execution errors to the loop iteration that caused them
            if (dbg >= 2) {
                hipError_t e = hipStreamSynchronize(st);
                if (e == hipSuccess && nch == 2) e = hipStreamSynchronize(side->stream);
                if (e != hipSuccess) return fail((int)e, "mpdx_plan: loop iteration %d (t=%d%s) faulted: %s", k, t, guided ? ", guided" : "", hipGetErrorString(e));
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail((int)e, "mpdx_plan: launch in loop iteration %d (t=%d%s) failed: %s", k, t, guided ? ", guided" : "", hipGetErrorString(e));
        }
    }
    if (nch == 2 && forked) {   // join: the caller's stream continues behind both chains
        HIP_TRY(hipEventRecord(side->join, side->stream));
        HIP_TRY(hipStreamWaitEvent(st, side->join, 0));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_unet_profile(mpdx_unet* u, const float* packed, const float* timetab, int T, const float* x, int t, int B, float* ws,
                      void* stream, int cap, float* ms_out, double* flops_out, const char** names_out, int* n_out) {
    if (!u || !packed || !timetab || !x || !ws || !ms_out || !n_out) return fail(MPDX_E_INVALID, "null argument");
    if (int rc = check_ready(u)) return rc;
    if (t < 0 || t >= T) return fail(MPDX_E_INVALID, "timestep %d outside [0,%d)", t, T);
    hipStream_t st = (hipStream_t)stream;
    bool fin_fused = false;
    const auto units = current_units(u, B, &fin_fused);
    const bool need_final = !fin_fused;
    const int nl = (int)units.size() + (need_final ? 1 : 0);
    if (cap < nl) return fail(MPDX_E_INVALID, "need room for %d launches", nl);
    static float* scratch = nullptr;  // eps sink owned by the library (measurement helper only)
    static size_t scratch_n = 0;
    const size_t need = (size_t)B * u->cfg.n_support_points * u->cfg.state_dim;
    if (scratch_n < need) { if (scratch) (void)hipFree(scratch); HIP_TRY(hipMalloc(&scratch, need * sizeof(float))); scratch_n = need; }
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.out = scratch; fa.mode = 0; fa.n_per_ctx = 1;
    std::vector<hipEvent_t> ev(2 * nl);
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    const float* row = timetab + (size_t)t * u->tt_row;
    static std::vector<std::string> fused_names;
    fused_names.resize(u->fused.size());
    int rc = 0;
    for (int i = 0; i < (int)units.size() && !rc; ++i) {
        HIP_TRY(hipEventRecord(ev[2 * i], st));
        const double fl = unit_flops(u, units[i], B);
        rc = run_unit(u, units[i], packed, row, x, ws, B, &fa, st);
        if (units[i].fused >= 0) {
            const auto& f = u->fused[units[i].fused];
            fused_names[units[i].fused] = "fused[" + u->layers[f.first].name.substr(0, u->layers[f.first].name.find(".blocks")) + "..+" +
                                          std::to_string(f.count) + (f.has_final ? " layers+final_conv.1+ddpm_step]" : " layers]");
            if (names_out) names_out[i] = fused_names[units[i].fused].c_str();
        } else if (names_out) names_out[i] = u->layers[units[i].layer].name.c_str();
        HIP_TRY(hipEventRecord(ev[2 * i + 1], st));
        if (flops_out) flops_out[i] = fl;
    }
    if (!rc && need_final) {
        HIP_TRY(hipEventRecord(ev[2 * (nl - 1)], st));
        rc = run_final(u, packed, fa, B, ws, st);
        HIP_TRY(hipEventRecord(ev[2 * (nl - 1) + 1], st));
        if (flops_out) flops_out[nl - 1] = 0.0;
        if (names_out) names_out[nl - 1] = "final_conv.1+ddpm_step";
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < nl && !rc; ++i) HIP_TRY(hipEventElapsedTime(&ms_out[i], ev[2 * i], ev[2 * i + 1]));
    for (auto& e : ev) (void)hipEventDestroy(e);
    *n_out = nl;
    return rc;
}

/* dev tool: one launch of layer `layer` with s_memtime stamps (7 per workgroup) of the first and the last workgroup */
int mpdx_layer_trace(mpdx_unet* u, const float* packed, const float* timetab, const float* x, int layer, int B, float* ws, void* stream,
                     long long* stamps32) {
#ifndef MPDX_DEV_HOOKS
    return fail(MPDX_E_STATE, "%s needs a development build of libmpdx.so (MPDX_BUILD_DEFS=-DMPDX_DEV_HOOKS MPDX_BUILD_OUT=build_ab/libmpdx_dev.so python -m mpd_public_amd.build, then MPDX_LIB=build_ab/libmpdx_dev.so): "
                "the production kernels carry no trace / ablation hooks", __func__);
#endif
    if (!u || layer < 0 || layer >= (int)u->layers.size() || !stamps32) return fail(MPDX_E_INVALID, "bad argument");
    if (int rc = check_ready(u)) return rc;
    hipStream_t st = (hipStream_t)stream;
    long long* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, 32 * sizeof(long long)));
    HIP_TRY(hipMemsetAsync(dev, 0, 32 * sizeof(long long), st));
    g_conv_trace = dev;
    int rc = run_layer(u, u->layers[layer], packed, timetab, x, ws, B, st);
    g_conv_trace = nullptr;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(stamps32, dev, 32 * sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    return rc;
}

/* dev tool: run fused segment `seg` once with per-phase s_memtime stamps of workgroup 0 / wave 0; stamps_out[n] */
int mpdx_fused_trace(mpdx_unet* u, const float* packed, const float* timetab, const float* x, int seg, int B, float* ws, void* stream,
                     long long* stamps_out, int cap, int* n_out, int* nops_out) {
#ifndef MPDX_DEV_HOOKS
    return fail(MPDX_E_STATE, "%s needs a development build of libmpdx.so (MPDX_BUILD_DEFS=-DMPDX_DEV_HOOKS MPDX_BUILD_OUT=build_ab/libmpdx_dev.so python -m mpd_public_amd.build, then MPDX_LIB=build_ab/libmpdx_dev.so): "
                "the production kernels carry no trace / ablation hooks", __func__);
#endif
    if (!u || seg < 0 || seg >= (int)u->fused.size()) return fail(MPDX_E_INVALID, "bad segment");
    if (int rc = check_ready(u)) return rc;
    hipStream_t st = (hipStream_t)stream;
    long long* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, 1024 * sizeof(long long)));
    HIP_TRY(hipMemsetAsync(dev, 0, 1024 * sizeof(long long), st));
    static float* scratch = nullptr;
    if (!scratch) HIP_TRY(hipMalloc(&scratch, (size_t)1 << 24));
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.out = scratch; fa.mode = 0; fa.n_per_ctx = 1;
    // the traced launch runs IN CONTEXT: two untraced U-Net passes, then a third pass in which only segment `seg` stamps - same
    // predecessors, cache and clock state as in production, no host synchronisation in between
    int rc = 0;
    for (int pass = 0; pass < 3 && !rc; ++pass) {
        if (pass == 2) { g_fused_trace = dev; g_fused_trace_seg = seg; }
        rc = run_unet_and_final(u, packed, timetab, 1 << 30, x, 0, B, ws, fa, st);
    }
    g_fused_trace_seg = -1;
    g_fused_trace = nullptr;
    HIP_TRY(hipStreamSynchronize(st));
    const int n = std::min(cap, 1024);   // 8 waves x 128 slots
    HIP_TRY(hipMemcpy(stamps_out, dev, n * sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    if (n_out) *n_out = n;
    if (nops_out) *nops_out = u->fused[seg].tmpl.nops;
    return rc;
}

/* in-situ timing: `reps` full U-Net passes; ONE event pair brackets launch units [unit_first, unit_last] of each pass
 * (so the bracketed kernels run in their real context - cold weights, real predecessor - and the event cost is
 * amortised over the run).  *ms_avg = average bracketed time per pass.  Synchronises. */
int mpdx_unet_time_units(mpdx_unet* u, const float* packed, const float* timetab, int T, const float* x, int t, int B, float* ws,
                         void* stream, int unit_first, int unit_last, int reps, float* ms_avg) {
    if (!u || !packed || !timetab || !x || !ws || !ms_avg || reps < 1) return fail(MPDX_E_INVALID, "bad argument");
    if (int rc = check_ready(u)) return rc;
    if (t < 0 || t >= T) return fail(MPDX_E_INVALID, "timestep %d outside [0,%d)", t, T);
    hipStream_t st = (hipStream_t)stream;
    const auto units = current_units(u, B, nullptr);
    if (unit_first < 0 || unit_last >= (int)units.size() || unit_first > unit_last) return fail(MPDX_E_INVALID, "bad unit range");
    static float* scratch = nullptr;
    static size_t scratch_n = 0;
    const size_t need = (size_t)B * u->cfg.n_support_points * u->cfg.state_dim;
    if (scratch_n < need) { if (scratch) (void)hipFree(scratch); HIP_TRY(hipMalloc(&scratch, need * sizeof(float))); scratch_n = need; }
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.out = scratch; fa.mode = 0; fa.n_per_ctx = 1;
    const float* row = timetab + (size_t)t * u->tt_row;
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    int rc = 0;
    for (int r = 0; r < reps && !rc; ++r) {
        bool final_done = false;
        for (int i = 0; i < (int)units.size() && !rc; ++i) {
            if (i == unit_first) HIP_TRY(hipEventRecord(ev[2 * r], st));
            rc = run_unit(u, units[i], packed, row, x, ws, B, &fa, st);
            if (units[i].fused >= 0) final_done |= u->fused[units[i].fused].has_final;
            if (i == unit_last) HIP_TRY(hipEventRecord(ev[2 * r + 1], st));
        }
        if (!rc && !final_done) rc = run_final(u, packed, fa, B, ws, st);
    }
    HIP_TRY(hipStreamSynchronize(st));
    double tot = 0.0;
    for (int r = 0; r < reps && !rc; ++r) { float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1])); tot += ms; }
    for (auto& e : ev) (void)hipEventDestroy(e);
    *ms_avg = (float)(tot / reps);
    return rc;
}

/* measurement helper (bench.py roofline leg, the DIFFERENTIAL form): `reps` back-to-back U-Net passes WITHOUT the launch units whose bit is set in skip_mask
 * (0: nothing skipped) between ONE HIP-event pair on the launch stream -> average ms per pass.  The cost of a launch class inside
 * the pass = (pass with everything) - (pass without the class): no event pair sits next to the measured launches (an event pair around a single
 * 35-us launch adds ~5 us of marker processing + dispatch gap that the un-instrumented stream does not have).  The skipped units' consumers read
 * whatever the workspace holds: timing only, the output is not meaningful. */
int mpdx_unet_time_without(mpdx_unet* u, const float* packed, const float* timetab, int T, const float* x, int t, int B, float* ws,
                           void* stream, unsigned long long skip_mask, int reps, float* ms_avg) {
    if (!u || !packed || !timetab || !x || !ws || !ms_avg || reps < 1) return fail(MPDX_E_INVALID, "bad argument");
    if (int rc = check_ready(u)) return rc;
    if (t < 0 || t >= T) return fail(MPDX_E_INVALID, "timestep %d outside [0,%d)", t, T);
    hipStream_t st = (hipStream_t)stream;
    const auto units = current_units(u, B, nullptr);
    if (units.size() > 64) return fail(MPDX_E_INVALID, "more than 64 launch units");
    static float* scratch = nullptr;
    static size_t scratch_n = 0;
    const size_t need = (size_t)B * u->cfg.n_support_points * u->cfg.state_dim;
    if (scratch_n < need) { if (scratch) (void)hipFree(scratch); HIP_TRY(hipMalloc(&scratch, need * sizeof(float))); scratch_n = need; }
    FinalArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.out = scratch; fa.mode = 0; fa.n_per_ctx = 1;
    const float* row = timetab + (size_t)t * u->tt_row;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = 0;
    auto one_pass = [&]() {
        bool final_done = false;
        for (int i = 0; i < (int)units.size() && !rc; ++i) {
            if (units[i].fused >= 0) final_done |= u->fused[units[i].fused].has_final;   // (a skipped program with the final op: no separate final kernel either)
            if ((skip_mask >> i) & 1ull) continue;
            rc = run_unit(u, units[i], packed, row, x, ws, B, &fa, st);
        }
        if (!rc && !final_done) rc = run_final(u, packed, fa, B, ws, st);
    };
    for (int r = 0; r < 3 && !rc; ++r) one_pass();   // warm-up (code objects, clocks)
    HIP_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps && !rc; ++r) one_pass();
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_avg = ms / (float)reps;
    return rc;
}

/* layer index of launch unit i (-1 for a fused unit / the final kernel): lets bench.py query the tile of a unit */
int mpdx_unet_unit_layer(const mpdx_unet* u, int B, int i) {
    if (!u) return -1;
    const auto units = current_units(u, B, nullptr);
    if (i < 0 || i >= (int)units.size()) return -1;
    return units[i].fused >= 0 ? -1 : units[i].layer;
}

/* which kernel runs fused segment `seg`: 0..5 = a static program (fused_program_kernel<FusedSeq...>; 0, 3, 5 read their LDS geometry from the
 * compile-time tables of fused_geom.hpp), -1 = the generic op-list kernel (runtime descriptors), -2 = no such segment */
int mpdx_unet_fused_program(const mpdx_unet* u, int seg) {
    if (!u || seg < 0 || seg >= (int)u->fused.size()) return -2;
    return u->fused[seg].program;
}

/* algorithmic bytes of launch unit i at batch B (weights once + boundary activations once); 0 for a bad index */
double mpdx_unet_unit_bytes(const mpdx_unet* u, int B, int i) {
    if (!u) return 0.0;
    const auto units = current_units(u, B, nullptr);
    if (i < 0 || i >= (int)units.size()) return 0.0;
    return unit_bytes(u, units[i], B);
}

/* 1 when launch unit i is a paired launch (blocks[0] + the block's residual 1x1 conv in one conv_pair_kernel) */
int mpdx_unet_unit_is_pair(const mpdx_unet* u, int B, int i) {
    if (!u) return 0;
    const auto units = current_units(u, B, nullptr);
    return (i >= 0 && i < (int)units.size() && units[i].fused < 0 && units[i].pair) ? 1 : 0;
}

int mpdx_bench_layer(mpdx_unet* u, const float* packed, const float* timetab, const float* x, int layer, int B, float* ws,
                     void* stream, int reps, int dbg, float* ms_per_launch) {
    if (!u || !packed || !timetab || !x || !ws || !ms_per_launch) return fail(MPDX_E_INVALID, "null argument");
    if (int rc = check_ready(u)) return rc;
    if (layer < 0 || layer >= (int)u->layers.size()) return fail(MPDX_E_INVALID, "bad layer index");
#ifndef MPDX_DEV_HOOKS
    if (dbg & 15) return fail(MPDX_E_STATE, "phase-ablation masks need a development build of libmpdx.so (MPDX_BUILD_DEFS=-DMPDX_DEV_HOOKS)");
#endif
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)
        if (int rc = run_layer(u, u->layers[layer], packed, timetab, x, ws, B, st, dbg & 15)) return rc;
    if (dbg & 16) {  // replay the same launches from a hipGraph (device-side launch cadence, no host in the loop)
        hipGraph_t graph;
        hipGraphExec_t exec;
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < reps; ++i)
            if (int rc = run_layer(u, u->layers[layer], packed, timetab, x, ws, B, st, dbg & 15)) return rc;
        HIP_TRY(hipStreamEndCapture(st, &graph));
        HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_TRY(hipGraphLaunch(exec, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipEventRecord(e0, st));
        HIP_TRY(hipGraphLaunch(exec, st));
        HIP_TRY(hipEventRecord(e1, st));
        HIP_TRY(hipStreamSynchronize(st));
        float msg = 0.f;
        HIP_TRY(hipEventElapsedTime(&msg, e0, e1));
        *ms_per_launch = msg / reps;
        (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return 0;
    }
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i)
        if (int rc = run_layer(u, u->layers[layer], packed, timetab, x, ws, B, st, dbg)) return rc;
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

int mpdx_unet_layer_tile(const mpdx_unet* u, int i, int B, char* buf, size_t buflen) {
    if (!u || !buf || i < 0 || i >= (int)u->layers.size()) return fail(MPDX_E_INVALID, "bad layer index");
    const Layer& l = u->layers[i];
    int MT, NT;
    ConvArgs dummy;
    memset(&dummy, 0, sizeof(dummy));
    const Layer* l2 = (i + 1 < (int)u->layers.size() && pair_tile(l, u->layers[i + 1], B, MT, NT)) ? &u->layers[i + 1] : nullptr;
    if (const int v = weight_stationary_variant(l, l2, dummy, B, 0)) {   // "ws": the weight-stationary persistent kernel (conv_ws.hpp)
        if (v == 6) snprintf(buf, buflen, "wsp 32x16/2x1+1x1");   // conv_wsp_kernel: a pair of waves per tile, whole K per wave
        else if (v >= 4) snprintf(buf, buflen, "wsn 16x16/8x1");   // conv_wsn_kernel: 8 waves = 8 position tiles, whole K per wave
        else snprintf(buf, buflen, "ws %dx16/1x8%s", v == 1 ? 32 : 16, v == 3 ? "+1x1" : "");
        return 0;
    }
    choose_tile(l, B, MT, NT);
    if (l.cout % MT) MT = 16;
    const bool ks = layer_ksplit(l);
    snprintf(buf, buflen, "%dx%d/%dx%d", MT, NT, ks ? 1 : NT / 16, ks ? 8 : 8 / (NT / 16));
    return 0;
}

int mpdx_randn(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream) {
    if (!out) return fail(MPDX_E_INVALID, "null argument");
    if (n == 0) return 0;
    const size_t nquad = (n + 3) / 4;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)std::min<size_t>((nquad + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, out, n,
                       seed, offset);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

