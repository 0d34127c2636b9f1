// train_host.hpp - host side of the training step (included at the end of mpdx.hip; kernels in train.hpp).
// Entry points: mpdx_train_* (include/mpdx.h).  Everything is enqueued on the caller's stream; no synchronisation.
#pragma once

namespace mpdx {

// one-off: the flat (reference-layout) offsets of every parameter, the dgrad convolution of every layer whose input needs a
// gradient, and the producer of every tensor a layer reads (the forward engine recycles 4 + n_levels slots; training keeps
// every layer's output)
static void build_train_plan(mpdx_unet* u) {
    if (u->train_ready) return;
    size_t fo = 0;
    for (auto& p : u->params) { p.foff = fo; fo += (p.n + 3) / 4 * 4; }
    u->flat_floats = fo;
    const int n = (int)u->layers.size();
    u->tl.assign(n, {});
    std::unordered_map<int, int> owner;   // slot -> layer that wrote it last
    size_t pt = 0;
    for (int i = 0; i < n; ++i) {
        const Layer& l = u->layers[i];
        auto& t = u->tl[i];
        auto prod = [&](int s) { return s == SRC_X ? -1 : (s == SRC_NONE ? -2 : owner.at(s)); };
        t.src1_l = prod(l.src1); t.src2_l = prod(l.src2); t.res_l = prod(l.res);
        owner[l.dst] = i;
        t.need_dgrad = (t.src1_l >= 0) || (t.src2_l >= 0);
        t.dgrad_woff = ~(size_t)0;
        if (t.need_dgrad) {
            Layer d;
            d.name = "dgrad(" + l.name + ")";
            d.mode = CONV_S1; d.epi = EPI_BIAS;
            d.ks = l.mode == CONV_S1 ? l.ks : (l.mode == CONV_DOWN ? 3 : 5);
            d.c1 = l.cout; d.c2 = 0; d.cout = l.c1 + l.c2;
            d.L_in = d.L_out = l.mode == CONV_UPT ? l.L_out : l.L_in;
            // a padded container: valid rows of the dgrad's output = of the layer's input (ConvTranspose: full resolution, decimated at the store)
            d.Lv_out = l.Lv_out == 0 ? 0 : (l.mode == CONV_UPT ? l.Lv_out : (l.mode == CONV_DOWN ? 2 * l.Lv_out : l.Lv_out));
            d.cin_pad = (d.c1 + 15) / 16 * 16;
            d.rs = pick_row_stride(d.cin_pad, CONV_S1, d.L_in, d.L_out, d.L_in + 2 * (d.ks / 2));
            t.dg = d;
            t.dgrad_woff = pt;
            pt += (size_t)(d.cout / 16) * (d.cin_pad / 16) * d.ks * 256;
        }
    }
    u->packedT_floats = pt;
    // packing table
    std::vector<PackDesc> descs;
    for (size_t k = 0; k < u->params.size(); ++k) {
        const Param& p = u->params[k];
        PackDesc d;
        memset(&d, 0, sizeof(d));
        d.src = p.foff; d.dst = p.off; d.dstT = ~0ull;
        d.n = p.n; d.pn = p.pn; d.kind = p.kind;
        d.cout = p.cout; d.cin = p.cin; d.ks = p.ksz; d.cin_pad = p.cin_pad; d.nslot = p.nslot;
        for (int i = 0; i < n; ++i)
            if (u->layers[i].w == (int)k && u->tl[i].need_dgrad) {
                const Layer& g = u->tl[i].dg;
                d.dstT = u->tl[i].dgrad_woff;
                d.t_cout = g.cout; d.t_cin = g.c1; d.t_ks = g.ks; d.t_cin_pad = g.cin_pad;
                d.t_mode = u->layers[i].mode == CONV_UPT ? 1 : 0;
                d.pnT = (size_t)(g.cout / 16) * (g.cin_pad / 16) * g.ks * 256;
            }
        descs.push_back(d);
    }
    u->pack_descs_host = descs;
    u->train_ready = true;
}

static int ensure_pack_descs(mpdx_unet* u) {
    if (u->pack_descs_dev) return 0;
    // the chunk table of pack_train_kernel: kPackGroups groups of one pack of one parameter per block
    std::vector<PackChunk> chunks;
    for (size_t k = 0; k < u->pack_descs_host.size(); ++k) {
        const PackDesc& d = u->pack_descs_host[k];
        if (d.pn >= (1ull << 32) || d.pnT >= (1ull << 32) || d.n >= (1ull << 32)) return fail(MPDX_E_INVALID, "parameter %zu: more than 2^32 packed floats", k);
        // (PackChunk::first: the chunk's first GROUP - nslot consecutive 256-float blocks; kPackGroups groups per chunk)
        const unsigned long long gf = 256ull * (unsigned long long)(d.kind == 0 ? 1 : d.nslot), gt = 256ull * (unsigned long long)std::max(d.t_ks, 1);
        if (d.kind != 0 && d.nslot > 5) return fail(MPDX_E_INVALID, "parameter %zu: %d tap slots (the repack kernel holds 5)", k, d.nslot);
        if (d.dstT != ~0ull && d.t_ks > 5) return fail(MPDX_E_INVALID, "parameter %zu: %d dgrad taps (the repack kernel holds 5)", k, d.t_ks);
        for (unsigned long long g = 0; g * gf < d.pn; g += kPackGroups) chunks.push_back(PackChunk{(int)k, 0, (unsigned)g, 0u});
        if (d.dstT != ~0ull)
            for (unsigned long long g = 0; g * gt < d.pnT; g += kPackGroups) chunks.push_back(PackChunk{(int)k, 1, (unsigned)g, 0u});
    }
    u->n_pack_chunks = chunks.size();
    HIP_TRY(hipMalloc(&u->pack_chunks_dev, chunks.size() * sizeof(PackChunk)));
    HIP_TRY(hipMemcpy(u->pack_chunks_dev, chunks.data(), chunks.size() * sizeof(PackChunk), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&u->pack_descs_dev, u->pack_descs_host.size() * sizeof(PackDesc)));
    HIP_TRY(hipMemcpy(u->pack_descs_dev, u->pack_descs_host.data(), u->pack_descs_host.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
    return 0;
}

// workspace layout for a batch of B (float offsets)
struct TrainWs {
    size_t xn, eps, dE, out0, pre0, grad0, tmpX, dU, zst, pvec, wpart, rpart, emb, h1, temb, tm, h1m, tb, dT, dtm, dh1, zeros, ticket, norm, lossp, total;
    size_t slotB;          // floats of one activation slot for the batch
    size_t wpart_floats;
    // deferred reductions (one launch each at the end of the backward pass): every layer keeps its own partial sums
    bool deferred;
    size_t wparts, pvecs;  // areas; carved up in launch order by the backward pass
};
static size_t wgrad_splits(int M, int N, int B) {
    const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
    int S = (256 + tiles - 1) / tiles;   // about one workgroup per CU
    // How many batch splits: more splits = shorter trajectory chains per block but more partial sums to write and re-read (wgrad_reduce_all).  With the
    // prefetching trajectory loop the measured optimum of the 256 x 256 layers (64 tiles, 4 splits by the rule above) is 2 / 2 / 4 / 8 splits at batch
    // 32 / 64 / 128 / 512 (gpurun_out/r04za/wgrad_rule*.txt: batch 32 0.68 -> 0.63 ms with half the splits, batch 512 2.42 -> 2.37 with twice) - the
    // rule's count times sqrt(B / 128), within [1/2, 2]; fewer splits for every layer, more only for the layers with many tiles.
    {
        const float f = std::min(2.0f, std::max(0.5f, sqrtf((float)B / 128.0f)));
        if (f < 1.0f || tiles >= 16) S = std::max(1, (int)lroundf((float)S * f));
    }
    S = std::max(1, std::min(S, B));
    const int per = (B + S - 1) / S;
    return (size_t)((B + per - 1) / per);
}
static TrainWs train_ws(const mpdx_unet* u, int B) {
    TrainWs w;
    const size_t n = u->layers.size();
    const int H = u->cfg.n_support_points, D = u->cfg.state_dim;
    w.slotB = u->slot_floats * (size_t)B;
    const size_t xs = ((size_t)B * std::max(H, u->Hc) * D + 3) / 4 * 4;   // (xn and dE live in the network's container layout)
    size_t o = 0;
    auto take = [&](size_t k) { const size_t r = o; o += (k + 3) / 4 * 4; return r; };
    w.xn = take(xs); w.eps = take(xs); w.dE = take(xs);
    w.out0 = take(n * w.slotB);
    w.pre0 = take(n * w.slotB);
    w.grad0 = take(n * w.slotB);
    w.tmpX = take(2 * w.slotB);
    w.dU = take(w.slotB);
    w.zst = take(2 * w.slotB);
    w.pvec = take((size_t)4 * B * 512);
    size_t wp = 0;
    for (const Layer& l : u->layers) {
        const int M = l.mode == CONV_UPT ? l.c1 + l.c2 : l.cout, N = l.mode == CONV_UPT ? l.cout : std::max(l.c1, l.c2);
        wp = std::max(wp, wgrad_splits(M, N, B) * M * N * (size_t)l.ks);
    }
    wp = std::max(wp, wgrad_splits(D, u->cfg.unet_input_dim, B) * D * (size_t)u->cfg.unet_input_dim);
    w.wpart_floats = wp;
    w.wpart = take(wp);
    {   // total partial-sum storage if no layer re-uses another's: deferred mode while it stays below 256 MB
        size_t tot = 0, pv = 0;
        for (const Layer& l : u->layers) {
            const int KS = l.ks;
            if (l.mode == CONV_UPT) tot += wgrad_splits(l.c1, l.cout, B) * l.c1 * l.cout * (size_t)KS;
            else {
                tot += wgrad_splits(l.cout, l.c1, B) * l.cout * l.c1 * (size_t)KS;
                if (l.c2 > 0) tot += wgrad_splits(l.cout, l.c2, B) * l.cout * l.c2 * (size_t)KS;
            }
            pv += (l.epi == EPI_GN_MISH ? (size_t)3 * B : (size_t)256 + 64) * l.cout;
        }
        tot += wgrad_splits(D, u->cfg.unet_input_dim, B) * D * (size_t)u->cfg.unet_input_dim;
        if (B < 64) tot *= 4;   // (the backward programs' layers may take up to 4 x the rule's splits at small batches: MPDX_WGRAD_PROG_MUL <= 4)
        pv += (size_t)(256 + 64) * D;
        static const bool off = getenv("MPDX_TRAIN_DEFERRED") && atoi(getenv("MPDX_TRAIN_DEFERRED")) == 0;
        w.deferred = !off && tot <= ((size_t)96 << 20);   // floats
        w.wparts = take(w.deferred ? tot : 4);
        w.pvecs = take(w.deferred ? pv : 4);
    }
    w.rpart = take((size_t)256 * 512);
    w.emb = take((size_t)B * 32); w.h1 = take((size_t)B * 128); w.temb = take((size_t)B * 32);
    w.tm = take((size_t)B * 32); w.h1m = take((size_t)B * 128);
    w.tb = take((size_t)B * u->tt_row); w.dT = take((size_t)B * u->tt_row);
    w.dtm = take((size_t)B * 32); w.dh1 = take((size_t)B * 128);
    w.zeros = take(1024);
    w.ticket = take(4);      // directly behind `zeros`: one memset clears both
    w.norm = take(1024 + 8);
    w.lossp = take(32);       // 16 doubles: the loss value's wave sums (train_loss_kernel); offsets are multiples of 4 floats: 8-byte aligned
    w.total = o;
    return w;
}

static int launch_layer(const Layer& l, ConvArgs& a, int B, hipStream_t st) { return launch_conv_layer(l, a, B, st); }

static int fill_geom(const Layer& l, int B, ConvArgs& a) {
    a.c1 = l.c1; a.c2 = l.c2;
    a.B = B; a.L_in = l.L_in; a.L_out = l.L_out; a.C_out = l.cout;
    a.cin_pad = l.cin_pad; a.rs = l.rs; a.gs = l.gs;
    a.Lv_out = l.Lv_out;
    auto lg2 = [](int v) { int k = 0; while ((1 << k) < v) ++k; return k; };
    a.lg_c4n = lg2(l.cin_pad / 4); a.lg_Lin = lg2(l.L_in); a.lg_Lout = lg2(l.L_out); a.lg_gs = l.gs > 0 ? lg2(l.gs) : 0;
    if ((1 << a.lg_c4n) != l.cin_pad / 4 || (1 << a.lg_Lin) != l.L_in || (1 << a.lg_Lout) != l.L_out || (l.gs > 0 && (1 << a.lg_gs) != l.gs))
        return fail(MPDX_E_INVALID, "layer %s: channel/length/group sizes must be powers of two", l.name.c_str());
    return 0;
}

// deferred reductions of one backward pass
struct Deferred {
    bool on = false;
    float* ws = nullptr;
    float* grads = nullptr;
    size_t wcur = 0, pcur = 0;   // next free float in the partial areas (offsets from ws)
    ReduceAllArgs red;
    ColsumAllArgs col;
};

// a weight-gradient GEMM ready to launch (alone, or inside bwd_pair_kernel)
struct WgradJob { WgradArgs a; dim3 grid; size_t lds; int KS; bool deferred; float* g; int n_tot, n_off, S; };

// s_div > 1 (weight gradients that run behind the chain in wgrad_multi_kernel, where ALL layers' blocks fill the GPU together): fewer batch splits per
// layer than the one-workgroup-per-CU rule of a layer on its own - the partial sums written and re-read by the reduction shrink by the same factor
static int make_wgrad(const float* A, int LA, int lda, int a_off, int M, const float* Bm, int LB, int ldb, int b_off, int N, int sb, int ob, int KS,
                      int B, float* part, float* g, int n_tot, int n_off, Deferred* df, WgradJob& j, int s_div = 1) {
    size_t S_use = wgrad_splits(M, N, B);
    if ((s_div > 1 || s_div < -1) && df && df->on) {   // (s_div < -1: MORE splits - the small batches, where a block's chain of trajectories is the launch's length)
        const size_t want = s_div > 1 ? std::max<size_t>(1, S_use / (size_t)s_div) : std::min<size_t>((size_t)B, S_use * (size_t)(-s_div));
        const int per = (int)((B + want - 1) / want);
        S_use = (size_t)((B + per - 1) / per);
    }
    if (df && df->on && df->red.n < 96) {   // this layer's partial sums get their own storage; reduced at the end of the pass
        const size_t S0 = S_use;
        part = df->ws + df->wcur;
        auto& e = df->red.e[df->red.n++];
        e.part = df->wcur; e.g = (unsigned long long)(g - df->grads); e.S = (int)S0; e.M = M; e.N = N; e.KS = KS; e.n_tot = n_tot; e.n_off = n_off;
        df->wcur += S0 * M * N * (size_t)KS;
    } else df = nullptr;
    WgradArgs& a = j.a;
    a.bias_part = nullptr; a.bias_from_b = 0;
    a.A = A; a.Bm = Bm; a.part = part;
    a.LA = LA; a.lda = lda; a.a_off = a_off; a.M = M;
    a.LB = LB; a.ldb = ldb; a.b_off = b_off; a.N = N;
    a.sb = sb; a.ob = ob; a.B = B;
    const int S = (int)S_use;
    a.b_per_split = (B + S - 1) / S;
    if (LA % 4) return fail(MPDX_E_INVALID, "wgrad: horizon %d is not a multiple of 4", LA);
    if (KS != 1 && KS != 3 && KS != 4 && KS != 5) return fail(MPDX_E_INVALID, "wgrad: %d taps", KS);
    j.lds = (size_t)(LA + LB + 4) * kWgRS * sizeof(float);
    j.grid = dim3((N + 31) / 32, (M + 31) / 32, S);
    j.KS = KS; j.deferred = df != nullptr; j.g = g; j.n_tot = n_tot; j.n_off = n_off; j.S = S;
    return 0;
}
// let a (deferred) weight-gradient job add up its convolution's bias gradient too: true if attached (else the caller runs launch_rowsum)
static bool attach_bias(WgradJob& j, Deferred* df, float* gbias, bool from_b) {
    static const bool off = getenv("MPDX_TRAIN_BIAS_FOLD") && atoi(getenv("MPDX_TRAIN_BIAS_FOLD")) == 0;   // dev A/B switch
    if (off || !df || !df->on || !j.deferred || df->col.n >= 120) return false;
    const int C = from_b ? j.a.N : j.a.M;
    j.a.bias_part = df->ws + df->pcur;
    j.a.bias_from_b = from_b ? 1 : 0;
    auto& e = df->col.e[df->col.n++];
    e.part = df->pcur; e.out = (unsigned long long)(gbias - df->grads); e.rows = j.S; e.C = C;
    df->pcur += (size_t)j.S * C;
    return true;
}
// the reduction of a job whose partial sums are not deferred
static void finish_wgrad(const WgradJob& j, hipStream_t st) {
    if (j.deferred) return;
    const size_t per = (size_t)j.a.M * j.a.N * j.KS;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((per + 255) / 256, 1024)), dim3(256), 0, st, (const float*)j.a.part, j.g, j.S,
                       j.a.M, j.a.N, j.KS, j.n_tot, j.n_off);
}
static void run_wgrad(const WgradJob& j, hipStream_t st) {
    switch (j.KS) {
        case 1: hipLaunchKernelGGL(wgrad_kernel<1>, j.grid, dim3(256), j.lds, st, j.a); break;
        case 3: hipLaunchKernelGGL(wgrad_kernel<3>, j.grid, dim3(256), j.lds, st, j.a); break;
        case 4: hipLaunchKernelGGL(wgrad_kernel<4>, j.grid, dim3(256), j.lds, st, j.a); break;
        default: hipLaunchKernelGGL(wgrad_kernel<5>, j.grid, dim3(256), j.lds, st, j.a); break;
    }
    finish_wgrad(j, st);
}
static int launch_wgrad(const float* A, int LA, int lda, int a_off, int M, const float* Bm, int LB, int ldb, int b_off, int N, int sb, int ob, int KS,
                        int B, float* part, float* g, int n_tot, int n_off, hipStream_t st, Deferred* df = nullptr) {
    WgradJob j;
    if (int rc = make_wgrad(A, LA, lda, a_off, M, Bm, LB, ldb, b_off, N, sb, ob, KS, B, part, g, n_tot, n_off, df, j)) return rc;
    run_wgrad(j, st);
    return 0;
}

// two wave groups per weight-gradient block inside bwd_pair_kernel's 512-thread workgroups (train.hpp wgrad_body, ngrp = 2): shapes of the prefetching loop
static bool wgrad_two_groups(const WgradJob& j) {
    static const bool off = getenv("MPDX_WGRAD_TWO") && atoi(getenv("MPDX_WGRAD_TWO")) == 0;   // dev A/B switch
    if (off) return false;
    const int LA = j.a.LA, LB = j.a.LB, nr = LA >> 3;
    return (LA & 7) == 0 && LB == ((j.KS == 3 || j.KS == 4) ? 2 * LA : LA) && (nr == 1 || nr == 2 || nr == 4 || nr == 8);
}

// does bwd_pair_kernel exist for the tile the forward engine picks for this input-gradient convolution?  (levels of more than 64 positions -
// n_support_points = 128 - run one trajectory per 128-position tile: per-layer launches there)
static bool bwd_pair_has_tile(const Layer& dgl, int B) {
    int MT, NT;
    choose_tile(dgl, B, MT, NT);
    if (dgl.cout % MT) MT = 16;
    if (dgl.cout % MT || NT % dgl.L_out) return false;
    return (MT == 32 || MT == 16) && (NT == 64 || NT == 32 || NT == 16);
}

// dgrad convolution + the layer's weight-gradient GEMM(s) in ONE launch (bwd_pair_kernel); the jobs must use distinct partial buffers
// GN_BWD: the dgrad blocks run the EPI_GN_BWD epilogue (cd carries its operands; `dgl` then has epi = EPI_GN_MISH and the group size
// of the Conv1dBlock below, so that the tile holds whole GroupNorm regions)
// dgl2 / cd2 (optional): a second, 1x1 input-gradient convolution on the same tile behind the first one's blocks (BwdPairArgs::cd2); returns kNoPair2 (nothing
// launched) when that convolution does not fit the first one's tile
constexpr int kNoPair2 = 99;
template <int KS_D, bool GN_BWD = false>
static int launch_bwd_pair(const Layer& dgl, ConvArgs& cd, int B, const WgradJob* jobs, int njobs, hipStream_t st, const Layer* dgl2 = nullptr, ConvArgs* cd2 = nullptr) {
    int MT, NT;
    choose_tile(dgl, B, MT, NT);
    if (dgl.cout % MT) MT = 16;
    if (dgl.cout % MT || NT % dgl.L_out) return fail(MPDX_E_INVALID, "layer %s: no tile for C_out=%d L=%d", dgl.name.c_str(), dgl.cout, dgl.L_out);
    if (dgl2 && (dgl2->cout % MT || dgl2->L_out != dgl.L_out || dgl2->ks != 1 || dgl2->mode != CONV_S1 || njobs > 3)) return kNoPair2;
    cd.n_tiles_n = (int)(((long)B * dgl.L_out + NT - 1) / NT);
    BwdPairArgs a;
    memset(&a, 0, sizeof(a));
    a.cd = cd;
    a.n_dgrad = (dgl.cout / MT) * cd.n_tiles_n;
    size_t lds = 0;
    if (dgl2) {
        cd2->n_tiles_n = cd.n_tiles_n;
        a.cd2 = *cd2;
        a.n_dgrad2 = (dgl2->cout / MT) * cd2->n_tiles_n;
    }
    int total = a.n_dgrad + a.n_dgrad2;
    for (int k = 0; k < njobs; ++k) {
        a.w[k] = jobs[k].a; a.ks_w[k] = jobs[k].KS;
        a.gx[k] = jobs[k].grid.x; a.gy[k] = jobs[k].grid.y;
        a.nw[k] = jobs[k].grid.x * jobs[k].grid.y * jobs[k].grid.z;
        total += a.nw[k];
        a.two[k] = wgrad_two_groups(jobs[k]) ? 1 : 0;
        lds = std::max(lds, a.two[k] ? std::max(2 * jobs[k].lds, (size_t)(256 * jobs[k].KS * 4 + 256) * sizeof(float)) : jobs[k].lds);
    }
#define MPDX_BP_TILE(mt, nt)                                                                              \
    if (MT == mt && NT == nt) {                                                                           \
        lds = std::max(lds, conv_block_lds_bytes<CONV_S1, KS_D, mt, nt, 8>(cd.L_in, cd.L_out, cd.rs));      \
        if (dgl2) lds = std::max(lds, conv_block_lds_bytes<CONV_S1, 1, mt, nt, 8>(cd2->L_in, cd2->L_out, cd2->rs)); \
        if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "backward pair needs %zu B of LDS", lds);        \
        auto kern = bwd_pair_kernel<KS_D, mt, nt, GN_BWD ? EPI_GN_BWD : EPI_BIAS>;                         \
        if (lds > 64 * 1024)                                                                              \
            if (int rc = raise_lds_limit((const void*)kern)) return rc;                                   \
        hipLaunchKernelGGL(kern, dim3(total), dim3(512), lds, st, a);                                     \
        for (int k = 0; k < njobs; ++k) finish_wgrad(jobs[k], st);                                        \
        return 0;                                                                                         \
    }
    MPDX_BP_TILE(32, 64) MPDX_BP_TILE(32, 32) MPDX_BP_TILE(16, 64) MPDX_BP_TILE(16, 32) MPDX_BP_TILE(32, 16) MPDX_BP_TILE(16, 16)
#undef MPDX_BP_TILE
    return fail(MPDX_E_INVALID, "no backward-pair instantiation for tile %dx%d", MT, NT);
}

// up to three weight-gradient GEMMs that have no input-gradient convolution to ride on (layers whose input needs no gradient, final_conv[1]) in ONE
// launch: bwd_pair_kernel with n_dgrad = 0 (any instantiation: the dgrad body is never entered)
static int launch_lone_wgrads(const WgradJob* jobs, int njobs, hipStream_t st) {
    if (njobs <= 0) return 0;
    BwdPairArgs a;
    memset(&a, 0, sizeof(a));
    size_t lds = 0;
    int total = 0;
    for (int k = 0; k < njobs; ++k) {
        a.w[k] = jobs[k].a; a.ks_w[k] = jobs[k].KS;
        a.gx[k] = jobs[k].grid.x; a.gy[k] = jobs[k].grid.y;
        a.nw[k] = jobs[k].grid.x * jobs[k].grid.y * jobs[k].grid.z;
        total += a.nw[k];
        a.two[k] = wgrad_two_groups(jobs[k]) ? 1 : 0;
        lds = std::max(lds, a.two[k] ? std::max(2 * jobs[k].lds, (size_t)(256 * jobs[k].KS * 4 + 256) * sizeof(float)) : jobs[k].lds);
    }
    auto kern = bwd_pair_kernel<1, 16, 16, EPI_BIAS>;
    if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "weight-gradient launch needs %zu B of LDS", lds);
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit((const void*)kern)) return rc;
    hipLaunchKernelGGL(kern, dim3(total), dim3(512), lds, st, a);
    for (int k = 0; k < njobs; ++k) finish_wgrad(jobs[k], st);
    return 0;
}

// ---- whole-trajectory backward programs (fused_bwd.hpp).  The DOWN program: the backward pass of downs[0..2] of the standard network (three levels of
// [blocks.0 | residual 1x1 | blocks.1] [blocks.0 | blocks.1 (identity residual)] [Downsample1d], 32 / 64 / 128 channels on 64 / 32 / 16 positions) in
// ONE launch.  Layer indices: level k occupies [6 k, 6 k + 6) = b0.0, r, b0.1, b1.0, b1.1, down.
struct BwdProgLayout { int off4[5]; int stat_off; size_t lds_bytes; };
static BwdProgLayout bwd_down_layout() {
    // five LDS slots of the largest buffer (20 rows x (128 + 4) floats = 660 float4): IN (the stride-2 layer's zero-stuffed dU), GB, DUa, DUb, GA
    BwdProgLayout L;
    const int slot4 = kBwdSlot4;   // >= 20 x 33, 36 x 17, 68 x 9 float4
    for (int k = 0; k < 5; ++k) L.off4[k] = k * slot4;
    L.stat_off = 5 * slot4 * 4;
    L.lds_bytes = (size_t)(L.stat_off + 384) * sizeof(float);
    return L;
}
// is the head of the network the three-level down path the program is written for?  1: layers [0, 18), every level ends in a Downsample1d (dim_mults (1, 2, 4, 8));
// 2: layers [0, 17), the third level is the innermost one and has none (dim_mults (1, 2, 4): the reference's UNET_DIM_MULTS option 0); 0: neither
static int bwd_down_applicable(const mpdx_unet* u) {
    if ((int)u->layers.size() < 19 || u->cfg.n_support_points != 64 || u->masked()) return 0;
    int variant = 1;
    for (int k = 0; k < 3; ++k) {
        const int C = 32 << k, Lk = 64 >> k, b = 6 * k;
        const int cin = k == 0 ? u->cfg.state_dim : C / 2;
        auto blk = [&](int i, int c_in) { const Layer& l = u->layers[i]; return l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_GN_MISH && l.c1 == c_in && l.c2 == 0 && l.cout == C && l.L_out == Lk && l.gs * 8 == C; };
        if (!blk(b + 0, cin) || !blk(b + 2, C) || !blk(b + 3, C) || !blk(b + 4, C)) return 0;
        const Layer& r = u->layers[b + 1];
        if (!(r.mode == CONV_S1 && r.ks == 1 && r.epi == EPI_BIAS && r.c1 == cin && r.cout == C && r.L_out == Lk)) return 0;
        const Layer& d = u->layers[b + 5];
        const auto& tl = u->tl;
        const bool has_down = d.mode == CONV_DOWN && d.ks == 3 && d.epi == EPI_BIAS && d.c1 == C && d.cout == C && d.L_in == Lk && d.L_out == Lk / 2 && tl[b + 5].src1_l == b + 4;
        if (!has_down) {
            if (k < 2) return 0;
            variant = 2;   // (layer 17 is mid_block1's first convolution: the per-layer path has put its gradient into grd(16) by the time the program runs)
        }
        if (u->layers[b + 0].tb_off < 0 || u->layers[b + 3].tb_off < 0 || u->layers[b + 2].tb_off >= 0 || u->layers[b + 4].tb_off >= 0) return 0;
        if (tl[b + 2].res_l != b + 1 || tl[b + 4].res_l != b + 2 || tl[b + 2].src1_l != b || tl[b + 3].src1_l != b + 2 || tl[b + 4].src1_l != b + 3) return 0;
        if (k > 0 && (tl[b].src1_l != b - 1 || tl[b + 1].src1_l != b - 1)) return 0;
        for (int i = b + (k == 0 ? 2 : 0); i < b + (has_down ? 6 : 5); ++i) if (!tl[i].need_dgrad) return 0;
    }
    if (variant == 2 && (int)u->layers.size() == 34) {   // 3: ... and the two middle blocks (layers 17 .. 20: identity residuals, 128 channels on 16 positions) in front
        static const bool mid_off = getenv("MPDX_TRAIN_BWD_MID") && atoi(getenv("MPDX_TRAIN_BWD_MID")) == 0;
        const auto& tl = u->tl;
        bool ok = !mid_off;
        for (int i = 17; i <= 20 && ok; ++i) {
            const Layer& l = u->layers[i];
            ok = l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_GN_MISH && l.c1 == 128 && l.c2 == 0 && l.cout == 128 && l.L_out == 16 && l.gs == 16 && tl[i].src1_l == i - 1 && tl[i].need_dgrad &&
                 ((i & 1) ? l.tb_off >= 0 : l.tb_off < 0);
        }
        if (ok && tl[18].res_l == 16 && tl[20].res_l == 18 && tl[17].res_l < 0 && tl[19].res_l < 0) variant = 3;
    }
    return variant;
}

// is the tail of the network [Upsample1d(128) | up level of 64 channels on 16 positions | up level of 32 on 32 | final_conv[0]] the UP program is written for?
// The four-level network: layers [33, 46) (ups[1], ups[2]); the three-level one: [21, 34) (ups[0], ups[1]).  Returns the first layer of the program, or -1.
static int bwd_up_applicable(const mpdx_unet* u) {
    const int n = (int)u->layers.size();
    if (n < 34 || u->cfg.n_support_points != 64 || u->masked()) return -1;
    const auto& tl = u->tl;
    const int fi = n - 1;   // final_conv[0] (final_conv[1], the 1x1, lives in the loss kernel and in final_conv[0]'s epilogue: it is no layer of the list)
    const Layer& f = u->layers[fi];
    if (!(f.mode == CONV_S1 && f.ks == 5 && f.epi == EPI_GN_MISH && f.c1 == 32 && f.c2 == 0 && f.cout == 32 && f.L_out == 64 && f.gs == 4 && f.tb_off < 0 && tl[fi].src1_l == fi - 1 && tl[fi].res_l < 0))
        return -1;
    const int bases[2] = {fi - 6, fi - 12}, Cs[2] = {32, 64}, Ls[2] = {32, 16}, skips[2] = {10, 16};
    for (int k = 0; k < 2; ++k) {
        const int b = bases[k], C = Cs[k], Lk = Ls[k];
        auto blk = [&](int i, int c1, int c2) { const Layer& l = u->layers[i]; return l.mode == CONV_S1 && l.ks == 5 && l.epi == EPI_GN_MISH && l.c1 == c1 && l.c2 == c2 && l.cout == C && l.L_out == Lk && l.gs * 8 == C; };
        if (!blk(b, 2 * C, 2 * C) || !blk(b + 2, C, 0) || !blk(b + 3, C, 0) || !blk(b + 4, C, 0)) return -1;
        const Layer& r = u->layers[b + 1];
        if (!(r.mode == CONV_S1 && r.ks == 1 && r.epi == EPI_BIAS && r.c1 == 2 * C && r.c2 == 2 * C && r.cout == C && r.L_out == Lk)) return -1;
        const Layer& up = u->layers[b + 5];
        if (!(up.mode == CONV_UPT && up.ks == 4 && up.epi == EPI_BIAS && up.c1 == C && up.cout == C && up.L_in == Lk && up.L_out == 2 * Lk)) return -1;
        if (u->layers[b].tb_off < 0 || u->layers[b + 3].tb_off < 0 || u->layers[b + 2].tb_off >= 0 || u->layers[b + 4].tb_off >= 0) return -1;
        if (tl[b].src1_l != b - 1 || tl[b + 1].src1_l != b - 1 || tl[b].src2_l != skips[k] || tl[b + 1].src2_l != skips[k]) return -1;
        if (tl[b + 2].res_l != b + 1 || tl[b + 4].res_l != b + 2 || tl[b + 2].src1_l != b || tl[b + 3].src1_l != b + 2 || tl[b + 4].src1_l != b + 3 || tl[b + 5].src1_l != b + 4) return -1;
        for (int i = b; i < b + 6; ++i) if (!tl[i].need_dgrad) return -1;
    }
    const int first = fi - 12;
    // the producer of the inner level's x half: 128 channels on 16 positions (the Upsample1d of the level below, or - three levels - mid_block2's blocks.1)
    const Layer& x = u->layers[first - 1];
    if (!(x.cout == 128 && x.L_out == 16 && u->layers[16].cout == 128 && u->layers[10].cout == 64 && tl[fi].need_dgrad)) return -1;
    return first;
}

// The steps of a backward chain (bwd_chain_kernel) collected in launch order; flush() launches them as ONE kernel (more than it can hold: several)
struct ChainBuilder {
    bool on = false;
    int B = 0;
    hipStream_t st = nullptr;
    ChainArgs a;
    int n_conv = 0, n_gn = 0;
    size_t lds = 0;
    int launches = 0, steps = 0;
    ChainBuilder() { memset(&a, 0, sizeof(a)); }
    int flush() {
        if (a.n == 0) return 0;
        if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "backward chain needs %zu B of LDS", lds);
        if (lds > 64 * 1024)
            if (int rc = raise_lds_limit((const void*)bwd_chain_kernel)) return rc;
        hipLaunchKernelGGL(bwd_chain_kernel, dim3(B), dim3(512), lds, st, a);
        ++launches;
        a.n = 0; n_conv = 0; n_gn = 0; lds = 0;
        return 0;
    }
    // can this input-gradient convolution run as a chain step?  (one trajectory per tile, 32-channel tiles, an instantiated body)
    static bool conv_ok(const Layer& dgl, int min_L) {
        return dgl.mode == CONV_S1 && (dgl.ks == 5 || dgl.ks == 3 || dgl.ks == 1) && dgl.L_in == dgl.L_out && (dgl.L_out == 64 || dgl.L_out == 32 || dgl.L_out == 16) &&
               dgl.L_out >= min_L && dgl.cout % 32 == 0;
    }
    int add_conv(const Layer& dgl, const ConvArgs& cd, bool gnbwd) {
        if (gnbwd && dgl.ks == 1) return fail(MPDX_E_INVALID, "chain: no GroupNorm-backward body for a 1-tap convolution");
        if (n_conv == kChainMaxConv || a.n == kChainMaxSteps)
            if (int rc = flush()) return rc;
        ChainStep& s = a.st[a.n++];
        s.kind = 0; s.sel = (short)chain_sel(dgl.ks, dgl.L_out, gnbwd ? 1 : 0); s.n_mt = (short)(dgl.cout / 32); s.idx = (short)n_conv;
        a.cd[n_conv] = cd;
        a.cd[n_conv].n_tiles_n = B;
        ++n_conv; ++steps;
        const size_t need = dgl.ks == 5 ? conv_block_lds_bytes<CONV_S1, 5, 32, 64, 8>(dgl.L_in, dgl.L_out, cd.rs)
                          : dgl.ks == 3 ? conv_block_lds_bytes<CONV_S1, 3, 32, 64, 8>(dgl.L_in, dgl.L_out, cd.rs)
                                        : conv_block_lds_bytes<CONV_S1, 1, 32, 64, 8>(dgl.L_in, dgl.L_out, cd.rs);
        // (the staged window of ONE trajectory: NT / L_out = 1, whatever NT the template names; the K-partial buffer: 8 x L_out x 36 floats)
        const size_t stage = (size_t)(dgl.L_in + 2 * (dgl.ks / 2)) * cd.rs * sizeof(float), red = (size_t)8 * dgl.L_out * 36 * sizeof(float);
        (void)need;
        lds = std::max(lds, std::max(stage, red));
        return 0;
    }
    int add_gn(const GnBwdArgs& g, int epl) {
        if (n_gn == kChainMaxGn || a.n == kChainMaxSteps)
            if (int rc = flush()) return rc;
        ChainStep& s = a.st[a.n++];
        s.kind = 1; s.sel = (short)epl; s.n_mt = 1; s.idx = (short)n_gn;
        a.gn[n_gn++] = g;
        ++steps;
        return 0;
    }
};

// any number of deferred weight-gradient GEMMs in launches of up to kWgradMultiMax jobs (wgrad_multi_kernel); the longest blocks first
static int launch_wgrads_multi(std::vector<WgradJob>& jobs, hipStream_t st) {
    if (jobs.empty()) return 0;
    std::stable_sort(jobs.begin(), jobs.end(), [](const WgradJob& x, const WgradJob& y) {
        const long wx = (long)x.a.b_per_split * x.a.LA * x.KS, wy = (long)y.a.b_per_split * y.a.LA * y.KS;
        return wx > wy;
    });
    for (size_t k0 = 0; k0 < jobs.size(); k0 += kWgradMultiMax) {
        const int nj = (int)std::min<size_t>(kWgradMultiMax, jobs.size() - k0);
        WgradMultiArgs a;
        memset(&a, 0, sizeof(a));
        a.n = nj;
        size_t lds = 0;
        int total = 0;
        for (int k = 0; k < nj; ++k) {
            const WgradJob& j = jobs[k0 + k];
            if (!j.deferred) return fail(MPDX_E_INVALID, "wgrad_multi: job without its own partial-sum buffer");
            a.w[k] = j.a; a.ks[k] = (signed char)j.KS;
            a.gx[k] = (short)j.grid.x; a.gy[k] = (short)j.grid.y;
            a.start[k] = total;
            total += (int)(j.grid.x * j.grid.y * j.grid.z);
            a.two[k] = wgrad_two_groups(j) ? 1 : 0;
            lds = std::max(lds, a.two[k] ? std::max(2 * j.lds, (size_t)(256 * j.KS * 4 + 256) * sizeof(float)) : j.lds);
        }
        for (int k = nj; k <= kWgradMultiMax; ++k) a.start[k] = total;
        if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "weight-gradient launch needs %zu B of LDS", lds);
        if (lds > 64 * 1024)
            if (int rc = raise_lds_limit((const void*)wgrad_multi_kernel)) return rc;
        hipLaunchKernelGGL(wgrad_multi_kernel, dim3(total), dim3(512), lds, st, a);
    }
    return 0;
}

// channel sums of a dense [rows][C] tensor -> out[C]
static void launch_rowsum(const float* x, size_t rows, int C, float* part, float* out, hipStream_t st, Deferred* df = nullptr) {
    const int nb = (int)std::min<size_t>(64, rows);
    const int rpb = (int)((rows + nb - 1) / nb);
    const int nblk = (int)((rows + rpb - 1) / rpb);
    if (df && df->on && df->col.n < 120) {
        part = df->ws + df->pcur;
        auto& e = df->col.e[df->col.n++];
        e.part = df->pcur; e.out = (unsigned long long)(out - df->grads); e.rows = nblk; e.C = C;
        df->pcur += (size_t)64 * C;
        hipLaunchKernelGGL(rowsum_part_kernel, dim3(nblk), dim3(256), 0, st, x, part, (int)rows, C, rpb);
        return;
    }
    hipLaunchKernelGGL(rowsum_part_kernel, dim3(nblk), dim3(256), 0, st, x, part, (int)rows, C, rpb);
    ColsumArgs c;
    memset(&c, 0, sizeof(c));
    c.part[0] = part; c.out[0] = out; c.B = nblk; c.C = C;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64, 1), dim3(256), 0, st, c);
}

static void launch_acc(float* dst, const float* src, int B, int L, int Cd, int Ls, int Cs, int c_off, int step, int store, hipStream_t st) {
    const size_t total = (size_t)B * L * Cd;
    hipLaunchKernelGGL(acc_slice_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048)), dim3(256), 0, st, dst, src, B, L, Cd, Ls, Cs, c_off, step, store);
}

}  // namespace mpdx

using namespace mpdx;

extern "C" {

size_t mpdx_train_flat_floats(mpdx_unet* u) {
    if (!u) return 0;
    build_train_plan(u);
    return u->flat_floats;
}
size_t mpdx_train_dgrad_pack_floats(mpdx_unet* u) {
    if (!u) return 0;
    build_train_plan(u);
    return std::max<size_t>(u->packedT_floats, 4);
}
size_t mpdx_train_workspace_floats(mpdx_unet* u, int B) {
    if (!u || B <= 0) return 0;
    build_train_plan(u);
    return train_ws(u, B).total;
}
int mpdx_train_param_offset(mpdx_unet* u, int idx, size_t* off, size_t* n) {
    if (!u || idx < 0 || idx >= (int)u->params.size() || !off || !n) return fail(MPDX_E_INVALID, "bad argument");
    build_train_plan(u);
    *off = u->params[idx].foff; *n = u->params[idx].n;
    return 0;
}

/* flat parameter vector (reference layout) -> forward pack (+ the dgrad pack when packedT is given): one launch */
int mpdx_train_pack(mpdx_unet* u, const float* flat, float* packed, float* packedT, void* stream) {
    if (!u || !flat || !packed) return fail(MPDX_E_INVALID, "null argument");
    build_train_plan(u);
    if (int rc = ensure_pack_descs(u)) return rc;
    hipLaunchKernelGGL(pack_train_kernel, dim3((unsigned)u->n_pack_chunks), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)u->pack_descs_dev,
                       (const PackChunk*)u->pack_chunks_dev, flat, packed, packedT);
    HIP_TRY(hipGetLastError());
    for (auto& p : u->params) if (!p.done) { p.done = true; u->n_done++; }
    u->pack_version++;
    return 0;
}

/* Draw mode of the training pass (an iteration captured into a hipGraph, trainer.TrainStep.step): with a non-null `step_counter_dev` (a device int
 * that counts the optimiser steps taken: mpdx_adam_step(step < 0) advances it) the NEXT mpdx_train_loss_backward calls treat `t_dev` and `noise` as
 * OUTPUTS and draw them on the device (Philox4x32-10 keyed by `seed`, stream position step * B + sample); null disarms. */
namespace mpdx {
struct TrainRng { unsigned long long seed; const int* counter; };
static std::mutex g_train_rng_mu;
static std::unordered_map<const mpdx_unet*, TrainRng> g_train_rng;
}  // namespace mpdx
int mpdx_train_draw(mpdx_unet* u, unsigned long long seed, const int* step_counter_dev) {
    if (!u) return fail(MPDX_E_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(g_train_rng_mu);
    if (step_counter_dev) g_train_rng[u] = TrainRng{seed, step_counter_dev};
    else g_train_rng.erase(u);
    return 0;
}

/* One p_losses evaluation WITH its gradient (diffusion_model_base.py:331-352 + loss.backward()):
 *   x_noisy = q_sample(x_start, t, noise) with hard conditions; x_recon = unet(x_noisy, t) with hard conditions;
 *   loss = mean(|x_recon - target|^p [* weights]);  grads_flat = d loss * loss_scale / d parameters  (every entry written).
 * `flat` / `packed` / `packedT`: the parameters and their two packs (mpdx_train_pack).  loss_out: one device float. */
int mpdx_train_loss_backward(mpdx_unet* u, const float* flat, const float* packed, const float* packedT, float* grads_flat, const float* x_start,
                             const float* noise, const long long* t_dev, const float* sqrt_alphas_cumprod_dev,
                             const float* sqrt_one_minus_alphas_cumprod_dev, const float* freqs16, const float* hard_start, const float* hard_goal,
                             const float* weights_hd, int T, int B, int predict_epsilon, int l1, float loss_scale, float* loss_out, float* ws,
                             void* stream) {
    if (!u || !flat || !packed || !packedT || !grads_flat || !x_start || !noise || !t_dev || !freqs16 || !loss_out || !ws || B <= 0)
        return fail(MPDX_E_INVALID, "bad argument");
    build_train_plan(u);
    if (int rc = check_ready(u)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const mpdx_unet_cfg& c = u->cfg;
    const int H = c.n_support_points, D = c.state_dim, n = (int)u->layers.size();
    const int Hc = u->Hc;   // rows per trajectory of every activation / gradient tensor: H, or its power-of-two container (24, 40, 48, 96 ...: rows [H, Hc) zero)
    const bool masked = u->masked();
    const TrainWs w = train_ws(u, B);
    float* const xn = ws + w.xn;
    float* const eps = ws + w.eps;
    float* const dE = ws + w.dE;
    auto out = [&](int i) { return ws + w.out0 + (size_t)i * w.slotB; };
    auto pre = [&](int i) { return ws + w.pre0 + (size_t)i * w.slotB; };
    auto grd = [&](int i) { return ws + w.grad0 + (size_t)i * w.slotB; };
    auto tensor = [&](int li) -> const float* { return li == -1 ? xn : (li < 0 ? nullptr : out(li)); };
    auto gflat = [&](int pidx) { return grads_flat + u->params[pidx].foff; };

    // ---- forward, every layer's output (and GroupNorm input) kept
    // (no memset of the gradient buffers: the first writer of each in the backward pass stores, the later ones add - `first_write`)
    // (q_sample with its hard conditions and the pass's zero words - the zero bias of the dgrad convolutions + the time backward's ticket - are
    //  per-sample side jobs of the first launch, time_train_fwd_kernel: round 3 spent a memset and a launch on them)
    if (!sqrt_alphas_cumprod_dev || !sqrt_one_minus_alphas_cumprod_dev) return fail(MPDX_E_INVALID, "schedule tables missing");
    TimeTrainArgs ta;
    memset(&ta, 0, sizeof(ta));
    TimeBwdArgs tb;
    memset(&tb, 0, sizeof(tb));
    {
        ta.flat = flat; ta.t = t_dev; ta.freqs = freqs16;
        ta.emb = ws + w.emb; ta.h1 = ws + w.h1; ta.temb = ws + w.temb; ta.tb = ws + w.tb;
        ta.tm = ws + w.tm; ta.h1m = ws + w.h1m;
        ta.w1 = u->params[u->pidx.at("time_mlp.encoder.1.weight")].foff; ta.b1 = u->params[u->pidx.at("time_mlp.encoder.1.bias")].foff;
        ta.w3 = u->params[u->pidx.at("time_mlp.encoder.3.weight")].foff; ta.b3 = u->params[u->pidx.at("time_mlp.encoder.3.bias")].foff;
        ta.row = u->tt_row; ta.nblk = (int)u->tt_w.size();
        if (ta.nblk > 40 || c.time_emb_dim != 32) return fail(MPDX_E_INVALID, "time MLP shape unsupported by the training kernels");
        for (int i = 0; i < ta.nblk; ++i) {
            ta.woff[i] = u->params[u->tt_w[i]].foff; ta.boff[i] = u->params[u->tt_b[i]].foff;
            ta.cout[i] = u->tt_cout[i]; ta.toff[i] = u->tt_off[i];
        }
        ta.x0 = x_start; ta.noise = noise; ta.sqrt_ac = sqrt_alphas_cumprod_dev; ta.sqrt_1mac = sqrt_one_minus_alphas_cumprod_dev;
        ta.hs = hard_start; ta.hg = hard_goal; ta.xn = xn; ta.zero_words = ws + w.zeros; ta.n_zero = 1024 + 4;
        ta.H = H; ta.D = D; ta.T = T;
        {
            std::lock_guard<std::mutex> lk(g_train_rng_mu);
            auto it = g_train_rng.find(u);
            if (it != g_train_rng.end()) {
                if ((H * D) & 3) return fail(MPDX_E_INVALID, "draw mode: H * D = %d is not a multiple of 4", H * D);
                ta.rng_seed = it->second.seed; ta.rng_counter = it->second.counter;
                ta.t_out = const_cast<long long*>(t_dev); ta.noise_out = const_cast<float*>(noise);
            }
        }
        ta.B = B; ta.packed = const_cast<float*>(packed); ta.jobs = nullptr; ta.n_jobs = 0;
        ta.Hc = masked ? Hc : 0;
        {   // the fused forward programs' weight streams: re-assembled by side blocks of this launch (the pack launch before it wrote `packed`)
            static const bool fused_fwd_off0 = getenv("MPDX_TRAIN_FUSED_FWD") && atoi(getenv("MPDX_TRAIN_FUSED_FWD")) == 0;
            static const bool ride_off = getenv("MPDX_TRAIN_RESTREAM_RIDE") && atoi(getenv("MPDX_TRAIN_RESTREAM_RIDE")) == 0;   // dev A/B switch
            if (!fused_fwd_off0 && !ride_off && fused_mask(B) != 0u && (w.total < ((size_t)1 << 31))) {
                const void* jb = nullptr;
                int nj = 0;
                if (int rc = claim_fused_stream_jobs(u, packed, &jb, &nj)) return rc;
                ta.jobs = (const CopyJobDev*)jb; ta.n_jobs = nj;
            }
        }
        hipLaunchKernelGGL(time_train_fwd_kernel, dim3(2 * B + kRestreamBlocksPerJob * ta.n_jobs), dim3(512), 0, st, ta);
        tb.flat = flat; tb.grad = grads_flat; tb.dT = ws + w.dT; tb.emb = ta.emb; tb.h1 = ta.h1; tb.temb = ta.temb; tb.tm = ta.tm; tb.h1m = ta.h1m;
        tb.dtm = ws + w.dtm; tb.dh1 = ws + w.dh1; tb.ticket = (unsigned*)(ws + w.ticket);
        if (ta.row > kTimeBwdMaxRow)   // time_bwd_all_kernel carves dTs | roff | red out of LDS at fixed offsets of kTimeBwdMaxRow
            return fail(MPDX_E_INVALID, "time table row of %d floats (the training kernels take %d)", ta.row, kTimeBwdMaxRow);
        tb.w1 = ta.w1; tb.b1 = ta.b1; tb.w3 = ta.w3; tb.b3 = ta.b3;
        tb.B = B; tb.row = ta.row; tb.nblk = ta.nblk;
        for (int i = 0; i < ta.nblk; ++i) { tb.woff[i] = ta.woff[i]; tb.boff[i] = ta.boff[i]; tb.cout[i] = ta.cout[i]; tb.toff[i] = ta.toff[i]; }
    }
    // forward: the fused level programs of the planning path (they additionally keep every op's output and GroupNorm input,
    // FusedArgs::save) for the outer levels, one launch per layer for the rest
    static const bool fused_fwd_off = getenv("MPDX_TRAIN_FUSED_FWD") && atoi(getenv("MPDX_TRAIN_FUSED_FWD")) == 0;
    const bool fused_fwd = !fused_fwd_off && fused_mask(B) != 0u && (w.total < ((size_t)1 << 31));
    bool eps_done = false;
    for (int i = 0; i < n; ++i) {
        const int seg = fused_fwd ? u->owner[i] : -1;
        if (seg >= 0 && ((fused_mask(B) >> seg) & 1u) && fused_save_variant(u->fused[seg])) {
            const mpdx_unet::Fused& f = u->fused[seg];
            if (i != f.first) continue;   // the segment's launch covers layers [first, first + count)
            if (int rc = ensure_fused_streams(u, packed, st)) return rc;
            FusedArgs a = f.tmpl;
            a.packed = packed;
            a.tt_row = ws + w.tb; a.tt_stride = u->tt_row;
            const auto& t0 = u->tl[f.first];
            a.gsrc1 = tensor(t0.src1_l); a.gsrc2 = tensor(t0.src2_l);
            a.gsrc3 = f.in3_consumer >= 0 ? tensor(u->tl[f.in3_consumer].src2_l) : a.gsrc1;
            a.B = B;
            a.save = ws;
            int k = 0;
            for (; k < (int)f.op_layer.size(); ++k) {
                const int li = f.op_layer[k];
                a.ops[k].gdst = -1;   // nothing reads the planning path's slots here
                a.ops[k].save_out = (int)(w.out0 + (size_t)li * w.slotB);
                a.ops[k].save_pre = u->layers[li].epi == EPI_GN_MISH ? (int)(w.pre0 + (size_t)li * w.slotB) : -1;
            }
            for (; k < a.nops; ++k) { a.ops[k].save_out = -1; a.ops[k].save_pre = -1; }
            if (f.has_final) {   // final_conv[1] -> eps, no DDPM step
                a.out = eps; a.fmode = 0; a.n_per_ctx = B;
                eps_done = true;
            }
            if (int rc = launch_fused_args(f, a, B, st, true)) return rc;
            continue;
        }
        auto layer_args = [&](int li, ConvArgs& a) -> int {
            const Layer& l = u->layers[li];
            const auto& t = u->tl[li];
            memset(&a, 0, sizeof(a));
            if (int rc = fill_geom(l, B, a)) return rc;
            a.src1 = tensor(t.src1_l); a.src2 = tensor(t.src2_l);
            a.wp = packed + u->params[l.w].off;
            a.bias = packed + u->params[l.b].off;
            a.gamma = l.gamma >= 0 ? packed + u->params[l.gamma].off : nullptr;
            a.beta = l.beta >= 0 ? packed + u->params[l.beta].off : nullptr;
            if (l.tb_off >= 0) { a.tbias = ws + w.tb + l.tb_off; a.tb_stride = u->tt_row; }
            a.res = tensor(t.res_l);
            a.dst = out(li);
            a.pre = l.epi == EPI_GN_MISH ? pre(li) : nullptr;
            return 0;
        };
        const Layer& l = u->layers[i];
        ConvArgs a;
        if (int rc = layer_args(i, a)) return rc;
        // blocks[0] and the same block's residual 1x1 convolution (the next layer; both read the block input) as ONE launch - the planning path's conv_pair_kernel
        // (round 6: two launches of ~4.8 us less per pass on the four-level network; MPDX_TRAIN_PAIR_FWD=0: one launch per layer)
        static const bool pair_fwd_off = getenv("MPDX_TRAIN_PAIR_FWD") && atoi(getenv("MPDX_TRAIN_PAIR_FWD")) == 0;
        int MT = 0, NT = 0;
        if (!pair_fwd_off && !masked && i + 1 < n && !(fused_fwd && u->owner[i + 1] >= 0 && ((fused_mask(B) >> u->owner[i + 1]) & 1u)) && u->tl[i + 1].src1_l == u->tl[i].src1_l &&
            u->tl[i + 1].src2_l == u->tl[i].src2_l && pair_tile(l, u->layers[i + 1], B, MT, NT)) {
            ConvArgs a2;
            if (int rc = layer_args(i + 1, a2)) return rc;
            a.n_tiles_n = a2.n_tiles_n = (int)(((long)B * l.L_out + NT - 1) / NT);
            const int rc = launch_conv_pair(MT, NT, a, a2, l, u->layers[i + 1], st);
            if (rc < 0) return rc;
            if (rc == 1) { ++i; continue; }
        }
        if (int rc = launch_layer(l, a, B, st)) return rc;
    }
    {   // final_conv[1] -> eps (the network output), hard conditions, loss value and its gradient
        FinalArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.h = out(n - 1);
        fa.w = packed + u->params[u->pidx.at("final_conv.1.weight")].off;
        fa.bias = packed + u->params[u->pidx.at("final_conv.1.bias")].off;
        fa.out = eps; fa.mode = 0; fa.n_per_ctx = 1;
        fa.B = B; fa.H = H; fa.D = D; fa.C = c.unet_input_dim;
        fa.Hc = masked ? Hc : 0;
        if (!eps_done) launch_final_step(fa, st);
        const float* target = predict_epsilon ? noise : x_start;
        // loss value + dE + the gradient wrt final_conv[0]'s output (back through final_conv[1]) in one launch
        {   // train_loss_kernel: state_dim <= 16 in its plain form (padded containers, odd widths), <= 32 in the LDS-staged form (1024 % C == 0, D C <= 1024)
            const bool staged = !masked && fa.C > 0 && 1024 % fa.C == 0 && 1024 / fa.C <= 64 && D * fa.C <= 1024 && (1024 / fa.C) * D <= 1024;
            if (fa.C < D || D > 32 || (D > 16 && !staged && !masked))   // (the padded-container form loops over d: any D)
                return fail(MPDX_E_INVALID, "training: unet_input_dim %d / state_dim %d (the loss kernel takes state_dim <= 32 <= unet_input_dim)", fa.C, D);
        }
        const size_t tot = (size_t)B * Hc * fa.C;
        hipLaunchKernelGGL(train_loss_kernel, dim3((unsigned)std::min<size_t>((tot + 1023) / 1024, 1024) + 16), dim3(1024), 0, st, (const float*)eps, target, weights_hd,
                           hard_start, hard_goal, l1, loss_scale, dE, flat + u->params[u->pidx.at("final_conv.1.weight")].foff, ws + w.grad0 + (size_t)(n - 1) * w.slotB,
                           B, H, D, fa.C, loss_out, (double*)(ws + w.lossp), (unsigned*)(ws + w.ticket) + 1, masked ? Hc : 0);   // (ticket word 1: zeroed by the pass's first launch)
    }
    HIP_TRY(hipGetLastError());

    // ---- backward
    float* const part = ws + w.wpart;
    float* const rpart = ws + w.rpart;
    Deferred df;
    df.on = w.deferred; df.ws = ws; df.grads = grads_flat; df.wcur = w.wparts; df.pcur = w.pvecs;
    df.red.ws = ws; df.red.grad = grads_flat; df.red.n = 0;
    df.col.ws = ws; df.col.grad = grads_flat; df.col.n = 0;
    std::vector<WgradJob> lone;   // deferred weight-gradient GEMMs without a dgrad convolution: launched together, three per launch
    {   // final_conv[1]
        const int C = c.unet_input_dim;
        const int wi = u->pidx.at("final_conv.1.weight"), bi = u->pidx.at("final_conv.1.bias");
        const size_t rows = (size_t)B * Hc;   // (dE in the container layout: its rows behind the horizon are zero)
        // (grd(n - 1) = dE W was written by train_loss_kernel)
        WgradJob fj;
        if (int rc = make_wgrad(dE, Hc, D, 0, D, out(n - 1), Hc, C, 0, C, 1, 0, 1, B, part, gflat(wi), C, 0, &df, fj)) return rc;
        const bool fb = attach_bias(fj, &df, gflat(bi), false);
        if (fj.deferred) lone.push_back(fj);   // rides with the other GEMMs that have no dgrad convolution (one launch behind the loop)
        else run_wgrad(fj, st);
        if (!fb) launch_rowsum(dE, rows, D, rpart, gflat(bi), st, &df);
    }
    // A Conv1dBlock j whose output feeds exactly one k5 convolution i (blocks[0] -> blocks[1] of a ResidualTemporalBlock) gets its
    // Mish + GroupNorm backward as the EPILOGUE of i's input-gradient convolution (EPI_GN_BWD): one launch less per residual block.
    // With several consumers the LAST one in backward order (the lowest layer index) carries the epilogue; the others have added
    // their gradients to grd(j) by then.
    std::vector<int> first_consumer(n, n);
    for (int i = n - 1; i >= 0; --i)
        for (int sl : {u->tl[i].src1_l, u->tl[i].src2_l, u->tl[i].res_l})
            if (sl >= 0) first_consumer[sl] = i;
    static const bool gnfuse_off = getenv("MPDX_TRAIN_GN_FUSE") && atoi(getenv("MPDX_TRAIN_GN_FUSE")) == 0;   // dev A/B switch
    // round 6 EXPERIMENT, OFF by default (MPDX_TRAIN_CHAIN: 0 off (default), 1 auto, 16 / 32 / 64: the smallest level length that chains): the launches of
    // the outer levels' backward chain COLLECTED and run as one bwd_chain_kernel launch per run of chainable steps.  Bit-identical gradients, but SLOWER
    // than the launches it replaces (batch 32: 0.608 -> 0.640 ms with L >= 32, 0.749 ms with L >= 16; batch 128 x D = 14: 0.894 -> 0.921 / 1.002 ms;
    // profiles/r06_train_chain_ab.txt): under a hipGraph a launch boundary costs ~1 us, a chain step still pays the body's own latency chain (operands
    // through L2, staging, K-split reduction, epilogue loads) AND runs a layer's channel tiles one after the other on ONE CU instead of side by side on
    // several.  What would pay is keeping the gradients in LDS between the steps (the forward programs' design) - not built.
    static const int chain_env = getenv("MPDX_TRAIN_CHAIN") ? atoi(getenv("MPDX_TRAIN_CHAIN")) : 0;
    static const int chain_max_b = getenv("MPDX_TRAIN_CHAIN_MAX_B") ? atoi(getenv("MPDX_TRAIN_CHAIN_MAX_B")) : 256;
    ChainBuilder chain;
    chain.B = B; chain.st = st;
    chain.on = chain_env != 0 && !masked && df.on && B <= chain_max_b;
    const int chain_min_L = chain_env >= 16 ? chain_env : (B >= 64 ? 16 : 32);
    std::vector<char> du_ready(n, 0);   // grd(j) already holds the gradient wrt layer j's CONVOLUTION output
    std::vector<char> written(n, 0);    // grd(j) has been written in this pass (launches execute in the order they are enqueued here)
    written[n - 1] = 1;                 // train_loss_kernel above
    auto first_write = [&](int j) { const bool f = !written[j]; written[j] = 1; return f; };
    // round 6: the backward pass of downs[0..2] as ONE whole-trajectory program (fused_bwd.hpp; MPDX_TRAIN_BWD_PROG=0 switches it off)
    static const int prog_env = getenv("MPDX_TRAIN_BWD_PROG") ? atoi(getenv("MPDX_TRAIN_BWD_PROG")) : 1;
    static const int prog_max_b = getenv("MPDX_TRAIN_BWD_PROG_MAX_B") ? atoi(getenv("MPDX_TRAIN_BWD_PROG_MAX_B")) : 512;
    const int down_variant = (prog_env != 0 && df.on && !masked && B <= prog_max_b && w.total < ((size_t)1 << 31)) ? bwd_down_applicable(u) : 0;
    const bool prog_down_on = down_variant != 0;
    const int dn_last = down_variant == 3 ? 20 : (down_variant == 2 ? 16 : 17);   // the program covers layers [0, dn_last]
    auto run_down_program = [&]() -> int {   // layers [0, 18) (three-level network: [0, 17)): returns 0 ok, < 0 error, 1 not applicable here (the per-layer path takes over)
        const int n_gn = down_variant == 3 ? 16 : 12;   // GroupNorm ops (three column-sum entries each); dn_last + 1 weight-gradient jobs
        if (!written[dn_last] || df.red.n + dn_last + 1 > 96 || df.col.n + n_gn * 3 + 6 > 120) return 1;
        if (down_variant == 3 && !written[16]) return 1;   // (the skip connection's gradient, an addend of op M5)
        static const int late_div_env = getenv("MPDX_WGRAD_LATE_DIV") ? std::max(1, atoi(getenv("MPDX_WGRAD_LATE_DIV"))) : 0;
        static const int small_mul = getenv("MPDX_WGRAD_PROG_MUL") ? atoi(getenv("MPDX_WGRAD_PROG_MUL")) : 1;   // batch < 64: split multiplier of the program layers' weight gradients
        const int sdiv = late_div_env ? late_div_env : (B >= 64 ? (B <= 128 ? 8 : 4) : (small_mul > 1 ? -std::min(small_mul, 4) : 4));   // (batch < 64: 4 - measured with the host out of the way, profiles/r06_train_b32_split_ab.txt)
        const BwdProgLayout lay = bwd_down_layout();
        BwdArgs a;
        memset(&a, 0, sizeof(a));
        a.packedT = packedT; a.flat = flat; a.ws = ws; a.B = B; a.dT_stride = u->tt_row; a.stat_off = lay.stat_off;
        a.dbg = getenv("MPDX_BWD_DBG") ? atoi(getenv("MPDX_BWD_DBG")) : 0;
        auto goff = [&](const float* p) { return (int)(p - ws); };
        enum { IN = 0, GB = 1, DUA = 2, DUB = 3, GA = 4 };
        auto rs4_of = [](int C) { return C / 4 + kBwdPad4; };
        a.gin = grd(17); a.in_L = 8; a.in_C = 128; a.in_stuff = 1; a.in_off4 = lay.off4[IN]; a.in_rs4 = rs4_of(128);
        if (down_variant >= 2) { a.gin = nullptr; a.in_L = 0; a.in_stuff = 0; }   // (no staged input: the first op takes grd(16) / grd(20) as its global addend)
        int nop = 0;
        auto gn_part = [&](int li, BwdOp& op) {   // the lower Conv1dBlock `li`: its GroupNorm input, parameters and the partial-sum rows of its gamma / beta / bias gradients
            const Layer& lj = u->layers[li];
            op.pre_g = goff(pre(li));
            op.gamma_f = (int)u->params[lj.gamma].foff; op.beta_f = (int)u->params[lj.beta].foff;
            op.part_g = (int)df.pcur;
            const int prm[3] = {lj.gamma, lj.beta, lj.b};
            for (int k = 0; k < 3; ++k) {
                auto& e = df.col.e[df.col.n++];
                e.part = df.pcur + (size_t)k * B * lj.cout; e.out = u->params[prm[k]].foff; e.rows = B; e.C = lj.cout;
            }
            df.pcur += (size_t)3 * B * lj.cout;
            op.dT_g = lj.tb_off >= 0 ? (int)(w.dT + lj.tb_off) : -1;
        };
        if (down_variant == 3) {   // M1 .. M4: the two middle blocks (fused_bwd.hpp bwd_down_mid_geom); M5 = the level loop's first op below
            const int r4 = rs4_of(128);
            auto mid_op = [&](int nc16) -> BwdOp& {
                BwdOp& op = a.ops[nop++];
                memset(&op, 0, sizeof(op));
                op.shape = bwd_shape_id(CONV_S1, 5, nc16, 0, 128, 16, 1);
                op.add_off4 = -1; op.gadd = -1; op.gy_off4 = -1; op.gy_g = -1; op.dst_off4 = -1; op.out_g = -1; op.part_g = -1; op.dT_g = -1;
                return op;
            };
            {   // M1: G(layer 20 out) from the up program -> GB (the identity residual of mid_block2 passes it on to layer 18's output); GroupNorm backward of layer 20
                BwdOp& op = mid_op(0);
                op.gadd = goff(grd(20));
                op.gy_off4 = lay.off4[GB]; op.gy_rs4 = r4;
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(20));
                gn_part(20, op);
            }
            {   // M2: dgrad of layer 20 -> G(19 out) (time bias), GroupNorm backward of 19
                BwdOp& op = mid_op(8);
                op.src_off4 = lay.off4[DUA]; op.src_rs4 = r4; op.wbase = (int)u->tl[20].dgrad_woff;
                op.dst_off4 = lay.off4[DUB]; op.dst_rs4 = r4; op.out_g = goff(grd(19));
                gn_part(19, op);
            }
            {   // M3: dgrad of 19 + GB -> G(18 out) -> GA; GroupNorm backward of 18
                BwdOp& op = mid_op(8);
                op.src_off4 = lay.off4[DUB]; op.src_rs4 = r4; op.wbase = (int)u->tl[19].dgrad_woff;
                op.add_off4 = lay.off4[GB]; op.add_rs4 = r4;
                op.gy_off4 = lay.off4[GA]; op.gy_rs4 = r4;
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(18));
                gn_part(18, op);
            }
            {   // M4: dgrad of 18 -> G(17 out) (time bias), GroupNorm backward of 17
                BwdOp& op = mid_op(8);
                op.src_off4 = lay.off4[DUA]; op.src_rs4 = r4; op.wbase = (int)u->tl[18].dgrad_woff;
                op.dst_off4 = lay.off4[DUB]; op.dst_rs4 = r4; op.out_g = goff(grd(17));
                gn_part(17, op);
            }
        }
        for (int k = 2; k >= 0; --k) {
            const int C = 32 << k, Lk = 64 >> k, b0 = 6 * k, r4 = rs4_of(C);
            auto base_op = [&](int shape_ks, int nc16, int ncr, int cout, int gn) -> BwdOp& {
                BwdOp& op = a.ops[nop++];
                memset(&op, 0, sizeof(op));
                op.shape = bwd_shape_id(CONV_S1, shape_ks, nc16, ncr, cout, Lk, gn);
                op.add_off4 = -1; op.gadd = -1; op.gy_off4 = -1; op.gy_g = -1; op.dst_off4 = -1; op.out_g = -1; op.part_g = -1; op.dT_g = -1;
                return op;
            };
            if (k == 2 && down_variant == 3) {   // M5: dgrad of layer 17 + GA (mid_block1's identity residual) + the skip connection's gradient -> G(16 out) -> GB; GroupNorm backward of 16
                BwdOp& op = base_op(5, C / 16, 0, C, 1);
                op.src_off4 = lay.off4[DUB]; op.src_rs4 = r4; op.wbase = (int)u->tl[17].dgrad_woff;
                op.add_off4 = lay.off4[GA]; op.add_rs4 = r4;
                op.gadd = goff(grd(b0 + 4));
                op.gy_off4 = lay.off4[GB]; op.gy_rs4 = r4;
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 4));
                gn_part(b0 + 4, op);
            } else if (k == 2 && down_variant == 2) {   // P1 of an innermost level (no Downsample1d): G(b1.1 out) is what the per-layer path accumulated in grd(16)
                BwdOp& op = base_op(5, 0, 0, C, 1);
                op.gadd = goff(grd(b0 + 4));
                op.gy_off4 = lay.off4[GB]; op.gy_rs4 = r4;
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 4));
                gn_part(b0 + 4, op);
            } else {   // P1: dgrad of the Downsample1d (its dU zero-stuffed in IN) + the skip connection's gradient -> G(b1.1 out) -> GB; GroupNorm backward of b1.1
                BwdOp& op = base_op(3, C / 16, 0, C, 1);
                op.src_off4 = lay.off4[IN]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 5].dgrad_woff;
                if (written[b0 + 4]) op.gadd = goff(grd(b0 + 4));
                op.gy_off4 = lay.off4[GB]; op.gy_rs4 = r4;
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 4));
                gn_part(b0 + 4, op);
            }
            {   // P2: dgrad of b1.1 -> G(b1.0 out) (its time-bias gradient), GroupNorm backward of b1.0
                BwdOp& op = base_op(5, C / 16, 0, C, 1);
                op.src_off4 = lay.off4[DUA]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 4].dgrad_woff;
                op.dst_off4 = lay.off4[DUB]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 3));
                gn_part(b0 + 3, op);
            }
            {   // P3: dgrad of b1.0 + the identity residual's G (GB) -> G(b0.1 out) -> GA + the residual 1x1's dY (global); GroupNorm backward of b0.1
                BwdOp& op = base_op(5, C / 16, 0, C, 1);
                op.src_off4 = lay.off4[DUB]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 3].dgrad_woff;
                op.add_off4 = lay.off4[GB]; op.add_rs4 = r4;
                op.gy_off4 = lay.off4[GA]; op.gy_rs4 = r4; op.gy_g = goff(grd(b0 + 1));
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 2));
                gn_part(b0 + 2, op);
            }
            {   // P4: dgrad of b0.1 -> G(b0.0 out) (time bias), GroupNorm backward of b0.0
                BwdOp& op = base_op(5, C / 16, 0, C, 1);
                op.src_off4 = lay.off4[DUA]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 2].dgrad_woff;
                op.dst_off4 = k > 0 ? lay.off4[DUB] : -1; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 0));
                gn_part(b0 + 0, op);
            }
            if (k > 0) {   // P5: dgrad of b0.0 + the residual 1x1's (from GA) -> dU of the level above's Downsample1d, zero-stuffed into IN
                const int Cp = C / 2;
                BwdOp& op = base_op(5, C / 16, C / 16, Cp, 0);
                op.src_off4 = lay.off4[DUB]; op.src_rs4 = r4;
                op.rsrc_off4 = lay.off4[GA]; op.rsrc_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 0].dgrad_woff; op.rwbase = (int)u->tl[b0 + 1].dgrad_woff;
                op.dst_off4 = lay.off4[IN]; op.dst_rs4 = rs4_of(Cp); op.dst_mode = 1; op.out_g = goff(grd(b0 - 1));
            }
        }
        a.nops = nop;
        for (int k = 0; k < nop; ++k)
            if (a.ops[k].shape < 0) return fail(MPDX_E_INVALID, "backward program: op %d has no shape", k);
        const bool last = down_variant == 2, mid = down_variant == 3;
        bool is_static = nop == (mid ? BwdSeqDown3Mid::N : BwdSeqDown3::N);
        for (int k = 0; k < nop && is_static; ++k)
            is_static = a.ops[k].shape == (mid ? BwdSeqDown3Mid::ids[k] : (last ? BwdSeqDown3Last::ids[k] : BwdSeqDown3::ids[k])) &&
                        bwd_geom_matches(a.ops[k], mid ? bwd_down_mid_geom(k) : bwd_down_geom(k, last), a.ops[k].shape == 2 || a.ops[k].shape == 5);
        if (!is_static) return fail(MPDX_E_STATE, "backward program (down): the layout differs from the static program's table");
        const void* kern = mid ? (const void*)fused_bwd_program_kernel<BwdSeqDown3Mid> : (last ? (const void*)fused_bwd_program_kernel<BwdSeqDown3Last> : (const void*)fused_bwd_program_kernel<BwdSeqDown3>);
        if (int rc = raise_lds_limit(kern)) return rc;
        if (mid) hipLaunchKernelGGL(fused_bwd_program_kernel<BwdSeqDown3Mid>, dim3(B), dim3(kFusedThreads), lay.lds_bytes, st, a);
        else if (last) hipLaunchKernelGGL(fused_bwd_program_kernel<BwdSeqDown3Last>, dim3(B), dim3(kFusedThreads), lay.lds_bytes, st, a);
        else hipLaunchKernelGGL(fused_bwd_program_kernel<BwdSeqDown3>, dim3(B), dim3(kFusedThreads), lay.lds_bytes, st, a);
        // the layers' weight gradients (their dY operands now sit in grd(i)) behind the chain; bias gradients of the three convolutions without GroupNorm
        for (int i = dn_last; i >= 0; --i) {
            const Layer& l = u->layers[i];
            const auto& t = u->tl[i];
            written[i] = 1; du_ready[i] = 1;
            const int sb = l.mode == CONV_DOWN ? 2 : 1, ob = l.mode == CONV_DOWN ? -1 : -(l.ks / 2);
            WgradJob j;
            if (int rc = make_wgrad(grd(i), l.L_out, l.cout, 0, l.cout, tensor(t.src1_l), l.L_in, l.c1, 0, l.c1, sb, ob, l.ks, B, part, gflat(l.w), l.c1, 0, &df, j, sdiv)) return rc;
            if (!j.deferred) return fail(MPDX_E_STATE, "backward program: no partial-sum storage left for layer %d", i);
            if (l.epi != EPI_GN_MISH && !attach_bias(j, &df, gflat(l.b), false)) return fail(MPDX_E_STATE, "backward program: no column-sum slot left for layer %d", i);
            lone.push_back(j);
        }
        return 0;
    };
    const int up_first = (prog_env != 0 && prog_env != 2 && df.on && !masked && B <= prog_max_b && w.total < ((size_t)1 << 31)) ? bwd_up_applicable(u) : -1;   // (2: the down program only)
    const bool prog_up_on = up_first >= 0;
    const int up_fi = n - 1;   // final_conv[0]
    auto run_up_program = [&]() -> int {   // layers [up_first, n) = [33, 46) ([21, 34) with three levels): final_conv[0] and the two outer up levels; 0 ok, < 0 error, 1 not applicable here
        if (!written[up_fi] || df.red.n + 17 > 96 || df.col.n + 9 * 3 + 4 > 120) return 1;
        static const int late_div_env = getenv("MPDX_WGRAD_LATE_DIV") ? std::max(1, atoi(getenv("MPDX_WGRAD_LATE_DIV"))) : 0;
        static const int small_mul = getenv("MPDX_WGRAD_PROG_MUL") ? atoi(getenv("MPDX_WGRAD_PROG_MUL")) : 1;
        const int sdiv = late_div_env ? late_div_env : (B >= 64 ? (B <= 128 ? 8 : 4) : (small_mul > 1 ? -std::min(small_mul, 4) : 4));   // (batch < 64: 4 - measured with the host out of the way, profiles/r06_train_b32_split_ab.txt)
        const BwdProgLayout lay = bwd_down_layout();   // (the same five slots: the largest buffer here is 68 rows x 36 floats = 612 float4)
        BwdArgs a;
        memset(&a, 0, sizeof(a));
        a.packedT = packedT; a.flat = flat; a.ws = ws; a.B = B; a.dT_stride = u->tt_row; a.stat_off = lay.stat_off;
        a.dbg = getenv("MPDX_BWD_DBG") ? atoi(getenv("MPDX_BWD_DBG")) : 0;
        auto goff = [&](const float* p) { return (int)(p - ws); };
        enum { IN = 0, GB = 1, DUA = 2, DUB = 3, GA = 4 };
        auto rs4_of = [](int C) { return C / 4 + kBwdPad4; };
        auto part3 = [&](int li, int& part_g) {   // partial-sum rows + column-sum entries of a Conv1dBlock's gamma / beta / bias gradients
            const Layer& lj = u->layers[li];
            part_g = (int)df.pcur;
            const int prm[3] = {lj.gamma, lj.beta, lj.b};
            for (int k = 0; k < 3; ++k) {
                auto& e = df.col.e[df.col.n++];
                e.part = df.pcur + (size_t)k * B * lj.cout; e.out = u->params[prm[k]].foff; e.rows = B; e.C = lj.cout;
            }
            df.pcur += (size_t)3 * B * lj.cout;
        };
        a.gin = nullptr; a.in_L = 64; a.in_C = 32; a.in_stuff = 0; a.in_off4 = lay.off4[IN]; a.in_rs4 = rs4_of(32);   // (no staged input: U0 reads the loss kernel's gradient itself)
        int nop = 0;
        auto new_op = [&](int mode, int ks, int nc16, int ncr, int cout, int L, int gn) -> BwdOp& {
            BwdOp& op = a.ops[nop++];
            memset(&op, 0, sizeof(op));
            op.shape = bwd_shape_id(mode, ks, nc16, ncr, cout, L, gn);
            op.add_off4 = -1; op.gadd = -1; op.gy_off4 = -1; op.gy_g = -1; op.dst_off4 = -1; op.out_g = -1; op.part_g = -1; op.dT_g = -1;
            return op;
        };
        auto gn_part = [&](int li, BwdOp& op) {
            const Layer& lj = u->layers[li];
            op.pre_g = goff(pre(li));
            op.gamma_f = (int)u->params[lj.gamma].foff; op.beta_f = (int)u->params[lj.beta].foff;
            part3(li, op.part_g);
            op.dT_g = lj.tb_off >= 0 ? (int)(w.dT + lj.tb_off) : -1;
        };
        {   // U0: final_conv[0]'s Mish + GroupNorm backward on the loss kernel's gradient (an op without a convolution: the gradient is its global addend);
            // dU -> IN and, in place, grd(45) (the operand of final_conv[0]'s weight gradient)
            BwdOp& op = new_op(CONV_S1, 5, 0, 0, 32, 64, 1);
            op.gadd = goff(grd(up_fi));
            op.dst_off4 = lay.off4[IN]; op.dst_rs4 = rs4_of(32); op.out_g = goff(grd(up_fi));
            gn_part(up_fi, op);
        }
        {   // U1: dgrad of final_conv[0] -> dU of ups[2]'s Upsample1d (64 positions)
            BwdOp& op = new_op(CONV_S1, 5, 2, 0, 32, 64, 0);
            op.src_off4 = lay.off4[IN]; op.src_rs4 = rs4_of(32);
            op.wbase = (int)u->tl[up_fi].dgrad_woff;
            op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = rs4_of(32); op.out_g = goff(grd(up_fi - 1));
        }
        const int bases[2] = {up_fi - 6, up_fi - 12}, Cs[2] = {32, 64}, Ls[2] = {32, 16}, skips[2] = {10, 16};
        int src_slot = DUA;   // where the level's Upsample1d dU sits (2 L positions)
        for (int k = 0; k < 2; ++k) {
            const int b0 = bases[k], C = Cs[k], Lk = Ls[k], r4 = rs4_of(C);
            {   // dgrad of the Upsample1d (its 5-tap pack at stride 2) -> G(b1.1 out) -> GB; GroupNorm backward of b1.1
                BwdOp& op = new_op(CONV_DOWN, 5, C / 16, 0, C, Lk, 1);
                op.src_off4 = lay.off4[src_slot]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 5].dgrad_woff;
                op.gy_off4 = lay.off4[GB]; op.gy_rs4 = r4;
                op.dst_off4 = lay.off4[DUB]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 4));
                gn_part(b0 + 4, op);
            }
            {   // dgrad of b1.1 -> G(b1.0 out) (time bias), GroupNorm backward of b1.0
                BwdOp& op = new_op(CONV_S1, 5, C / 16, 0, C, Lk, 1);
                op.src_off4 = lay.off4[DUB]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 4].dgrad_woff;
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 3));
                gn_part(b0 + 3, op);
            }
            {   // dgrad of b1.0 + the identity residual's G -> G(b0.1 out) -> GA + the residual 1x1's dY; GroupNorm backward of b0.1
                BwdOp& op = new_op(CONV_S1, 5, C / 16, 0, C, Lk, 1);
                op.src_off4 = lay.off4[DUA]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 3].dgrad_woff;
                op.add_off4 = lay.off4[GB]; op.add_rs4 = r4;
                op.gy_off4 = lay.off4[GA]; op.gy_rs4 = r4; op.gy_g = goff(grd(b0 + 1));
                op.dst_off4 = lay.off4[DUB]; op.dst_rs4 = r4; op.out_g = goff(grd(b0 + 2));
                gn_part(b0 + 2, op);
            }
            {   // dgrad of b0.1 -> G(b0.0 out) (time bias), GroupNorm backward of b0.0
                BwdOp& op = new_op(CONV_S1, 5, C / 16, 0, C, Lk, 1);
                op.src_off4 = lay.off4[DUB]; op.src_rs4 = r4;
                op.wbase = (int)u->tl[b0 + 2].dgrad_woff;
                op.dst_off4 = lay.off4[DUA]; op.dst_rs4 = r4; op.out_g = goff(grd(b0));
                gn_part(b0, op);
            }
            // dgrad of b0.0 + the residual 1x1's (from GA): the gradient of the channel concat [x | skip], one op per half (2 C channels each)
            const int nblk = (C / 16) * 5, ncr = C / 16, rows_half = 2 * C / 16;
            for (int half = 0; half < 2; ++half) {
                BwdOp& op = new_op(CONV_S1, 5, C / 16, C / 16, 2 * C, Lk, 0);
                op.src_off4 = lay.off4[DUA]; op.src_rs4 = r4;
                op.rsrc_off4 = lay.off4[GA]; op.rsrc_rs4 = r4;
                op.wbase = (int)u->tl[b0].dgrad_woff + half * rows_half * nblk * 256;
                op.rwbase = (int)u->tl[b0 + 1].dgrad_woff + half * rows_half * ncr * 256;
                if (half == 0) {   // x: the level below's Upsample1d output (its dU); the next level of this program reads it from LDS
                    op.out_g = goff(grd(b0 - 1));
                    if (k == 0) { op.dst_off4 = lay.off4[IN]; op.dst_rs4 = rs4_of(2 * C); }
                } else op.out_g = goff(grd(skips[k]));   // the skip connection's gradient (first writer: stored)
            }
            src_slot = IN;
        }
        a.nops = nop;
        for (int k = 0; k < nop; ++k)
            if (a.ops[k].shape < 0) return fail(MPDX_E_INVALID, "backward program (up): op %d has no shape", k);
        bool is_static = nop == BwdSeqUp2::N;
        for (int k = 0; k < nop && is_static; ++k) is_static = a.ops[k].shape == BwdSeqUp2::ids[k] && bwd_geom_matches(a.ops[k], bwd_up_geom(k), a.ops[k].shape == 11 || a.ops[k].shape == 14);
        if (!is_static) return fail(MPDX_E_STATE, "backward program (up): the layout differs from the static program's table");
        if (int rc = raise_lds_limit((const void*)fused_bwd_program_kernel<BwdSeqUp2>)) return rc;
        hipLaunchKernelGGL(fused_bwd_program_kernel<BwdSeqUp2>, dim3(B), dim3(kFusedThreads), lay.lds_bytes, st, a);
        written[up_first - 1] = written[16] = written[10] = 1;
        for (int i = up_fi; i >= up_first; --i) {
            const Layer& l = u->layers[i];
            const auto& t = u->tl[i];
            written[i] = 1; du_ready[i] = 1;
            const int Cin = l.c1 + l.c2;
            WgradJob jb[2];
            int nj = 0;
            if (l.mode == CONV_UPT) {
                if (int rc = make_wgrad(tensor(t.src1_l), l.L_in, l.c1, 0, l.c1, grd(i), l.L_out, l.cout, 0, l.cout, 2, -1, 4, B, part, gflat(l.w), l.cout, 0, &df, jb[nj++], sdiv)) return rc;
            } else {
                const int ob = -(l.ks / 2);
                if (int rc = make_wgrad(grd(i), l.L_out, l.cout, 0, l.cout, tensor(t.src1_l), l.L_in, l.c1, 0, l.c1, 1, ob, l.ks, B, part, gflat(l.w), Cin, 0, &df, jb[nj++], sdiv)) return rc;
                if (l.c2 > 0)
                    if (int rc = make_wgrad(grd(i), l.L_out, l.cout, 0, l.cout, tensor(t.src2_l), l.L_in, l.c2, 0, l.c2, 1, ob, l.ks, B, part, gflat(l.w), Cin, l.c1, &df, jb[nj++], sdiv)) return rc;
            }
            for (int k = 0; k < nj; ++k) if (!jb[k].deferred) return fail(MPDX_E_STATE, "backward program: no partial-sum storage left for layer %d", i);
            if (l.epi != EPI_GN_MISH && !attach_bias(jb[0], &df, gflat(l.b), l.mode == CONV_UPT)) return fail(MPDX_E_STATE, "backward program: no column-sum slot left for layer %d", i);
            for (int k = 0; k < nj; ++k) lone.push_back(jb[k]);
        }
        return 0;
    };
    bool ran_up = false, ran_down = false;
    // round 6: the dgrad launch of a ResidualTemporalBlock's blocks[1] (with the GroupNorm backward of blocks[0] in its epilogue) WAITS one layer for the block's
    // residual 1x1 convolution (the next layer in backward order): its 1x1 dgrad - and at batch < 48 both layers' weight-gradient blocks - ride on the same
    // launch (BwdPairArgs::cd2): one launch less per such block.  MPDX_TRAIN_PAIR_RES=0: one launch per layer as before
    static const bool pair_res_off = getenv("MPDX_TRAIN_PAIR_RES") && atoi(getenv("MPDX_TRAIN_PAIR_RES")) == 0;
    struct Pending { bool on = false; int i_next = -1; Layer dg; ConvArgs a; WgradJob jobs[3]; int njobs = 0; } pend;
    auto flush_pending = [&]() -> int {
        if (!pend.on) return 0;
        pend.on = false;
        return launch_bwd_pair<5, true>(pend.dg, pend.a, B, pend.jobs, pend.njobs, st);
    };
    for (int i = n - 1; i >= 0; --i) {
        if (pend.on && i != pend.i_next)
            if (int rc = flush_pending()) return rc;
        if (prog_up_on && i == up_fi) {
            if (int rc = chain.flush()) return rc;
            const int rc = run_up_program();
            if (rc < 0 || rc > 1) return rc;
            if (rc == 0) { ran_up = true; i = up_first; continue; }   // layers [up_first, n) are done: on with the layer below
        }
        if (prog_down_on && i == dn_last) {
            if (int rc = chain.flush()) return rc;
            const int rc = run_down_program();
            if (rc < 0 || rc > 1) return rc;
            if (rc == 0) { ran_down = true; break; }   // layers [0, dn_last] are done
        }
        const Layer& l = u->layers[i];
        const auto& t = u->tl[i];
        const int Cin = l.c1 + l.c2;
        float* gy = grd(i);
        const float* dy = gy;   // gradient wrt the convolution output (after the GroupNorm/Mish backward for Conv1dBlocks)
        if (!written[i]) {   // nothing downstream of this layer carries a gradient: it is zero
            if (getenv("MPDX_DEBUG_TRAIN")) fprintf(stderr, "[mpdx] backward: layer %d %s has no gradient-carrying consumer (zeroed)\n", i, l.name.c_str());
            if (int rc = chain.flush()) return rc;
            HIP_TRY(hipMemsetAsync(gy, 0, w.slotB * sizeof(float), st));
            written[i] = 1;
        }
        if (l.epi == EPI_GN_MISH && !du_ready[i]) {
            GnBwdArgs g;
            memset(&g, 0, sizeof(g));
            if (t.res_l >= 0) { g.gres = grd(t.res_l); g.gres_store = first_write(t.res_l) ? 1 : 0; }
            g.gy = gy; g.pre = pre(i); g.gamma = flat + u->params[l.gamma].foff; g.beta = flat + u->params[l.beta].foff;
            g.du = ws + w.dU;
            g.pg = ws + w.pvec; g.pb = g.pg + (size_t)B * 512; g.pbias = g.pb + (size_t)B * 512;
            const bool dcol = df.on && df.col.n + 3 <= 120;
            if (dcol) {
                g.pg = ws + df.pcur; g.pb = g.pg + (size_t)B * l.cout; g.pbias = g.pb + (size_t)B * l.cout;
                const int prm[3] = {l.gamma, l.beta, l.b};
                for (int k = 0; k < 3; ++k) {
                    auto& e = df.col.e[df.col.n++];
                    e.part = df.pcur + (size_t)k * B * l.cout; e.out = u->params[prm[k]].foff; e.rows = B; e.C = l.cout;
                }
                df.pcur += (size_t)3 * B * l.cout;
            }
            if (l.tb_off >= 0) { g.dT = ws + w.dT + l.tb_off; g.dT_stride = u->tt_row; }
            g.B = B; g.L = l.L_out; g.C = l.cout; g.gs = l.gs; g.n_groups = l.cout / l.gs;
            { int k = 0; while ((1 << k) < l.gs) ++k; g.lg_gs = k; }
            if (l.cout > 512) return fail(MPDX_E_INVALID, "layer %s: more than 512 channels", l.name.c_str());
            const int re = l.gs * l.L_out, regions = B * g.n_groups;
            const dim3 ggrid((regions + 3) / 4);
            g.Lv = l.Lv_out;
            const bool mrows = l.Lv_out > 0 && l.Lv_out < l.L_out;   // a padded container: the general kernel carries the row mask
            const bool gn_chain = chain.on && dcol && !mrows && (re == 256 || re == 128) && g.n_groups <= 8 && l.L_out >= chain_min_L &&
                                  (l.L_out == 64 || l.L_out == 32 || l.L_out == 16);
            if (gn_chain) {   // a step of the chain; du IN PLACE (grd(i): it outlives the pass, so the layer's weight gradients can run behind the chains)
                g.du = gy;
                if (int rc = chain.add_gn(g, re == 256 ? 4 : 2)) return rc;
            } else {
            if (int rc = chain.flush()) return rc;
            // du IN PLACE for the two kernels whose body allows it (a lane reads its elements before it writes them): grd(i) outlives the pass, the shared
            // dU scratch does not - so this layer's weight gradients can run behind the chain too (dy == gy below).  Round 6: the one 256 -> 256 layer whose
            // GroupNorm backward is its own launch kept its weight-gradient blocks riding on its dgrad launch - 22.6 us against its six siblings' 12.5 at batch 128
            static const bool gn_inplace_off = getenv("MPDX_TRAIN_GN_INPLACE") && atoi(getenv("MPDX_TRAIN_GN_INPLACE")) == 0;
            if (!gn_inplace_off && !mrows && (re == 256 || re == 128)) g.du = gy;
            if (re == 256 && !mrows) hipLaunchKernelGGL(gn_mish_bwd_kernel<4>, ggrid, dim3(256), 0, st, g);
            else if (re == 128 && !mrows) hipLaunchKernelGGL(gn_mish_bwd_kernel<2>, ggrid, dim3(256), 0, st, g);
            else if (re == 256 && l.gs >= 4) hipLaunchKernelGGL((gn_mish_bwd_gen_kernel<4, 1>), ggrid, dim3(256), 0, st, g);
            else if (re == 128 && l.gs >= 2) hipLaunchKernelGGL((gn_mish_bwd_gen_kernel<2, 1>), ggrid, dim3(256), 0, st, g);
            // horizons other than 64 (power-of-two containers 16 ... 128): regions of 64 / 512 / 1024 / 2048 elements
            else if (re == 64) hipLaunchKernelGGL((gn_mish_bwd_gen_kernel<1, 1>), ggrid, dim3(256), 0, st, g);
            else if (re == 512 && l.gs >= 4) hipLaunchKernelGGL((gn_mish_bwd_gen_kernel<4, 2>), ggrid, dim3(256), 0, st, g);
            else if (re == 1024 && l.gs >= 4) hipLaunchKernelGGL((gn_mish_bwd_gen_kernel<4, 4>), ggrid, dim3(256), 0, st, g);
            else if (re == 2048 && l.gs >= 4) hipLaunchKernelGGL((gn_mish_bwd_gen_kernel<4, 8>), ggrid, dim3(256), 0, st, g);
            else return fail(MPDX_E_INVALID, "layer %s: GroupNorm region of %d elements (group of %d channels)", l.name.c_str(), re, l.gs);
            }
            ColsumArgs cs;
            memset(&cs, 0, sizeof(cs));
            cs.part[0] = g.pg; cs.out[0] = gflat(l.gamma);
            cs.part[1] = g.pb; cs.out[1] = gflat(l.beta);
            cs.part[2] = g.pbias; cs.out[2] = gflat(l.b);
            cs.B = B; cs.C = l.cout;
            if (!dcol) hipLaunchKernelGGL(colsum_kernel, dim3((l.cout + 63) / 64, 3), dim3(256), 0, st, cs);
            dy = g.du;
        }
        // weight gradient(s) and input gradient: everything below depends only on dy
        float* gw = gflat(l.w);
        WgradJob jobs[2];
        int njobs = 0;
        static const bool pair_off0 = getenv("MPDX_TRAIN_PAIR") && atoi(getenv("MPDX_TRAIN_PAIR")) == 0;
        static const int late_env0 = getenv("MPDX_TRAIN_WGRAD_LATE") ? atoi(getenv("MPDX_TRAIN_WGRAD_LATE")) : -1;
        // fewer batch splits for the weight gradients that run behind the chain (measured, profiles/r06_train_late_div_ab.txt: batch 128 x D = 14 0.898 / 0.84 / 0.82 /
        // 0.81 ms with 1 / 4 / 8 / 16; batch 512 2.027 / 1.94 / 1.96 / 2.01): 8 up to batch 128, 4 beyond
        static const int late_div_env = getenv("MPDX_WGRAD_LATE_DIV") ? std::max(1, atoi(getenv("MPDX_WGRAD_LATE_DIV"))) : 0;
        const int late_div = late_div_env ? late_div_env : (B < 64 ? 4 : (B <= 128 ? 8 : 4));
        // will this layer's weight gradients run behind the chain (decided below, once the jobs exist: the same conditions)?  Then with fewer batch splits.
        const bool late_cand = (late_env0 < 0 ? B >= 48 : late_env0 != 0) && t.need_dgrad && !pair_off0 && df.on && df.red.n + 2 <= 96 && bwd_pair_has_tile(t.dg, B) && dy == gy;
        const int sdiv = late_cand ? late_div : 1;
        if (l.mode == CONV_UPT) {
            if (int rc = make_wgrad(tensor(t.src1_l), l.L_in, l.c1, 0, l.c1, dy, l.L_out, l.cout, 0, l.cout, 2, -1, 4, B, part, gw, l.cout, 0, &df, jobs[njobs++], sdiv)) return rc;
        } else {
            const int sb = l.mode == CONV_DOWN ? 2 : 1, ob = l.mode == CONV_DOWN ? -1 : -(l.ks / 2);
            if (int rc = make_wgrad(dy, l.L_out, l.cout, 0, l.cout, tensor(t.src1_l), l.L_in, l.c1, 0, l.c1, sb, ob, l.ks, B, part, gw, Cin, 0, &df, jobs[njobs++], sdiv)) return rc;
            if (l.c2 > 0)
                if (int rc = make_wgrad(dy, l.L_out, l.cout, 0, l.cout, tensor(t.src2_l), l.L_in, l.c2, 0, l.c2, sb, ob, l.ks, B, part, gw, Cin, l.c1, &df, jobs[njobs++], sdiv)) return rc;
        }
        if (l.epi != EPI_GN_MISH) {   // bias gradient = channel sums of dY: rides on the first weight-gradient job, else its own two launches
            if (!attach_bias(jobs[0], &df, gflat(l.b), l.mode == CONV_UPT)) {
                if (int rc = chain.flush()) return rc;
                launch_rowsum(gy, (size_t)B * l.L_out, l.cout, rpart, gflat(l.b), st, &df);
            }
        }
        static const bool pair_off = getenv("MPDX_TRAIN_PAIR") && atoi(getenv("MPDX_TRAIN_PAIR")) == 0;
        // one launch for all of them needs every job on its own partial buffer (the deferred mode)
        const bool paired = t.need_dgrad && !pair_off && jobs[0].deferred && (njobs == 1 || jobs[1].deferred) && bwd_pair_has_tile(t.dg, B);
        // round 6 (MPDX_TRAIN_WGRAD_LATE, dev A/B switch): a layer's weight gradients leave the chain when their dU operand outlives the pass - it does
        // whenever it sits in the layer's own gradient slot (grd(i): written once, never recycled), not in the shared dU scratch of an un-fused
        // GroupNorm backward - and run with everybody else's in wgrad_multi_kernel behind the chain
        static const int late_env = getenv("MPDX_TRAIN_WGRAD_LATE") ? atoi(getenv("MPDX_TRAIN_WGRAD_LATE")) : -1;   // -1: by batch (measured: batch 32 no gain, 128 -3 %, 512 -4.6 %)
        const bool late_on = late_env < 0 ? B >= 48 : late_env != 0;   // (batch 48: 0.58 -> 0.543 ms, batch 32: within noise: profiles/r06_train_b32_late_ab.txt)
        static const bool resamp_fold_off_c = getenv("MPDX_TRAIN_RESAMPLE_FOLD") && atoi(getenv("MPDX_TRAIN_RESAMPLE_FOLD")) == 0;
        // this layer's input-gradient convolution as a step of the backward chain?
        const bool chain_d = chain.on && paired && dy == gy && ChainBuilder::conv_ok(t.dg, chain_min_L) && !resamp_fold_off_c &&
                             (l.mode != CONV_UPT || t.src1_l >= 0);
        bool late = false;
        if ((late_on || chain_d) && paired && dy == gy) {
            late = true;
            for (int k = 0; k < njobs; ++k) lone.push_back(jobs[k]);
            njobs = 0;
        }
        if (!paired)
            for (int k = 0; k < njobs; ++k) {
                if (!t.need_dgrad && !pair_off && jobs[k].deferred) lone.push_back(jobs[k]);
                else {
                    if (int rc = chain.flush()) return rc;
                    run_wgrad(jobs[k], st);
                }
            }
        (void)late;
        if (t.need_dgrad) {
            const Layer& dgl = t.dg;
            const float* din = dy;
            static const bool resamp_fold_off = getenv("MPDX_TRAIN_RESAMPLE_FOLD") && atoi(getenv("MPDX_TRAIN_RESAMPLE_FOLD")) == 0;   // dev A/B switch
            const bool fold = !resamp_fold_off;
            if (l.mode == CONV_DOWN && !fold) {
                if (int rc = chain.flush()) return rc;
                const size_t tot = (size_t)B * 2 * l.L_out * l.cout;
                hipLaunchKernelGGL(zero_stuff_kernel, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 2048)), dim3(256), 0, st, dy, ws + w.zst, B, l.L_out, l.cout);
                din = ws + w.zst;
            }
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            if (int rc = fill_geom(dgl, B, a)) return rc;
            a.src1 = din;
            if (l.mode == CONV_DOWN && fold) a.stuff = 1;   // the staging reads dy zero-stuffed (ConvArgs::stuff)
            a.wp = packedT + t.dgrad_woff;
            a.bias = ws + w.zeros;
            if (l.mode == CONV_UPT && fold && t.src1_l >= 0) {   // even output rows straight into the gradient of the layer's input (ConvArgs::decim)
                a.dst = grd(t.src1_l); a.decim = 1;
                if (!first_write(t.src1_l)) a.accum |= 1;
            } else if (l.mode == CONV_UPT) a.dst = ws + w.tmpX;   // full-resolution result, every second position is the gradient
            else {   // added straight into the gradient buffer(s) of the layer's input(s)
                a.dst = t.src1_l >= 0 ? grd(t.src1_l) : nullptr;
                if (t.src1_l >= 0 && !first_write(t.src1_l)) a.accum |= 1;
                if (l.c2 > 0) {
                    a.c_split = l.c1; a.dst2 = t.src2_l >= 0 ? grd(t.src2_l) : nullptr;
                    if (t.src2_l >= 0 && !first_write(t.src2_l)) a.accum |= 2;
                }
            }
            const int j = t.src1_l;
            bool gn_fused = false;
            if (paired && !masked && !gnfuse_off && ((l.mode == CONV_S1 && l.ks == 5) || (l.mode == CONV_DOWN && dgl.ks == 3)) && l.c2 == 0 && j >= 0 && j != n - 1 && first_consumer[j] == i && t.res_l != j &&
                !(prog_down_on && j == dn_last) &&   // (the down program's first op is that layer's GroupNorm backward: it wants G, not dU)
                u->layers[j].epi == EPI_GN_MISH && u->layers[j].cout == l.c1 && df.on && df.col.n + 3 <= 120) {
                const Layer& lj = u->layers[j];
                const int re = lj.gs * lj.L_out;
                if ((re == 256 || re == 128) && lj.L_out == dgl.L_out) {
                    Layer dg2 = dgl;
                    dg2.epi = EPI_GN_MISH; dg2.gs = lj.gs;
                    a.dst = grd(j); a.dst2 = nullptr; a.c_split = 0;   // (a.accum bit 0 as set above: the other consumers' gradients are in grd(j))
                    if (u->tl[j].res_l >= 0) { a.bw_gres = grd(u->tl[j].res_l); a.bw_gres_store = first_write(u->tl[j].res_l) ? 1 : 0; }
                    a.res = pre(j);
                    a.gamma = flat + u->params[lj.gamma].foff; a.beta = flat + u->params[lj.beta].foff;
                    a.gs = lj.gs; a.lg_gs = 0;
                    while ((1 << a.lg_gs) < lj.gs) ++a.lg_gs;
                    a.bw_pg = ws + df.pcur; a.bw_pb = a.bw_pg + (size_t)B * lj.cout; a.bw_pbias = a.bw_pb + (size_t)B * lj.cout;
                    const int prm[3] = {lj.gamma, lj.beta, lj.b};
                    for (int k = 0; k < 3; ++k) {
                        auto& e = df.col.e[df.col.n++];
                        e.part = df.pcur + (size_t)k * B * lj.cout; e.out = u->params[prm[k]].foff; e.rows = B; e.C = lj.cout;
                    }
                    df.pcur += (size_t)3 * B * lj.cout;
                    if (lj.tb_off >= 0) { a.bw_dT = ws + w.dT + lj.tb_off; a.bw_dT_stride = u->tt_row; }
                    if (chain_d) {
                        if (int rc = chain.add_conv(dg2, a, true)) return rc;
                    } else {
                        if (int rc = chain.flush()) return rc;
                        // is the next layer in backward order this block's residual 1x1 convolution?  Then this launch waits for it (see `pend`)
                        bool defer = false;
                        if (!pair_res_off && dgl.ks == 5 && i >= 1 && t.res_l == i - 1 && njobs <= 1) {
                            const Layer& r = u->layers[i - 1];
                            defer = r.mode == CONV_S1 && r.ks == 1 && r.epi == EPI_BIAS && u->tl[i - 1].need_dgrad && r.L_out == l.L_out && !(prog_down_on && i - 1 <= dn_last);
                        }
                        if (defer) {
                            pend.on = true; pend.i_next = i - 1; pend.dg = dg2; pend.a = a; pend.njobs = njobs;
                            for (int k = 0; k < njobs; ++k) pend.jobs[k] = jobs[k];
                        } else if (int rc = dgl.ks == 5 ? launch_bwd_pair<5, true>(dg2, a, B, jobs, njobs, st) : launch_bwd_pair<3, true>(dg2, a, B, jobs, njobs, st)) return rc;
                    }
                    du_ready[j] = 1;
                    gn_fused = true;
                }
            }
            if (gn_fused) {
            } else if (chain_d) {
                if (int rc = chain.add_conv(dgl, a, false)) return rc;
            } else if (paired) {
                if (int rc0 = chain.flush()) return rc0;
                int rc = kNoPair2;
                if (pend.on && dgl.ks == 1 && pend.njobs + njobs <= 3) {   // the residual 1x1's dgrad (and weight-gradient blocks) ride on the waiting blocks[1] launch
                    WgradJob all[3];
                    int na = 0;
                    for (int k = 0; k < pend.njobs; ++k) all[na++] = pend.jobs[k];
                    for (int k = 0; k < njobs; ++k) all[na++] = jobs[k];
                    rc = launch_bwd_pair<5, true>(pend.dg, pend.a, B, all, na, st, &dgl, &a);
                    if (rc != kNoPair2) pend.on = false;
                }
                if (rc == kNoPair2) {
                    if (int rc1 = flush_pending()) return rc1;
                } else if (rc) return rc;
                if (rc != kNoPair2) {
                } else
                if (dgl.ks == 5) rc = launch_bwd_pair<5>(dgl, a, B, jobs, njobs, st);
                else if (dgl.ks == 3) rc = launch_bwd_pair<3>(dgl, a, B, jobs, njobs, st);
                else rc = launch_bwd_pair<1>(dgl, a, B, jobs, njobs, st);
                if (rc) return rc;
            } else {
                if (int rc = chain.flush()) return rc;
                if (int rc = launch_layer(dgl, a, B, st)) return rc;
            }
            if (l.mode == CONV_UPT && t.src1_l >= 0 && !a.decim) if (int rc = chain.flush()) return rc;
            if (l.mode == CONV_UPT && t.src1_l >= 0 && !a.decim) launch_acc(grd(t.src1_l), ws + w.tmpX, B, l.L_in, l.c1, dgl.L_out, Cin, 0, 2, first_write(t.src1_l) ? 1 : 0, st);
        }
    }
    if (int rc = flush_pending()) return rc;
    if (int rc = chain.flush()) return rc;
    if (getenv("MPDX_DEBUG_TRAIN"))   // (tests/test_gpu_train.py reads this line: the programs must RUN on both networks the reference trains)
        fprintf(stderr, "[mpdx] backward programs: up %d (layers [%d, %d)), down %d (variant %d, layers [0, %d])\n", ran_up ? 1 : 0, up_first, n, ran_down ? 1 : 0, down_variant, dn_last);
    if (getenv("MPDX_DEBUG_TRAIN")) fprintf(stderr, "[mpdx] backward: %d chain launch(es) of %d steps, %zu weight-gradient jobs behind them\n", chain.launches, chain.steps, lone.size());
    {
        static const bool multi_off = getenv("MPDX_TRAIN_WGRAD_MULTI") && atoi(getenv("MPDX_TRAIN_WGRAD_MULTI")) == 0;   // dev A/B switch
        if (!multi_off) {
            if (int rc = launch_wgrads_multi(lone, st)) return rc;
        } else
            for (size_t k = 0; k < lone.size(); k += 3)
                if (int rc = launch_lone_wgrads(lone.data() + k, (int)std::min<size_t>(3, lone.size() - k), st)) return rc;
    }
    if (df.red.n) {
        int blocks = 0;
        for (int k = 0; k < df.red.n; ++k) {
            df.red.cstart[k] = blocks;
            auto& e = df.red.e[k];
            e.zsl = reduce_zsl(e);
            const size_t opb = 1024 / (size_t)std::max(1, e.zsl);   // outputs per block
            blocks += (int)(((size_t)e.M * e.N * e.KS + opb - 1) / opb);
        }
        df.red.cstart[df.red.n] = blocks;
        static const bool join_off = getenv("MPDX_TRAIN_REDUCE_JOIN") && atoi(getenv("MPDX_TRAIN_REDUCE_JOIN")) == 0;   // dev A/B switch
        if (df.col.n && !join_off) {   // the column sums ride on the same launch (side blocks behind the reduction's)
            ReduceColsumArgs rc;   // (8 KB of kernel arguments; the launch copies them)
            rc.red = df.red; rc.col = df.col; rc.n_red_blocks = blocks;
            hipLaunchKernelGGL(wgrad_reduce_colsum_kernel, dim3(blocks + 2 * df.col.n), dim3(256), 0, st, rc);
            df.col.n = 0;
        } else hipLaunchKernelGGL(wgrad_reduce_all_kernel, dim3(blocks), dim3(256), 0, st, df.red);
    }
    if (df.col.n) hipLaunchKernelGGL(colsum_all_kernel, dim3(2, df.col.n), dim3(256), 0, st, df.col);
    static const bool tail_join = getenv("MPDX_TIME_TAIL_SPLIT") && atoi(getenv("MPDX_TIME_TAIL_SPLIT")) == 0;   // dev A/B switch: the tail inside the launch
    tb.split_tail = tail_join ? 0 : 1;
    hipLaunchKernelGGL(time_bwd_all_kernel, dim3(B + (tb.row + 31) / 32), dim3(1024), 0, st, tb);   // the time conditioning's backward
    if (tb.split_tail) hipLaunchKernelGGL(time_tail_kernel, dim3(kTimeTailBlocks), dim3(512), 0, st, tb);   // ... and its encoder tail, 8 blocks
    HIP_TRY(hipGetLastError());
    return 0;
}

/* clip_grad_norm_ (max_norm > 0) + Adam step on flat vectors; scratch: >= 1032 floats; step: 1-based step count, or < 0: the count lives on the device
 * (int at scratch + 4, the number of steps taken so far; this call advances it) - the form a step replayed as a hipGraph needs, its kernel arguments being frozen */
int mpdx_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2, float eps,
                   int step, float max_norm, float* scratch, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch || n == 0 || step == 0) return fail(MPDX_E_INVALID, "bad argument");
    if (lr < 0.f && step > 0) return fail(MPDX_E_INVALID, "lr < 0 (the learning rate from scratch[5]) needs the device-resident step count (step < 0)");
    hipStream_t st = (hipStream_t)stream;
    const float* clip = nullptr;
    int n_part = 0;
    int* cnt = step < 0 ? (int*)(scratch + 4) : nullptr;
    if (max_norm > 0.f || cnt) {   // (device-counter mode: the launch also advances the counter, clipping or not)
        const int nb = (int)std::min<size_t>((n + 255) / 256, 1024);
        hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, grads, n, scratch + 8, cnt);
        if (max_norm > 0.f) {
            clip = scratch + 8;   // the partial sums; adam_kernel finishes the norm itself (norm_finish_kernel's order) and publishes scratch[0..1]
            n_part = nb;
        }
    }
    const float bc1 = step > 0 ? 1.0f - (float)pow((double)beta1, step) : 1.0f, bc2 = step > 0 ? 1.0f - (float)pow((double)beta2, step) : 1.0f;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, bc1, sqrtf(bc2), clip, n_part, max_norm, scratch, (const int*)cnt);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_ema_update(float* ema, const float* params, size_t n, float beta, void* stream) {
    if (!ema || !params || n == 0) return fail(MPDX_E_INVALID, "bad argument");
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, ema, params, n, beta);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
