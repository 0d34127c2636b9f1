// host.hpp - the host-side model (layer plan, parameter table, fused segments) and the launcher interface between the translation
// units of libmpdx.so.  The library is built from one host TU (mpdx.hip: model building, the planning loop, the C ABI, the small
// streaming kernels) and one TU per kernel family - k_conv.hip (conv_block.hpp), k_ws.hip (conv_ws.hpp), k_fused.hip /
// k_fused_train.hip (fused_level.hpp), k_guide.hip (guide.hpp), k_train.hip (train.hpp + train_host.hpp), k_planner.hip
// (planner.hpp + planner_host.hpp) - so that an edit to one kernel family recompiles that family only (mpd_public_amd/build.py
// compiles the TUs in parallel and keeps the objects).  A kernel template is instantiated in exactly ONE TU, behind a plain function
// declared here; no device code crosses a TU (no -fgpu-rdc).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mpdx.h"
#include "conv_block.hpp"
#include "conv_ws.hpp"
#include "fused_level.hpp"
#include "train_types.hpp"

namespace mpdx {

// ------------------------------------------------------------------------------------------------ error plumbing (mpdx.hip)
int fail(int code, const char* fmt, ...);
#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return ::mpdx::fail((int)e_, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
int raise_lds_limit(const void* kern);   // hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel)
int debug_level();                       // MPDX_DEBUG

// final_conv[1] (Conv1d(32 -> D, k=1), temporal_unet.py:113-116) fused with the DDPM posterior step
// (diffusion_model_base.py:121-155, sample_functions.py:31-62) and hard conditioning (sample_functions.py:5-8).
// One thread per (trajectory, horizon index); the arithmetic is rounded op by op exactly as the reference's
// separate elementwise ATen kernels are (no fma contraction), so given the same eps the update is bit-identical.
struct FinalArgs {
    const float* h;      // [B][H][C] output of final_conv[0]
    const float* w;      // [D][C]
    const float* bias;   // [D]
    const float* x_in;   // [B][H][D]
    const float* noise;  // [B][H][D] or null
    const float* hs;     // hard start [B][D] or null
    const float* hg;     // hard goal  [B][D] or null
    float* out;          // eps (mode 0) or x_next (mode 1/2)
    float* chain;        // optional second destination
    uint32_t* absmax;    // optional per-context max|out| (bit pattern)
    int B, H, D, C;
    int Hc;              // rows per trajectory of h (the container; 0: H)
    int mode;            // 0: eps only; 1: full step; 2: posterior mean only (guide insertion point); 3: DDIM update
    int n_per_ctx;
    mpdx_step_coefs k;
    NoiseRng rng;        // rng.on: the step's noise is drawn in place (noise pointer ignored)
};

// ------------------------------------------------------------------------------------------------ host-side model
enum ParamKind { PK_VEC = 0, PK_CONV = 1, PK_CONVT = 2 };

struct Param {
    std::string name;
    int32_t shape[3] = {0, 0, 0};
    int32_t ndim = 0;
    size_t n = 0;       // floats in the reference tensor
    size_t off = 0;     // offset (floats) in the packed buffer
    size_t foff = 0;    // offset (floats) in the flat reference-layout parameter vector (training, train_host.hpp)
    size_t pn = 0;      // floats in the packed buffer
    int kind = PK_VEC;
    int cout = 0, cin = 0, ksz = 0, cin_pad = 0, nslot = 0;
    bool done = false;
};

enum { SRC_X = -1, SRC_NONE = -2 };

struct Layer {
    int mode = CONV_S1, ks = 5, epi = EPI_GN_MISH;
    int c1 = 0, c2 = 0, cout = 0, L_in = 0, L_out = 0, gs = 0;
    int Lv_out = 0;           // valid rows of the output when the horizon runs in a power-of-two container (0: all L_out rows; ConvArgs::Lv_out)
    int src1 = SRC_NONE, src2 = SRC_NONE, dst = 0, res = SRC_NONE;  // workspace slots
    int w = -1, b = -1, gamma = -1, beta = -1;                       // param indices
    int tb_off = -1;                                                 // offset in a time-table row
    int cin_pad = 0, rs = 0;
    std::string name;
};

}  // namespace mpdx

struct mpdx_unet {
    mpdx_unet_cfg cfg;
    std::vector<mpdx::Param> params;
    std::unordered_map<std::string, int> pidx;
    std::vector<mpdx::Layer> layers;
    size_t packed_floats = 0;
    size_t slot_floats = 0;   // per-trajectory floats of one activation slot
    int n_slots = 0;
    int tt_row = 0;           // floats per time-table row
    std::vector<int> tt_w, tt_b, tt_cout, tt_off;  // cond_mlp param indices per block
    int final_slot = 0;       // slot holding final_conv[0]'s output
    int n_done = 0;
    // horizons that are not powers of two: the network runs in a container of Hc = next power of two rows per trajectory (ConvArgs::Lv_out);
    // the network input is copied into workspace slot `xpad_slot` ([B][Hc][D], zero rows behind the H real ones) at the head of a pass
    int Hc = 0, xpad_slot = -1;
    bool masked() const { return Hc != cfg.n_support_points; }
    // launch units: fused whole-trajectory segments (fused_level.hpp) or single layers
    struct CopyJob { size_t src, dst; int n0, ss0, ds0, n1, ss1, ds1, n_inner; };   // strided copy inside `packed` (float units)
    struct Fused {
        int first = 0, count = 0;       // layer range [first, first+count)
        bool has_final = false;         // final_conv[1] + DDPM step folded in
        int in1 = 0, in2 = 0;           // input slots (SRC_X / SRC_NONE allowed)
        int in3 = mpdx::SRC_NONE;       // slot of a skip tensor concatenated INSIDE the program (FusedArgs::gsrc3)
        int gout_slot[3] = {-1, -1, -1};
        size_t lds_bytes = 0;
        mpdx::FusedArgs tmpl;
        int program = -1;               // index of the matching static program (fused_program_kernel), -1: generic op-list kernel
        std::vector<CopyJob> jobs;      // assemble the stream-ordered weight copies + the contiguous parameter block
        std::vector<int> op_layer;      // conv op k computes layer op_layer[k] (a folded residual conv has no op of its own)
        int in3_consumer = -1;          // layer whose second source is the in3 skip tensor
    };
    std::vector<Fused> fused;
    void* jobs_dev = nullptr;           // device copy of every segment's CopyJobs (restream_all_kernel)
    int n_jobs = 0;
    const float* streams_for = nullptr; // `packed` buffer the fused streams were last assembled in
    int pack_version = 0, streams_version = -1;
    struct Unit { int fused; int layer; bool pair; };   // fused >= 0: fused[fused]; else layers[layer] (pair: + layers[layer+1] in one launch)
    std::vector<int> owner;                  // layer -> fused segment (-1: per-layer launch)
    // mpdx_plan's second chain (small batches run as two concurrent half-batch chains): side stream + fork / join events, created on first use
    struct PlanSide { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
    PlanSide side;
    // training (train_host.hpp)
    struct TrainLayer {
        int src1_l = -2, src2_l = -2, res_l = -2;   // layer that produced the tensor (-1: the network input, -2: none)
        bool need_dgrad = false;
        size_t dgrad_woff = 0;                      // offset of the dgrad weights in packedT
        mpdx::Layer dg;                             // the stride-1 convolution that computes the input gradient
    };
    bool train_ready = false;
    size_t flat_floats = 0, packedT_floats = 0;
    std::vector<TrainLayer> tl;
    std::vector<mpdx::PackDesc> pack_descs_host;
    void* pack_descs_dev = nullptr;
    void* pack_chunks_dev = nullptr;   // PackChunk table of pack_train_kernel
    size_t n_pack_chunks = 0;
};

namespace mpdx {

// ---- mpdx.hip
int pick_row_stride(int cin_pad, int mode, int L_in, int L_out, int LP);
void choose_tile(const Layer& l, int B, int& MT, int& NT);
bool layer_ksplit(const Layer& l);
int check_ready(const mpdx_unet* u);
unsigned fused_mask(int B);
bool fused_save_variant(const mpdx_unet::Fused& f);
int ensure_fused_streams(mpdx_unet* u, const float* packed, hipStream_t st);
int claim_fused_stream_jobs(mpdx_unet* u, const float* packed, const void** jobs, int* n);   // training: the copies ride on the pass's first launch
int launch_final_step(const FinalArgs& fa, hipStream_t st);   // final_step_kernel (fa.B/H/D/C set)
// ---- k_conv.hip: every conv_block_kernel / conv_pair_kernel instantiation
int launch_conv_layer(const Layer& l, ConvArgs& a, int B, hipStream_t st);
bool pair_tile(const Layer& l1, const Layer& l2, int B, int& MT, int& NT);   // (mpdx.hip) do blocks[0] + the block's residual 1x1 conv run as one launch, on which tile?
int launch_conv_pair(int MT, int NT, const ConvArgs& a1, const ConvArgs& a2, const Layer& l1, const Layer& l2, hipStream_t st);   // 1 launched, 0 does not fit, -1 error
// ---- k_ws.hip: the weight-stationary persistent kernels
int launch_weight_stationary(int variant, const Layer& l, ConvArgs& a, const ConvArgs& a2, int B, hipStream_t st);
// ---- k_fused.hip (planning programs) / k_fused_train.hip (the variants that keep activations for the backward pass)
int launch_fused_args(const mpdx_unet::Fused& f, const FusedArgs& a, int B, hipStream_t st, bool save = false);
int launch_fused_train(const mpdx_unet::Fused& f, const FusedArgs& a, int B, hipStream_t st);
// ---- k_guide.hip
struct NoiseRng;
int launch_guide(const mpdx_guide_params* gp, float* x, float* grad_out, const float* hs, const float* hg, const uint32_t* amax_in,
                 uint32_t* amax_out, int n_per_ctx, int B, int H, int D, hipStream_t st, const float* noise = nullptr, float noise_scale = 0.f,
                 float noise_extra = 0.f, float* chain = nullptr, float guide_scale = 1.0f, const NoiseRng* rng = nullptr);

}  // namespace mpdx
