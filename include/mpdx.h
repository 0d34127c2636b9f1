/*
 * mpdx.h - C ABI of libmpdx.so: the MI355X (gfx950) guided reverse-diffusion trajectory sampler.
 *
 * The reference (jacarvalho/mpd-public) has NO native / FFI interface for this path: it is three Python callable
 * protocols.  Each entry point below states the reference Python interface it replaces
 * (file:line under /root/reference) - this is what a maintainer's ctypes stub binds.
 *
 * Conventions
 *   - every function returns int: 0 = ok, >0 = hipError_t, <0 = MPDX_E_* ; never throws across the ABI;
 *     mpdx_last_error() returns a thread-local human-readable message for the last non-zero return.
 *   - the CALLER owns every device buffer (PyTorch caching allocator in practice) and passes raw device pointers;
 *     the library never allocates caller-visible device memory and never synchronises: it only enqueues work on
 *     the hipStream_t it is given (void* here so that the header needs no HIP include).
 *   - tensors are contiguous fp32 row-major.  Trajectories are [B, H, D] exactly as the reference's x.
 *   - one handle per model; a handle is host-side metadata only (layer table, offsets); not thread-safe per handle.
 */
#ifndef MPDX_H
#define MPDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPDX_E_INVALID   (-1) /* bad argument / unsupported configuration */
#define MPDX_E_NOTFOUND  (-2) /* unknown parameter name */
#define MPDX_E_STATE     (-3) /* call order violated (e.g. forward before all parameters were packed) */

#define MPDX_MAX_LEVELS 8

typedef struct mpdx_unet mpdx_unet; /* opaque */

/* TemporalUnet.__init__ arguments that shape the network (mpd/models/diffusion_models/temporal_unet.py:22-35),
 * for the only configuration the reference builds: conditioning_type=None, self_attention=False. */
typedef struct mpdx_unet_cfg {
    int32_t state_dim;                  /* D */
    int32_t n_support_points;           /* H: 64 in every shipped configuration; 16 ... 128 with H % 2^(n_levels-1) == 0 (powers of two run as they are, 24 / 40 / 48 / 96 ... in the next power-of-two container, zero rows kept zero) */
    int32_t unet_input_dim;             /* 32 */
    int32_t n_levels;                   /* len(dim_mults) */
    int32_t dim_mults[MPDX_MAX_LEVELS]; /* (1,2,4,8) or (1,2,4): UNET_DIM_MULTS, temporal_unet.py:14-17 */
    int32_t time_emb_dim;               /* 32 */
} mpdx_unet_cfg;

const char* mpdx_last_error(void);
int mpdx_version(void);

/* ---- model construction: replaces TemporalUnet(**unet_configs) + load_state_dict (inference.py:132-148) ---- */
int    mpdx_unet_create(const mpdx_unet_cfg* cfg, mpdx_unet** out);
void   mpdx_unet_destroy(mpdx_unet* u);
/* number of state-dict tensors the model expects; their names/shapes (identical to the reference's keys) */
int    mpdx_unet_num_params(const mpdx_unet* u);
int    mpdx_unet_param_info(const mpdx_unet* u, int idx, const char** name, int32_t shape[3], int32_t* ndim);
/* sizes (in floats) of the caller-allocated device buffers */
size_t mpdx_unet_packed_floats(const mpdx_unet* u);             /* repacked weights (MFMA fragment order)       */
size_t mpdx_unet_timetab_floats(const mpdx_unet* u, int T);     /* per-timestep conditioning table [T, sum C_out] */
size_t mpdx_unet_workspace_floats(const mpdx_unet* u, int B);   /* activations for a batch of B trajectories    */
/* repack one state-dict tensor (device pointer, reference layout) into `packed` */
int    mpdx_unet_pack_param(mpdx_unet* u, const char* name, const float* src_dev, size_t n_floats,
                            float* packed_dev, void* stream);
/* TimeEncoder + every block's cond_mlp depend only on the integer t (layers.py:229-255,336-340): tabulate them.
 * freqs16 = the 16 sinusoid frequencies exp(-k*ln(1e4)/15) computed by the host exactly as layers.py:249-251. */
int    mpdx_unet_build_timetab(mpdx_unet* u, const float* packed_dev, const float* freqs16_dev, int T,
                               float* timetab_dev, void* stream);

/* ---- eps-model call: replaces model(x, t, context=None) (diffusion_model_base.py:147; temporal_unet.py:118) ----
 * x, eps: [B,H,D]; t: the (batch-constant) integer timestep, 0 <= t < T. */
int mpdx_unet_forward(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, int T,
                      const float* x, int t, float* eps, int B, float* ws, void* stream);

/* ---- one reverse step: replaces ddpm_sample_fn's arithmetic (sample_functions.py:17-62) together with
 * p_mean_variance / predict_start_from_noise / q_posterior (diffusion_model_base.py:121-155) and
 * apply_hard_conditioning (sample_functions.py:5-8).  Scalars are the t-th entries of the registered buffers. */
typedef struct mpdx_step_coefs {
    float sqrt_recip_alphas_cumprod;    /* diffusion_model_base.py:90  */
    float sqrt_recipm1_alphas_cumprod;  /* :91 */
    float posterior_mean_coef1;         /* :100 */
    float posterior_mean_coef2;         /* :102 */
    float noise_scale;                  /* exp(0.5*posterior_log_variance_clipped[t]) ; 0 when t == 0 */
    float noise_std_extra;              /* noise_std_extra_schedule_fn(t) (0.5 at inference.py:243), 1 if None */
    int32_t predict_epsilon;            /* :121-132 */
    int32_t clip_denoised;              /* :149-150 */
    float ddim_k1;                      /* ddim_sample (:184-259): sqrt(alphas_cumprod[t_next]), 1 on the last pair */
    float ddim_k2;                      /* sqrt(1 - alphas_cumprod[t_next] - sigma^2) with eta = 0, 0 on the last pair */
    float guide_scale;                  /* factor on every guide increment of this step: 1, or model_var = exp(posterior_log_variance_clipped[t])
                                         * when scale_grad_by_std (sample_functions.py:77-78) */
} mpdx_step_coefs;

/* x_io[B,H,D] is updated in place to  hard_cond( mean + noise_scale*noise*noise_std_extra ).
 * noise may be NULL (treated as 0).  hard_start/hard_goal: [B,D] values written at horizon index 0 / H-1
 * (NULL = no hard conditioning).  mean_only == 1: the posterior mean (before noise, before hard conditioning) is
 * written instead - the point where the reference inserts the guide (sample_functions.py:39-48).
 * mean_only == 2: the DDIM update of ddim_sample (diffusion_model_base.py:216-237, eta = 0):
 *   x <- hard_cond( ddim_k1 * x_start + ddim_k2 * pred_noise ), x_start NOT clamped (as the reference).
 * chain_out (optional): a second [B,H,D] destination that receives the same values (chain.append, :175-176).
 * absmax_out (optional, uint32 per context): atomicMax of the bit pattern of max|x| over each context's
 * n_per_ctx trajectories - the whole-tensor range test of LimitsNormalizer.unnormalize (normalization.py:160). */
int mpdx_ddpm_step(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, int T,
                   float* x_io, const float* noise, const float* hard_start, const float* hard_goal,
                   const mpdx_step_coefs* coefs, int t, int mean_only, float* chain_out,
                   uint32_t* absmax_out, int n_per_ctx, int B, float* ws, void* stream);

/* finish a guided step: x = hard_cond(x + noise_scale*noise*noise_std_extra), optional chain copy */
int mpdx_add_noise(float* x_io, const float* noise, const float* hard_start, const float* hard_goal,
                   float noise_scale, float noise_std_extra, float* chain_out, int B, int H, int D, void* stream);

/* apply_hard_conditioning (sample_functions.py:5-8) for ARBITRARY horizon indices:  x[:, horizon_idx[k], :] = values[k]  ([B,D] device tables) for
 * k < n <= 16, in order (python semantics: negative indices count from the end, a later entry wins on a repeated index); chain_out (optional) receives
 * the same writes.  horizon_idx / values are HOST arrays.  The step kernels fold indices 0 and H-1 into their epilogue (hard_start / hard_goal above);
 * this is the step-by-step loop's path for every other index (the fused mpdx_plan takes 0 and H-1 only). */
int mpdx_hard_conds(float* x_io, float* chain_out, int n, const int32_t* horizon_idx, const float* const* values, int B, int H, int D, void* stream);

/* ---- forward loss (what the reference's validation loop evaluates under no_grad; the pass WITH its gradient is mpdx_train_loss_backward below).
 * q_sample (diffusion_model_base.py:320-330) followed by apply_hard_conditioning (:335): per-trajectory timesteps t_dev[B]
 * (int64, clamped to [0,T)), schedule buffers on the device. */
int mpdx_q_sample(const float* x_start, const float* noise, const long long* t_dev, const float* sqrt_alphas_cumprod_dev,
                  const float* sqrt_one_minus_alphas_cumprod_dev, const float* hard_start, const float* hard_goal, float* out, int B, int H,
                  int D, int T, void* stream);
/* WeightedL2 (l1 = 0) / WeightedL1 (l1 = 1) of helpers.py:71-99 applied to apply_hard_conditioning(pred) vs targ
 * (diffusion_model_base.py:343-350): out1[0] = mean over B*H*D of the (optionally weights_hd[H*D]-weighted) error. */
int mpdx_weighted_loss(const float* pred, const float* targ, const float* weights_hd, const float* hard_start, const float* hard_goal,
                       int l1, float* out1, int B, int H, int D, void* stream);

/* ---- training step: replaces  loss = model.loss(x, context, hard_conds); loss.backward();
 * clip_grad_norm_; optimizer.step(); EMA.update_model_average   of mpd/trainer/trainer.py:186-283 with
 * GaussianDiffusionModel.p_losses (diffusion_model_base.py:331-352) and WeightedL1/L2 (helpers.py:71-99).
 * Parameters, gradients, Adam moments and the EMA copy are FLAT fp32 vectors in reference (state-dict) layout: parameter idx
 * (mpdx_unet_param_info order) lives at [off, off + n) with (off, n) from mpdx_train_param_offset, gaps are zero.  A host
 * framework can alias its parameter tensors onto the vector (mpd_public_amd/trainer.py does, so torch optimisers keep working). */
size_t mpdx_train_flat_floats(mpdx_unet* u);
size_t mpdx_train_dgrad_pack_floats(mpdx_unet* u);              /* transposed / tap-flipped packs for the input-gradient convolutions */
size_t mpdx_train_workspace_floats(mpdx_unet* u, int B);     /* every activation of a batch of B + gradient buffers */
int    mpdx_train_param_offset(mpdx_unet* u, int idx, size_t* off, size_t* n);
/* flat parameters -> forward pack (what mpdx_unet_pack_param builds, all parameters, one launch) and, if packedT != NULL,
 * the dgrad pack.  Call after every optimiser step. */
int    mpdx_train_pack(mpdx_unet* u, const float* flat, float* packed, float* packedT, void* stream);
/* one p_losses evaluation and its gradient wrt every parameter:
 *   x_start, noise [B,H,D]; t_dev [B] int64 timesteps; sqrt_alphas_cumprod / sqrt_one_minus_alphas_cumprod [T] device tables;
 *   freqs16 the SinusoidalPosEmb frequencies (as mpdx_unet_build_timetab); hard_start / hard_goal [B,D] or NULL;
 *   weights_hd [H,D] loss weights or NULL; l1: 1 = WeightedL1, 0 = WeightedL2; loss_scale multiplies the gradient;
 *   loss_out: one device float <- the loss; grads_flat <- d(loss_scale * loss)/d(parameters), every parameter entry written. */
int    mpdx_train_loss_backward(mpdx_unet* u, const float* flat, const float* packed, const float* packedT, float* grads_flat,
                                const float* x_start, const float* noise, const long long* t_dev, const float* sqrt_alphas_cumprod_dev,
                                const float* sqrt_one_minus_alphas_cumprod_dev, const float* freqs16, const float* hard_start,
                                const float* hard_goal, const float* weights_hd, int T, int B, int predict_epsilon, int l1, float loss_scale,
                                float* loss_out, float* ws, void* stream);
/* torch.nn.utils.clip_grad_norm_(max_norm) if max_norm > 0, then torch.optim.Adam.step() (no weight decay, no amsgrad);
 * step counts from 1; scratch: >= 1032 floats (scratch[0] <- the gradient norm before clipping, scratch[1] <- the clip factor).
 * step < 0: the count lives on the device - the int at scratch + 4 holds the number of steps taken so far and this call advances it (the form a
 * training step captured into a hipGraph needs: kernel arguments are frozen at capture, trainer.TrainStep.step); with step < 0, lr < 0 takes the
 * learning rate from the float at scratch + 5 as well (an LR schedule then neither re-captures nor re-keys anything) */
int    mpdx_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                      float eps, int step, float max_norm, float* scratch, void* stream);
/* EMA.update_model_average (trainer.py:67-85): ema = beta * ema + (1 - beta) * params */
/* draw mode of mpdx_train_loss_backward (replaces `t = torch.randint(...)`, `noise = torch.randn_like(x)` of diffusion_model_base.py:356 / :337 when an
 * iteration is replayed as a hipGraph): with a non-null step_counter_dev - the device int mpdx_adam_step(step < 0) advances - the following loss passes
 * treat t_dev and noise as OUTPUTS drawn on the device (Philox4x32-10 keyed by seed, stream position step * B + sample); NULL disarms */
int    mpdx_train_draw(mpdx_unet* u, unsigned long long seed, const int* step_counter_dev);
int    mpdx_ema_update(float* ema, const float* params, size_t n, float beta, void* stream);

/* standard-normal generator for the production path (Philox4x32-10 + Box-Muller); replaces torch.randn /
 * torch.randn_like (diffusion_model_base.py:165, sample_functions.py:51).  Parity runs inject noise instead. */
int mpdx_randn(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);

/* ---- cost guidance: replaces guide(x) = GuideManagerTrajectoriesWithVelocity.forward (guides.py:173-211) and one
 * iteration of guide_gradient_steps (sample_functions.py:74-81).  The cost terms are the ones inference.py:188-225
 * builds: one CostCollision per collision field of the task + CostGPTrajectory, weights as at :204,213.
 * Their arithmetic is un-vendored in the reference (empty submodules): restated, see oracle/costs.py. */
#define MPDX_MAX_FIELDS 4
#define MPDX_FIELD_OBJECTS   0 /* sdf to sphere/box primitives (task.df_collision_objects / extra objects) */
#define MPDX_FIELD_WORKSPACE 1 /* workspace-boundary planes */
#define MPDX_FIELD_SELF      2 /* robot self collision (Panda) */
#define MPDX_ROBOT_POINTMASS 0
#define MPDX_ROBOT_PANDA     1

typedef struct mpdx_field {
    int32_t kind;                       /* MPDX_FIELD_* */
    float   weight;                     /* weight_grad_cost_collision (inference.py:55) */
    int32_t sphere_off, n_spheres;      /* float offset into prims, 4 floats each: cx,cy,cz,r (cz unused in 2-D) */
    int32_t box_off, n_boxes;           /* 6 floats each: cx,cy,cz,hx,hy,hz */
    float   ws_min[3], ws_max[3];       /* MPDX_FIELD_WORKSPACE */
} mpdx_field;

typedef struct mpdx_guide_params {
    int32_t robot;                      /* MPDX_ROBOT_* */
    int32_t q_dim;                      /* 2, 3 (point mass) or 7 (Panda); state dim D = 2*q_dim (pos + vel) */
    int32_t ws_dim;                     /* workspace dimension 2 or 3 */
    int32_t interpolate;                /* interpolate_trajectories_for_collision (guides.py:152) */
    int32_t n_interp;                   /* num_interpolated_points_for_collision; effective reference value 128 */
    int32_t clip_grad;                  /* clip_grad (guides.py:151), rule 'norm' */
    float   max_grad_norm;              /* 1.0 */
    float   mins[16], maxs[16];         /* LimitsNormalizer limits of the trajectory field (normalization.py:92-93) */
    float   cutoff_margin;              /* obstacle_cutoff_margin (inference.py:110) */
    float   link_margin;                /* point-mass collision radius */
    int32_t n_fields;
    mpdx_field fields[MPDX_MAX_FIELDS];
    int32_t use_gp;                     /* CostGPTrajectory present */
    float   gp_weight;                  /* weight_grad_cost_smoothness (inference.py:56) */
    float   dt;                         /* trajectory_duration / n_support_points (inference.py:120) */
    float   sigma_gp;                   /* 1.0 */
    const float* prims;                 /* device pointer: primitive table */
    int32_t n_prim_floats;
    /* --- switches for what the reference's empty submodules leave undecidable / for options of guides.py --- */
    int32_t clip_rule;                  /* 0: clip_grad_rule 'norm' (guides.py:224-230); 1: 'value' (guides.py:232-236) */
    float   max_grad_value;             /* 0.1 (guides.py:151) */
    int32_t gp_half_factor;             /* 0: cost_GP = sum e^T Qinv e (default);  1: 1/2 sum e^T Qinv e (GPMP2's convention) */
    int32_t identity_normalizer;        /* 0: LimitsNormalizer (mins / maxs above, whole-tensor range test; normalization.py:156-167);
                                         * 1: x is ALREADY in robot units (Identity :111-116; no range test): the trajectory optimiser of
                                         *    generate_trajectories (scripts/generate_data/generate_trajectories.py:94-117) works on raw trajectories;
                                         * 2: GaussianNormalizer (:140-141): x * stds + means with means in `mins`, stds in `maxs`, no range test */
} mpdx_guide_params;

/* one guide iteration on x[B,H,D] (normalised).  grad_out == NULL: x <- hard_cond(x + guide(x)) in place and
 * absmax_out[ctx] <- atomicMax(max|x_new|) ; grad_out != NULL: grad_out <- guide(x), x untouched.
 * absmax_in[ctx] holds the bit pattern of max|x| over the context's n_per_ctx trajectories (the whole-tensor range
 * test of LimitsNormalizer.unnormalize, normalization.py:160). */
int mpdx_guide_step(const mpdx_guide_params* gp, float* x, float* grad_out, const float* hard_start, const float* hard_goal,
                    const uint32_t* absmax_in, uint32_t* absmax_out, int n_per_ctx, int B, int H, int D, void* stream);
/* the same with the increment multiplied by guide_scale before it is added (scale_grad_by_std, sample_functions.py:77-78) */
int mpdx_guide_step_scaled(const mpdx_guide_params* gp, float* x, float* grad_out, const float* hard_start, const float* hard_goal,
                           const uint32_t* absmax_in, uint32_t* absmax_out, int n_per_ctx, int B, int H, int D, float guide_scale,
                           void* stream);
/* measurement helper (bench.py `guided` sub-record): `reps` back-to-back guide launches in gradient-only mode (grad_out
 * <- guide(x); x and the flags untouched, so every launch does the same work) bracketed by ONE HIP-event pair on `stream`;
 * *ms_avg = average time per launch.  Replaces nothing in the reference: guides.py:173-211 is what one launch computes.  Synchronises. */
int mpdx_guide_time(const mpdx_guide_params* gp, float* x, float* grad_out, const uint32_t* absmax_in, int n_per_ctx, int B, int H, int D,
                    int reps, void* stream, float* ms_avg);
/* dev tool: cycle stamps (16 slots per wave x 8 waves, workgroup 0) of one guide launch (gradient-only mode).
 * The three *_trace entry points and the ablation masks of mpdx_bench_layer work only in a library built with -DMPDX_DEV_HOOKS
 * (the production kernels carry no hooks); otherwise they return MPDX_E_STATE. */
int mpdx_guide_trace(const mpdx_guide_params* gp, float* x, const uint32_t* absmax_in, int B, int H, int D, void* stream,
                     long long* stamps128);
/* absmax_out[ctx] <- atomicMax over the context's trajectories (caller zeroes absmax_out first) */
int mpdx_absmax(const float* x, uint32_t* absmax_out, int n_per_ctx, int B, int H, int D, void* stream);

/* ---- post-loop metrics: the arithmetic behind task.get_trajs_collision_and_free / compute_fraction_free_trajs /
 * compute_collision_intensity_trajs and compute_smoothness / compute_path_length (inference.py:288-297,311-316;
 * un-vendored, restated).  x_unnormalised [B,H,D]; out4 [B,4] = {#colliding interpolated waypoints, path length,
 * smoothness, #waypoints checked}; n_check = interpolated waypoints per trajectory used for collision checking. */
int mpdx_traj_metrics(const mpdx_guide_params* gp, const float* x_unnormalised, float* out4, int n_check, int B, int H, int D,
                      void* stream);
/* the same, plus the per-waypoint collision flags the count is made of: mask [B, n_check] bytes (1 = the interpolated waypoint
 * collides), or NULL.  compute_collision_intensity_trajs (inference.py:295-297) is the mean of these flags; the parity tests use them to
 * compare the kernel's decisions with an fp64 evaluation waypoint by waypoint. */
int mpdx_traj_metrics_mask(const mpdx_guide_params* gp, const float* x_unnormalised, float* out4, uint8_t* mask, int n_check, int B,
                           int H, int D, void* stream);

/* ---- baseline planners of the dataset-generation script: replaces `HybridPlanner(RRTConnect x n via
 * MultiSampleBasedPlanner, GPMP2).optimize()` of scripts/generate_data/generate_trajectories.py:68-120.  The planners are
 * un-vendored in the reference (mp_baselines submodule empty: PARITY UNPINNED); the published algorithms are restated
 * (oracle/gpmp.py).  Trajectories and configurations are in RAW robot units (gp->identity_normalizer semantics).
 *
 * GPMP2 (Mukadam et al. 2018) - one Levenberg-Marquardt iteration per call for B trajectories:
 *   F = 1/2 sum e_i^T Q^-1 e_i / sigma_gp^2 + 1/2 sum c^2 / sigma_obs^2   (constant-velocity GP prior of gp->dt / gp->sigma_gp between the
 *   supports; hinge collision factors of gp->fields on the gp->n_interp interpolated points; start and goal states fixed).
 *   state[b] = {F(x_b), lambda_b, accepted steps, F(last candidate)}: initialise to {3e38, lambda_init, 0, 0} and delta to 0.
 *   Each call judges the pending candidate x + delta (accept if it lowers F: x <- x + delta, lambda *= lambda_down; else lambda *= lambda_up),
 *   then (solve != 0) linearises at x and writes the next proposal  delta = -step (K^-1 + J^T J / sigma_obs^2 + lambda diag)^-1 grad F
 *   (block-tridiagonal system, banded LDL^T in LDS).  Call iters times with solve = 1 and once more with solve = 0.
 *   A trajectory whose accepted step no longer lowers F (relative 1e-7) or whose lambda reached lambda_max is CONVERGED: its lambda is
 *   stored negated and further calls return at once for it.
 *   adaptive == 0: every candidate is accepted and lambda stays fixed (damped Gauss-Newton with a fixed step). */
typedef struct mpdx_gpmp_opts {
    float   sigma_obs;
    float   lambda_up, lambda_down, lambda_min, lambda_max;
    float   step;
    int32_t adaptive;
} mpdx_gpmp_opts;
int mpdx_gpmp_step(const mpdx_guide_params* gp, const mpdx_gpmp_opts* opts, float* x, float* delta, float* state, int B, int H, int D,
                   int solve, void* stream);

/* RRT-Connect (Kuffner & LaValle 2000) for n independent problems, the WHOLE search in one launch (one workgroup per problem):
 *   start, goal [n, q]; nodes [n, 2, max_nodes, q] / parent [n, 2, max_nodes] the two trees (tree 0 from the start, tree 1 from the goal);
 *   count [n, 2] nodes per tree; link [n, 2] the node indices where the trees met (-1: not solved within max_iters / max_nodes);
 *   iters [n] iterations used.  Samples are uniform in [q_lo, q_hi] (Philox keyed by seed, problem, iteration); an edge is checked on
 *   n_edge_checks interpolated configurations with the link radius only (as mpdx_traj_metrics). */
typedef struct mpdx_rrt_opts {
    float    q_lo[8], q_hi[8];
    float    step;
    int32_t  max_nodes, max_iters, max_connect_steps, n_edge_checks;
    uint64_t seed;
} mpdx_rrt_opts;
int mpdx_rrt_connect(const mpdx_guide_params* gp, const mpdx_rrt_opts* opts, const float* start, const float* goal, float* nodes,
                     int32_t* parent, int32_t* count, int32_t* link, int32_t* iters, int n, void* stream);

/* RRT-Connect post-processing on the device: per problem the path start ... goal is extracted from the two trees mpdx_rrt_connect grew, shortcut
 * greedily (`rounds` passes, edges checked on n_edge_checks interpolated configurations) and resampled uniformly in arc length to H support
 * points with central-difference velocities -> trajs_out[n][H][2 q_dim]; path_len[n] (or NULL) = nodes of the shortcut path.  An unsolved problem
 * becomes the straight line.  Replaces the path extraction / smoothing that MultiSampleBasedPlanner + HybridPlanner do between RRTConnect and GPMP2
 * (scripts/generate_data/generate_trajectories.py:68-105; un-vendored: the published algorithms, parity unpinned). */
int mpdx_rrt_paths(const mpdx_guide_params* gp, const float* start, const float* goal, const float* nodes, const int32_t* parent, const int32_t* link,
                   float* trajs_out, int32_t* path_len, int n, int max_nodes, int H, float dt, int n_edge_checks, int rounds, void* stream);

/* ---- the whole planning loop: replaces GaussianDiffusionModel.p_sample_loop driven by run_inference
 * (diffusion_model_base.py:157-182,285-316) with sample_fn=ddpm_sample_fn.  Everything is enqueued on `stream`
 * without a single host synchronisation: the t-dependent branches of the reference (`t_single < 0`,
 * `t_single < t_start_guide`, sample_functions.py:28-29,39) depend only on the integer loop index.
 *   coefs  : host array [T], entry t = the scalars of timestep t (see mpdx_step_coefs)
 *   x      : [B,H,D]; in: x_T ~ N(0,I) (hard conditioning is applied here, :165-166); out: the final trajectories
 *   noise  : [T + n_without_noise, B,H,D] the randn_like draw of each loop iteration, in loop order; or NULL: every iteration
 *            draws its noise in place from the Philox stream (rng_seed, rng_offset) - iteration k takes elements
 *            [(k+1) n, (k+2) n), n = B*H*D, of the stream whose first n elements are x_T as written by
 *            mpdx_randn(x, n, rng_seed, rng_offset): bit-identical to passing the pre-generated tensor, without its
 *            (T + n_without_noise) * n * 4 bytes (2.4 GB for a 6400-trajectory Panda shard)
 *   chain  : NULL or [T + n_without_noise + 1, B,H,D] <- x after every iteration, index 0 = conditioned x_T
 *            (the 'diffsteps b h d' layout run_inference returns, :310)
 *   guide  : NULL (planner_alg 'diffusion_prior') or the cost guide; applied n_guide_steps times on the posterior
 *            mean of every iteration whose loop index i < t_start_guide (sample_functions.py:39-48), each increment times
 *            coefs[t].guide_scale; n_guide_steps == 0 skips guidance (the reference's empty range(0) loop)
 *   guide_flags : device scratch, (T + n_without_noise) * (n_guide_steps + 1) * ceil(B / n_per_ctx) uint32
 *   n_per_ctx   : trajectories per start/goal context (n_samples); contexts are consecutive blocks of the batch   */
int mpdx_plan(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, int T, const mpdx_step_coefs* coefs,
              int n_without_noise, float* x, const float* noise, const float* hard_start, const float* hard_goal,
              float* chain, int B, float* ws, const mpdx_guide_params* guide, int n_guide_steps, int t_start_guide,
              uint32_t* guide_flags, int n_per_ctx, uint64_t rng_seed, uint64_t rng_offset, void* stream);

/* ---- measurement helpers (bench.py's roofline leg; not used by the planning path) ----
 * One U-Net pass with a hipEvent pair around every kernel launch, on `stream`.  This call DOES synchronise the
 * stream (it reads the events).  ms_out[i] = duration of launch i; flops_out[i] = its algorithmic FLOPs
 * (2*C_out*B*L_out*C_in*taps summed over the layers of the launch, 0 for the final 1x1+step kernel); names_out[i] ->
 * layer name, or "fused[...]" for a whole-trajectory fused segment (disable fusion with MPDX_FUSED=0).
 * Returns the number of launches written (<= cap) in *n_out. */
int mpdx_unet_profile(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, int T, const float* x, int t,
                      int B, float* ws, void* stream, int cap, float* ms_out, double* flops_out,
                      const char** names_out, int* n_out);
/* dev tool: per-phase s_memtime stamps (7 each) of the first and the last workgroup of one launch of layer `layer` */
int mpdx_layer_trace(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, const float* x, int layer, int B, float* ws,
                     void* stream, long long* stamps32);
/* dev tool: per-phase s_memtime stamps (workgroup 0; 8 waves x 128 slots) of one launch of fused segment `seg` */
int mpdx_fused_trace(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, const float* x, int seg, int B,
                     float* ws, void* stream, long long* stamps_out, int cap, int* n_out, int* nops_out);
/* in-situ timing of launch units [unit_first, unit_last] inside `reps` real U-Net passes (one event pair per pass around the
 * run: real predecessors and cold weights, event cost amortised over the run); *ms_avg = bracketed time per pass. */
int mpdx_unet_time_units(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, int T, const float* x, int t, int B,
                         float* ws, void* stream, int unit_first, int unit_last, int reps, float* ms_avg);
/* the DIFFERENTIAL form: `reps` back-to-back U-Net passes WITHOUT the launch units whose bit is set in skip_mask (0: nothing skipped)
 * between ONE event pair -> *ms_avg per pass.  A launch class costs (pass with everything) - (pass without the class): no event sits next to the
 * measured launches.  Timing only: the skipped units' consumers read whatever the workspace holds. */
int mpdx_unet_time_without(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, int T, const float* x, int t, int B,
                           float* ws, void* stream, unsigned long long skip_mask, int reps, float* ms_avg);
/* layer index behind launch unit i of mpdx_unet_profile at batch B (-1: fused whole-trajectory segment or the final kernel) */
int mpdx_unet_unit_layer(const mpdx_unet* u, int B, int i);
/* introspection: the kernel that runs fused segment `seg` of this network - 0..6: a static whole-trajectory program (0, 3, 5, 6 with
 * compile-time LDS geometry, csrc/fused_geom.hpp; four levels: 5 + 3, three levels: 0 + 6 + 3), -1: the generic op-list kernel, -2: no such segment.  Replaces nothing in the reference. */
int mpdx_unet_fused_program(const mpdx_unet* u, int seg);
/* measurement helper: ALGORITHMIC bytes of launch unit i of a U-Net pass at batch B - its weights / parameters once plus the activations
 * that cross its boundary once (bench.py: roofline.traffic_over_algorithmic).  Replaces nothing in the reference. */
double mpdx_unet_unit_bytes(const mpdx_unet* u, int B, int i);
/* 1 when launch unit i is a paired launch (blocks[0] + the same block's residual 1x1 conv in one conv_pair_kernel) */
int mpdx_unet_unit_is_pair(const mpdx_unet* u, int B, int i);
/* `reps` back-to-back launches of layer `layer` between two events; dbg = ablation mask (1 skip staging, 2 skip
 * MFMA loop, 4 skip epilogue, 8 skip weight loads: -DMPDX_DEV_HOOKS builds only), 16 = replay from a hipGraph; synchronises. */
int mpdx_bench_layer(mpdx_unet* u, const float* packed_dev, const float* timetab_dev, const float* x, int layer, int B,
                     float* ws, void* stream, int reps, int dbg, float* ms_per_launch);
/* tile the dispatcher picks for launch i at batch B: writes "MTxNT/WNxWK" into buf */
int mpdx_unet_layer_tile(const mpdx_unet* u, int i, int B, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* MPDX_H */
