// guide.hpp - one cost-guidance update of the trajectory batch, hand-derived gradients (no autograd on device).
//
// Replaces one iteration of guide_gradient_steps (mpd/models/diffusion_models/sample_functions.py:74-81):
//     x = x + guide(x);  apply_hard_conditioning(x)
// with guide = GuideManagerTrajectoriesWithVelocity.forward (mpd/models/diffusion_models/guides.py:173-211):
//     unnormalise (LimitsNormalizer.unnormalize incl. its whole-tensor range test, mpd/datasets/normalization.py:156-167)
//     -> interpolate 64 -> 128 points (interpolate_points_v1, guides.py:184) -> per cost term: d cost / d x_unnormalised
//     -> per-waypoint norm clip over all D dims of (g + 1e-6) (guides.py:224-230) -> zero first/last waypoint (:202-203)
//     -> weight (:206) -> negate (:210).  The gradient is w.r.t. the UNNORMALISED x and is added to the NORMALISED x
//     (reproduced, not fixed - SURVEY.md 3.3).
// The cost arithmetic itself (CostCollision / CostGPTrajectory / robot FK / SDF fields) lives in un-vendored
// submodules of the reference; it is restated from the published formulas (oracle/costs.py header) - PARITY UNPINNED.
//
// Mapping.  One 512-thread workgroup (8 waves) per trajectory; lane h of wave 0 = support point h (H <= 64).  The [H, D]
// state is read with coalesced loads; the horizon window needed by the interpolation and by the GP prior's 3-point
// finite-difference stencil is staged in LDS; the collision work on the interpolated points (lane = point) is split over
// the 8 waves (point mass: point half x field; Panda: point half x link-sphere group, FK once per point in LDS); the
// transpose of the interpolation (scatter of point gradients to support points) is a deterministic LDS gather (no
// atomics); max|x| for the next iteration's range test is a wave reduction + one atomicMax per trajectory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpdx.h"
#include "conv_block.hpp"

namespace mpdx {

// Franka Panda, modified DH (Craig): T_i = Rot_x(alpha_{i-1}) Trans_x(a_{i-1}) Rot_z(theta_i) Trans_z(d_i)
__device__ static const float kPandaA[7] = {0.0f, 0.0f, 0.0f, 0.0825f, -0.0825f, 0.0f, 0.088f};
__device__ static const float kPandaD[7] = {0.333f, 0.0f, 0.316f, 0.0f, 0.384f, 0.0f, 0.0f};
__device__ static const float kPandaCA[7] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};       // cos(alpha)
__device__ static const float kPandaSA[7] = {0.0f, -1.0f, 1.0f, 1.0f, -1.0f, 1.0f, 1.0f};     // sin(alpha)
// collision spheres: frame (1..7), offset along the frame's z axis, radius (synthetic geometry)
constexpr int kPandaNS = 11;
__device__ static const int kPandaSF[kPandaNS] = {1, 1, 3, 3, 4, 5, 5, 5, 7, 7, 7};
__device__ static const float kPandaSO[kPandaNS] = {-0.15f, 0.0f, -0.15f, 0.0f, 0.0f, -0.25f, -0.12f, 0.0f, 0.0f, 0.107f, 0.17f};
__device__ static const float kPandaSR[kPandaNS] = {0.10f, 0.10f, 0.09f, 0.09f, 0.09f, 0.08f, 0.08f, 0.08f, 0.07f, 0.06f, 0.06f};
constexpr int kPandaNP = 12;  // self-collision pairs
__device__ static const int kPandaPA[kPandaNP] = {8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 10};
__device__ static const int kPandaPB[kPandaNP] = {0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};

struct GuideArgs {
    mpdx_guide_params gp;
    float* x;               // [B][H][D] normalised trajectories (updated in place unless grad_out)
    float* grad_out;        // optional: write the guide increment instead of applying it (API path guide(x))
    const float* hs;        // hard start [B][D] or null
    const float* hg;        // hard goal  [B][D] or null
    const uint32_t* amax_in;   // per-context max|x| bit pattern of the INPUT (range test)
    uint32_t* amax_out;        // per-context max|x| of the OUTPUT (next iteration), or null
    int B, H, D, n_per_ctx;
    // last guide iteration of a step: also finish the step (sample_functions.py:51-62 + hard conditioning + chain.append)
    const float* noise;     // [B][H][D] or null
    float noise_scale, noise_extra;
    float guide_scale;      // factor on the increment (1, or model_var when scale_grad_by_std: sample_functions.py:77-78)
    float* chain;           // optional second destination
    NoiseRng rng;           // rng.on: the step's noise is drawn in place (noise pointer ignored)
    long long* trace;       // dev tool: cycle stamps, 16 slots per wave of workgroup 0 (null in production)
};

// d cost / d p  for  cost = relu(margin - min_prims sdf(p)).
// The scan over primitives only tracks the minimum signed distance and its index: batched (4 primitives' data are read
// back to back: one LDS wait per batch), branch-free (selects), hardware v_sqrt_f32 (1 ulp) instead of the IEEE-exact
// sqrtf sequence.  The gradient is evaluated once, for the arg-min primitive, after the scan.  (The first version -
// per primitive: load, exact sqrt, exact divide, compare, divergent branch - cost ~330 cycles per primitive and made
// the 15-sphere objects field of the Panda 60-70 k cycles per launch.)
// Returns the hinge value relu(margin - sdf) (0 when inactive); `force` = its gradient w.r.t. p.
template <int DIM>
__device__ __forceinline__ float objects_force(const float* __restrict__ prims, const mpdx_field& f, const float (&p)[DIM], float margin,
                                               float (&force)[DIM]) {
    float best = 3.0e38f;
    int bi = -1;  // arg-min: sphere index, or n_spheres + box index
    const float* sp = prims + f.sphere_off;
    for (int s0 = 0; s0 < f.n_spheres; s0 += 4) {
        float c[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int si = (s0 + u < f.n_spheres) ? s0 + u : f.n_spheres - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) c[u][j] = sp[si * 4 + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float n2 = 0.f;
#pragma unroll
            for (int j = 0; j < DIM; ++j) { const float d = p[j] - c[u][j]; n2 += d * d; }
            const float sd = (s0 + u < f.n_spheres) ? __builtin_amdgcn_sqrtf(n2) - c[u][3] : 3.0e38f;
            const bool better = sd < best;
            best = better ? sd : best;
            bi = better ? s0 + u : bi;
        }
    }
    const float* bp = prims + f.box_off;
    for (int s0 = 0; s0 < f.n_boxes; s0 += 2) {
        float c[2][6];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int si = (s0 + u < f.n_boxes) ? s0 + u : f.n_boxes - 1;
#pragma unroll
            for (int j = 0; j < 6; ++j) c[u][j] = bp[si * 6 + j];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float mx = -3.0e38f, n2 = 0.f;
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
                const float d = fabsf(p[j] - c[u][j]) - c[u][3 + j];
                mx = fmaxf(mx, d);
                const float r = fmaxf(d, 0.f);
                n2 += r * r;
            }
            const float sd = (s0 + u < f.n_boxes) ? fminf(mx, 0.f) + __builtin_amdgcn_sqrtf(n2) : 3.0e38f;
            const bool better = sd < best;
            best = better ? sd : best;
            bi = better ? f.n_spheres + s0 + u : bi;
        }
    }
    // gradient of the arg-min primitive (only if the hinge is active)
#pragma unroll
    for (int j = 0; j < DIM; ++j) force[j] = 0.f;
    if ((margin - best) > 0.f && bi >= 0) {
        if (bi < f.n_spheres) {
            float d[DIM], n2 = 0.f;
#pragma unroll
            for (int j = 0; j < DIM; ++j) { d[j] = p[j] - sp[bi * 4 + j]; n2 += d[j] * d[j]; }
            const float inv = n2 > 0.f ? __builtin_amdgcn_rsqf(n2) : 0.f;
#pragma unroll
            for (int j = 0; j < DIM; ++j) force[j] = -d[j] * inv;
        } else {
            const int bb = bi - f.n_spheres;
            float d[DIM], sg[DIM], mx = -3.0e38f, n2 = 0.f;
            int jm = 0;
#pragma unroll
            for (int j = 0; j < DIM; ++j) {
                const float cc = p[j] - bp[bb * 6 + j];
                sg[j] = cc > 0.f ? 1.f : (cc < 0.f ? -1.f : 0.f);
                d[j] = fabsf(cc) - bp[bb * 6 + 3 + j];
                jm = d[j] > mx ? j : jm;
                mx = fmaxf(mx, d[j]);
                const float r = fmaxf(d[j], 0.f);
                n2 += r * r;
            }
            const bool outside = mx > 0.f;
            const float inv = outside ? __builtin_amdgcn_rsqf(n2) : 0.f;
#pragma unroll
            for (int j = 0; j < DIM; ++j)  // outside: gradient of |relu(d)|; inside (or on the surface): gradient of max_j d_j
                force[j] = -(outside ? sg[j] * fmaxf(d[j], 0.f) * inv : (j == jm ? sg[j] : 0.f));
        }
        return margin - best;
    }
    return 0.f;
}

// objects_force for NP points at once (the 2-3 link spheres of a Panda sphere group): ONE scan of the primitive table - a batch's LDS
// reads are shared by the points, whose distance chains are independent (3 x the instruction-level parallelism of one point's scan,
// which is a serial min / arg-min chain).  Per point exactly the arithmetic of objects_force: same bits.
template <int DIM, int NPT, int UB = 2>
__device__ __forceinline__ void objects_force_n(const float* __restrict__ prims, const mpdx_field& f, const float (&p)[NPT][DIM], const float (&margin)[NPT],
                                                float (&force)[NPT][DIM]) {
    float best[NPT];
    int bi[NPT];
#pragma unroll
    for (int n = 0; n < NPT; ++n) { best[n] = 3.0e38f; bi[n] = -1; }
    const float* sp = prims + f.sphere_off;
    for (int s0 = 0; s0 < f.n_spheres; s0 += UB) {
        float c[UB][4];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int si = (s0 + u < f.n_spheres) ? s0 + u : f.n_spheres - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) c[u][j] = sp[si * 4 + j];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
#pragma unroll
            for (int n = 0; n < NPT; ++n) {
                float n2 = 0.f;
#pragma unroll
                for (int j = 0; j < DIM; ++j) { const float d = p[n][j] - c[u][j]; n2 += d * d; }
                const float sd = (s0 + u < f.n_spheres) ? __builtin_amdgcn_sqrtf(n2) - c[u][3] : 3.0e38f;
                const bool better = sd < best[n];
                best[n] = better ? sd : best[n];
                bi[n] = better ? s0 + u : bi[n];
            }
        }
    }
    const float* bp = prims + f.box_off;
    for (int s0 = 0; s0 < f.n_boxes; s0 += 2) {
        float c[2][6];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int si = (s0 + u < f.n_boxes) ? s0 + u : f.n_boxes - 1;
#pragma unroll
            for (int j = 0; j < 6; ++j) c[u][j] = bp[si * 6 + j];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int n = 0; n < NPT; ++n) {
                float mx = -3.0e38f, n2 = 0.f;
#pragma unroll
                for (int j = 0; j < DIM; ++j) {
                    const float d = fabsf(p[n][j] - c[u][j]) - c[u][3 + j];
                    mx = fmaxf(mx, d);
                    const float r = fmaxf(d, 0.f);
                    n2 += r * r;
                }
                const float sd = (s0 + u < f.n_boxes) ? fminf(mx, 0.f) + __builtin_amdgcn_sqrtf(n2) : 3.0e38f;
                const bool better = sd < best[n];
                best[n] = better ? sd : best[n];
                bi[n] = better ? f.n_spheres + s0 + u : bi[n];
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NPT; ++n) {
#pragma unroll
        for (int j = 0; j < DIM; ++j) force[n][j] = 0.f;
        if ((margin[n] - best[n]) > 0.f && bi[n] >= 0) {
            if (bi[n] < f.n_spheres) {
                float d[DIM], n2 = 0.f;
#pragma unroll
                for (int j = 0; j < DIM; ++j) { d[j] = p[n][j] - sp[bi[n] * 4 + j]; n2 += d[j] * d[j]; }
                const float inv = n2 > 0.f ? __builtin_amdgcn_rsqf(n2) : 0.f;
#pragma unroll
                for (int j = 0; j < DIM; ++j) force[n][j] = -d[j] * inv;
            } else {
                const int bb = bi[n] - f.n_spheres;
                float d[DIM], sg[DIM], mx = -3.0e38f, n2 = 0.f;
                int jm = 0;
#pragma unroll
                for (int j = 0; j < DIM; ++j) {
                    const float cc = p[n][j] - bp[bb * 6 + j];
                    sg[j] = cc > 0.f ? 1.f : (cc < 0.f ? -1.f : 0.f);
                    d[j] = fabsf(cc) - bp[bb * 6 + 3 + j];
                    jm = d[j] > mx ? j : jm;
                    mx = fmaxf(mx, d[j]);
                    const float r = fmaxf(d[j], 0.f);
                    n2 += r * r;
                }
                const bool outside = mx > 0.f;
                const float inv = outside ? __builtin_amdgcn_rsqf(n2) : 0.f;
#pragma unroll
                for (int j = 0; j < DIM; ++j) force[n][j] = -(outside ? sg[j] * fmaxf(d[j], 0.f) * inv : (j == jm ? sg[j] : 0.f));
            }
        }
    }
}

template <int DIM>
__device__ __forceinline__ void workspace_force(const mpdx_field& f, const float (&p)[DIM], float margin, float (&force)[DIM]) {
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const float lo = p[j] - f.ws_min[j], hi = f.ws_max[j] - p[j];
        force[j] = ((margin - lo) > 0.f ? -1.f : 0.f) + ((margin - hi) > 0.f ? 1.f : 0.f);
    }
}

// Panda forward kinematics (modified DH): frame origins O_k and z axes Z_k in the world frame
// NJ < 7: only the first NJ frames (a sphere group whose spheres sit on frames <= NJ needs no more; rows k >= NJ are left untouched)
template <int QD, int NJ = 7>
__device__ __forceinline__ void panda_fk(const float (&q)[QD], float (&O)[7][3], float (&Z)[7][3]) {
    float R[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}, T[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        float st, ct;
        __sincosf(q[k < QD ? k : 0], &st, &ct);  // |q| <= pi (joint limits): the hardware sin/cos is accurate to ~1e-6 here
        const float ca = kPandaCA[k], sa = kPandaSA[k], aa = kPandaA[k], dd = kPandaD[k];
        const float L[3][3] = {{ct, -st, 0.f}, {st * ca, ct * ca, -sa}, {st * sa, ct * sa, ca}};
        const float Lt[3] = {aa, -sa * dd, ca * dd};
        float Rn[3][3], Tn[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Rn[r][c] = R[r][0] * L[0][c] + R[r][1] * L[1][c] + R[r][2] * L[2][c];
            Tn[r] = R[r][0] * Lt[0] + R[r][1] * Lt[1] + R[r][2] * Lt[2] + T[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) R[r][c] = Rn[r][c];
            T[r] = Tn[r];
            O[k][r] = Tn[r];
            Z[k][r] = Rn[r][2];
        }
    }
}

// minimum signed distance of p to the field's primitives (collision checking)
template <int DIM>
__device__ __forceinline__ float objects_sdf(const float* __restrict__ prims, const mpdx_field& f, const float (&p)[DIM]) {
    float best = 3.0e38f;
    const float* sp = prims + f.sphere_off;
    for (int s = 0; s < f.n_spheres; ++s) {
        float n2 = 0.f;
#pragma unroll
        for (int j = 0; j < DIM; ++j) { const float d = p[j] - sp[s * 4 + j]; n2 += d * d; }
        best = fminf(best, sqrtf(n2) - sp[s * 4 + 3]);
    }
    const float* bp = prims + f.box_off;
    for (int s = 0; s < f.n_boxes; ++s) {
        float mx = -3.0e38f, n2 = 0.f;
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            const float d = fabsf(p[j] - bp[s * 6 + j]) - bp[s * 6 + 3 + j];
            mx = fmaxf(mx, d);
            const float r = fmaxf(d, 0.f);
            n2 += r * r;
        }
        best = fminf(best, fminf(mx, 0.f) + sqrtf(n2));
    }
    return best;
}

// Post-loop trajectory metrics (the arithmetic of task.get_trajs_collision_and_free / compute_collision_intensity_trajs
// and torch_robotics.trajectory.metrics.compute_smoothness / compute_path_length, called at inference.py:288-297,
// 311-316 - un-vendored, restated): per trajectory
//   out[b][0] = number of interpolated waypoints in collision (a link sphere penetrates an object, leaves the
//               workspace, or self-collides; margin = link radius, no cutoff margin)
//   out[b][1] = path length  sum_h |q_{h+1} - q_h|      out[b][2] = smoothness  sum_h |v_{h+1} - v_h|
//   out[b][3] = number of interpolated waypoints checked
//   mask[b][i] (optional) = 1 if interpolated waypoint i collides (what out[b][0] counts)
// x is UNNORMALISED [B,H,D] (inference.py:285 un-normalises before computing metrics).
template <int QD, int DIM, int ROBOT>
__global__ __launch_bounds__(64) void traj_metrics_kernel(const mpdx_guide_params gp, const float* __restrict__ x, float* __restrict__ out,
                                                          int B, int H, int n_check, uint8_t* __restrict__ mask) {
    constexpr int D = 2 * QD;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, b = blockIdx.x;
    float* sx = sm;
    float* sprim = sx + H * D;
    for (int i = lane; i < gp.n_prim_floats; i += 64) sprim[i] = gp.prims[i];
    for (int i = lane; i < H * D; i += 64) sx[i] = x[(size_t)b * H * D + i];
    __syncthreads();
    float plen = 0.f, smooth = 0.f;
    for (int h = lane; h < H - 1; h += 64) {   // (H <= 64: one segment per lane, as before)
        float a2 = 0.f, v2 = 0.f;
#pragma unroll
        for (int j = 0; j < QD; ++j) {
            const float dq = sx[(h + 1) * D + j] - sx[h * D + j], dv = sx[(h + 1) * D + QD + j] - sx[h * D + QD + j];
            a2 += dq * dq; v2 += dv * dv;
        }
        plen += sqrtf(a2); smooth += sqrtf(v2);
    }
    const int N = n_check;
    const float scale = (N > 1) ? (float)(H - 1) / (float)(N - 1) : 0.f;
    float ncoll = 0.f;
    for (int i = lane; i < N; i += 64) {
        const float u = scale * (float)i;
        int i0 = (int)u;
        if (i0 > H - 1) i0 = H - 1;
        const int i1 = i0 + 1 < H ? i0 + 1 : H - 1;
        const float l1 = u - (float)i0, l0 = 1.0f - l1;
        float q[QD];
#pragma unroll
        for (int j = 0; j < QD; ++j) q[j] = l0 * sx[i0 * D + j] + l1 * sx[i1 * D + j];
        bool hit = false;
        if (ROBOT == MPDX_ROBOT_POINTMASS) {
            float p[DIM];
#pragma unroll
            for (int j = 0; j < DIM; ++j) p[j] = q[j];
            for (int f = 0; f < gp.n_fields; ++f) {
                if (gp.fields[f].kind == MPDX_FIELD_OBJECTS) hit |= objects_sdf<DIM>(sprim, gp.fields[f], p) < gp.link_margin;
                else if (gp.fields[f].kind == MPDX_FIELD_WORKSPACE) {
#pragma unroll
                    for (int j = 0; j < DIM; ++j) hit |= (p[j] - gp.fields[f].ws_min[j] < gp.link_margin) || (gp.fields[f].ws_max[j] - p[j] < gp.link_margin);
                }
            }
        } else {
            float O[7][3], Z[7][3];
            panda_fk(q, O, Z);
            float P[kPandaNS][3];
#pragma unroll
            for (int s = 0; s < kPandaNS; ++s)
#pragma unroll
                for (int r = 0; r < 3; ++r) P[s][r] = O[kPandaSF[s] - 1][r] + kPandaSO[s] * Z[kPandaSF[s] - 1][r];
            for (int f = 0; f < gp.n_fields; ++f) {
                const int kind = gp.fields[f].kind;
                if (kind == MPDX_FIELD_SELF) {
#pragma unroll
                    for (int pr = 0; pr < kPandaNP; ++pr) {
                        const float dx = P[kPandaPA[pr]][0] - P[kPandaPB[pr]][0], dy = P[kPandaPA[pr]][1] - P[kPandaPB[pr]][1],
                                    dz = P[kPandaPA[pr]][2] - P[kPandaPB[pr]][2];
                        hit |= sqrtf(dx * dx + dy * dy + dz * dz) < kPandaSR[kPandaPA[pr]] + kPandaSR[kPandaPB[pr]];
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < kPandaNS; ++s) {
                        const float p3[3] = {P[s][0], P[s][1], P[s][2]};
                        if (kind == MPDX_FIELD_OBJECTS) hit |= objects_sdf<3>(sprim, gp.fields[f], p3) < kPandaSR[s];
                        else {
#pragma unroll
                            for (int j = 0; j < 3; ++j) hit |= (p3[j] - gp.fields[f].ws_min[j] < kPandaSR[s]) || (gp.fields[f].ws_max[j] - p3[j] < kPandaSR[s]);
                        }
                    }
                }
            }
        }
        ncoll += hit ? 1.f : 0.f;
        if (mask) mask[(size_t)b * N + i] = hit ? 1 : 0;   // per-waypoint collision flags (mpdx_traj_metrics_mask)
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        ncoll += __shfl_xor(ncoll, s, 64);
        plen += __shfl_xor(plen, s, 64);
        smooth += __shfl_xor(smooth, s, 64);
    }
    if (lane == 0) {
        out[(size_t)b * 4 + 0] = ncoll; out[(size_t)b * 4 + 1] = plen; out[(size_t)b * 4 + 2] = smooth; out[(size_t)b * 4 + 3] = (float)N;
    }
}

// clip_gradient (guides.py:213-236) of one waypoint's gradient g[0..N) (the remaining D - N dims of the term are zero):
//   rule 'norm'  : g * clip(|g + 1e-6|, 0, max_norm) / |g + 1e-6|, norm over ALL D dims
//   rule 'value' : clip(g, -max_value, max_value) per element
template <int N>
__device__ __forceinline__ void clip_waypoint_grad(const mpdx_guide_params& gp, float (&g)[N], int n_zero_dims) {
    if (!gp.clip_grad) return;
    if (gp.clip_rule == 1) {
#pragma unroll
        for (int j = 0; j < N; ++j) g[j] = fminf(fmaxf(g[j], -gp.max_grad_value), gp.max_grad_value);
        return;
    }
    float n2 = (float)n_zero_dims * (1e-6f * 1e-6f);
#pragma unroll
    for (int j = 0; j < N; ++j) n2 += (g[j] + 1e-6f) * (g[j] + 1e-6f);
    const float n = sqrtf(n2);
    const float ratio = fminf(fmaxf(n, 0.f), gp.max_grad_norm) / n;
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = ratio * g[j];
}

// GP prior (constant-velocity, GPMP2: 3-point stencil over the horizon) added to the gathered collision gradient, then
//     x = x + (-grad);  [+ the step's noise term on the last guide iteration];  hard conditioning;  max|x| for the next
// range test.  One wave, lane = support point.  tr: optional two cycle stamps (dev tool).
// snoise (or null): the step's noise of THIS trajectory, drawn by idle waves into LDS (guide_draw_noise: element k of the trajectory at
// snoise[k + (e0 & 3)]) - the same Philox values philox_normal_at gives, one counter per FOUR elements instead of one per element.
template <int QD>
__device__ __forceinline__ void guide_gp_apply(const GuideArgs& a, int b, int ctx, int lane, int h, int H, bool live, const float (&xn)[2 * QD],
                                               const float (&xu)[2 * QD], const float* sx, float (&total)[2 * QD], size_t base, long long* tr,
                                               const float* snoise = nullptr, const float* shc = nullptr) {
    constexpr int D = 2 * QD;
    const bool interior = live && h > 0 && h < H - 1;
    if (a.gp.use_gp) {
        const float dt = a.gp.dt, s2 = (a.gp.gp_half_factor ? 0.5f : 1.0f) / (a.gp.sigma_gp * a.gp.sigma_gp);
        const float c_qq = 24.0f / (dt * dt * dt), c_qv = 12.0f / (dt * dt), c_vv = 8.0f / dt;
        // a_h = d c_h / d e_q,  b_h = d c_h / d e_v  for the segment (h, h+1); (ap, bp) the same for the segment (h-1, h).  Both are
        // computed from the LDS-staged state (the neighbour's registers hold exactly these floats): no cross-lane traffic, so the
        // support index may cross a wave boundary (H up to 128: two waves of supports)
        float g[D];
#pragma unroll
        for (int j = 0; j < QD; ++j) {
            float eq = 0.f, ev = 0.f, eqp = 0.f, evp = 0.f;
            if (live && h < H - 1) {
                eq = sx[(h + 1) * D + j] - xu[j] - dt * xu[QD + j];
                ev = sx[(h + 1) * D + QD + j] - xu[QD + j];
            }
            if (live && h > 0) {
                eqp = sx[h * D + j] - sx[(h - 1) * D + j] - dt * sx[(h - 1) * D + QD + j];
                evp = sx[h * D + QD + j] - sx[(h - 1) * D + QD + j];
            }
            const float av = (c_qq * eq - c_qv * ev) * s2, bv = (-c_qv * eq + c_vv * ev) * s2;
            const float ap = (c_qq * eqp - c_qv * evp) * s2, bp = (-c_qv * eqp + c_vv * evp) * s2;
            g[j] = ap - av;
            g[QD + j] = bp - bv - dt * av;
        }
        clip_waypoint_grad<D>(a.gp, g, 0);
        if (interior) {
#pragma unroll
            for (int d = 0; d < D; ++d) total[d] += a.gp.gp_weight * g[d];
        }
    }

    if (tr) tr[0] = (long long)__builtin_readcyclecounter();  // GP term done
    // ---- apply:  x = x + (-grad);  hard conditioning;  max|x| for the next range test
    float vmax = 0.f;
    if (live) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float inc = __fmul_rn(a.guide_scale, -1.0f * total[d]);
            if (a.grad_out) {
                a.grad_out[base + d] = inc;
            } else {
                float r = __fadd_rn(xn[d], inc);
                if (a.rng.on) {
                    const float z = snoise ? snoise[h * D + d] : philox_normal_at(a.rng.seed, a.rng.offset, a.rng.elem0 + base + d);
                    r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.noise_scale, z), a.noise_extra));
                }
                else if (a.noise) r = __fadd_rn(r, __fmul_rn(__fmul_rn(a.noise_scale, a.noise[base + d]), a.noise_extra));
                // hard conditions: from LDS when the prologue staged this trajectory's two rows (shc: [start | goal] x D; round 6 - the global loads sat on
                // the kernel's tail: 7 us of a 200-us Panda launch at B = 6400, tools/guide_inplan_probe.py), else from global memory
                if (a.hs && h == 0) r = shc ? shc[d] : a.hs[(size_t)b * D + d];
                if (a.hg && h == H - 1) r = shc ? shc[D + d] : a.hg[(size_t)b * D + d];
                a.x[base + d] = r;
                if (a.chain) a.chain[base + d] = r;
                vmax = fmaxf(vmax, fabsf(r));
            }
        }
    }
    if (a.amax_out && !a.grad_out) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, s, 64));
        if (lane == 0) atomicMax(a.amax_out + ctx, __float_as_uint(vmax));
    }
    if (tr) tr[1] = (long long)__builtin_readcyclecounter();  // applied
}

// The step's noise for one trajectory (H * D consecutive elements of the plan's Philox stream, first element e0), drawn by `nthr`
// threads (thread `t` of them) into LDS: one Philox counter yields FOUR consecutive elements, so H * D / 4 (+1 when e0 is not a multiple
// of 4) evaluations serve the trajectory - the support wave used to evaluate one counter PER ELEMENT (14 per lane for the Panda: the
// noise launches took 330 us against 200 us for the others at B = 6400, profiles/r05_cfg5_kernel_stats_a.csv).  sn[k] = element e0 - (e0 & 3) + k.
__device__ __forceinline__ void guide_draw_noise(const NoiseRng& rng, unsigned long long e0, int n_elems, float* sn, int t, int nthr) {
    const unsigned long long q0 = e0 >> 2;
    const int nq = (int)(((e0 + (unsigned long long)n_elems + 3ull) >> 2) - q0);
    for (int k = t; k < nq; k += nthr) {
        float z[4];
        philox_normal4(rng.seed, rng.offset + q0 + (unsigned long long)k, z);
        *(f32x4*)(sn + 4 * k) = (f32x4){z[0], z[1], z[2], z[3]};
    }
}

// Point-mass robots (QD = DIM = 2 or 3).  WPT = waves per trajectory.  The collision part (SDF force per interpolated point
// and per field) is split over WPT waves as (point slice) x (field): PW = min(WPT,2) point slices, WPT/PW field slots;
// wave 0 then gathers, clips, adds the GP term and applies the update.  WPT = 8 (2 point halves x up to 4 fields): the
// single-wave version is a long serial latency chain (2-D: 19 us per launch; 8 waves: 9 us).  The Panda has its own
// kernel below (guide_step_panda_kernel).
template <int QD, int DIM, int ROBOT, int WPT>
__global__ __launch_bounds__(64 * WPT) void guide_step_kernel(const GuideArgs a) {
    static_assert(ROBOT == MPDX_ROBOT_POINTMASS, "the Panda has its own kernel (guide_step_panda_kernel)");
    constexpr int D = 2 * QD;
    constexpr int MAXF = MPDX_MAX_FIELDS;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const mpdx_guide_params& gp = a.gp;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PW = WPT >= 2 ? 2 : 1, FW = WPT / PW;
    const int b = blockIdx.x;
    const int H = a.H;
    const int N = gp.interpolate ? gp.n_interp : H;
    // support points: lane of wave sw = support sw * 64 + lane; H <= 64: one support wave (wave 0), H <= 128: two (waves 0, 1)
    const int nsw = (H + 63) >> 6;
    const int hs_ = (wv < nsw ? wv : 0) * 64 + lane;   // this thread's support index (waves >= nsw shadow block 0: their copy is unused)
    const bool live = hs_ < H;
    int tr_i = 0;
#define G_STAMP() do { if (MPDX_TRACE_PTR(a.trace) && b == 0 && lane == 0) a.trace[wv * 16 + tr_i] = (long long)__builtin_readcyclecounter(); ++tr_i; } while (0)
    G_STAMP();  // 0 entry
    // LDS carve: unnormalised state [H][D] | point forces A,B [MAXF][N][QD] each | primitive table
    float* sx = sm;
    float* sA = sx + H * D;
    float* sB = sA + MAXF * N * QD;
    float* sprim = sB + MAXF * N * QD;
    for (int i = threadIdx.x; i < gp.n_prim_floats; i += 64 * WPT) sprim[i] = gp.prims[i];

    // ---- load + unnormalise (normalization.py:156-167)
    const int ctx = b / a.n_per_ctx;
    const bool clipall = __uint_as_float(a.amax_in[ctx]) > 1.0001f;  // x.max() > 1+eps or x.min() < -1-eps (eps = 1e-4)
    float xn[D], xu[D];
    const size_t base = ((size_t)b * H + (live ? hs_ : 0)) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xn[d] = live ? a.x[base + d] : 0.f;
        const float c = clipall ? fminf(fmaxf(xn[d], -1.f), 1.f) : xn[d];
        const float u01 = __fadd_rn(c, 1.0f) * 0.5f;
        // identity_normalizer: 0 limits (normalization.py:156-167), 1 Identity (:111-116), 2 GaussianNormalizer (:140-141: x * stds + means - the
        // host passes means in `mins`, stds in `maxs`; no range test)
        xu[d] = gp.identity_normalizer == 1 ? xn[d] : gp.identity_normalizer == 2 ? __fadd_rn(__fmul_rn(xn[d], gp.maxs[d]), gp.mins[d])
                                                    : __fadd_rn(__fmul_rn(u01, __fsub_rn(gp.maxs[d], gp.mins[d])), gp.mins[d]);
        if (live && wv < nsw) sx[hs_ * D + d] = xu[d];
    }
    __syncthreads();
    G_STAMP();  // 1 state unnormalised + staged

    // ---- collision terms on the interpolated positions
    const float scale = (N > 1) ? (float)(H - 1) / (float)(N - 1) : 0.f;  // align_corners=True
    for (int i = (wv % PW) * 64 + lane; i < N; i += 64 * PW) {
        int i0 = i, i1 = i;
        float l0 = 1.f, l1 = 0.f;
        if (gp.interpolate) {
            const float u = scale * (float)i;
            i0 = (int)u;
            if (i0 > H - 1) i0 = H - 1;
            i1 = i0 + 1 < H ? i0 + 1 : H - 1;
            l1 = u - (float)i0;
            l0 = 1.0f - l1;
        }
        float q[QD];
#pragma unroll
        for (int j = 0; j < QD; ++j) q[j] = l0 * sx[i0 * D + j] + l1 * sx[i1 * D + j];

        if (ROBOT == MPDX_ROBOT_POINTMASS) {
            float p[DIM];
#pragma unroll
            for (int j = 0; j < DIM; ++j) p[j] = q[j];
            const float margin = gp.link_margin + gp.cutoff_margin;
            for (int f = 0; f < gp.n_fields; ++f) {
                if ((f % FW) != wv / PW) continue;
                float force[DIM];
                if (gp.fields[f].kind == MPDX_FIELD_OBJECTS) objects_force<DIM>(sprim, gp.fields[f], p, margin, force);
                else if (gp.fields[f].kind == MPDX_FIELD_WORKSPACE) workspace_force<DIM>(gp.fields[f], p, margin, force);
                else {
#pragma unroll
                    for (int j = 0; j < DIM; ++j) force[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < QD; ++j) {
                    sA[(f * N + i) * QD + j] = l0 * force[j];
                    sB[(f * N + i) * QD + j] = l1 * force[j];
                }
            }
        }
    }
    G_STAMP();  // 2 this wave's (point slice, field) done
    __syncthreads();
    G_STAMP();  // 3 all waves done
    // ---- gather to support points (transpose of the interpolation), clip, zero ends, weight: wave f handles field f and
    //      leaves its clipped, weighted gradient in LDS; wave 0 then adds the fields in order (same additions as one wave
    //      looping over the fields)
    float* sC = sB;   // [MAXF][H][QD]: overlays sB after the barrier below (every gatherer has its sums in registers by then)
    constexpr int NSB = 2;   // support blocks of 64 (H <= 128)
    float cg[NSB][QD];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int j = 0; j < QD; ++j) cg[sb][j] = 0.f;
    const bool gatherer = wv < gp.n_fields;
    if (gatherer) {
        const int f = wv;
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            const int hg_ = sb * 64 + lane;   // the support this lane gathers in block sb
            if (sb >= nsw || hg_ >= H) continue;
            int ilo = hg_, ihi = hg_;
            if (gp.interpolate && scale > 0.f) {
                ilo = (int)((float)(hg_ - 1) / scale) - 1;   // points of the segments (hg-1, hg) and (hg, hg+1): any number per segment
                ihi = (int)((float)(hg_ + 1) / scale) + 1;
                if (ilo < 0) ilo = 0;
                if (ihi > N - 1) ihi = N - 1;
            }
            float g[QD];
#pragma unroll
            for (int j = 0; j < QD; ++j) g[j] = 0.f;
            for (int i = ilo; i <= ihi; ++i) {
                int i0 = i, i1 = i;
                if (gp.interpolate) {
                    const float u = scale * (float)i;
                    i0 = (int)u;
                    if (i0 > H - 1) i0 = H - 1;
                    i1 = i0 + 1 < H ? i0 + 1 : H - 1;
                }
                if (i0 == hg_) {
#pragma unroll
                    for (int j = 0; j < QD; ++j) g[j] += sA[(f * N + i) * QD + j];
                }
                if (i1 == hg_ && gp.interpolate) {
#pragma unroll
                    for (int j = 0; j < QD; ++j) g[j] += sB[(f * N + i) * QD + j];
                }
            }
            // clip over ALL D dims of (g + 1e-6): the velocity dims of a collision gradient are 0
            clip_waypoint_grad<QD>(gp, g, QD);
            if (hg_ > 0 && hg_ < H - 1) {
#pragma unroll
                for (int j = 0; j < QD; ++j) cg[sb][j] = gp.fields[f].weight * g[j];
            }
        }
    }
    static_assert(WPT >= MAXF, "one gathering wave per field");
    __syncthreads();   // every gatherer has read its sA / sB slabs
    if (gatherer) {
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
            const int hg_ = sb * 64 + lane;
            if (sb >= nsw || hg_ >= H) continue;
#pragma unroll
            for (int j = 0; j < QD; ++j) sC[(wv * H + hg_) * QD + j] = cg[sb][j];
        }
    }
    __syncthreads();
    if (wv >= nsw) return;   // the support wave(s) finish the trajectory (no further workgroup barriers below)
    float total[D];
#pragma unroll
    for (int d = 0; d < D; ++d) total[d] = 0.f;
    if (live) {
        for (int f = 0; f < gp.n_fields; ++f) {
#pragma unroll
            for (int j = 0; j < QD; ++j) total[j] += sC[(f * H + hs_) * QD + j];
        }
    }
    G_STAMP();  // 4 gathered + clipped
    guide_gp_apply<QD>(a, b, ctx, lane, hs_, H, live, xn, xu, sx, total, base, (MPDX_TRACE_PTR(a.trace) && b == 0 && lane == 0) ? a.trace + wv * 16 + 5 : nullptr);
#undef G_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// Panda variant of the guide update.  Same arithmetic as guide_step_kernel<7,3,PANDA,8>, different work split:
// stamps of that kernel showed all four SIMDs of the CU saturated with VALU work for ~75 k cycles, most of it redundant
// (every (point half, field) wave recomputed the full FK) or badly balanced (the 15-obstacle field is 3x the others).
//   phase 1  FK once per interpolated point (2 of the 8 waves), frame origins / z axes / link-sphere centres -> LDS
//   phase 2  wave = (point half, sphere group): for EVERY field, the forces on its 2-3 link spheres (or 3 self-collision
//            pairs), folded per joint j into total force / moment of the spheres that joint moves,
//                g_j = z_j . ( sum P_s x F_s  -  O_j x sum F_s ),     s over spheres with frame(s) >= j
//            (d P_s / d theta_j = z_j x (P_s - O_j)); partial joint gradients per (field, group) -> LDS
//   phase 3  wave f gathers field f to the support points (transpose of the interpolation, summing the groups in a fixed
//            order), clips by norm, weights
//   phase 4  wave 0: sum over fields, GP prior, apply (guide_gp_apply)
constexpr int kPandaFKS = 75;   // floats per interpolated point in LDS: O[7][3] | Z[7][3] | P[11][3]  (odd stride: no bank conflicts)
// Sphere groups (round 5).  The kernel is VALU-issue bound once two workgroups share a CU (stamps + ISA census, profiles/r05_guide_*):
// what counts is the number of instructions, so the split follows the kinematic chain - a group's joint gradients and (in the dense
// variant) its forward kinematics stop at the highest frame its spheres sit on:
//     group 0: spheres 0-3  (frames 1,1,3,3) -> joints 0-2      group 2: spheres 7-9 (frames 5,7,7) -> all 7 joints
//     group 1: spheres 4-6  (frames 4,5,5)   -> joints 0-4      group 3: sphere 10 (frame 7) + ALL 12 self-collision pairs
// (round 4: {0-2, 3-5, 6-8, 9-10} with 3 pairs each: every group touched frame 7 through its pairs, i.e. 4 x the full FK and 4 x a
// 7-joint fold for the self field).  Entries of sG that are zero by construction are neither written nor read.
constexpr int kPandaParts = 4;
constexpr int panda_group_first(int part) { return part == 0 ? 0 : part == 1 ? 4 : part == 2 ? 7 : part == 3 ? 10 : 11; }
constexpr int panda_group_joints(int part) { return part == 0 ? 3 : part == 1 ? 5 : 7; }      // joints that move the group's spheres
// (the self field lives in part 3 alone, with all 7 joints)

// phase 2 of guide_step_panda_kernel for sphere / pair group PART and point half `half`
// FKREG: the forward kinematics of the point are evaluated HERE from the LDS-staged state (sx, H, D, scale) instead of being read from
// the per-point FK table of phase 1 (sfk): 4 x the FK work, no 38-KB table - the large-batch variant of the kernel (below).
template <int PART, bool FKREG>
__device__ __forceinline__ void panda_group_forces(const mpdx_guide_params& gp, const float* sprim, const float* sfk, float* sG, int half, int lane, int N,
                                                   long long* tr, const float* sx = nullptr, int H = 0, float scale = 0.f) {
    constexpr int QD = 7, NP = kPandaParts, D = 14;
    constexpr int s_beg = panda_group_first(PART), s_end = panda_group_first(PART + 1), NG = s_end - s_beg;
    constexpr int NJ = panda_group_joints(PART);
    constexpr bool SELF = PART == NP - 1;   // this group also carries the self-collision pairs
    for (int i = half * 64 + lane; i < N; i += 128) {
        float O[7][3], Z[7][3], P[kPandaNS][3];
        if constexpr (FKREG) {
            int i0 = i, i1 = i;
            float l0 = 1.f, l1 = 0.f;
            if (gp.interpolate) {
                const float u = scale * (float)i;
                i0 = (int)u;
                if (i0 > H - 1) i0 = H - 1;
                i1 = i0 + 1 < H ? i0 + 1 : H - 1;
                l1 = u - (float)i0;
                l0 = 1.0f - l1;
            }
            float q[QD];
#pragma unroll
            for (int j = 0; j < QD; ++j) q[j] = l0 * sx[i0 * D + j] + l1 * sx[i1 * D + j];
            panda_fk<QD, NJ>(q, O, Z);
#pragma unroll
            for (int s = 0; s < kPandaNS; ++s) {
                if (kPandaSF[s] > NJ) continue;   // (compile time after unrolling: frames beyond the group's chain are never read)
#pragma unroll
                for (int r = 0; r < 3; ++r) P[s][r] = O[kPandaSF[s] - 1][r] + kPandaSO[s] * Z[kPandaSF[s] - 1][r];
            }
        } else {
            const float* fk = sfk + i * kPandaFKS;
#pragma unroll
            for (int k = 0; k < NJ; ++k) {
#pragma unroll
                for (int r = 0; r < 3; ++r) { O[k][r] = fk[k * 3 + r]; Z[k][r] = fk[21 + k * 3 + r]; }
            }
#pragma unroll
            for (int s = 0; s < kPandaNS; ++s) {  // only the spheres this group touches stay live
#pragma unroll
                for (int r = 0; r < 3; ++r) P[s][r] = fk[42 + s * 3 + r];
            }
        }
        // fold of per-frame forces / moments into joint gradients, stored as sG[(f, PART)][i][0 .. NJ)
        auto fold_store = [&](int f, const float (&FF)[7][3], const float (&FM)[7][3]) {
            float Ft[3] = {0.f, 0.f, 0.f}, Mt[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = NJ - 1; k >= 0; --k) {  // joint k moves every sphere on frames >= k  (joints >= NJ move none of this group's)
#pragma unroll
                for (int r = 0; r < 3; ++r) { Ft[r] += FF[k][r]; Mt[r] += FM[k][r]; }
                const float cx = O[k][1] * Ft[2] - O[k][2] * Ft[1], cy = O[k][2] * Ft[0] - O[k][0] * Ft[2], cz = O[k][0] * Ft[1] - O[k][1] * Ft[0];
                sG[((f * NP + PART) * N + i) * QD + k] = Z[k][0] * (Mt[0] - cx) + Z[k][1] * (Mt[1] - cy) + Z[k][2] * (Mt[2] - cz);
            }
        };
        if constexpr (SELF) {   // the self-collision field(s) first, in a loop of their own: the eight sphere centres the pairs touch die here
            for (int f = 0; f < gp.n_fields; ++f) {
                if (gp.fields[f].kind != MPDX_FIELD_SELF) continue;
                float FF[7][3], FM[7][3];
#pragma unroll
                for (int k = 0; k < 7; ++k) { FF[k][0] = FF[k][1] = FF[k][2] = 0.f; FM[k][0] = FM[k][1] = FM[k][2] = 0.f; }
#pragma unroll
                for (int pr = 0; pr < kPandaNP; ++pr) {
                    const int sa_ = kPandaPA[pr], sb_ = kPandaPB[pr];
                    const int fa = kPandaSF[sa_] - 1, fb = kPandaSF[sb_] - 1;
                    const float dx = P[sa_][0] - P[sb_][0], dy = P[sa_][1] - P[sb_][1], dz = P[sa_][2] - P[sb_][2];
                    const float d2 = dx * dx + dy * dy + dz * dz;
                    const float dist = __builtin_amdgcn_sqrtf(d2);
                    const float inv = (kPandaSR[sa_] + kPandaSR[sb_] - dist > 0.f && dist > 0.f) ? __builtin_amdgcn_rsqf(d2) : 0.f;
                    const float fx = dx * inv, fy = dy * inv, fz = dz * inv;  // d cost / d P_b = +f,  d cost / d P_a = -f
                    FF[fa][0] -= fx; FF[fa][1] -= fy; FF[fa][2] -= fz;
                    FF[fb][0] += fx; FF[fb][1] += fy; FF[fb][2] += fz;
                    FM[fa][0] -= P[sa_][1] * fz - P[sa_][2] * fy; FM[fa][1] -= P[sa_][2] * fx - P[sa_][0] * fz; FM[fa][2] -= P[sa_][0] * fy - P[sa_][1] * fx;
                    FM[fb][0] += P[sb_][1] * fz - P[sb_][2] * fy; FM[fb][1] += P[sb_][2] * fx - P[sb_][0] * fz; FM[fb][2] += P[sb_][0] * fy - P[sb_][1] * fx;
                }
                fold_store(f, FF, FM);
                if (tr) tr[f] = (long long)__builtin_readcyclecounter();
            }
        }
        for (int f = 0; f < gp.n_fields; ++f) {
            const int kind = gp.fields[f].kind;
            if (kind != MPDX_FIELD_OBJECTS && kind != MPDX_FIELD_WORKSPACE) continue;   // (self: above, group 3 only; sG of the other parts is never read)
            float FF[7][3], FM[7][3];  // per frame: force on its spheres, their moment about the world origin
#pragma unroll
            for (int k = 0; k < 7; ++k) { FF[k][0] = FF[k][1] = FF[k][2] = 0.f; FM[k][0] = FM[k][1] = FM[k][2] = 0.f; }
            // the group's link spheres: one scan of the primitive table for all of them
            float pg[NG][3], mg[NG], fg[NG][3];
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                pg[n][0] = P[s_beg + n][0]; pg[n][1] = P[s_beg + n][1]; pg[n][2] = P[s_beg + n][2];
                mg[n] = kPandaSR[s_beg + n] + gp.cutoff_margin;
            }
            if (kind == MPDX_FIELD_OBJECTS) objects_force_n<3, NG>(sprim, gp.fields[f], pg, mg, fg);
            else {
#pragma unroll
                for (int n = 0; n < NG; ++n) workspace_force<3>(gp.fields[f], pg[n], mg[n], fg[n]);
            }
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                const int fr = kPandaSF[s_beg + n] - 1;
                const float(&p3)[3] = pg[n];
                const float(&fo)[3] = fg[n];
                FF[fr][0] += fo[0]; FF[fr][1] += fo[1]; FF[fr][2] += fo[2];
                FM[fr][0] += p3[1] * fo[2] - p3[2] * fo[1]; FM[fr][1] += p3[2] * fo[0] - p3[0] * fo[2]; FM[fr][2] += p3[0] * fo[1] - p3[1] * fo[0];
            }
            fold_store(f, FF, FM);
            if (tr) tr[f] = (long long)__builtin_readcyclecounter();
        }
    }
}

// DENSE (large batches): no FK table in LDS (the forces phase evaluates the FK in registers, four times per point) and registers capped
// at 128: 70 KB of LDS and 4 waves per SIMD -> TWO workgroups per CU, where the latency-bound phases of one overlap the other's
// (the default variant: 107 KB, 166 VGPRs, one workgroup per CU; at B = 100 there is one workgroup per CU anyway).  Same arithmetic.
template <bool DENSE>
__global__ __launch_bounds__(512, DENSE ? 4 : 2) void guide_step_panda_kernel(const GuideArgs a) {
    constexpr int QD = 7, D = 14, MAXF = MPDX_MAX_FIELDS, NP = kPandaParts, WPT = 8;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const mpdx_guide_params& gp = a.gp;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int H = a.H;
    const int N = gp.interpolate ? gp.n_interp : H;
    // support points: lane of wave sw = support sw * 64 + lane; H <= 64: one support wave (wave 0), H <= 128: two (waves 0, 1)
    const int nsw = (H + 63) >> 6;
    const int hs_ = (wv < nsw ? wv : 0) * 64 + lane;
    const bool live = hs_ < H;
    int tr_i = 0;
#define G_STAMP() do { if (MPDX_TRACE_PTR(a.trace) && b == 0 && lane == 0) a.trace[wv * 16 + tr_i] = (long long)__builtin_readcyclecounter(); ++tr_i; } while (0)
    G_STAMP();  // 0 entry
    float* sx = sm;                           // [H][D]  unnormalised state
    float* sfk = sx + H * D;                  // [N][kPandaFKS]   (not in the DENSE variant)
    float* sG = sfk + (DENSE ? 0 : N * kPandaFKS);   // [MAXF][NP][N][QD]  partial joint gradients
    float* sC = sG + MAXF * NP * N * QD;      // [MAXF][H][QD]      clipped, weighted per-field support-point gradients
    float* snz = sC + ((MAXF * H * QD + (int)((sC - sm) & 3) + 3) & ~3) - (int)((sC - sm) & 3);   // 16-byte aligned: [H * D + 4] the step's noise of this
    float* snz_x = snz + H * D + 4;           //                    trajectory (last guide iteration of a step, rng.on) | [H * D] the normalised state
    float* sprim = snz_x + H * D;
    for (int i = threadIdx.x; i < gp.n_prim_floats; i += 64 * WPT) sprim[i] = gp.prims[i];
    float* shc = sprim + ((gp.n_prim_floats + 3) & ~3);   // [2][D] this trajectory's hard conditions (apply mode), staged here: their loads fly with the state's
    if (!a.grad_out && (int)threadIdx.x < 2 * D) {
        const int which = (int)threadIdx.x >= D ? 1 : 0, d = (int)threadIdx.x - which * D;
        const float* p = which ? a.hg : a.hs;
        if (p) shc[which * D + d] = p[(size_t)b * D + d];
    }

    // ---- load + unnormalise (normalization.py:156-167), by the whole workgroup: the trajectory's H * D floats are CONTIGUOUS - one coalesced
    //      16-byte load per thread into LDS (sxn), then element-wise unnormalisation (two elements per thread).  (Rounds 1-4: the support
    //      wave read its D floats per lane with D strided loads while the other seven waves waited at the barrier - 4.7-6.3 k cycles per
    //      workgroup, tools/guide_trace.py.)  The state is NOT kept in registers across the force phases (28 VGPRs that every wave held -
    //      and, in the 128-register dense variant, spilled to scratch: 112 B per lane, 148 MB written per launch at B = 6400): phase 4
    //      reads the normalised state back from sxn and the unnormalised one from sx.
    const int ctx = b / a.n_per_ctx;
    {
        const int n = H * D;
        const float* const xb = a.x + (size_t)b * n;   // (16-byte aligned: H * D * 4 is a multiple of 16 for even H)
        for (int i = threadIdx.x; i < (n >> 2); i += 64 * WPT) *(f32x4*)(snz_x + 4 * i) = *(const f32x4*)(xb + 4 * i);
        for (int i = (n & ~3) + threadIdx.x; i < n; i += 64 * WPT) snz_x[i] = xb[i];
        // the limits, one per lane (static index -> scalar loads), fetched per element through the crossbar below
        float mn = 0.f, mx = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { mn = lane == d ? gp.mins[d] : mn; mx = lane == d ? gp.maxs[d] : mx; }
        const bool clipall = __uint_as_float(a.amax_in[ctx]) > 1.0001f;
        __syncthreads();
        for (int i = threadIdx.x; i < ((n + 64 * WPT - 1) / (64 * WPT)) * (64 * WPT); i += 64 * WPT) {   // (whole waves take part in the shuffles)
            const int ic = i < n ? i : 0, d = ic % D;
            const float lo = __shfl(mn, d, 64), hi = __shfl(mx, d, 64);
            const float xnd = snz_x[ic];
            const float c = clipall ? fminf(fmaxf(xnd, -1.f), 1.f) : xnd;
            const float u01 = __fadd_rn(c, 1.0f) * 0.5f;
            const float xud = gp.identity_normalizer == 1 ? xnd : gp.identity_normalizer == 2 ? __fadd_rn(__fmul_rn(xnd, hi), lo)   // (Gaussian: lo = mean, hi = std)
                                                            : __fadd_rn(__fmul_rn(u01, __fsub_rn(hi, lo)), lo);
            if (i < n) sx[i] = xud;
        }
    }
    __syncthreads();
    G_STAMP();  // 1 state unnormalised + staged

    // ---- phase 1: interpolate + FK, once per point
    const float scale = (N > 1) ? (float)(H - 1) / (float)(N - 1) : 0.f;  // align_corners=True
    for (int i = wv * 64 + lane; i < N && !DENSE; i += 64 * WPT) {
        int i0 = i, i1 = i;
        float l0 = 1.f, l1 = 0.f;
        if (gp.interpolate) {
            const float u = scale * (float)i;
            i0 = (int)u;
            if (i0 > H - 1) i0 = H - 1;
            i1 = i0 + 1 < H ? i0 + 1 : H - 1;
            l1 = u - (float)i0;
            l0 = 1.0f - l1;
        }
        float q[QD];
#pragma unroll
        for (int j = 0; j < QD; ++j) q[j] = l0 * sx[i0 * D + j] + l1 * sx[i1 * D + j];
        float O[7][3], Z[7][3];
        panda_fk(q, O, Z);
        float* fk = sfk + i * kPandaFKS;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { fk[k * 3 + r] = O[k][r]; fk[21 + k * 3 + r] = Z[k][r]; }
        }
#pragma unroll
        for (int s = 0; s < kPandaNS; ++s) {
#pragma unroll
            for (int r = 0; r < 3; ++r) fk[42 + s * 3 + r] = O[kPandaSF[s] - 1][r] + kPandaSO[s] * Z[kPandaSF[s] - 1][r];
        }
    }
    if (!DENSE) __syncthreads();
    G_STAMP();  // 2 FK in LDS

    // ---- phase 2: forces of this wave's sphere / pair group, every field, folded to joint gradients
    {
        const int half = wv & 1;
        long long* trf = (MPDX_TRACE_PTR(a.trace) && b == 0 && lane == 0) ? a.trace + wv * 16 + 8 : nullptr;  // slots 8..: per-field stamps
        switch (wv >> 1) {  // the group is a template parameter: sphere -> frame is static, no per-joint masks
            case 0: panda_group_forces<0, DENSE>(gp, sprim, sfk, sG, half, lane, N, trf, sx, H, scale); break;
            case 1: panda_group_forces<1, DENSE>(gp, sprim, sfk, sG, half, lane, N, trf, sx, H, scale); break;
            case 2: panda_group_forces<2, DENSE>(gp, sprim, sfk, sG, half, lane, N, trf, sx, H, scale); break;
            default: panda_group_forces<3, DENSE>(gp, sprim, sfk, sG, half, lane, N, trf, sx, H, scale); break;
        }
    }
    G_STAMP();  // 3 this wave's forces done
    __syncthreads();
    G_STAMP();  // 4 all waves done

    // the step's noise (last guide iteration, drawn in place): by the waves that do not gather, under the gather
    const unsigned long long ne0 = a.rng.elem0 + (unsigned long long)b * H * D;
    if (a.rng.on && !a.grad_out && wv >= MAXF) guide_draw_noise(a.rng, ne0, H * D, snz, threadIdx.x - 64 * MAXF, 64 * (WPT - MAXF));

    // ---- phase 3: wave f gathers field f to the support points (one or two blocks of 64), clips, weights
    if (wv < gp.n_fields) {
        const int f = wv;
        // the parts of sG[(f, .)] that hold data are known per field kind (panda_part_joints): two straight-line variants of the gather,
        // chosen once per wave (a per-element test inside the loop doubled this phase: 6.3 k -> 12 k cycles, tools/guide_trace.py)
        auto gather = [&](auto self_c) {
            constexpr bool SELFK = decltype(self_c)::value;
            for (int sb = 0; sb < nsw; ++sb) {
                const int hg_ = sb * 64 + lane;
                if (hg_ >= H) continue;
                int ilo = hg_, ihi = hg_;
                if (gp.interpolate && scale > 0.f) {
                    ilo = (int)((float)(hg_ - 1) / scale) - 1;   // points of the segments (hg-1, hg) and (hg, hg+1): any number per segment
                    ihi = (int)((float)(hg_ + 1) / scale) + 1;
                    if (ilo < 0) ilo = 0;
                    if (ihi > N - 1) ihi = N - 1;
                }
                float g[QD];
#pragma unroll
                for (int j = 0; j < QD; ++j) g[j] = 0.f;
                for (int i = ilo; i <= ihi; ++i) {
                    int i0 = i, i1 = i;
                    float l0 = 1.f, l1 = 0.f;
                    if (gp.interpolate) {
                        const float u = scale * (float)i;
                        i0 = (int)u;
                        if (i0 > H - 1) i0 = H - 1;
                        i1 = i0 + 1 < H ? i0 + 1 : H - 1;
                        l1 = u - (float)i0;
                        l0 = 1.0f - l1;
                    }
                    const bool m0 = i0 == hg_, m1 = i1 == hg_ && gp.interpolate;
                    if (m0 || m1) {
#pragma unroll
                        for (int j = 0; j < QD; ++j) {
                            float v = 0.f;
#pragma unroll
                            for (int pt = 0; pt < NP; ++pt)
                                if (SELFK ? pt == NP - 1 : j < panda_group_joints(pt)) v += sG[((f * NP + pt) * N + i) * QD + j];   // (compile time)
                            if (m0) g[j] += l0 * v;
                            if (m1) g[j] += l1 * v;
                        }
                    }
                }
                // clip over ALL D dims of (g + 1e-6): the velocity dims of a collision gradient are 0
                clip_waypoint_grad<QD>(gp, g, QD);
                const bool interior = hg_ > 0 && hg_ < H - 1;
#pragma unroll
                for (int j = 0; j < QD; ++j) sC[(f * H + hg_) * QD + j] = interior ? gp.fields[f].weight * g[j] : 0.f;
            }
        };
        if (gp.fields[f].kind == MPDX_FIELD_SELF) gather(std::true_type{}); else gather(std::false_type{});
    }
    __syncthreads();
    G_STAMP();  // 5 gathered + clipped
    if (wv >= nsw) return;

    // ---- phase 4 (the support wave(s)): sum over fields, GP prior, apply
    float total[D], xu[D], xn[D];
    const size_t base = ((size_t)b * H + (live ? hs_ : 0)) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        total[d] = 0.f;
        xu[d] = live ? sx[hs_ * D + d] : 0.f;        // = the unnormalised state the prologue staged
        xn[d] = live ? snz_x[hs_ * D + d] : 0.f;     // = the normalised state as loaded (nothing has written x since)
    }
    if (live) {
        for (int f = 0; f < gp.n_fields; ++f) {
#pragma unroll
            for (int j = 0; j < QD; ++j) total[j] += sC[(f * H + hs_) * QD + j];
        }
    }
    guide_gp_apply<QD>(a, b, ctx, lane, hs_, H, live, xn, xu, sx, total, base, (MPDX_TRACE_PTR(a.trace) && b == 0 && lane == 0) ? a.trace + wv * 16 + 6 : nullptr,
                       snz + (int)(ne0 & 3ull), shc);
#undef G_STAMP
}


inline size_t guide_lds_bytes(const mpdx_guide_params& gp, int H, int D, bool dense = false) {
    const int N = gp.interpolate ? gp.n_interp : H;
    if (gp.robot == MPDX_ROBOT_PANDA)
        return (size_t)(H * D + (dense ? 0 : N * kPandaFKS) + MPDX_MAX_FIELDS * kPandaParts * N * 7 + MPDX_MAX_FIELDS * H * 7 + (2 * H * D + 4 + 3) + gp.n_prim_floats + 3 + 2 * D) * sizeof(float);
    return (size_t)(H * D + 2 * MPDX_MAX_FIELDS * N * (D / 2) + gp.n_prim_floats) * sizeof(float);
}

}  // namespace mpdx
