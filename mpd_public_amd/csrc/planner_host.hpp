// planner_host.hpp - C-ABI entry points of the baseline planners (csrc/planner.hpp); included at the end of mpdx.hip.
#pragma once
#include "planner.hpp"

namespace mpdx {
// the argument checks launch_guide applies to a guide parameter block, for the planners that copy it into their kernel arguments
// (gn_point / config_hit index gp.fields; the GPMP2 kernel's support-window arithmetic assumes n_interp >= H).  H == 0: no horizon (RRT).
static int check_planner_params(const mpdx_guide_params* gp, int H, int D) {
    if (D != 2 * gp->q_dim || D > 16) return fail(MPDX_E_INVALID, "state dim %d != 2*q_dim (%d)", D, gp->q_dim);
    if (gp->n_fields < 0 || gp->n_fields > MPDX_MAX_FIELDS) return fail(MPDX_E_INVALID, "n_fields %d", gp->n_fields);
    if (gp->n_prim_floats > 0 && !gp->prims) return fail(MPDX_E_INVALID, "primitive table missing");
    if (H > 0 && gp->interpolate && (gp->n_interp < 2 || gp->n_interp < H || gp->n_interp > 8 * H))
        return fail(MPDX_E_INVALID, "n_interp %d unsupported for H=%d", gp->n_interp, H);
    return 0;
}
}  // namespace mpdx

extern "C" {

int mpdx_gpmp_step(const mpdx_guide_params* gp, const mpdx_gpmp_opts* o, float* x, float* delta, float* state, int B, int H, int D, int solve,
                   void* stream) {
    using namespace mpdx;
    if (!gp || !o || !x || !delta || !state || B <= 0) return fail(MPDX_E_INVALID, "null argument");
    if (H < 4 || D != 2 * gp->q_dim) return fail(MPDX_E_INVALID, "GPMP2 step: H=%d D=%d q_dim=%d", H, D, gp->q_dim);
    if (!gp->use_gp || !(gp->dt > 0.f) || !(o->sigma_obs > 0.f))
        return fail(MPDX_E_INVALID, "GPMP2 step needs the GP prior (dt, sigma_gp) and sigma_obs > 0");
    if (int rc = check_planner_params(gp, H, D)) return rc;
    GpmpArgs a;
    memset(&a, 0, sizeof(a));
    a.gp = *gp; a.x = x; a.delta = delta; a.state = state; a.B = B; a.H = H;
    a.sigma_obs = o->sigma_obs; a.lam_up = o->lambda_up; a.lam_down = o->lambda_down; a.lam_min = o->lambda_min; a.lam_max = o->lambda_max;
    a.step = o->step; a.adaptive = o->adaptive; a.solve = solve;
    const int N = gp->interpolate ? gp->n_interp : H;
    hipStream_t st = (hipStream_t)stream;
#define MPDX_GPMP(QD_, DIM_, ROBOT_)                                                                                     \
    {                                                                                                                    \
        const size_t lds = gpmp_lds_bytes<QD_>(H, N, gp->n_prim_floats);                                                 \
        if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "GPMP2 step needs %zu B of LDS (H=%d, %d points)", lds, H, N);  \
        auto kern = gpmp_lm_kernel<QD_, DIM_, ROBOT_>;                                                                   \
        if (lds > 64 * 1024)                                                                                             \
            if (int rc = raise_lds_limit((const void*)kern)) return rc;                                                  \
        hipLaunchKernelGGL(kern, dim3(B), dim3(kGpmpThreads), lds, st, a);                                               \
    }
    if (gp->robot == MPDX_ROBOT_PANDA && gp->q_dim == 7) MPDX_GPMP(7, 3, MPDX_ROBOT_PANDA)
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 2) MPDX_GPMP(2, 2, MPDX_ROBOT_POINTMASS)
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 3) MPDX_GPMP(3, 3, MPDX_ROBOT_POINTMASS)
    else return fail(MPDX_E_INVALID, "GPMP2 step: unsupported robot %d / q_dim %d", gp->robot, gp->q_dim);
#undef MPDX_GPMP
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_rrt_connect(const mpdx_guide_params* gp, const mpdx_rrt_opts* o, const float* start, const float* goal, float* nodes, int32_t* parent,
                     int32_t* count, int32_t* link, int32_t* iters, int n, void* stream) {
    using namespace mpdx;
    if (!gp || !o || !start || !goal || !nodes || !parent || !count || !link || !iters || n <= 0) return fail(MPDX_E_INVALID, "null argument");
    if (o->max_nodes < 2 || o->n_edge_checks < 2 || o->n_edge_checks > kRrtThreads || !(o->step > 0.f))
        return fail(MPDX_E_INVALID, "RRT-Connect: max_nodes %d, n_edge_checks %d, step %g", o->max_nodes, o->n_edge_checks, (double)o->step);
    if (o->max_iters < 0 || o->max_connect_steps < 0)
        return fail(MPDX_E_INVALID, "RRT-Connect: max_iters %d, max_connect_steps %d", o->max_iters, o->max_connect_steps);
    if (int rc = check_planner_params(gp, 0, 2 * gp->q_dim)) return rc;
    RrtArgs a;
    memset(&a, 0, sizeof(a));
    a.gp = *gp; a.start = start; a.goal = goal; a.nodes = nodes; a.parent = parent; a.count = count; a.link = link; a.iters = iters;
    for (int j = 0; j < 8; ++j) { a.q_lo[j] = o->q_lo[j]; a.q_hi[j] = o->q_hi[j]; }
    a.step = o->step; a.max_nodes = o->max_nodes; a.max_iters = o->max_iters; a.max_connect = o->max_connect_steps; a.n_checks = o->n_edge_checks;
    a.seed = o->seed;
    hipStream_t st = (hipStream_t)stream;
#define MPDX_RRT(QD_, DIM_, ROBOT_)                                                                                      \
    {                                                                                                                    \
        const size_t lds = rrt_lds_bytes<QD_>(o->max_nodes, gp->n_prim_floats);                                          \
        if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "RRT-Connect: %d nodes x %d dims need %zu B of LDS", o->max_nodes, QD_, lds); \
        auto kern = rrt_connect_kernel<QD_, DIM_, ROBOT_>;                                                               \
        if (lds > 64 * 1024)                                                                                             \
            if (int rc = raise_lds_limit((const void*)kern)) return rc;                                                  \
        hipLaunchKernelGGL(kern, dim3(n), dim3(kRrtThreads), lds, st, a);                                                \
    }
    if (gp->robot == MPDX_ROBOT_PANDA && gp->q_dim == 7) MPDX_RRT(7, 3, MPDX_ROBOT_PANDA)
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 2) MPDX_RRT(2, 2, MPDX_ROBOT_POINTMASS)
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 3) MPDX_RRT(3, 3, MPDX_ROBOT_POINTMASS)
    else return fail(MPDX_E_INVALID, "RRT-Connect: unsupported robot %d / q_dim %d", gp->robot, gp->q_dim);
#undef MPDX_RRT
    HIP_TRY(hipGetLastError());
    return 0;
}

int mpdx_rrt_paths(const mpdx_guide_params* gp, const float* start, const float* goal, const float* nodes, const int32_t* parent, const int32_t* link,
                   float* trajs_out, int32_t* path_len, int n, int max_nodes, int H, float dt, int n_edge_checks, int rounds, void* stream) {
    using namespace mpdx;
    if (!gp || !start || !goal || !nodes || !parent || !link || !trajs_out || n <= 0) return fail(MPDX_E_INVALID, "null argument");
    if (max_nodes < 2 || H < 2 || H > 1024 || !(dt > 0.f) || n_edge_checks < 2 || n_edge_checks > kRrtThreads || rounds < 0)
        return fail(MPDX_E_INVALID, "RRT paths: max_nodes %d, H %d, dt %g, n_edge_checks %d, rounds %d", max_nodes, H, (double)dt, n_edge_checks, rounds);
    if (int rc = check_planner_params(gp, 0, 2 * gp->q_dim)) return rc;
    RrtPathArgs a;
    memset(&a, 0, sizeof(a));
    a.gp = *gp; a.start = start; a.goal = goal; a.nodes = nodes; a.parent = parent; a.link = link; a.out = trajs_out; a.path_len = path_len;
    a.max_nodes = max_nodes; a.H = H; a.n_checks = n_edge_checks; a.rounds = rounds; a.dt = dt;
    hipStream_t st = (hipStream_t)stream;
#define MPDX_RRTP(QD_, DIM_, ROBOT_)                                                                                     \
    {                                                                                                                    \
        const size_t lds = rrt_path_lds_bytes<QD_>(max_nodes, H, gp->n_prim_floats);                                     \
        if (lds > 160 * 1024) return fail(MPDX_E_INVALID, "RRT paths: %d nodes x %d dims need %zu B of LDS", max_nodes, QD_, lds); \
        auto kern = rrt_path_kernel<QD_, DIM_, ROBOT_>;                                                                  \
        if (lds > 64 * 1024)                                                                                             \
            if (int rc = raise_lds_limit((const void*)kern)) return rc;                                                  \
        hipLaunchKernelGGL(kern, dim3(n), dim3(kRrtThreads), lds, st, a);                                                \
    }
    if (gp->robot == MPDX_ROBOT_PANDA && gp->q_dim == 7) MPDX_RRTP(7, 3, MPDX_ROBOT_PANDA)
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 2) MPDX_RRTP(2, 2, MPDX_ROBOT_POINTMASS)
    else if (gp->robot == MPDX_ROBOT_POINTMASS && gp->q_dim == 3) MPDX_RRTP(3, 3, MPDX_ROBOT_POINTMASS)
    else return fail(MPDX_E_INVALID, "RRT paths: unsupported robot %d / q_dim %d", gp->robot, gp->q_dim);
#undef MPDX_RRTP
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
