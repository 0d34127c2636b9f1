// planner.hpp - the reference's BASELINE planners (scripts/generate_data/generate_trajectories.py:68-120: `HybridPlanner` =
// RRT-Connect initialisation + GPMP2 optimisation, both from the un-vendored mp_baselines submodule - PARITY UNPINNED) as HIP kernels.
//
//   gpmp_lm_kernel      one Levenberg-Marquardt / Gauss-Newton iteration of GPMP2 (Mukadam et al., IJRR 2018, section 4) for a
//                       batch of trajectories: MAP estimate of
//                           F(theta) = 1/2 sum_i e_i^T Q^-1 e_i / sigma_gp^2  +  1/2 sum_{points, factors} c^2 / sigma_obs^2
//                       e_i = theta_{i+1} - Phi theta_i (constant-velocity GP prior, Q^-1 = [[12/dt^3,-6/dt^2],[-6/dt^2,4/dt]] (x) I),
//                       c = hinge obstacle / workspace / self-collision factors of every link sphere on the interpolated
//                       trajectory (the SAME factors the guide differentiates: oracle/costs.py), start and goal states fixed.
//                       The normal equations  (K^-1 + J^T J / sigma_obs^2 + lambda diag) delta = -grad  are BLOCK TRIDIAGONAL over the
//                       H - 2 free support states (an interpolated point touches two neighbouring supports): assembled in LDS as
//                       d x d blocks (d = state dim) and solved there by block cyclic reduction (gpmp_bcr_solve) - one workgroup
//                       per trajectory, no global traffic besides the trajectory itself.
//   rrt_connect_kernel  RRT-Connect (Kuffner & LaValle 2000), one workgroup per problem, the WHOLE bidirectional search in one
//                       launch: sampling (Philox), nearest neighbour over the tree (lanes over nodes + wave arg-min), steering,
//                       edge collision checks (lanes over interpolated configurations, the metrics kernel's FK / SDF functions)
//                       and the greedy connect loop - no host round trip per extension.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "guide.hpp"

namespace mpdx {

struct GpmpArgs {
    mpdx_guide_params gp;   // robot, fields, margins, dt, sigma_gp, interpolation (identity normaliser: raw robot units)
    float* x;               // [B][H][D] current trajectories (in/out)
    float* delta;           // [B][H][D] last proposed step (in: candidate = x + delta; out: the new proposal)
    float* state;           // [B][4]: F(x), lambda, accepted steps, F(last candidate)
    int B, H;
    float sigma_obs;
    float lam_up, lam_down, lam_min, lam_max;
    float step;             // fraction of the Gauss-Newton step taken (1; mp_baselines-style fixed damping uses < 1)
    int adaptive;           // 1: accept a candidate only if it lowers F and adapt lambda (Levenberg-Marquardt); 0: always accept, lambda fixed
    int solve;              // 0: only judge the pending candidate (the call after the last iteration)
};

// accumulate one factor (hinge value c, Jacobian row j) into the point's packed normal-equation terms:
// M (lower triangle of sum j j^T), v (sum c j), c2 (sum c^2)
template <int QD>
__device__ __forceinline__ void gn_accumulate(float c, const float (&j)[QD], float (&M)[QD * (QD + 1) / 2], float (&v)[QD], float& c2) {
    int e = 0;
#pragma unroll
    for (int r = 0; r < QD; ++r) {
#pragma unroll
        for (int cc = 0; cc <= r; ++cc) M[e++] += j[r] * j[cc];
        v[r] += c * j[r];
    }
    c2 += c * c;
}

// Linearise every collision factor of interpolated point i whose index parity / group matches `half` (2 threads per point).
// q: the point's configuration.  Writes MSZ = QD(QD+1)/2 + QD + 1 floats.
// PART of NPARTS (2 or 4): the point's factors are split over that many threads (of different waves) - Panda: ranges of link spheres and of
// self-collision pairs, point mass: fields round-robin
template <int QD, int DIM, int ROBOT, int PART, int NPARTS>
__device__ __forceinline__ void gn_point(const mpdx_guide_params& gp, const float* sprim, const float (&q)[QD], float* out) {
    static_assert(NPARTS == 2 || NPARTS == 4, "two or four parts");
    constexpr int NT = QD * (QD + 1) / 2;
    float M[NT], v[QD], c2 = 0.f;
#pragma unroll
    for (int e = 0; e < NT; ++e) M[e] = 0.f;
#pragma unroll
    for (int j = 0; j < QD; ++j) v[j] = 0.f;
    if constexpr (ROBOT == MPDX_ROBOT_POINTMASS) {
        float p[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) p[j] = q[j];
        const float margin = gp.link_margin + gp.cutoff_margin;
        for (int f = 0; f < gp.n_fields; ++f) {
            if ((f % NPARTS) != PART) continue;
            if (gp.fields[f].kind == MPDX_FIELD_OBJECTS) {
                float fo[DIM];
                const float c = objects_force<DIM>(sprim, gp.fields[f], p, margin, fo);
                if (c > 0.f) {
                    float jr[QD];
#pragma unroll
                    for (int j = 0; j < QD; ++j) jr[j] = j < DIM ? fo[j] : 0.f;
                    gn_accumulate<QD>(c, jr, M, v, c2);
                }
            } else if (gp.fields[f].kind == MPDX_FIELD_WORKSPACE) {
#pragma unroll
                for (int j = 0; j < DIM; ++j) {   // one factor per face: Jacobian -e_j / +e_j
                    const float clo = margin - (p[j] - gp.fields[f].ws_min[j]), chi = margin - (gp.fields[f].ws_max[j] - p[j]);
                    const int dj = j * (j + 1) / 2 + j;
                    if (clo > 0.f) { M[dj] += 1.f; v[j] -= clo; c2 += clo * clo; }
                    if (chi > 0.f) { M[dj] += 1.f; v[j] += chi; c2 += chi * chi; }
                }
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
        // The factors of ONE link sphere s share its position Jacobian J_s (3 x 7, column k = z_k x (P_s - O_k) for joints k <= frame(s)):
        // a factor with value c and gradient f (w.r.t. the sphere centre) has the Jacobian row f^T J_s, so the sphere's contribution is
        //     M += J_s^T (sum_i f_i f_i^T) J_s,      v += J_s^T (sum_i c_i f_i),      c2 += sum_i c_i^2
        // - the 3 x 3 / 3-vector sums over the sphere's factors (objects fields, six workspace faces) are a few FMAs per factor, and the
        // 7 x 7 congruence is done ONCE per sphere, branch-free.  Round 4 ran a 35-FMA rank-1 update of M per ACTIVE factor at ~48 inlined call
        // sites behind divergent branches: 5.7 k VALU instructions, 256 + 256 registers and 1.26 KB of scratch per lane for this kernel.
        auto sphere_jac = [&](int s, float (&J)[3][QD]) {
            const int fr = kPandaSF[s] - 1;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (k <= fr) {
                    const float rx = P[s][0] - O[k][0], ry = P[s][1] - O[k][1], rz = P[s][2] - O[k][2];
                    J[0][k] = Z[k][1] * rz - Z[k][2] * ry; J[1][k] = Z[k][2] * rx - Z[k][0] * rz; J[2][k] = Z[k][0] * ry - Z[k][1] * rx;
                } else { J[0][k] = J[1][k] = J[2][k] = 0.f; }
            }
        };
        // static sphere / pair ranges (P[s], kPandaSF[s] are compile-time): halves {0-5, 6-10} / {0-5, 6-11}, quarters {0-2, 3-5, 6-8, 9-10} / {0-2, 3-5, 6-8, 9-11}
        constexpr int s_beg = NPARTS == 2 ? (PART ? 6 : 0) : 3 * PART, s_end = NPARTS == 2 ? (PART ? kPandaNS : 6) : (PART == 3 ? kPandaNS : 3 * PART + 3);
        constexpr int p_beg = NPARTS == 2 ? (PART ? 6 : 0) : 3 * PART, p_end = NPARTS == 2 ? (PART ? kPandaNP : 6) : (PART == 3 ? kPandaNP : 3 * PART + 3);
#pragma unroll
        for (int s = s_beg; s < s_end; ++s) {
            const int fr = kPandaSF[s] - 1;
            const float p3[3] = {P[s][0], P[s][1], P[s][2]};
            const float margin = kPandaSR[s] + gp.cutoff_margin;
            float A[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};   // A: xx xy yy xz yz zz of sum f f^T; g = sum c f
            for (int f = 0; f < gp.n_fields; ++f) {
                const int kind = gp.fields[f].kind;
                if (kind == MPDX_FIELD_OBJECTS) {
                    float fo[3];
                    const float c = objects_force<3>(sprim, gp.fields[f], p3, margin, fo);   // (c == 0: fo == 0 - nothing is added)
                    A[0] += fo[0] * fo[0]; A[1] += fo[0] * fo[1]; A[2] += fo[1] * fo[1]; A[3] += fo[0] * fo[2]; A[4] += fo[1] * fo[2]; A[5] += fo[2] * fo[2];
                    g[0] += c * fo[0]; g[1] += c * fo[1]; g[2] += c * fo[2];
                    c2 += c * c;
                } else if (kind == MPDX_FIELD_WORKSPACE) {
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {   // one factor per face: c = margin -+ (P - bound), gradient -+ e_ax
                        const float clo = fmaxf(margin - (p3[ax] - gp.fields[f].ws_min[ax]), 0.f), chi = fmaxf(margin - (gp.fields[f].ws_max[ax] - p3[ax]), 0.f);
                        const int d = ax == 0 ? 0 : (ax == 1 ? 2 : 5);
                        A[d] += (clo > 0.f ? 1.f : 0.f) + (chi > 0.f ? 1.f : 0.f);
                        g[ax] += chi - clo;
                        c2 += clo * clo + chi * chi;
                    }
                }
            }
            float J[3][QD];
            sphere_jac(s, J);
            // T = A J (3 x 7), then M += J^T T (lower triangle), v += J^T g; columns beyond the sphere's frame are zero (compile time)
#pragma unroll
            for (int c = 0; c < QD; ++c) {
                if (c > fr) continue;
                const float t0 = A[0] * J[0][c] + A[1] * J[1][c] + A[3] * J[2][c];
                const float t1 = A[1] * J[0][c] + A[2] * J[1][c] + A[4] * J[2][c];
                const float t2 = A[3] * J[0][c] + A[4] * J[1][c] + A[5] * J[2][c];
#pragma unroll
                for (int r = c; r < QD; ++r) {
                    if (r > fr) continue;
                    M[r * (r + 1) / 2 + c] += J[0][r] * t0 + J[1][r] * t1 + J[2][r] * t2;
                }
                v[c] += J[0][c] * g[0] + J[1][c] * g[1] + J[2][c] * g[2];
            }
        }
        bool has_self = false;
        for (int f = 0; f < gp.n_fields; ++f) has_self |= gp.fields[f].kind == MPDX_FIELD_SELF;
        if (has_self) {
#pragma unroll
            for (int pr = p_beg; pr < p_end; ++pr) {
                const int sa_ = kPandaPA[pr], sb_ = kPandaPB[pr];
                const float dx = P[sa_][0] - P[sb_][0], dy = P[sa_][1] - P[sb_][1], dz = P[sa_][2] - P[sb_][2];
                const float d2 = dx * dx + dy * dy + dz * dz;
                const float dist = __builtin_amdgcn_sqrtf(d2);
                const float cr = kPandaSR[sa_] + kPandaSR[sb_] - dist;
                const bool act = cr > 0.f && dist > 0.f;
                const float c = act ? cr : 0.f;
                const float inv = act ? __builtin_amdgcn_rsqf(d2) : 0.f;
                const float u[3] = {dx * inv, dy * inv, dz * inv};   // d c / d P_a = -u, d c / d P_b = +u  (inactive: u = 0 adds nothing)
                float Ja[3][QD], Jb[3][QD], jr[QD];
                sphere_jac(sa_, Ja);
                sphere_jac(sb_, Jb);
#pragma unroll
                for (int k = 0; k < QD; ++k) jr[k] = u[0] * (Jb[0][k] - Ja[0][k]) + u[1] * (Jb[1][k] - Ja[1][k]) + u[2] * (Jb[2][k] - Ja[2][k]);
                gn_accumulate<QD>(c, jr, M, v, c2);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NT; ++e) out[e] = M[e];
#pragma unroll
    for (int j = 0; j < QD; ++j) out[NT + j] = v[j];
    out[NT + QD] = c2;
}

constexpr int kGpmpThreads = 512;   // (round 5: 256 -> 512: the assembly and the solve's substitution / Schur phases are task-parallel over the threads - 102.6 -> 80.6 us per Panda iteration)

template <int QD>
inline size_t gpmp_lds_bytes(int H, int N, int n_prim_floats) {
    constexpr int D = 2 * QD, MSZ = QD * (QD + 1) / 2 + QD + 1, LSZ = D * (D + 1) / 2;
    const size_t n = (size_t)(H - 2), terms = (size_t)2 * N * MSZ, pool = n * LSZ;   // the Cholesky pool overlays the per-point terms
    return (size_t)(2 * H * D + (terms > pool ? terms : pool) + 2 * n * D * D + n * D + 64 + (H + 2) + n_prim_floats) * sizeof(float);
}

// ---- the Levenberg-Marquardt system: symmetric positive definite, block tridiagonal (n = H - 2 free supports, d x d blocks, d = 2 q_dim)
//   Dm[i] = A[i][i] (row-major d x d, lower triangle used), Cm[i] = A[i+1][i] (row-major d x d), rhs[i] (d)
// solved by BLOCK CYCLIC REDUCTION.  Rounds 3 / 4 first factorised the band column by column: 2 n d - 1 = 1 735 DEPENDENT steps for the Panda,
// each at least one LDS round trip (tools/gpmp_phase_probe.py, s_memtime stamps: a workgroup-wide column update with one barrier per step
// 750 cycles per step; one wave without barriers 900 cycles per factorisation step + 185 per substitution step = 393 us of a 528-us
// iteration).  Cyclic reduction eliminates every other block of the current chain at once - ceil(log2 n) = 6 levels for n = 62 - and each
// level is three workgroup-wide phases of d x d dense algebra:
//   P1  L_e = chol(D_e)                                      one lane per eliminated block, the 105-entry triangle in registers
//   P2  Y_lo = L^-1 A[e][a], Y_hi = L^-1 A[e][b], w = L^-1 r_e   one lane per column (in place; Y_hi is kept transposed, as A[b][e] was)
//   P3  kept blocks: D_q -= Y^T Y (both eliminated neighbours), r_q -= Y^T w; new couplings A[q+2][q] = -Y_hi^T Y_lo (stored in the
//       eliminated block's slot Dm[e]: its diagonal block was consumed by P1)
// and the substitution walks the levels back: x_e = L^-T (w - Y_lo x_a - Y_hi x_b).  Positions p of level l (stride s = 2^l) are the
// blocks (p + 1) s - 1; even positions are eliminated, odd ones kept.  The Cholesky factors (packed, reciprocal diagonal) of all n blocks
// go to `Lp`, which overlays the per-point linearisation terms (dead once the system is assembled).
template <int D>
struct Bcr {
    static constexpr int DD = D * D, LSZ = D * (D + 1) / 2;
    static __device__ __forceinline__ int idx(int p, int s) { return (p + 1) * s - 1; }
    // storage of the coupling between positions p and p + 1 of the level with stride s (rows: p + 1, columns: p)
    static __device__ __forceinline__ float* coup(float* Dm, float* Cm, int s, int p) {
        return s == 1 ? Cm + (size_t)p * DD : Dm + (size_t)(((2 * p + 3) * s) / 2 - 1) * DD;
    }
};

#ifdef MPDX_GPMP_STAMPS   // dev probe (tools/gpmp_phase_probe.py): s_memtime after P1 / P2 / P3 of every level (slots 1 + 3 lev ..), 30 = forward done, 31 + lev = substitution level done
__device__ long long g_bcr_st[64];
#define BCR_STAMP(i) do { if (tid == 0 && blockIdx.x == 0) g_bcr_st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BCR_STAMP(i)
#endif
template <int QD>
__device__ __forceinline__ void gpmp_bcr_solve(float* Dm, float* Cm, float* rhs, float* Lp, const int n, const int tid) {
    constexpr int D = 2 * QD, DD = D * D, LSZ = D * (D + 1) / 2, NTHR = kGpmpThreads;
    using G = Bcr<D>;
    int nlev = 0;
    BCR_STAMP(0);
    for (int nl = n, s = 1; nl >= 1; nl >>= 1, s <<= 1, ++nlev) {
        const int ne = (nl + 1) >> 1, nk = nl >> 1;
        // ---- P1: Cholesky of the eliminated diagonal blocks (registers; l_kk is stored as its reciprocal)
        for (int t = tid; t < ne; t += NTHR) {
            const int e = G::idx(2 * t, s);
            const float* Dp = Dm + (size_t)e * DD;
            float a[LSZ];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) a[i * (i + 1) / 2 + j] = Dp[i * D + j];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const float rk = __builtin_amdgcn_rsqf(a[k * (k + 1) / 2 + k]);
                a[k * (k + 1) / 2 + k] = rk;
#pragma unroll
                for (int i = k + 1; i < D; ++i) a[i * (i + 1) / 2 + k] *= rk;
#pragma unroll
                for (int j = k + 1; j < D; ++j)
#pragma unroll
                    for (int i = j; i < D; ++i) a[i * (i + 1) / 2 + j] = __builtin_fmaf(-a[i * (i + 1) / 2 + k], a[j * (j + 1) / 2 + k], a[i * (i + 1) / 2 + j]);
            }
            float* L = Lp + (size_t)e * LSZ;
#pragma unroll
            for (int i = 0; i < LSZ; ++i) L[i] = a[i];
        }
        __syncthreads();
        BCR_STAMP(1 + 3 * nlev);
        // ---- P2: forward substitutions, one lane per right-hand-side column
        for (int task = tid; task < ne * (2 * D + 1); task += NTHR) {
            const int t = task / (2 * D + 1), c = task - t * (2 * D + 1), p = 2 * t, e = G::idx(p, s);
            float* v;
            int stride = 1;
            if (c < D) {
                if (p == 0) continue;
                v = G::coup(Dm, Cm, s, p - 1) + c; stride = D;          // column c of A[e][a]
            } else if (c < 2 * D) {
                if (p + 1 >= nl) continue;
                v = G::coup(Dm, Cm, s, p) + (c - D) * D;                // row c - D of A[b][e] = column of A[e][b]
            } else v = rhs + (size_t)e * D;
            const float* L = Lp + (size_t)e * LSZ;
            float y[D];
#pragma unroll
            for (int i = 0; i < D; ++i) y[i] = v[i * stride];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float acc = y[i];
#pragma unroll
                for (int j = 0; j < i; ++j) acc = __builtin_fmaf(-L[i * (i + 1) / 2 + j], y[j], acc);
                y[i] = acc * L[i * (i + 1) / 2 + i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) v[i * stride] = y[i];
        }
        __syncthreads();
        BCR_STAMP(2 + 3 * nlev);
        // ---- P3: Schur complements onto the kept blocks and the next level's couplings
        if constexpr (D > 8 && D < 16) {
            // d x d x d products on the matrix cores: one wave = one product, four v_mfma_f32_16x16x4_f32 (d padded to 16 by zero operands; exact fp32
            // FMAs).  A kept block's two products share one accumulator; column d of the B operand (free: d < 16) carries the eliminated block's
            // right-hand side, so that  r_q -= Y^T w  falls out of the same instructions as column d of the result.  (The scalar form below - two LDS
            // loads per FMA - was ~40 % of the solve for the Panda.)
            const int lane = tid & 63, wave = tid >> 6;
            const int c16 = lane & 15, kq = lane >> 4;   // operand row / column index, k sub-index; result: rows 4 kq + r, column c16
            const int nc = (nl - 1) >> 1;
            for (int task = wave; task < nk + nc; task += NTHR / 64) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (task < nk) {
                    const int q = 2 * task + 1, kb = G::idx(q, s);
                    const float* Y1 = G::coup(Dm, Cm, s, q - 1);      // Y_hi^T of the eliminated block below: [i][k]
                    const float* w1 = rhs + (size_t)G::idx(q - 1, s) * D;
                    const bool up = q + 1 < nl;
                    const float* Y2 = up ? G::coup(Dm, Cm, s, q) : Y1;   // Y_lo of the eliminated block above: [k][j]
                    const float* w2 = up ? rhs + (size_t)G::idx(q + 1, s) * D : w1;
                    const int cc = c16 < D ? c16 : 0;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = 4 * kk + kq, kc = k < D ? k : 0;
                        const bool kin = k < D;
                        const float y1 = Y1[cc * D + kc], y2 = Y2[kc * D + cc], v1 = w1[kc], v2 = w2[kc];
                        const float a1 = (kin && c16 < D) ? y1 : 0.f, b1 = kin ? (c16 < D ? y1 : (c16 == D ? v1 : 0.f)) : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
                        const float a2 = (kin && c16 < D && up) ? y2 : 0.f, b2 = (kin && up) ? (c16 < D ? y2 : (c16 == D ? v2 : 0.f)) : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * kq + r;
                        if (i < D && c16 <= i) Dm[(size_t)kb * DD + i * D + c16] -= acc[r];          // lower triangle of D_q
                        else if (i < D && c16 == D) rhs[(size_t)kb * D + i] -= acc[r];                // r_q
                    }
                } else {   // eliminated position p = 2 t (t >= 1) with a neighbour on both sides: A[p+1][p-1] = -Y_hi^T Y_lo -> Dm[e]
                    const int p = 2 * (task - nk + 1);
                    const float* Yh = G::coup(Dm, Cm, s, p);
                    const float* Yl = G::coup(Dm, Cm, s, p - 1);
                    const int cc = c16 < D ? c16 : 0;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = 4 * kk + kq, kc = k < D ? k : 0;
                        const bool ok = k < D && c16 < D;
                        const float av = Yh[cc * D + kc], bv = Yl[kc * D + cc];
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? av : 0.f, ok ? bv : 0.f, acc, 0, 0, 0);
                    }
                    float* dst = Dm + (size_t)G::idx(p, s) * DD;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * kq + r;
                        if (i < D && c16 < D) dst[i * D + c16] = -acc[r];
                    }
                }
            }
        } else {
        for (int task = tid; task < nk * (LSZ + D); task += NTHR) {
            const int u = task / (LSZ + D), r = task - u * (LSZ + D), q = 2 * u + 1, kb = G::idx(q, s);
            const float* Y1 = G::coup(Dm, Cm, s, q - 1);                 // Y_hi^T of the eliminated block below: [i][k]
            const bool up = q + 1 < nl;
            const float* Y2 = up ? G::coup(Dm, Cm, s, q) : Y1;          // Y_lo of the eliminated block above: [k][j]
            float acc = 0.f;
            if (r < LSZ) {
                int i = (int)((__builtin_amdgcn_sqrtf(8.0f * (float)r + 1.0f) - 1.0f) * 0.5f);
                if ((i + 1) * (i + 2) / 2 <= r) ++i;
                if (i * (i + 1) / 2 > r) --i;
                const int j = r - i * (i + 1) / 2;
#pragma unroll
                for (int k = 0; k < D; ++k) acc = __builtin_fmaf(Y1[i * D + k], Y1[j * D + k], acc);
                if (up) {
#pragma unroll
                    for (int k = 0; k < D; ++k) acc = __builtin_fmaf(Y2[k * D + i], Y2[k * D + j], acc);
                }
                Dm[(size_t)kb * DD + i * D + j] -= acc;
            } else {
                const int i = r - LSZ;
                const float* w1 = rhs + (size_t)G::idx(q - 1, s) * D;
#pragma unroll
                for (int k = 0; k < D; ++k) acc = __builtin_fmaf(Y1[i * D + k], w1[k], acc);
                if (up) {
                    const float* w2 = rhs + (size_t)G::idx(q + 1, s) * D;
#pragma unroll
                    for (int k = 0; k < D; ++k) acc = __builtin_fmaf(Y2[k * D + i], w2[k], acc);
                }
                rhs[(size_t)kb * D + i] -= acc;
            }
        }
        // eliminated positions p = 2 t, t >= 1, with a neighbour on both sides: A[p+1][p-1] = -Y_hi^T Y_lo -> Dm[e]
        const int nc = (nl - 1) >> 1;   // p = 2, 4, ... with p + 1 < nl
        for (int task = tid; task < nc * DD; task += NTHR) {
            const int t = task / DD + 1, r = task - (t - 1) * DD, i = r / D, j = r - i * D, p = 2 * t;
            const float* Yh = G::coup(Dm, Cm, s, p);
            const float* Yl = G::coup(Dm, Cm, s, p - 1);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = __builtin_fmaf(Yh[i * D + k], Yl[k * D + j], acc);
            Dm[(size_t)G::idx(p, s) * DD + r] = -acc;
        }
        }
        __syncthreads();
        BCR_STAMP(3 + 3 * nlev);
    }
    BCR_STAMP(30);
    // ---- substitution, deepest level first
    for (int lev = nlev - 1; lev >= 0; --lev) {
        const int s = 1 << lev, nl = n >> lev, ne = (nl + 1) >> 1;
        for (int task = tid; task < ne * D; task += NTHR) {
            const int t = task / D, k = task - t * D, p = 2 * t, e = G::idx(p, s);
            float acc = rhs[(size_t)e * D + k];
            if (p > 0) {
                const float* Yl = G::coup(Dm, Cm, s, p - 1) + k * D;
                const float* xa = rhs + (size_t)G::idx(p - 1, s) * D;
#pragma unroll
                for (int j = 0; j < D; ++j) acc = __builtin_fmaf(-Yl[j], xa[j], acc);
            }
            if (p + 1 < nl) {
                const float* Yh = G::coup(Dm, Cm, s, p) + k;
                const float* xb = rhs + (size_t)G::idx(p + 1, s) * D;
#pragma unroll
                for (int i = 0; i < D; ++i) acc = __builtin_fmaf(-Yh[i * D], xb[i], acc);
            }
            rhs[(size_t)e * D + k] = acc;   // (every lane reads only its own component of w_e: in place)
        }
        __syncthreads();
        for (int t = tid; t < ne; t += NTHR) {
            const int e = G::idx(2 * t, s);
            const float* L = Lp + (size_t)e * LSZ;
            float* xe = rhs + (size_t)e * D;
            float x[D];
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = xe[i];
#pragma unroll
            for (int k = D - 1; k >= 0; --k) {
                float acc = x[k];
#pragma unroll
                for (int i = k + 1; i < D; ++i) acc = __builtin_fmaf(-L[i * (i + 1) / 2 + k], x[i], acc);
                x[k] = acc * L[k * (k + 1) / 2 + k];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) xe[i] = x[i];
        }
        __syncthreads();
        BCR_STAMP(31 + lev);
    }
}

template <int QD, int DIM, int ROBOT>
__global__ __launch_bounds__(kGpmpThreads, 1) void gpmp_lm_kernel(const GpmpArgs a) {
    constexpr int D = 2 * QD, DD = D * D, LSZ = D * (D + 1) / 2, NT = QD * (QD + 1) / 2, MSZ = NT + QD + 1, NTHR = kGpmpThreads;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const mpdx_guide_params& gp = a.gp;
    const int tid = threadIdx.x, b = blockIdx.x, H = a.H;
    const int N = gp.interpolate ? gp.n_interp : H;
    const int n = H - 2, NR = n * D;                 // free support states, unknowns
    float* sx = sm;                                  // [H][D] linearisation point
    float* sc = sx + H * D;                          // [H][D] candidate
    float* sM = sc + H * D;                          // [2][N][MSZ] per-point normal-equation terms
    const size_t n_terms = (size_t)2 * N * MSZ, n_pool = (size_t)n * LSZ;
    float* Dm = sM + (n_terms > n_pool ? n_terms : n_pool);   // [n][D][D] diagonal blocks (lower triangles); the Cholesky pool of the solve overlays sM
    float* Cm = Dm + (size_t)n * DD;                 // [n][D][D] Cm[i] = A[i+1][i]
    float* rhs = Cm + (size_t)n * DD;                // [NR]
    float* red = rhs + NR;                           // [64] reduction scratch
    int* ps = (int*)(red + 64);                      // [H + 1] first interpolation point of every support segment
    float* sprim = (float*)(ps + H + 2);
    for (int i = tid; i < gp.n_prim_floats; i += NTHR) sprim[i] = gp.prims[i];
    const size_t base = (size_t)b * H * D;
    for (int i = tid; i < H * D; i += NTHR) {
        const float xv = a.x[base + i];
        const int h = i / D;
        sx[i] = xv;
        sc[i] = (h > 0 && h < H - 1) ? xv + a.delta[base + i] : xv;
    }
    float F_cur = a.state[(size_t)b * 4 + 0], lam = a.state[(size_t)b * 4 + 1], n_acc = a.state[(size_t)b * 4 + 2];
#ifdef MPDX_GPMP_STAMPS   // dev probe (tools/gpmp_phase_probe.py): s_memtime at the phase boundaries, written over the (unused) delta rows 0 / H-1 of trajectory 0
    long long stamp[6];
    stamp[0] = __builtin_amdgcn_s_memtime();
#define GPMP_STAMP(i) stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define GPMP_STAMP(i)
#endif
    if (lam < 0.f) return;   // converged earlier (marked by a negative lambda): x is final, delta is zero - nothing left to do
    __syncthreads();

    const float scale = (N > 1) ? (float)(H - 1) / (float)(N - 1) : 0.f;
    const float s_gp = 1.0f / (gp.sigma_gp * gp.sigma_gp), s_ob = 1.0f / (a.sigma_obs * a.sigma_obs);
    const float dt = gp.dt;
    const float qa = 12.0f / (dt * dt * dt), qb = -6.0f / (dt * dt), qc = 4.0f / dt;   // Q^-1 = [[qa, qb], [qb, qc]] (x) I

    // linearise the collision factors at `src` and return F(src) (uniform over the workgroup)
    auto linearise = [&](const float* src) -> float {
        auto point_q = [&](int i, float (&q)[QD]) {
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
#pragma unroll
            for (int j = 0; j < QD; ++j) q[j] = l0 * src[i0 * D + j] + l1 * src[i1 * D + j];
        };
        if (4 * N <= NTHR && (size_t)2 * N * MSZ <= (size_t)2 * n * DD) {   // (one task per thread: 4 N of them)
            // FOUR threads per point (round 5; N is a multiple of 64 in practice: the part is wave-uniform): parts 0 / 1 write the two [N][MSZ] term
            // arrays, parts 2 / 3 write theirs into the system's storage (Dm | Cm: not assembled yet) and add them behind a barrier - the term arrays
            // stay two (four would not fit 160 KB of LDS next to the Panda's block-tridiagonal system).  Two threads per point left half of the 512
            // threads idle for the iteration's largest phase.
            const int part = tid / N, i = tid - part * N;   // (threads behind 4 N: no task)
            float* const mine = part < 2 ? sM + ((size_t)part * N + i) * MSZ : Dm + ((size_t)(part - 2) * N + i) * MSZ;
            if (part < 4) {
                float q[QD];
                point_q(i, q);
                if (part == 0) gn_point<QD, DIM, ROBOT, 0, 4>(gp, sprim, q, mine);
                else if (part == 1) gn_point<QD, DIM, ROBOT, 1, 4>(gp, sprim, q, mine);
                else if (part == 2) gn_point<QD, DIM, ROBOT, 2, 4>(gp, sprim, q, mine);
                else gn_point<QD, DIM, ROBOT, 3, 4>(gp, sprim, q, mine);
            }
            __syncthreads();
            if (part == 2 || part == 3) {
                float* dst = sM + ((size_t)(part - 2) * N + i) * MSZ;
                for (int e = 0; e < MSZ; ++e) dst[e] += mine[e];
            }
        } else
        for (int idx = tid; idx < 2 * N; idx += NTHR) {
            const int half = idx >= N ? 1 : 0, i = idx - half * N;   // N is a multiple of 64 in practice: the half is wave-uniform
            float q[QD];
            point_q(i, q);
            if (half) gn_point<QD, DIM, ROBOT, 1, 2>(gp, sprim, q, sM + (size_t)idx * MSZ);
            else gn_point<QD, DIM, ROBOT, 0, 2>(gp, sprim, q, sM + (size_t)idx * MSZ);
        }
        // cost: GP prior (one thread per factor) + collision (c2 of every point half)
        float part = 0.f;
        for (int i = tid; i < H - 1; i += NTHR) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < QD; ++j) {
                const float eq = src[(i + 1) * D + j] - src[i * D + j] - dt * src[i * D + QD + j];
                const float ev = src[(i + 1) * D + QD + j] - src[i * D + QD + j];
                s += qa * eq * eq + 2.0f * qb * eq * ev + qc * ev * ev;
            }
            part += 0.5f * s_gp * s;
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * N; idx += NTHR) part += 0.5f * s_ob * sM[(size_t)idx * MSZ + NT + QD];   // every [half][point] slot once
        part = wave_sum(part);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        float F = 0.f;
#pragma unroll
        for (int w = 0; w < NTHR / 64; ++w) F += red[w];
        __syncthreads();
        return F;
    };

    // ---- judge the pending candidate (ONE call site of the linearisation: the loop body runs once when the candidate is accepted,
    //      twice when it is rejected - the factors must then be linearised again at the point we keep)
    float F_cand = 0.f;
    const bool first = !(F_cur < 3.0e37f);
    bool converged = false;   // adaptive mode: an accepted step that no longer lowers F (relative 1e-7), or lambda at its ceiling
    const float* src = sc;
    for (int pass = 0; pass < 2; ++pass) {
        const float F_eval = linearise(src);
        if (pass == 1) { F_cur = F_eval; break; }
        F_cand = F_eval;
        bool accept = first || !a.adaptive || (F_cand < F_cur);
        if (!(F_cand == F_cand)) accept = first;   // NaN candidate: never accept (keep the current point)
        if (accept) {
            for (int i = tid; i < H * D; i += NTHR) sx[i] = sc[i];
            if (!first) {
                n_acc += 1.f;
                if (a.adaptive) { lam = fmaxf(lam * a.lam_down, a.lam_min); converged = (F_cur - F_cand) <= 1e-7f * F_cur; }
            }
            F_cur = F_cand;
            __syncthreads();
            break;
        }
        if (lam >= a.lam_max) { converged = true; break; }
        lam = fminf(lam * a.lam_up, a.lam_max);
        if (!a.solve) break;
        src = sx;
    }
    if (converged) {
        for (int i = tid; i < H * D; i += NTHR) { a.x[base + i] = sx[i]; a.delta[base + i] = 0.f; }
        if (tid == 0) { a.state[(size_t)b * 4 + 0] = F_cur; a.state[(size_t)b * 4 + 1] = -lam; a.state[(size_t)b * 4 + 2] = n_acc; a.state[(size_t)b * 4 + 3] = F_cand; }
        return;
    }
    if (!a.solve) {
        for (int i = tid; i < H * D; i += NTHR) { a.x[base + i] = sx[i]; a.delta[base + i] = 0.f; }
        if (tid == 0) { a.state[(size_t)b * 4 + 0] = F_cur; a.state[(size_t)b * 4 + 1] = lam; a.state[(size_t)b * 4 + 2] = n_acc; a.state[(size_t)b * 4 + 3] = F_cand; }
        return;
    }

    GPMP_STAMP(1);
    // ---- assemble the block-tridiagonal normal equations (every structural entry is written once; the rest is zero)
    for (int i = tid; i < 2 * n * DD; i += NTHR) Dm[i] = 0.f;
    // ps[h] = first point whose lower support is >= h (the points of segment [h, h + 1) are ps[h] .. ps[h + 1] - 1): the same float
    // expression as point_w, so the table and the weights agree to the bit
    for (int h = tid; h <= H; h += NTHR) {
        int i = h;
        if (gp.interpolate && scale > 0.f) {
            i = (int)((float)h / scale);
            if (i > N) i = N;
            auto lo = [&](int k) { int v = (int)(scale * (float)k); return v > H - 1 ? H - 1 : v; };
            while (i > 0 && lo(i - 1) >= h) --i;
            while (i < N && lo(i) < h) ++i;
        }
        ps[h] = h >= H ? N : i;
    }
    __syncthreads();
    constexpr int T0 = NT, T1 = T0 + QD, T2 = T1 + QD, T3 = T2 + QD, T4 = T3 + QD * QD, T5 = T4 + 3 * QD;
    for (int task = tid; task < n * T5; task += NTHR) {
        const int e = task / n, bi = task - e * n;   // entry class e (slow index: a wave's lanes share the branch below), free block bi (support h = bi + 1)
        const int h = bi + 1;
        // points that touch support h: those whose lower support is h - 1 or h
        const int plo = ps[h - 1], phi = ps[h + 1] - 1;
        auto point_w = [&](int i, int& i0, int& i1, float& l0, float& l1) {
            i0 = i; i1 = i; l0 = 1.f; l1 = 0.f;
            if (gp.interpolate) {
                const float u = scale * (float)i;
                i0 = (int)u;
                if (i0 > H - 1) i0 = H - 1;
                i1 = i0 + 1 < H ? i0 + 1 : H - 1;
                l1 = u - (float)i0;
                l0 = 1.0f - l1;
            }
        };
        auto msum = [&](int i, int off) { return sM[(size_t)i * MSZ + off] + sM[((size_t)N + i) * MSZ + off]; };
        if (e < T0) {            // pos-pos lower triangle of the diagonal block
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= e) ++r;
            const int c = e - r * (r + 1) / 2;
            float acc = 0.f;
            for (int i = plo; i <= phi; ++i) {
                int i0, i1; float l0, l1;
                point_w(i, i0, i1, l0, l1);
                float w = 0.f;
                if (i0 == h) w += l0 * l0;
                if (i1 == h && i1 != i0) w += l1 * l1;
                if (w != 0.f) acc += w * msum(i, e);
            }
            float val = s_ob * acc + (r == c ? s_gp * 2.0f * qa : 0.f);
            if (r == c) val *= (1.0f + lam);
            Dm[(size_t)bi * DD + r * D + c] = val;
        } else if (e < T1) {     // vel-vel diagonal (prior only: Q^-1 + Phi^T Q^-1 Phi = diag(24/dt^3, 8/dt))
            const int j = e - T0;
            Dm[(size_t)bi * DD + (QD + j) * (D + 1)] = s_gp * 2.0f * qc * (1.0f + lam);
        } else if (e < T3) {     // right-hand side = -gradient
            const bool vel = e >= T2;
            const int j = vel ? e - T2 : e - T1;
            // prior: g_h = Q^-1 e_{h-1} - Phi^T Q^-1 e_h
            const float eq0 = sx[h * D + j] - sx[(h - 1) * D + j] - dt * sx[(h - 1) * D + QD + j], ev0 = sx[h * D + QD + j] - sx[(h - 1) * D + QD + j];
            const float eq1 = sx[(h + 1) * D + j] - sx[h * D + j] - dt * sx[h * D + QD + j], ev1 = sx[(h + 1) * D + QD + j] - sx[h * D + QD + j];
            const float a0 = qa * eq0 + qb * ev0, b0 = qb * eq0 + qc * ev0, a1 = qa * eq1 + qb * ev1, b1 = qb * eq1 + qc * ev1;
            float g = s_gp * (vel ? (b0 - b1 - dt * a1) : (a0 - a1));
            if (!vel) {
                float acc = 0.f;
                for (int i = plo; i <= phi; ++i) {
                    int i0, i1; float l0, l1;
                    point_w(i, i0, i1, l0, l1);
                    float w = 0.f;
                    if (i0 == h) w += l0;
                    if (i1 == h && i1 != i0) w += l1;
                    if (w != 0.f) acc += w * msum(i, NT + j);
                }
                g += s_ob * acc;
            }
            rhs[bi * D + (vel ? QD : 0) + j] = -g;
        } else if (bi + 1 < n) {  // coupling block (support h+1, support h): rows of block bi+1, columns of block bi
            if (e < T4) {        // pos-pos: prior -qa on the diagonal + l0 l1 M of the points inside the segment (h, h+1)
                const int r = (e - T3) / QD, c = (e - T3) - r * QD;
                const int tri = r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r;
                float acc = 0.f;
                for (int i = plo; i <= phi; ++i) {
                    int i0, i1; float l0, l1;
                    point_w(i, i0, i1, l0, l1);
                    if (i0 == h && i1 == h + 1) acc += l0 * l1 * msum(i, tri);
                }
                Cm[(size_t)bi * DD + r * D + c] = s_ob * acc + (r == c ? -s_gp * qa : 0.f);
            } else {             // -Q^-1 Phi = [[-12/dt^3, -6/dt^2], [6/dt^2, 2/dt]] per joint
                const int k = (e - T4) / QD, j = (e - T4) - k * QD;
                if (k == 0) Cm[(size_t)bi * DD + j * D + (QD + j)] = s_gp * (-(qa * dt + qb));            // (pos_j of h+1, vel_j of h)
                else if (k == 1) Cm[(size_t)bi * DD + (QD + j) * D + j] = s_gp * (-qb);                 // (vel_j of h+1, pos_j of h)
                else Cm[(size_t)bi * DD + (QD + j) * D + (QD + j)] = s_gp * (-(qb * dt + qc));           // (vel_j of h+1, vel_j of h)
            }
        }
    }
    __syncthreads();

    GPMP_STAMP(2);
    gpmp_bcr_solve<QD>(Dm, Cm, rhs, sM, n, tid);
    GPMP_STAMP(3);
    // ---- write back: the (possibly updated) current point, the new proposal, the state
    for (int i = tid; i < H * D; i += NTHR) {
        const int h = i / D;
        a.x[base + i] = sx[i];
        a.delta[base + i] = (h > 0 && h < H - 1) ? a.step * rhs[(h - 1) * D + (i - h * D)] : 0.f;
    }
    if (tid == 0) { a.state[(size_t)b * 4 + 0] = F_cur; a.state[(size_t)b * 4 + 1] = lam; a.state[(size_t)b * 4 + 2] = n_acc; a.state[(size_t)b * 4 + 3] = F_cand; }
#ifdef MPDX_GPMP_STAMPS
    __syncthreads();
    if (tid == 0 && b == 0) {
        GPMP_STAMP(4);
        // [linearise + judge, assemble, solve, write back] in s_memtime ticks (shader clock)
        for (int i = 0; i < 4; ++i) a.delta[base + i] = (float)(stamp[i + 1] - stamp[i]);
        for (int i = 0; i < 40; ++i) a.delta[base + 8 + i] = (float)(g_bcr_st[i] - g_bcr_st[0]);   // (dev build only: overwrites trajectory 0's first proposals)
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------------------------
// RRT-Connect.  One workgroup (256 threads) per problem; both trees' node coordinates live in LDS (2 x max_nodes x QD floats) and
// are mirrored to global memory (with the parent links) for the host's path extraction.  All control flow is workgroup-uniform:
// every decision is an LDS broadcast.
struct RrtArgs {
    mpdx_guide_params gp;       // robot + collision fields (edges are checked with the LINK radius only, as traj_metrics_kernel)
    const float* start;         // [n][QD]
    const float* goal;          // [n][QD]
    float* nodes;               // [n][2][max_nodes][QD]   tree 0 grows from the start, tree 1 from the goal
    int* parent;                // [n][2][max_nodes]
    int* count;                 // [n][2]
    int* link;                  // [n][2] node indices where the trees met (-1: not solved)
    int* iters;                 // [n] iterations used
    float q_lo[8], q_hi[8];     // sampling box
    float step;
    int max_nodes, max_iters, max_connect, n_checks;
    unsigned long long seed;
};

__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint64_t ctr, float (&u)[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u[0] = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f); u[1] = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    u[2] = ((float)(c2 >> 8) + 0.5f) * (1.0f / 16777216.0f); u[3] = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// does configuration q collide?  `part` / `nparts` split the Panda's link spheres and self-collision pairs over threads
template <int QD, int DIM, int ROBOT>
__device__ __forceinline__ bool config_hit(const mpdx_guide_params& gp, const float* sprim, const float (&q)[QD], int part, int nparts) {
    bool hit = false;
    if constexpr (ROBOT == MPDX_ROBOT_POINTMASS) {
        if (part != 0) return false;
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
                    if (pr % nparts != part) continue;
                    const float dx = P[kPandaPA[pr]][0] - P[kPandaPB[pr]][0], dy = P[kPandaPA[pr]][1] - P[kPandaPB[pr]][1],
                                dz = P[kPandaPA[pr]][2] - P[kPandaPB[pr]][2];
                    hit |= sqrtf(dx * dx + dy * dy + dz * dz) < kPandaSR[kPandaPA[pr]] + kPandaSR[kPandaPB[pr]];
                }
            } else {
#pragma unroll
                for (int s = 0; s < kPandaNS; ++s) {
                    if (s % nparts != part) continue;
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
    return hit;
}

constexpr int kRrtThreads = 256;

template <int QD, int DIM, int ROBOT>
__global__ __launch_bounds__(kRrtThreads) void rrt_connect_kernel(const RrtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const mpdx_guide_params& gp = a.gp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int M = a.max_nodes;
    float* tree = sm;                          // [2][M][QD]
    float* sq = tree + 2 * M * QD;             // scratch: [0] qr/target, [1] qn/cur, [2] qnew/nxt  (QD floats each, padded to 8)
    float* sred = sq + 3 * 8;                  // [4][2] per-wave arg-min
    int* sint = (int*)(sred + 8);              // [0] nearest index
    float* sprim = (float*)(sint + 8);
    for (int i = tid; i < gp.n_prim_floats; i += kRrtThreads) sprim[i] = gp.prims[i];
    float* gnodes = a.nodes + (size_t)b * 2 * M * QD;
    int* gpar = a.parent + (size_t)b * 2 * M;
    if (tid < QD) {
        tree[tid] = a.start[(size_t)b * QD + tid]; tree[M * QD + tid] = a.goal[(size_t)b * QD + tid];
        gnodes[tid] = tree[tid]; gnodes[(size_t)M * QD + tid] = tree[M * QD + tid];
    }
    if (tid == 0) { gpar[0] = -1; gpar[M] = -1; }
    int cnt0 = 1, cnt1 = 1;                    // node counts, uniform (every thread tracks them; scalars: no dynamic register indexing)
    auto cnt_of = [&](int t) { return t == 0 ? cnt0 : cnt1; };
    int link0 = -1, link1 = -1, used = 0;
    __syncthreads();

    const int nchk = a.n_checks;
    const int nparts = (ROBOT == MPDX_ROBOT_PANDA) ? (kRrtThreads / nchk < 12 ? kRrtThreads / nchk : 12) : 1;
    // workgroup-wide: is the straight segment qa -> qb (LDS, QD floats each) collision free on nchk interpolated configurations?
    auto edge_free = [&](const float* qa, const float* qb) -> bool {
        const int c = tid % nchk, part = tid / nchk;
        bool hit = false;
        if (part < nparts) {
            const float w = nchk > 1 ? (float)c / (float)(nchk - 1) : 0.f;
            float q[QD];
#pragma unroll
            for (int j = 0; j < QD; ++j) q[j] = (1.0f - w) * qa[j] + w * qb[j];
            hit = config_hit<QD, DIM, ROBOT>(gp, sprim, q, part, nparts);
        }
        // workgroup-wide OR through the dynamic LDS region (__syncthreads_or brings a static __shared__ word: it shifts the dynamic
        // base and makes the 160-KiB opt-in fail)
        const bool wave_hit = __ballot(hit) != 0ull;
        if (lane == 0) sint[4 + wave] = wave_hit ? 1 : 0;
        __syncthreads();
        const bool any = (sint[4] | sint[5] | sint[6] | sint[7]) != 0;
        __syncthreads();
        return !any;
    };
    // workgroup-wide: index of the node of tree t nearest to target (LDS); lowest index wins ties
    auto nearest = [&](int t, const float* target) -> int {
        float best = 3.0e38f;
        int bi = 0;
        const float* tn = tree + (size_t)t * M * QD;
        const int nt_ = cnt_of(t);
        for (int i = tid; i < nt_; i += kRrtThreads) {
            float d2 = 0.f;
#pragma unroll
            for (int j = 0; j < QD; ++j) { const float d = tn[i * QD + j] - target[j]; d2 += d * d; }
            if (d2 < best) { best = d2; bi = i; }
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const float ob = __shfl_xor(best, s, 64);
            const int oi = __shfl_xor(bi, s, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { sred[wave * 2] = best; ((int*)sred)[wave * 2 + 1] = bi; }
        __syncthreads();
        float fb = sred[0];
        int fi = ((int*)sred)[1];
#pragma unroll
        for (int w = 1; w < kRrtThreads / 64; ++w) {
            const float ob = sred[w * 2];
            const int oi = ((int*)sred)[w * 2 + 1];
            if (ob < fb || (ob == fb && oi < fi)) { fb = ob; fi = oi; }
        }
        __syncthreads();
        return fi;
    };
    // steer from `from` towards `to` by at most `step`: result -> out (LDS); returns whether `to` was reached.  Uniform.
    auto steer = [&](const float* from, const float* to, float* out) -> bool {
        float d2 = 0.f;
#pragma unroll
        for (int j = 0; j < QD; ++j) { const float d = to[j] - from[j]; d2 += d * d; }
        const float dist = sqrtf(d2);
        const bool reach = dist <= a.step;
        const float sc = reach ? 1.0f : a.step / fmaxf(dist, 1e-12f);
        float o[QD];
#pragma unroll
        for (int j = 0; j < QD; ++j) o[j] = reach ? to[j] : from[j] + (to[j] - from[j]) * sc;
        __syncthreads();   // everyone has read `from` / `to` (out may alias scratch that others still read)
        if (tid == 0) {
#pragma unroll
            for (int j = 0; j < QD; ++j) out[j] = o[j];
        }
        __syncthreads();
        return reach;
    };
    auto add_node = [&](int t, const float* q, int par) -> int {   // uniform; caller guarantees room in tree t
        const int idx = cnt_of(t);
        if (tid < QD) {
            tree[((size_t)t * M + idx) * QD + tid] = q[tid];
            gnodes[((size_t)t * M + idx) * QD + tid] = q[tid];
        }
        if (tid == 0) gpar[(size_t)t * M + idx] = par;
        if (t == 0) cnt0 = idx + 1; else cnt1 = idx + 1;
        __syncthreads();
        return idx;
    };

    float* qr = sq, *qcur = sq + 8, *qnew = sq + 16;
    bool done = false;
    for (int it = 1; it <= a.max_iters && !done; ++it) {
        used = it;
        const int ta = it & 1, tb = 1 - ta;
        if (cnt0 >= M || cnt1 >= M) break;   // a full tree ends the search (unsolved) instead of corrupting links
        if (tid == 0) {
            float u[8];
            float u4[4];
            philox_uniform4(a.seed, ((uint64_t)b << 32) | (uint64_t)(2 * it), u4);
            u[0] = u4[0]; u[1] = u4[1]; u[2] = u4[2]; u[3] = u4[3];
            philox_uniform4(a.seed, ((uint64_t)b << 32) | (uint64_t)(2 * it + 1), u4);
            u[4] = u4[0]; u[5] = u4[1]; u[6] = u4[2]; u[7] = u4[3];
#pragma unroll
            for (int j = 0; j < QD; ++j) qr[j] = a.q_lo[j] + (a.q_hi[j] - a.q_lo[j]) * u[j];
        }
        __syncthreads();
        const int ia = nearest(ta, qr);
        const float* qn = tree + ((size_t)ta * M + ia) * QD;
        steer(qn, qr, qnew);
        if (!edge_free(qn, qnew)) continue;
        const int inew = add_node(ta, qnew, ia);
        // connect: walk the other tree from its nearest node towards qnew until blocked, full or there
        int cur_idx = nearest(tb, qnew);
        if (tid < QD) qcur[tid] = tree[((size_t)tb * M + cur_idx) * QD + tid];
        __syncthreads();
        for (int sstep = 0; sstep < a.max_connect; ++sstep) {
            float* nxt = qr;   // the sample is no longer needed
            const bool reach = steer(qcur, qnew, nxt);
            if (!edge_free(qcur, nxt)) break;
            if (reach) {       // the reached target IS qnew: record the link, do not duplicate the node
                link0 = ta == 0 ? inew : cur_idx;
                link1 = ta == 0 ? cur_idx : inew;
                done = true;
                break;
            }
            if (cnt_of(tb) >= M) break;
            cur_idx = add_node(tb, nxt, cur_idx);
            if (tid < QD) qcur[tid] = nxt[tid];
            __syncthreads();
        }
    }
    if (tid == 0) {
        a.count[(size_t)b * 2] = cnt0; a.count[(size_t)b * 2 + 1] = cnt1;
        a.link[(size_t)b * 2] = link0; a.link[(size_t)b * 2 + 1] = link1;
        a.iters[b] = used;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// RRT-Connect post-processing on the device (round 4; round 3 did it in host loops over GPU edge checks: ~170 ms of the 200 ms a
// 100-problem narrow-passage batch took): per problem ONE workgroup
//   1. extracts the path start ... meeting node | other branch ... goal from the two trees (parent links staged in LDS and chased there),
//   2. shortcuts it greedily (`rounds` passes: from node i jump to the LAST later node it sees - the host algorithm of
//      generate_trajectories.shortcut_path, same result; edges checked on n_checks interpolated configurations with config_hit),
//   3. resamples it uniformly in arc length to H support points, velocities by central differences, zero at both ends
//      (generate_trajectories.resample_path), and writes the [H][2 QD] state trajectory.
// An unsolved problem (link < 0) or a path longer than kPathMax nodes becomes the straight line start -> goal (the optimiser may repair it).
constexpr int kPathMax = 1024;
struct RrtPathArgs {
    mpdx_guide_params gp;
    const float* start;         // [n][QD]
    const float* goal;          // [n][QD]
    const float* nodes;         // [n][2][max_nodes][QD]
    const int* parent;          // [n][2][max_nodes]
    const int* link;            // [n][2]
    float* out;                 // [n][H][2 QD]
    int* path_len;              // [n] nodes of the shortcut path (2: straight line), or null
    int max_nodes, H, n_checks, rounds;
    float dt;
};

template <int QD, int DIM, int ROBOT>
__global__ __launch_bounds__(kRrtThreads) void rrt_path_kernel(const RrtPathArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const mpdx_guide_params& gp = a.gp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int M = a.max_nodes, H = a.H;
    float* pth = sm;                              // [kPathMax][QD] path nodes
    float* pos = pth + kPathMax * QD;             // [H][QD] resampled positions
    float* sarc = pos + H * QD;                   // [kPathMax] cumulative arc length
    int* sidx = (int*)(sarc + kPathMax);          // [kPathMax] node indices / keep list
    int* spar = sidx + kPathMax;                  // [2][M] parent links
    int* sint = spar + 2 * M;                     // [16] scalars: 0 m, 1 n0, 4..7 per-wave hit flags
    float* sprim = (float*)(sint + 16);
    for (int i = tid; i < gp.n_prim_floats; i += kRrtThreads) sprim[i] = gp.prims[i];
    const int* gpar = a.parent + (size_t)b * 2 * M;
    for (int i = tid; i < 2 * M; i += kRrtThreads) spar[i] = gpar[i];
    const int l0 = a.link[(size_t)b * 2], l1 = a.link[(size_t)b * 2 + 1];
    __syncthreads();
    // ---- 1. node index list: tree 0 from the meeting node back to the start (reversed below), then tree 1 from its meeting node to the goal
    if (tid == 0) {
        int n0 = 0, n1 = 0;
        if (l0 >= 0 && l1 >= 0) {
            for (int k = l0; k >= 0 && n0 <= kPathMax; k = spar[k]) ++n0;
            for (int k = l1; k >= 0 && n1 <= kPathMax; k = spar[M + k]) ++n1;
        }
        if (n0 == 0 || n0 + n1 > kPathMax) { sint[0] = 0; sint[1] = 0; }
        else {
            int w = n0 - 1;
            for (int k = l0; k >= 0; k = spar[k]) sidx[w--] = k;              // start ... meeting node
            w = n0;
            for (int k = l1; k >= 0; k = spar[M + k]) sidx[w++] = M + k;      // other tree's branch ... goal
            sint[0] = n0 + n1; sint[1] = n0;
        }
    }
    __syncthreads();
    int m = sint[0];
    const float* gnodes = a.nodes + (size_t)b * 2 * M * QD;
    if (m == 0) {   // straight line
        if (tid < QD) { pth[tid] = a.start[(size_t)b * QD + tid]; pth[QD + tid] = a.goal[(size_t)b * QD + tid]; }
        m = 2;
    } else {
        for (int i = tid; i < m * QD; i += kRrtThreads) { const int nd = i / QD, j = i - nd * QD; pth[i] = gnodes[(size_t)sidx[nd] * QD + j]; }
    }
    __syncthreads();
    // ---- 2. greedy shortcutting
    const int nchk = a.n_checks;
    const int nparts = (ROBOT == MPDX_ROBOT_PANDA) ? (kRrtThreads / nchk < 12 ? kRrtThreads / nchk : 12) : 1;
    auto edge_free = [&](const float* qa, const float* qb) -> bool {   // workgroup-wide, as in rrt_connect_kernel
        const int c = tid % nchk, part = tid / nchk;
        bool hit = false;
        if (part < nparts) {
            const float w = nchk > 1 ? (float)c / (float)(nchk - 1) : 0.f;
            float q[QD];
#pragma unroll
            for (int j = 0; j < QD; ++j) q[j] = (1.0f - w) * qa[j] + w * qb[j];
            hit = config_hit<QD, DIM, ROBOT>(gp, sprim, q, part, nparts);
        }
        const bool wave_hit = __ballot(hit) != 0ull;
        if (lane == 0) sint[4 + wave] = wave_hit ? 1 : 0;
        __syncthreads();
        const bool any = (sint[4] | sint[5] | sint[6] | sint[7]) != 0;
        __syncthreads();
        return !any;
    };
    for (int round = 0; round < a.rounds && m > 2; ++round) {
        int nk = 1, i = 0;           // keep list in sidx (uniform control flow: every thread tracks i, nk)
        if (tid == 0) sidx[0] = 0;
        while (i < m - 1) {
            int j = m - 1;
            for (; j > i + 1; --j)
                if (edge_free(pth + i * QD, pth + j * QD)) break;   // the LAST later node that i sees; none: the next node
            if (tid == 0) sidx[nk] = j;
            ++nk;
            i = j;
        }
        __syncthreads();
        if (nk == m) break;
        // compact the path in place (ascending indices: node k moves to a position <= its own; one float per thread and pass)
        for (int k = 1; k < nk; ++k) {
            const int src = sidx[k];
            float v = 0.f;
            if (tid < QD) v = pth[src * QD + tid];
            __syncthreads();
            if (tid < QD) pth[k * QD + tid] = v;
            __syncthreads();
        }
        m = nk;
    }
    if (tid == 0 && a.path_len) a.path_len[b] = m;
    // ---- 3. arc-length resampling to H support points + central-difference velocities
    if (tid == 0) {
        float acc = 0.f;
        sarc[0] = 0.f;
        for (int k = 1; k < m; ++k) {
            float d2 = 0.f;
#pragma unroll
            for (int j = 0; j < QD; ++j) { const float d = pth[k * QD + j] - pth[(k - 1) * QD + j]; d2 += d * d; }
            acc += sqrtf(d2);
            sarc[k] = acc;
        }
    }
    __syncthreads();
    const float total = sarc[m - 1];
    for (int h = tid; h < H; h += kRrtThreads) {
        const float u = H > 1 ? total * (float)h / (float)(H - 1) : 0.f;
        int k = 1;
        while (k < m - 1 && sarc[k] <= u) ++k;     // first node with s[k] > u (searchsorted right), clamped to [1, m - 1]
        const float den = fmaxf(sarc[k] - sarc[k - 1], 1e-12f);
        const float w = fminf(fmaxf((u - sarc[k - 1]) / den, 0.f), 1.f);
#pragma unroll
        for (int j = 0; j < QD; ++j) {
            float v = pth[(k - 1) * QD + j] * (1.0f - w) + pth[k * QD + j] * w;
            if (h == 0) v = pth[j];
            if (h == H - 1) v = pth[(m - 1) * QD + j];
            pos[h * QD + j] = v;
        }
    }
    __syncthreads();
    float* o = a.out + (size_t)b * H * 2 * QD;
    for (int i = tid; i < H * QD; i += kRrtThreads) {
        const int h = i / QD, j = i - h * QD;
        o[h * 2 * QD + j] = pos[i];
        o[h * 2 * QD + QD + j] = (h > 0 && h < H - 1) ? (pos[(h + 1) * QD + j] - pos[(h - 1) * QD + j]) / (2.0f * a.dt) : 0.f;
    }
}

template <int QD>
inline size_t rrt_path_lds_bytes(int max_nodes, int H, int n_prim_floats) {
    return (size_t)(kPathMax * QD + H * QD + kPathMax + kPathMax + 2 * max_nodes + 16 + n_prim_floats) * sizeof(float);
}

template <int QD>
inline size_t rrt_lds_bytes(int max_nodes, int n_prim_floats) {
    return (size_t)(2 * max_nodes * QD + 3 * 8 + 8 + 8 + n_prim_floats) * sizeof(float);
}

}  // namespace mpdx
