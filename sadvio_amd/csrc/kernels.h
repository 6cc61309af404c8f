// kernels.h — hand-written gfx950 kernels of the BA iteration (FP64, wave64).
//
// One Levenberg-Marquardt step attempt ("slot") is three launches on one stream, no host round trip:
//   k_build   landmark-major tiles: linearise every observation (K1/K2), eliminate each landmark
//             (3x3 damped H_ll inverse in registers), accumulate the reduced pose system
//             S = sum_l Jp^T (I - Jl M^-1 Jl^T) Jp, g = sum_l Jp^T (I - Jl M^-1 Jl^T) r in an LDS tile
//             (ds_add_f64), flush the tile's non-zeros to HBM with global_atomic_add_f64   (K5)
//   k_solve   one workgroup per window: pose-only factors (K4), Jacobi scale + LM diagonal,
//             LDS Cholesky + triangular solves, candidate poses                                (K6)
//   k_backsub landmark-major tiles again: delta_l = -M^-1 Jl^T (r + Jp delta_p), candidate cost,
//             model cost change                                                                (K7)
// The LM accept/reject logic of Ceres 2.2 (TrustRegionMinimizer / LevenbergMarquardtStrategy) runs on
// the device (`lm_decide`), evaluated redundantly by every workgroup of the next k_build from the
// previous slot's accumulators, so the whole <= 20-iteration solve is one stream submission.
#pragma once
#include "ba_types.h"
#include "device_math.h"

namespace sadvio {

struct DevPtrs {
    const WinDev* win;
    const Tile* tiles;
    const double* kf_T0;
    const int* kf_fidx;
    double* xp; double* xv; double* xba; double* xbg;  // [2][...]
    long long xp_stride, xv_stride, xl_stride;
    const double* kf_vel; const double* kf_ba; const double* kf_bg;
    const double* cam_K; const double* cam_T; const double* cam_isig;
    const double* lmk_p; double* xl; double* s_lmk;
    const unsigned char* lmk_const;
    const int* lmk_ob; const int* lmk_oe;
    const int* obs_kf; const int* obs_cam; const double* obs_meas;
    const PriorDev* priors;
    double* S; double* gred; double* gfull; double* hdiag; double* delta; double* s_pose;
    LmState* states;  // [n_win][slots+2]
    IterAcc* acc;     // [n_win][slots+1]
    int state_stride;
    int n_win;
    SolveOpts o;
};

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

__device__ __forceinline__ void atomic_max_u64(unsigned long long* p, unsigned long long v) { atomicMax(p, v); }

// Ceres 2.2 TrustRegionMinimizer iteration bookkeeping, restated (see oracle/solver.c for the
// line-by-line CPU restatement this must agree with).
__device__ __forceinline__ LmState lm_decide(LmState s, const IterAcc& a, const SolveOpts& o) {
    if (s.done) return s;
    double x_cost = 0.5 * a.lin_cost;
    if (s.iter == 0) s.initial_cost = x_cost;
    s.x_cost = x_cost;
    s.iter += 1;
    bool valid = (a.chol_fail == 0) && (a.mcc > 0.0);
    if (!valid) {
        s.n_invalid += 1;
        s.n_unsuccess += 1;
        if (s.n_invalid >= o.max_num_consecutive_invalid_steps) { s.done = 1; s.termination = 5; return s; }
        s.radius *= 0.5;
    } else {
        s.n_invalid = 0;
        double step_norm = sqrt(a.step_norm2);
        double cand_cost = 0.5 * a.cand_cost;
        if (step_norm <= o.parameter_tolerance * (s.x_norm + o.parameter_tolerance)) {
            s.done = 1; s.termination = 2; return s;
        }
        double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= o.function_tolerance * x_cost) { s.done = 1; s.termination = 1; return s; }
        double rel = cost_change / a.mcc;
        if (rel > o.min_relative_decrease) {
            s.cur ^= 1;
            s.x_norm = sqrt(a.cand_norm2);
            s.x_cost = cand_cost;
            double t = 2.0 * rel - 1.0;
            s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            s.radius = fmin(o.max_radius, s.radius);
            s.decrease_factor = 2.0;
            s.n_success += 1;
        } else {
            s.radius = s.radius / s.decrease_factor;
            s.decrease_factor *= 2.0;
            s.n_unsuccess += 1;
        }
    }
    if (s.iter >= o.max_num_iterations) { s.done = 1; s.termination = 0; }
    else if (s.radius <= o.min_radius) { s.done = 1; s.termination = 4; }
    return s;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// ---- shared tile prologue: per-observation linearisation + per-landmark elimination -------------
// LDS carve used by k_build and k_backsub.
struct TileLds {
    double* poseTab;   // [n_kf][POSE_TAB] at x
    double* obsStage;  // [MAX_TILE_OBS][OBS_STAGE]
    double* lmkStage;  // [MAX_TILE_LMK][LMK_STAGE]
    int* obsPa;        // [MAX_TILE_OBS] pose tile row (free_kf*6) or -1
    int* obsLmk;       // [MAX_TILE_OBS] local landmark index
    double* red;       // [16] reduction scratch
};

template <int FACTOR>
__device__ __forceinline__ void linearize_obs(const DevPtrs& P, const WinDev& W, const double* poseTab, int o,
                                              int lmk, const double* pw, double* r, double* Jp, double* Jl,
                                              bool lmk_free) {
    int kf = P.obs_kf[o], cam = P.obs_cam[o];
    const double* tab = poseTab + (kf - W.kf_base) * POSE_TAB;
    const double* K = P.cam_K + 4 * cam;
    const double* Tsf = P.cam_T + 12 * cam;
    double isig = P.cam_isig[cam];
    if (FACTOR == 0) {
        const double* m = P.obs_meas + 2 * (long long)o;
        pixel_factor<true>(tab, K, Tsf, pw, m[0], m[1], isig, r, Jp, Jl);
    } else {
        const double* m = P.obs_meas + 3 * (long long)o;
        double b[3] = {m[0], m[1], m[2]};
        angular_factor<true>(tab, Tsf, pw, b, isig, r, Jp, Jl);
    }
    if (P.kf_fidx[kf] < 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) Jp[i] = 0.0;
    }
    if (!lmk_free) {
#pragma unroll
        for (int i = 0; i < 6; i++) Jl[i] = 0.0;
    }
}

// Stage pose table, linearise the tile's observations into LDS, eliminate its landmarks.
// After this: obsStage[a] = {Jp, Jl, r, N = Jl Minv}, lmkStage[l] = {Minv(6), gl(3)}.
// Returns this thread's partial (sum r^2, fixed r^2, max |g_l|) through out params.
template <int FACTOR>
__device__ __forceinline__ void tile_prologue(const DevPtrs& P, const WinDev& W, const Tile& T, const TileLds& L,
                                              int cur, double radius, bool write_scale, double& cost_part,
                                              double& fixed_part, double& gmax_part) {
    const int tid = threadIdx.x;
    const double* xp = P.xp + (long long)cur * P.xp_stride;
    const double* xl = P.xl + (long long)cur * P.xl_stride;
    for (int k = tid; k < W.n_kf; k += blockDim.x) {
        int g = W.kf_base + k;
        double d6[6];
#pragma unroll
        for (int i = 0; i < 6; i++) d6[i] = xp[6 * (long long)g + i];
        pose_table_entry(P.kf_T0 + 12 * (long long)g, d6, L.poseTab + k * POSE_TAB);
    }
    // map observation -> local landmark (thread per landmark writes its range)
    int nl = T.lmk1 - T.lmk0;
    for (int l = tid; l < nl; l += blockDim.x) {
        int gl = T.lmk0 + l;
        for (int o = P.lmk_ob[gl]; o < P.lmk_oe[gl]; o++) L.obsLmk[o - T.obs0] = l;
    }
    __syncthreads();
    int no = T.obs1 - T.obs0;
    cost_part = 0; fixed_part = 0; gmax_part = 0;
    for (int a = tid; a < no; a += blockDim.x) {
        int o = T.obs0 + a;
        int l = L.obsLmk[a];
        int gl = T.lmk0 + l;
        bool lfree = !(P.lmk_const && P.lmk_const[gl]);
        double pw[3] = {P.lmk_p[3 * (long long)gl] + xl[3 * (long long)gl], P.lmk_p[3 * (long long)gl + 1] + xl[3 * (long long)gl + 1],
                        P.lmk_p[3 * (long long)gl + 2] + xl[3 * (long long)gl + 2]};
        double r[2], Jp[12], Jl[6];
        linearize_obs<FACTOR>(P, W, L.poseTab, o, gl, pw, r, Jp, Jl, lfree);
        int fi = P.kf_fidx[P.obs_kf[o]];
        L.obsPa[a] = fi < 0 ? -1 : fi * 6;
        double* st = L.obsStage + a * OBS_STAGE;
        if (fi < 0 && !lfree) {  // every parameter block constant: fixed cost, not part of the program
            fixed_part += r[0] * r[0] + r[1] * r[1];
            r[0] = 0; r[1] = 0;
        } else {
            cost_part += r[0] * r[0] + r[1] * r[1];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = Jp[i];
#pragma unroll
        for (int i = 0; i < 6; i++) st[12 + i] = Jl[i];
        st[18] = r[0]; st[19] = r[1];
    }
    __syncthreads();
    // per landmark: H_ll, g_l, LM damping, inverse
    for (int l = tid; l < nl; l += blockDim.x) {
        int gl = T.lmk0 + l;
        double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        int o0 = P.lmk_ob[gl] - T.obs0, o1 = P.lmk_oe[gl] - T.obs0;
        for (int a = o0; a < o1; a++) {
            const double* st = L.obsStage + a * OBS_STAGE;
            const double* Jl = st + 12;
            double r0 = st[18], r1 = st[19];
            H[0] += Jl[0] * Jl[0] + Jl[3] * Jl[3];
            H[1] += Jl[0] * Jl[1] + Jl[3] * Jl[4];
            H[2] += Jl[0] * Jl[2] + Jl[3] * Jl[5];
            H[3] += Jl[1] * Jl[1] + Jl[4] * Jl[4];
            H[4] += Jl[1] * Jl[2] + Jl[4] * Jl[5];
            H[5] += Jl[2] * Jl[2] + Jl[5] * Jl[5];
            g[0] += Jl[0] * r0 + Jl[3] * r1;
            g[1] += Jl[1] * r0 + Jl[4] * r1;
            g[2] += Jl[2] * r0 + Jl[5] * r1;
        }
        bool active = (o1 > o0) && !(P.lmk_const && P.lmk_const[gl]);
        double* ls = L.lmkStage + l * LMK_STAGE;
        if (active) {
            double s[3];
            if (write_scale) {
                s[0] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[0])) : 1.0;
                s[1] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[3])) : 1.0;
                s[2] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[5])) : 1.0;
                P.s_lmk[3 * (long long)gl] = s[0]; P.s_lmk[3 * (long long)gl + 1] = s[1]; P.s_lmk[3 * (long long)gl + 2] = s[2];
            } else {
                s[0] = P.s_lmk[3 * (long long)gl]; s[1] = P.s_lmk[3 * (long long)gl + 1]; s[2] = P.s_lmk[3 * (long long)gl + 2];
            }
            double ir = 1.0 / radius;
            double s0 = s[0] * s[0], s1 = s[1] * s[1], s2 = s[2] * s[2];
            double M[6] = {H[0], H[1], H[2], H[3], H[4], H[5]};
            M[0] += fmin(fmax(s0 * H[0], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s0;
            M[3] += fmin(fmax(s1 * H[3], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s1;
            M[5] += fmin(fmax(s2 * H[5], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s2;
            double Mi[6];
            sym3_inverse(M, Mi);
#pragma unroll
            for (int i = 0; i < 6; i++) ls[i] = Mi[i];
            gmax_part = fmax(gmax_part, fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))));
        } else {
#pragma unroll
            for (int i = 0; i < 6; i++) ls[i] = 0.0;
        }
        ls[6] = g[0]; ls[7] = g[1]; ls[8] = g[2];
    }
    __syncthreads();
    // N_a = Jl_a Minv (2x3)
    for (int a = tid; a < no; a += blockDim.x) {
        double* st = L.obsStage + a * OBS_STAGE;
        const double* Mi = L.lmkStage + L.obsLmk[a] * LMK_STAGE;
        const double* Jl = st + 12;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            double j0 = Jl[3 * q], j1 = Jl[3 * q + 1], j2 = Jl[3 * q + 2];
            st[20 + 3 * q] = j0 * Mi[0] + j1 * Mi[1] + j2 * Mi[2];
            st[20 + 3 * q + 1] = j0 * Mi[1] + j1 * Mi[3] + j2 * Mi[4];
            st[20 + 3 * q + 2] = j0 * Mi[2] + j1 * Mi[4] + j2 * Mi[5];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ size_t tile_lds_carve(char* smem, int n_kf, TileLds& L) {
    size_t off = 0;
    L.poseTab = (double*)(smem + off); off += sizeof(double) * (size_t)n_kf * POSE_TAB;
    L.obsStage = (double*)(smem + off); off += sizeof(double) * MAX_TILE_OBS * OBS_STAGE;
    L.lmkStage = (double*)(smem + off); off += sizeof(double) * MAX_TILE_LMK * LMK_STAGE;
    L.red = (double*)(smem + off); off += sizeof(double) * 16;
    L.obsPa = (int*)(smem + off); off += sizeof(int) * MAX_TILE_OBS;
    L.obsLmk = (int*)(smem + off); off += sizeof(int) * MAX_TILE_OBS;
    off = (off + 15) & ~(size_t)15;
    return off;
}
inline size_t tile_lds_bytes(int n_kf) {
    size_t off = sizeof(double) * ((size_t)n_kf * POSE_TAB + MAX_TILE_OBS * OBS_STAGE + MAX_TILE_LMK * LMK_STAGE + 16) +
                 sizeof(int) * 2 * MAX_TILE_OBS;
    return (off + 15) & ~(size_t)15;
}

// ---- K5: build the reduced system ----------------------------------------------------------------
// LDS_TILE = true : the window's pose block (<= MAX_LDS_NPOSE) is accumulated in an LDS lower-triangular
//                   tile with ds_add_f64 and flushed once per workgroup;
// LDS_TILE = false: contributions go straight to HBM with global_atomic_add_f64 (large windows).
template <int FACTOR, bool LDS_TILE>
__global__ __launch_bounds__(BUILD_THREADS) void k_build(DevPtrs P, int slot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Tile T = P.tiles[blockIdx.x];
    const WinDev W = P.win[T.w];
    const int tid = threadIdx.x;
    LmState st;
    if (slot == 0) st = P.states[(long long)T.w * P.state_stride];
    else st = lm_decide(P.states[(long long)T.w * P.state_stride + slot - 1], P.acc[(long long)T.w * P.state_stride + slot - 1], P.o);
    if (slot > 0 && (int)blockIdx.x == W.tile_begin && tid == 0) P.states[(long long)T.w * P.state_stride + slot] = st;
    if (st.done) return;

    TileLds L;
    size_t off = tile_lds_carve(smem, W.n_kf, L);
    double* Stile = (double*)(smem + off);
    const int Nt = W.Npose;
    const int tri_n = Nt * (Nt + 1) / 2;
    double* gT = Stile + (LDS_TILE ? tri_n : 0);  // reduced gradient
    double* gfT = gT + (LDS_TILE ? Nt : 0);       // full gradient
    double* hdT = gfT + (LDS_TILE ? Nt : 0);      // diag(H_pp)
    if (LDS_TILE)
        for (int i = tid; i < tri_n + 3 * Nt; i += blockDim.x) Stile[i] = 0.0;

    double cost_part, fixed_part, gmax_part;
    tile_prologue<FACTOR>(P, W, T, L, st.cur, st.radius, slot == 0, cost_part, fixed_part, gmax_part);

    double* Sg = P.S + W.S_off;
    double* gredg = P.gred + W.red_off;
    double* gfullg = P.gfull + W.red_off;
    double* hdg = P.hdiag + W.red_off;
    const int dpf = W.dpf, Np = W.Np;
    const int no = T.obs1 - T.obs0;

    // gradient + diag(H_pp): thread per (obs, i)
    for (int it = tid; it < no * 6; it += blockDim.x) {
        int a = it / 6, i = it - 6 * a;
        int pa = L.obsPa[a];
        if (pa < 0) continue;
        const double* s = L.obsStage + a * OBS_STAGE;
        const double* gl = L.lmkStage + L.obsLmk[a] * LMK_STAGE + 6;
        double j0 = s[i], j1 = s[6 + i];
        // r~ = r - N g_l
        double rt0 = s[18] - (s[20] * gl[0] + s[21] * gl[1] + s[22] * gl[2]);
        double rt1 = s[19] - (s[23] * gl[0] + s[24] * gl[1] + s[25] * gl[2]);
        double gr = j0 * rt0 + j1 * rt1;
        double gf = j0 * s[18] + j1 * s[19];
        double hd = j0 * j0 + j1 * j1;
        if (LDS_TILE) {
            atomic_add_f64(&gT[pa + i], gr);
            atomic_add_f64(&gfT[pa + i], gf);
            atomic_add_f64(&hdT[pa + i], hd);
        } else {
            int row = (pa / 6) * dpf + i;
            atomic_add_f64(&gredg[row], gr);
            atomic_add_f64(&gfullg[row], gf);
            atomic_add_f64(&hdg[row], hd);
        }
    }
    // S blocks: item = (a, b-offset, row i): row i of Jp_a^T W_ab Jp_b, W_ab = delta_ab I - N_a Jl_b^T
    const int kmax = T.kmax;
    const int items = no * kmax * 6;
    for (int it = tid; it < items; it += blockDim.x) {
        int a = it / (kmax * 6);
        int rem = it - a * kmax * 6;
        int bo = rem / 6, i = rem - 6 * bo;
        int pa = L.obsPa[a];
        if (pa < 0) continue;
        int gl = T.lmk0 + L.obsLmk[a];
        int b = P.lmk_ob[gl] - T.obs0 + bo;
        if (b >= P.lmk_oe[gl] - T.obs0) continue;
        int pb = L.obsPa[b];
        if (pb < 0 || pb > pa) continue;  // lower triangle only (block row >= block col)
        const double* sa = L.obsStage + a * OBS_STAGE;
        const double* sb = L.obsStage + b * OBS_STAGE;
        const double* Na = sa + 20;
        const double* Jlb = sb + 12;
        double w00 = -(Na[0] * Jlb[0] + Na[1] * Jlb[1] + Na[2] * Jlb[2]);
        double w01 = -(Na[0] * Jlb[3] + Na[1] * Jlb[4] + Na[2] * Jlb[5]);
        double w10 = -(Na[3] * Jlb[0] + Na[4] * Jlb[1] + Na[5] * Jlb[2]);
        double w11 = -(Na[3] * Jlb[3] + Na[4] * Jlb[4] + Na[5] * Jlb[5]);
        if (a == b) { w00 += 1.0; w11 += 1.0; }
        double c0 = sa[i] * w00 + sa[6 + i] * w10;
        double c1 = sa[i] * w01 + sa[6 + i] * w11;
        int row = pa + i;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            int col = pb + j;
            if (col > row) continue;  // within a diagonal block keep the lower part
            double v = c0 * sb[j] + c1 * sb[6 + j];
            if (LDS_TILE) atomic_add_f64(&Stile[tri(row, col)], v);
            else atomic_add_f64(&Sg[(long long)((pa / 6) * dpf + i) * Np + (pb / 6) * dpf + j], v);
        }
    }
    // cost / gradient-max reduction
    double c = wave_sum(cost_part), f = wave_sum(fixed_part), gm = wave_max(gmax_part);
    IterAcc* acc = P.acc + (long long)T.w * P.state_stride + slot;
    if ((tid & 63) == 0) {
        if (c != 0.0) atomic_add_f64(&acc->lin_cost, c);
        if (slot == 0 && f != 0.0) atomic_add_f64(&acc->fixed_cost, f);
        atomic_max_u64(&acc->gmax_bits, (unsigned long long)__double_as_longlong(gm));
    }
    if (LDS_TILE) {
        __syncthreads();
        // flush non-zeros: tile (row, col) -> global (row/6*dpf + row%6, col/6*dpf + col%6)
        for (int idx = tid; idx < tri_n; idx += blockDim.x) {
            double v = Stile[idx];
            if (v == 0.0) continue;
            // invert tri(): row = floor((sqrt(8 idx + 1) - 1) / 2)
            int row = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
            while (tri(row + 1, 0) <= idx) row++;
            while (tri(row, 0) > idx) row--;
            int col = idx - tri(row, 0);
            atomic_add_f64(&Sg[(long long)((row / 6) * dpf + row % 6) * Np + (col / 6) * dpf + col % 6], v);
        }
        for (int i = tid; i < Nt; i += blockDim.x) {
            int row = (i / 6) * dpf + i % 6;
            if (gT[i] != 0.0) atomic_add_f64(&gredg[row], gT[i]);
            if (gfT[i] != 0.0) atomic_add_f64(&gfullg[row], gfT[i]);
            if (hdT[i] != 0.0) atomic_add_f64(&hdg[row], hdT[i]);
        }
    }
}

// ---- K6: reduced solve, one workgroup per window -------------------------------------------------
// Packed lower-triangular S in LDS; right-looking Cholesky; forward/backward substitution.
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(DevPtrs P, int slot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = blockIdx.x;
    const WinDev W = P.win[w];
    const int tid = threadIdx.x;
    LmState* stp = P.states + (long long)w * P.state_stride + slot;
    IterAcc* acc = P.acc + (long long)w * P.state_stride + slot;
    __shared__ LmState st;
    __shared__ int s_fail;
    __shared__ double s_red[8];
    if (tid == 0) { st = *stp; s_fail = 0; }
    __syncthreads();
    const int Np = W.Np;
    double* Sg = P.S + W.S_off;
    double* gredg = P.gred + W.red_off;
    double* gfullg = P.gfull + W.red_off;
    double* hdg = P.hdiag + W.red_off;
    const int tri_n = Np * (Np + 1) / 2;
    double* A = (double*)smem;   // packed lower
    double* y = A + tri_n;       // rhs / solution
    double* gf = y + Np;         // full gradient
    double* hd = gf + Np;        // diag(H)
    double* lam = hd + Np;       // LM diagonal
    if (st.done) {
        // keep the accumulators clean for whoever runs next
        return;
    }
    const int cur = st.cur;
    // load + clear global accumulators
    for (int idx = tid; idx < Np * Np; idx += blockDim.x) {
        int row = idx / Np, col = idx - row * Np;
        if (col <= row) {
            A[tri(row, col)] = Sg[idx];
        }
        Sg[idx] = 0.0;
    }
    for (int i = tid; i < Np; i += blockDim.x) {
        y[i] = gredg[i]; gf[i] = gfullg[i]; hd[i] = hdg[i];
        gredg[i] = 0.0; gfullg[i] = 0.0; hdg[i] = 0.0;
    }
    __syncthreads();
    // pose-only factors at x: PosePriordx (K4). One thread per prior; LDS atomics.
    double cost_part = 0.0, fixed_part = 0.0;
    const double* xp = P.xp + (long long)cur * P.xp_stride;
    for (int k = W.prior_begin + tid; k < W.prior_end; k += blockDim.x) {
        const PriorDev pr = P.priors[k];
        int fi = P.kf_fidx[pr.kf];
        double d6[6], r[6], J[36];
#pragma unroll
        for (int i = 0; i < 6; i++) d6[i] = xp[6 * (long long)pr.kf + i];
        pose_prior_factor(P.kf_T0 + 12 * (long long)pr.kf, pr.T_prior, pr.inf, d6, r, J);
        double c = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) c += r[i] * r[i];
        if (fi < 0) { fixed_part += c; continue; }
        cost_part += c;
        int base = fi * W.dpf;
        for (int a = 0; a < 6; a++) {
            double g = 0, h = 0;
            for (int q = 0; q < 6; q++) { g += J[6 * q + a] * r[q]; h += J[6 * q + a] * J[6 * q + a]; }
            atomic_add_f64(&y[base + a], g);
            atomic_add_f64(&gf[base + a], g);
            atomic_add_f64(&hd[base + a], h);
            for (int b = 0; b <= a; b++) {
                double hh = 0;
                for (int q = 0; q < 6; q++) hh += J[6 * q + a] * J[6 * q + b];
                atomic_add_f64(&A[tri(base + a, base + b)], hh);
            }
        }
    }
    if (cost_part != 0.0) atomic_add_f64(&acc->lin_cost, cost_part);
    if (slot == 0 && fixed_part != 0.0) atomic_add_f64(&acc->fixed_cost, fixed_part);
    __syncthreads();
    // gradient tolerance (TrustRegionMinimizer::GradientToleranceReached) on the gradient at x
    double gm = 0.0;
    for (int i = tid; i < Np; i += blockDim.x) gm = fmax(gm, fabs(gf[i]));
    gm = wave_max(gm);
    if ((tid & 63) == 0) s_red[tid >> 6] = gm;
    __syncthreads();
    if (tid == 0) {
        double g = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) g = fmax(g, s_red[k]);
        g = fmax(g, __longlong_as_double((long long)acc->gmax_bits));
        acc->gmax_bits = (unsigned long long)__double_as_longlong(g);
        if (g <= P.o.gradient_tolerance) {
            st.done = 1; st.termination = 3;
            st.x_cost = 0.5 * acc->lin_cost;
            if (st.iter == 0) st.initial_cost = st.x_cost;
            *stp = st;
        }
    }
    __syncthreads();
    if (st.done) return;
    // Jacobi scaling (iteration 0) and LM diagonal
    double* sp = P.s_pose + W.red_off;
    for (int i = tid; i < Np; i += blockDim.x) {
        double s;
        if (slot == 0) { s = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(hd[i])) : 1.0; sp[i] = s; }
        else s = sp[i];
        double s2 = s * s;
        double l = fmin(fmax(s2 * hd[i], P.o.min_lm_diagonal), P.o.max_lm_diagonal) / st.radius / s2;
        lam[i] = l;
        A[tri(i, i)] += l;
    }
    __syncthreads();
    // right-looking Cholesky (packed lower, in place)
    for (int k = 0; k < Np; k++) {
        double d = A[tri(k, k)];
        if (!(d > 0.0) || !isfinite(d)) {
            if (tid == 0) s_fail = 1;
            __syncthreads();
            break;
        }
        double inv = 1.0 / sqrt(d);
        __syncthreads();
        for (int i = k + tid; i < Np; i += blockDim.x) A[tri(i, k)] *= inv;  // i == k gives sqrt(d)
        __syncthreads();
        int rem = Np - k - 1;
        // trailing update: thread per (i, j), k < j <= i
        int cnt = rem * (rem + 1) / 2;
        for (int e = tid; e < cnt; e += blockDim.x) {
            int ii = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while (tri(ii + 1, 0) <= e) ii++;
            while (tri(ii, 0) > e) ii--;
            int jj = e - tri(ii, 0);
            int i = k + 1 + ii, j = k + 1 + jj;
            A[tri(i, j)] -= A[tri(i, k)] * A[tri(j, k)];
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) acc->chol_fail = 1;
        return;
    }
    // forward / backward substitution (thread 0..: column-oriented, Np syncs each)
    for (int k = 0; k < Np; k++) {
        if (tid == 0) y[k] = y[k] / A[tri(k, k)];
        __syncthreads();
        double yk = y[k];
        for (int i = k + 1 + tid; i < Np; i += blockDim.x) y[i] -= A[tri(i, k)] * yk;
        __syncthreads();
    }
    for (int k = Np - 1; k >= 0; k--) {
        if (tid == 0) y[k] = y[k] / A[tri(k, k)];
        __syncthreads();
        double yk = y[k];
        for (int i = tid; i < k; i += blockDim.x) y[i] -= A[tri(k, i)] * yk;
        __syncthreads();
    }
    // delta = -y ; candidate poses ; norms ; pose-only model cost
    double* dl = P.delta + W.red_off;
    double sn = 0.0, cn = 0.0;
    bool bad = false;
    for (int i = tid; i < Np; i += blockDim.x) {
        double d = -y[i];
        dl[i] = d;
        y[i] = d;
        sn += d * d;
        if (!isfinite(d)) bad = true;
    }
    __syncthreads();
    double* xpc = P.xp + (long long)(1 - cur) * P.xp_stride;
    for (int k = tid; k < W.n_kf; k += blockDim.x) {
        int g = W.kf_base + k;
        int fi = P.kf_fidx[g];
        for (int i = 0; i < 6; i++) {
            double v = xp[6 * (long long)g + i] + (fi < 0 ? 0.0 : y[fi * W.dpf + i]);
            xpc[6 * (long long)g + i] = v;
            if (fi >= 0) cn += v * v;
        }
        if (W.dpf == 15) {
            double* xs[3] = {P.xv, P.xba, P.xbg};
            for (int q = 0; q < 3; q++) {
                const double* src = xs[q] + (long long)cur * P.xv_stride;
                double* dst = xs[q] + (long long)(1 - cur) * P.xv_stride;
                for (int i = 0; i < 3; i++) {
                    double v = src[3 * (long long)g + i] + (fi < 0 ? 0.0 : y[fi * 15 + 6 + 3 * q + i]);
                    dst[3 * (long long)g + i] = v;
                    if (fi >= 0) cn += v * v;
                }
            }
        }
    }
    // priors: model cost change and candidate cost
    double mcc = 0.0, cc = 0.0;
    for (int k = W.prior_begin + tid; k < W.prior_end; k += blockDim.x) {
        const PriorDev pr = P.priors[k];
        int fi = P.kf_fidx[pr.kf];
        if (fi < 0) continue;
        double d6[6], r[6], J[36], rc[6];
        for (int i = 0; i < 6; i++) d6[i] = xp[6 * (long long)pr.kf + i];
        pose_prior_factor(P.kf_T0 + 12 * (long long)pr.kf, pr.T_prior, pr.inf, d6, r, J);
        for (int q = 0; q < 6; q++) {
            double m = 0;
            for (int a = 0; a < 6; a++) m += J[6 * q + a] * y[fi * W.dpf + a];
            mcc += -m * (r[q] + 0.5 * m);
        }
        for (int i = 0; i < 6; i++) d6[i] += y[fi * W.dpf + i];
        pose_prior_factor(P.kf_T0 + 12 * (long long)pr.kf, pr.T_prior, pr.inf, d6, rc, nullptr);
        for (int q = 0; q < 6; q++) cc += rc[q] * rc[q];
    }
    sn = wave_sum(sn); cn = wave_sum(cn); mcc = wave_sum(mcc); cc = wave_sum(cc);
    if ((tid & 63) == 0) {
        if (sn != 0.0) atomic_add_f64(&acc->step_norm2, sn);
        if (cn != 0.0) atomic_add_f64(&acc->cand_norm2, cn);
        if (mcc != 0.0) atomic_add_f64(&acc->mcc, mcc);
        if (cc != 0.0) atomic_add_f64(&acc->cand_cost, cc);
    }
    if (bad) acc->chol_fail = 1;
}

// ---- K7: back-substitution + candidate cost -------------------------------------------------------
template <int FACTOR>
__global__ __launch_bounds__(BUILD_THREADS) void k_backsub(DevPtrs P, int slot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Tile T = P.tiles[blockIdx.x];
    const WinDev W = P.win[T.w];
    const int tid = threadIdx.x;
    const LmState st = P.states[(long long)T.w * P.state_stride + slot];
    IterAcc* acc = P.acc + (long long)T.w * P.state_stride + slot;
    if (st.done || acc->chol_fail) return;
    TileLds L;
    size_t off = tile_lds_carve(smem, W.n_kf, L);
    double* candTab = (double*)(smem + off);  // [n_kf][12] R|t at the candidate poses
    double* dlStage = candTab + W.n_kf * 12;  // [MAX_TILE_LMK][3]
    const int cur = st.cur;
    double cost_part, fixed_part, gmax_part;
    tile_prologue<FACTOR>(P, W, T, L, cur, st.radius, false, cost_part, fixed_part, gmax_part);
    const double* xpc = P.xp + (long long)(1 - cur) * P.xp_stride;
    for (int k = tid; k < W.n_kf; k += blockDim.x) {
        int g = W.kf_base + k;
        double d6[6], tab[POSE_TAB];
        for (int i = 0; i < 6; i++) d6[i] = xpc[6 * (long long)g + i];
        pose_table_entry(P.kf_T0 + 12 * (long long)g, d6, tab);
        for (int i = 0; i < POSE_TAB; i++) (void)0;
        for (int i = 0; i < 12; i++) candTab[k * 12 + i] = tab[i];
    }
    const double* dp = P.delta + W.red_off;
    const double* xl = P.xl + (long long)cur * P.xl_stride;
    double* xlc = P.xl + (long long)(1 - cur) * P.xl_stride;
    const int nl = T.lmk1 - T.lmk0, no = T.obs1 - T.obs0;
    // predicted residual e_a = r_a + Jp_a dp_a  -> stored over r in the stage (keep r in [18..19], e in regs)
    // per landmark: dl = -Minv sum_a Jl_a^T e_a
    double sn = 0.0, cn = 0.0;
    for (int l = tid; l < nl; l += blockDim.x) {
        int gl = T.lmk0 + l;
        int o0 = P.lmk_ob[gl] - T.obs0, o1 = P.lmk_oe[gl] - T.obs0;
        const double* Mi = L.lmkStage + l * LMK_STAGE;
        double t[3] = {0, 0, 0};
        for (int a = o0; a < o1; a++) {
            const double* s = L.obsStage + a * OBS_STAGE;
            int pa = L.obsPa[a];
            double e0 = s[18], e1 = s[19];
            if (pa >= 0) {
                const double* d = dp + (pa / 6) * W.dpf;
#pragma unroll
                for (int i = 0; i < 6; i++) { e0 += s[i] * d[i]; e1 += s[6 + i] * d[i]; }
            }
            t[0] += s[12] * e0 + s[15] * e1;
            t[1] += s[13] * e0 + s[16] * e1;
            t[2] += s[14] * e0 + s[17] * e1;
        }
        double d0 = -(Mi[0] * t[0] + Mi[1] * t[1] + Mi[2] * t[2]);
        double d1 = -(Mi[1] * t[0] + Mi[3] * t[1] + Mi[4] * t[2]);
        double d2 = -(Mi[2] * t[0] + Mi[4] * t[1] + Mi[5] * t[2]);
        dlStage[3 * l] = d0; dlStage[3 * l + 1] = d1; dlStage[3 * l + 2] = d2;
        double c0 = xl[3 * (long long)gl] + d0, c1 = xl[3 * (long long)gl + 1] + d1, c2 = xl[3 * (long long)gl + 2] + d2;
        xlc[3 * (long long)gl] = c0; xlc[3 * (long long)gl + 1] = c1; xlc[3 * (long long)gl + 2] = c2;
        bool active = (o1 > o0) && !(P.lmk_const && P.lmk_const[gl]);
        if (active) {
            sn += d0 * d0 + d1 * d1 + d2 * d2;
            cn += c0 * c0 + c1 * c1 + c2 * c2;
        }
    }
    __syncthreads();
    // per observation: model cost and candidate residual
    double mcc = 0.0, cc = 0.0;
    for (int a = tid; a < no; a += blockDim.x) {
        const double* s = L.obsStage + a * OBS_STAGE;
        int pa = L.obsPa[a];
        int l = L.obsLmk[a];
        int gl = T.lmk0 + l;
        bool lfree = !(P.lmk_const && P.lmk_const[gl]);
        if (pa < 0 && !lfree) continue;
        const double* dlv = dlStage + 3 * l;
        double m0 = s[12] * dlv[0] + s[13] * dlv[1] + s[14] * dlv[2];
        double m1 = s[15] * dlv[0] + s[16] * dlv[1] + s[17] * dlv[2];
        if (pa >= 0) {
            const double* d = dp + (pa / 6) * W.dpf;
#pragma unroll
            for (int i = 0; i < 6; i++) { m0 += s[i] * d[i]; m1 += s[6 + i] * d[i]; }
        }
        mcc += -m0 * (s[18] + 0.5 * m0) - m1 * (s[19] + 0.5 * m1);
        // candidate residual
        int o = T.obs0 + a;
        int kf = P.obs_kf[o], cam = P.obs_cam[o];
        double pw[3] = {P.lmk_p[3 * (long long)gl] + xlc[3 * (long long)gl], P.lmk_p[3 * (long long)gl + 1] + xlc[3 * (long long)gl + 1],
                        P.lmk_p[3 * (long long)gl + 2] + xlc[3 * (long long)gl + 2]};
        double r[2];
        const double* ctab = candTab + (kf - W.kf_base) * 12;
        if (FACTOR == 0) {
            const double* m = P.obs_meas + 2 * (long long)o;
            pixel_factor<false>(ctab, P.cam_K + 4 * cam, P.cam_T + 12 * cam, pw, m[0], m[1], P.cam_isig[cam], r, nullptr, nullptr);
        } else {
            const double* m = P.obs_meas + 3 * (long long)o;
            double b[3] = {m[0], m[1], m[2]};
            // angular_factor<false> only reads tab[0..11]
            angular_factor<false>(ctab, P.cam_T + 12 * cam, pw, b, P.cam_isig[cam], r, nullptr, nullptr);
        }
        cc += r[0] * r[0] + r[1] * r[1];
    }
    sn = wave_sum(sn); cn = wave_sum(cn); mcc = wave_sum(mcc); cc = wave_sum(cc);
    if ((tid & 63) == 0) {
        if (sn != 0.0) atomic_add_f64(&acc->step_norm2, sn);
        if (cn != 0.0) atomic_add_f64(&acc->cand_norm2, cn);
        if (mcc != 0.0) atomic_add_f64(&acc->mcc, mcc);
        if (cc != 0.0) atomic_add_f64(&acc->cand_cost, cc);
    }
}

// Last decision of the solve: state[slots] = decide(state[slots-1], acc[slots-1]).
__global__ void k_final(DevPtrs P, int slots) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= P.n_win) return;
    LmState s = lm_decide(P.states[(long long)w * P.state_stride + slots - 1], P.acc[(long long)w * P.state_stride + slots - 1], P.o);
    P.states[(long long)w * P.state_stride + slots] = s;
}

// Parity probe: per-observation residual / Jacobians at deltas held in buffer 0.
template <int FACTOR>
__global__ void k_linearize_probe(DevPtrs P, int w, double* r2, double* Jp12, double* Jl6) {
    const WinDev W = P.win[w];
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= W.n_obs) return;
    int o = W.obs_base + a;
    // find the landmark of this observation: binary search over lmk_ob
    int lo = W.lmk_base, hi = W.lmk_base + W.n_lmk - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (P.lmk_ob[mid] <= o) lo = mid; else hi = mid - 1;
    }
    // skip landmarks with empty ranges that share the same begin
    while (!(P.lmk_ob[lo] <= o && o < P.lmk_oe[lo]) && lo > W.lmk_base) lo--;
    int gl = lo;
    int kf = P.obs_kf[o], cam = P.obs_cam[o];
    double d6[6], tab[POSE_TAB];
    for (int i = 0; i < 6; i++) d6[i] = P.xp[6 * (long long)kf + i];
    pose_table_entry(P.kf_T0 + 12 * (long long)kf, d6, tab);
    double pw[3] = {P.lmk_p[3 * (long long)gl] + P.xl[3 * (long long)gl], P.lmk_p[3 * (long long)gl + 1] + P.xl[3 * (long long)gl + 1],
                    P.lmk_p[3 * (long long)gl + 2] + P.xl[3 * (long long)gl + 2]};
    double r[2], Jp[12], Jl[6];
    if (FACTOR == 0) {
        const double* m = P.obs_meas + 2 * (long long)o;
        pixel_factor<true>(tab, P.cam_K + 4 * cam, P.cam_T + 12 * cam, pw, m[0], m[1], P.cam_isig[cam], r, Jp, Jl);
    } else {
        const double* m = P.obs_meas + 3 * (long long)o;
        double b[3] = {m[0], m[1], m[2]};
        angular_factor<true>(tab, P.cam_T + 12 * cam, pw, b, P.cam_isig[cam], r, Jp, Jl);
    }
    r2[2 * (long long)a] = r[0]; r2[2 * (long long)a + 1] = r[1];
    for (int i = 0; i < 12; i++) Jp12[12 * (long long)a + i] = Jp[i];
    for (int i = 0; i < 6; i++) Jl6[6 * (long long)a + i] = Jl[i];
}

}  // namespace sadvio
