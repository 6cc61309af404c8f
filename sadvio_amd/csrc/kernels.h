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
#include "chol16.h"

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
    const unsigned char* obs_slot;  // index of the observation's key-frame in its tile's list
    const int* tile_kf;             // per-tile key-frame lists (global indices)
    const int* tile_row;            // matching row in the tile's LDS system (6 * rank among free) or -1
    // first-round packets (k_pre_packets; null for submissions of more than PRE_MAX_TILES tiles): per (tile, 8 consecutive lanes — a landmark's
    // lane group is 8 .. 64 wide) | landmark | its first observation | its observation count + (1 << 16 if the landmark exists) | position of the
    // 8 lanes' first one inside the group |,
    // per (tile, k < PRE_KF) | global key-frame or -1 | its free index |: the opening loads of k_build / k_backsub hang on blockIdx alone
    const int4* pre_lane;
    const int2* pre_kf;
    double* ptab;                   // [2][n_kf_tot][POSE_TAB] pose tables of the two delta buffers
    long long ptab_stride;
    const PriorDev* priors;
    double* prior_lin;              // [2][n_prior_tot][PRIOR_LIN] g = J^T r (6) | H = J^T J lower (21) | |r|^2 of every pose prior at the x of
    long long prior_lin_stride;     // each delta buffer: written at x = 0 by k_init_tables, at the candidate by k_solve's back half
    int n_prior_tot;
    const ImuDev* imus;
    double* imu_scratch;  // [2][n_imu_tot][IMU_ROW], see ba_types.h: the linearisation of every IMU factor pair at the deltas of buffer 0 | 1
    long long imu_scratch_stride;
    double* S; double* gred; double* gfull; double* hdiag; double* delta; double* s_pose;
    LmState* states;  // [n_win][slots+2]
    IterAcc* acc;     // [n_win][slots+1] window totals (written by single workgroups only)
    TileAcc* tacc;    // [2][n_tiles] per-tile partials, double-buffered by slot parity (no atomics)
    FinalRec* final_out;  // [n_win]
    int* big_info;        // [n_win] potrf info of the windows solved out of LDS
    // a window sharded over `world` GPUs (landmark partition): per-rank partial sums, exchanged with the reduced
    // system by one all-reduce per phase; rank r writes slot r and zeroes the others (sum == gather)
    int imu_direct;       // 1: imu_pair_lin_wg adds the IMU pairs' entries to the reduced system itself (one device); 0: k_solve's item loop
    int decide_kernel;    // 1: the LM decision of a slot is taken by k_decide (many tiles); 0: by every k_build workgroup
    int world, rank;
    double* rank_b;       // [n_win][world][4] lin_cost, fixed_cost, gmax, -      (after k_build)
    double* rank_s;       // [n_win][world][4] cand_cost, mcc, step_norm2, cand_norm2 (after k_backsub)
    const int* lmk_red;   // [n_lmk_tot] offset of a prior-kept landmark in its window's reduced vector, else -1 (may be null)
    const int* kept_obs;  // [n_kept][3] (device observation index, global landmark, window)
    int n_kept;
    double* dp_data;      // dense priors, see WinDev::dp_off
    const SparseDev* sparse;  // sparse prior factors
    const int* sp_list;       // indices (into sparse) of the factors the solve evaluates itself, per window slice
    double* sp_scratch;       // [2][n_sparse][SPARSE_J] r, J of the listed factors at the deltas of buffer 0 | 1 (as imu_scratch)
    long long sp_scratch_stride;
    int n_imu_tot, n_sp_list;  // extra workgroups of k_build / k_backsub: IMU factor pairs, then the listed sparse-prior factors
    const int* dp_ints;
    const int* chunk_ob;      // [n_chunks + 1] first observation of each chunk (lm_kernels.h)
    const int* chunk_lm;      // [n_chunks + 1] first landmark of each chunk
    const int* tile_perm;     // [n_tiles] launch order of the throughput kernels: tiles by decreasing chunk count
    const unsigned char* obs_lslot;  // [n_obs_tot] index of the observation's landmark inside its chunk
    double* lm_hg;            // [2][n_lmk_tot][9] H_ll | g_l of every landmark at the point of each delta buffer (k_lm_pass)
    double* lm_dt;            // [2][n_tiles][LM_DT] per tile: key-frame sums (sum Jp^T Jp, sum Jp^T r) + cost totals at each buffer's point
    long long lm_hg_stride, lm_dt_stride;
    const int* lm_sub;        // [n_sub][2] work list of k_lm_pass: tile | sub-block (LM_PASS_THREADS landmarks) of the tile, largest tiles first
    int lm_sub_per_item;      // sub-blocks one work item (workgroup) of k_lm_pass runs through
    int lm_ksub;              // record slots per tile (= the largest tile's sub-block count): lm_dt / lm_sacc are indexed tile * lm_ksub + sub-block
    double* lm_sacc;          // [2][n_tiles * lm_ksub][4] per sub-block and slot parity: cand_cost | mcc | step_norm2 | cand_norm2 (unused slots stay zero)
    const LineDev* lines;     // linexd landmarks: few, kept in the reduced system
    const LineObsDev* lobs;
    double* xline;            // [2][n_line_tot][6] line deltas, double-buffered like xp
    double* line_scratch;     // [n_lobs_tot][LINE_ROW]
    long long xline_stride;
    long long n_xp, n_xv, n_xl;  // doubles in the (double-buffered) delta arrays, zeroed by k_reset
    int n_tiles;
    int state_stride;
    int n_win;
    long long* t_start; // wall_clock64 at the first kernel of the solve (k_reset)
    double* trace;      // [n_win][state_stride][8] per-iteration log (sadvio_ba_get_trace), written by the slot's single decider
    long long* dbg_ts;  // [128] phase timestamps (wall_clock64, 100 MHz) of workgroup 0 when debug & 4096
    int debug;  // SADVIO_DEBUG env: bit 12 (4096) = in-kernel phase timestamps of workgroup 0 into dbg_ts (results unaffected)
    SolveOpts o;
};

// In-kernel phase timestamps (SADVIO_DEBUG & 4096) exist only in a build with -DSADVIO_KERNEL_TS: even as not-taken
// branches they cost the latency-bound kernels ~1-2 us per launch (measured A/B on k_solve).
#ifdef SADVIO_KERNEL_TS
#define SADVIO_TS(slot_, idx_) do { if ((P.debug & 4096) && blockIdx.x == 0 && threadIdx.x == 0 && slot == (slot_)) P.dbg_ts[idx_] = wall_clock64(); } while (0)
#define SADVIO_TS_PTR(cond_) ((cond_) ? P.dbg_ts : nullptr)
#else
#define SADVIO_TS(slot_, idx_) do { } while (0)
#define SADVIO_TS_PTR(cond_) nullptr
#endif

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j
// element (i >= j) of a window's reduced matrix in HBM: packed lower triangle (ld == 0) or full row-major
// ld == 0: the 16 x 16 tile-packed lower triangle of chol16.h (= the LDS image of k_solve<0>, so that its gather is a linear copy)
__device__ __forceinline__ long long s_index(int ld, int i, int j) { return ld ? (long long)i * ld + j : (long long)c16_index(i, j); }

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
    if (s.iter >= o.max_num_iterations) { s.done = 1; s.termination = 0; }   // NO_CONVERGENCE (the solver-time limit is checked by k_solve)
    else if (s.radius <= o.min_radius) { s.done = 1; s.termination = 4; }
    return s;
}

// Iteration log of a window (the IterationSummary fields Ceres reports, in the layout of the oracle's log): row i = the
// state after i step attempts: [cost, cost_change, radius, step_norm, relative_decrease, successful, max |gradient| at
// that state (-1 where it was not linearised: after the last attempt), model_cost_change]. Called by the one thread that
// publishes the decision of `slot` (prev = state before the attempt, a = its totals, next = the decision).
__device__ __forceinline__ void trace_write(const DevPtrs& P, int w, int slot, const LmState& prev, const IterAcc& a, const LmState& next) {
    if (!P.trace || prev.done) return;
    double* R = P.trace + ((long long)w * P.state_stride + slot) * 8;
    if (slot == 0) { R[0] = 0.5 * a.lin_cost; R[1] = 0.0; R[2] = prev.radius; R[3] = 0.0; R[4] = 0.0; R[5] = 1.0; R[7] = 0.0; }
    R[6] = __longlong_as_double((long long)a.gmax_bits);
    double* N = R + 8;
    const bool valid = a.chol_fail == 0 && a.mcc > 0.0;
    const double cc = valid ? 0.5 * (a.lin_cost - a.cand_cost) : 0.0;
    const bool judged = valid && !(next.done && (next.termination == 1 || next.termination == 2));
    N[0] = next.x_cost; N[1] = cc; N[2] = next.radius; N[3] = valid ? sqrt(a.step_norm2) : 0.0;
    N[4] = judged ? cc / a.mcc : 0.0; N[5] = next.n_success > prev.n_success ? 1.0 : 0.0; N[6] = -1.0; N[7] = a.mcc;
}

// Wave-wide sum / max with DPP + the gfx950 v_permlane{16,32}_swap (plain VALU): every lane receives the result. The
// __shfl_down forms go through ds_bpermute, ~100 cycles per step on the LDS pipe: six steps per value were ~1 us of every
// kernel tail that reduces a handful of partials.
template <int CTRL>
__device__ __forceinline__ double dpp_f64_(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// v_permlane16_swap / v_permlane32_swap of a register with itself return (own value, value of lane ^ 16 / ^ 32) in an order
// that depends on the half: symmetric combinations (sum, max) need not know which is which
__device__ __forceinline__ void swap16_pair_f64(double v, double& x0, double& x1) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x0 = __hiloint2double(b[0], a[0]); x1 = __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ void swap32_pair_f64(double v, double& x0, double& x1) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x0 = __hiloint2double(b[0], a[0]); x1 = __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64_<0xB1>(v);   // xor 1
    v += dpp_f64_<0x4E>(v);   // xor 2
    v += dpp_f64_<0x141>(v);  // row_half_mirror
    v += dpp_f64_<0x140>(v);  // row_mirror
    double x0, x1;
    swap16_pair_f64(v, x0, x1); v = x0 + x1;
    swap32_pair_f64(v, x0, x1); v = x0 + x1;
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64_<0xB1>(v));
    v = fmax(v, dpp_f64_<0x4E>(v));
    v = fmax(v, dpp_f64_<0x141>(v));
    v = fmax(v, dpp_f64_<0x140>(v));
    double x0, x1;
    swap16_pair_f64(v, x0, x1); v = fmax(x0, x1);
    swap32_pair_f64(v, x0, x1); v = fmax(x0, x1);
    return v;
}


// Sum the k_backsub partials of a window's tiles (slot parity `par`): executed by one wave. Every kernel that re-derives
// the LM decision of a slot must get the SAME BITS, so the order is canonical for windows of at most 4 * BUILD_THREADS tiles
// (the only ones whose decision is taken by more than one kernel): tile t belongs to virtual thread t % 256, which adds its
// (at most four) tiles pairwise; the 64 virtual threads of a virtual wave are summed with wave_sum; the four virtual waves
// pairwise. k_build does exactly this with its 256 real threads (all loads in flight at once); a single wave emulates it.
__device__ __forceinline__ void wave_sum_backsub_partials(const DevPtrs& P, int par, int w, int tile0, int ntiles, int ln,
                                                          double* out4) {
    double c = 0.0, m = 0.0, sn = 0.0, cn = 0.0;
    if (P.world > 1) {
        const double* rs = P.rank_s + (long long)w * P.world * 4;
        for (int r = ln; r < P.world; r += 64) { c += rs[4 * r]; m += rs[4 * r + 1]; sn += rs[4 * r + 2]; cn += rs[4 * r + 3]; }
        c = wave_sum(c); m = wave_sum(m); sn = wave_sum(sn); cn = wave_sum(cn);
    } else if (ntiles <= 4 * BUILD_THREADS) {
        const TileAcc* ta = P.tacc + (long long)par * P.n_tiles + tile0;
        double W[4][4];
#pragma unroll
        for (int vw = 0; vw < 4; vw++) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = vw * 64 + ln + u * BUILD_THREADS;
                const bool in = t < ntiles;
                const TileAcc* e = ta + (in ? t : 0);
                v[u][0] = in ? e->cand_cost : 0.0; v[u][1] = in ? e->mcc : 0.0; v[u][2] = in ? e->step_norm2 : 0.0; v[u][3] = in ? e->cand_norm2 : 0.0;
            }
#pragma unroll
            for (int f = 0; f < 4; f++) W[vw][f] = wave_sum((v[0][f] + v[1][f]) + (v[2][f] + v[3][f]));
        }
        c = (W[0][0] + W[1][0]) + (W[2][0] + W[3][0]); m = (W[0][1] + W[1][1]) + (W[2][1] + W[3][1]);
        sn = (W[0][2] + W[1][2]) + (W[2][2] + W[3][2]); cn = (W[0][3] + W[1][3]) + (W[2][3] + W[3][3]);
    } else {
        const TileAcc* ta = P.tacc + (long long)par * P.n_tiles + tile0;
        for (int t = ln; t < ntiles; t += 64) {
            c += ta[t].cand_cost; m += ta[t].mcc; sn += ta[t].step_norm2; cn += ta[t].cand_norm2;
        }
        c = wave_sum(c); m = wave_sum(m); sn = wave_sum(sn); cn = wave_sum(cn);
    }
    if (ln == 0) { out4[0] = c; out4[1] = m; out4[2] = sn; out4[3] = cn; }
}

// ---- landmark-group machinery shared by k_build and k_backsub ------------------------------------
// ---- cross-lane primitives without the LDS pipe -------------------------------------------------------
// ds_bpermute / ds_swizzle shuffles and ds_add_f64 cost ~100-170 cycles per wave-instruction here (measured),
// DPP modifiers and the gfx950 v_permlane{16,32}_swap are plain VALU. dpp_ctrl: quad_perm 0x00-0xFF,
// row_shl:n 0x100+n, row_shr:n 0x110+n, row_ror:n 0x120+n, row_mirror 0x140, row_half_mirror 0x141.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// v[lane] + v[lane ^ 16] / v[lane ^ 32] in every lane
__device__ __forceinline__ double xor16_sum(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double xor32_sum(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ int xor16_other(int v) {  // the value held by lane ^ 16
    auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 16) ? a[0] : a[1]);
}
__device__ __forceinline__ int xor32_other(int v) {
    auto a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 32) ? a[0] : a[1]);
}

// Sum over the G lanes of a landmark group (G = power of two, 8 <= G <= 64, groups aligned to G lanes);
// every lane of the group receives the total.
__device__ __forceinline__ double group_sum(double v, int G) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]  (xor 2)
    v += dpp_f64<0x141>(v);  // row_half_mirror: the other quad of the 8-lane group
    if (G >= 16) v += dpp_f64<0x140>(v);  // row_mirror: the other half of the 16-lane row
    if (G >= 32) v = xor16_sum(v);
    if (G >= 64) v = xor32_sum(v);
    return v;
}
// Sum over the 64 / 8 groups of a wave, lane q of every group receives the total of lanes q (G = 8 only).
__device__ __forceinline__ double across_groups8_sum(double v) {
    v += dpp_f64<0x128>(v);  // row_ror:8 (xor 8)
    v = xor16_sum(v);
    return xor32_sum(v);
}

// ceres::HuberLoss(a) + Corrector (rho'' <= 0 branch): returns rho(|r|^2) and the scale sqrt(rho') for r and J.
__device__ __forceinline__ double huber_rho(double a, double s, double& scale) {
    scale = 1.0;
    if (a > 0.0 && s > a * a) {
        const double rr = sqrt(s);
        scale = sqrt(fmax(2.2250738585072014e-308, a / rr));
        return 2.0 * a * rr - a * a;
    }
    return s;
}

struct ObsLin {
    double r[2], Jp[12], Jl[6];
    double rho;   // loss-corrected cost of the block (= |r|^2 without a loss function)
    int row;      // row of the observing key-frame in the tile's LDS system, -1 if constant
    int slot;     // index into the tile's key-frame list
    bool valid;   // this lane owns an observation
    bool counted; // the residual block belongs to the reduced program (some parameter block is free)
};

// Lane-local linearisation of observation `o` of landmark `gl` from LDS tables.
// RARE = false compiles the robust loss out (plain window BA: huber_a = 0, no prior-kept landmarks).
// The constants of one observation, loaded ahead of use (k_build fetches its first landmark round while the LM decision of
// the previous slot is still being summed).
struct ObsPre {
    int slot, craw;
    double m[3];
};
template <int FACTOR>
__device__ __forceinline__ ObsPre obs_prefetch(const DevPtrs& P, int o) {
    ObsPre q;
    q.slot = P.obs_slot[o];
    q.craw = P.obs_cam[o];
    if (FACTOR == 0) { const double* m = P.obs_meas + 2 * (long long)o; q.m[0] = m[0]; q.m[1] = m[1]; q.m[2] = 0.0; }
    else { const double* m = P.obs_meas + 3 * (long long)o; q.m[0] = m[0]; q.m[1] = m[1]; q.m[2] = m[2]; }
    return q;
}

template <int FACTOR, bool RARE>
__device__ __forceinline__ void lane_linearize(const DevPtrs& P, const double* poseTab, const double* camTab,
                                               const int* rowTab, int cam_base, const ObsPre& ob, const double* pw, bool keep_jl,
                                               bool lcounted, ObsLin& L) {
    const int slot = ob.slot;
    const int craw = ob.craw;
    const int cam = craw < 0 ? 0 : craw - cam_base;
    const double* tab = poseTab + slot * POSE_TAB;
    const double* ct = camTab + cam * 17;  // K[4] Tsf[12] isig
    L.slot = slot;
    L.row = rowTab[slot];
    if (RARE && craw < 0) {
        // pseudo-observation: one half of a PoseToLandmarkFactor of the sparsified prior (no loss function on it)
        const int code = -1 - craw;
        const SparseDev& f = P.sparse[code >> 1];
        p2l_pseudo_obs<true>(tab, pw, f.delta, f.W, code & 1, L.r, L.Jp, L.Jl);
        L.rho = L.r[0] * L.r[0] + L.r[1] * L.r[1];
        if (L.row < 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) L.Jp[i] = 0.0;
        }
        if (!keep_jl) {
#pragma unroll
            for (int i = 0; i < 6; i++) L.Jl[i] = 0.0;
        }
        L.counted = (L.row >= 0) || lcounted;
        return;
    }
    if (FACTOR == 0) {
        pixel_factor<true>(tab, ct, ct + 4, pw, ob.m[0], ob.m[1], ct[16], L.r, L.Jp, L.Jl);
    } else {
        double b[3] = {ob.m[0], ob.m[1], ob.m[2]};
        angular_factor<true>(tab, ct + 4, pw, b, ct[16], L.r, L.Jp, L.Jl);
    }
    L.rho = L.r[0] * L.r[0] + L.r[1] * L.r[1];
    if (RARE) {
        double sc;
        L.rho = huber_rho(P.o.huber_a, L.rho, sc);
        if (sc != 1.0) {
            L.r[0] *= sc; L.r[1] *= sc;
#pragma unroll
            for (int i = 0; i < 12; i++) L.Jp[i] *= sc;
#pragma unroll
            for (int i = 0; i < 6; i++) L.Jl[i] *= sc;
        }
    }
    if (L.row < 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) L.Jp[i] = 0.0;
    }
    if (!keep_jl) {
#pragma unroll
        for (int i = 0; i < 6; i++) L.Jl[i] = 0.0;
    }
    L.counted = (L.row >= 0) || lcounted;
    if (!L.counted) { /* fixed cost: caller accounts r, then it leaves the program */ }
}

// Per-landmark elimination, every lane of the group redundantly: H_ll, g_l (group sums), LM damping, the inverse Cholesky
// factor Li of the damped block (sym3_chol_inverse: packed lower 6-vector; named Mi below). Returns whether the landmark has a
// parameter block in the program.
__device__ __forceinline__ bool group_eliminate(const DevPtrs& P, const ObsLin& L, int G, int gl, bool lmk_valid,
                                                bool lfree, int nobs, double radius, bool write_scale, bool leader,
                                                double* Mi, double* g) {
    double H[6];
    H[0] = L.Jl[0] * L.Jl[0] + L.Jl[3] * L.Jl[3];
    H[1] = L.Jl[0] * L.Jl[1] + L.Jl[3] * L.Jl[4];
    H[2] = L.Jl[0] * L.Jl[2] + L.Jl[3] * L.Jl[5];
    H[3] = L.Jl[1] * L.Jl[1] + L.Jl[4] * L.Jl[4];
    H[4] = L.Jl[1] * L.Jl[2] + L.Jl[4] * L.Jl[5];
    H[5] = L.Jl[2] * L.Jl[2] + L.Jl[5] * L.Jl[5];
    g[0] = L.Jl[0] * L.r[0] + L.Jl[3] * L.r[1];
    g[1] = L.Jl[1] * L.r[0] + L.Jl[4] * L.r[1];
    g[2] = L.Jl[2] * L.r[0] + L.Jl[5] * L.r[1];
#pragma unroll
    for (int i = 0; i < 6; i++) H[i] = group_sum(H[i], G);
#pragma unroll
    for (int i = 0; i < 3; i++) g[i] = group_sum(g[i], G);
    const bool active = lmk_valid && lfree && nobs > 0;
    if (!active) {
#pragma unroll
        for (int i = 0; i < 6; i++) Mi[i] = 0.0;
        return false;
    }
    double s[3];
    if (write_scale) {
        s[0] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[0])) : 1.0;
        s[1] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[3])) : 1.0;
        s[2] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[5])) : 1.0;
        if (leader) {
            P.s_lmk[3 * (long long)gl] = s[0]; P.s_lmk[3 * (long long)gl + 1] = s[1]; P.s_lmk[3 * (long long)gl + 2] = s[2];
        }
    } else {
        s[0] = P.s_lmk[3 * (long long)gl]; s[1] = P.s_lmk[3 * (long long)gl + 1]; s[2] = P.s_lmk[3 * (long long)gl + 2];
    }
    const double ir = 1.0 / radius;
    const double s0 = s[0] * s[0], s1 = s[1] * s[1], s2 = s[2] * s[2];
    double M[6] = {H[0], H[1], H[2], H[3], H[4], H[5]};
    M[0] += fmin(fmax(s0 * H[0], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s0;
    M[3] += fmin(fmax(s1 * H[3], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s1;
    M[5] += fmin(fmax(s2 * H[5], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s2;
    sym3_chol_inverse(M, Mi);
    return true;
}

// Stage the tile's key-frame pose tables (buffer `buf`) and the window's cameras into LDS.
__device__ __forceinline__ void stage_tables(const DevPtrs& P, const Tile& T, int buf, double* poseTab, double* camTab,
                                             int* rowTab) {
    const int tid = threadIdx.x;
    const double* src = P.ptab + (long long)buf * P.ptab_stride;
    for (int i = tid; i < T.n_kf * POSE_TAB; i += blockDim.x) {
        const int k = i / POSE_TAB, e = i - k * POSE_TAB;
        poseTab[i] = src[(long long)P.tile_kf[T.kf_off + k] * POSE_TAB + e];
    }
    for (int i = tid; i < T.n_cam * 17; i += blockDim.x) {
        const int c = i / 17, e = i - 17 * c;
        const int gc = T.cam_base + c;
        camTab[i] = e < 4 ? P.cam_K[4 * (long long)gc + e] : (e < 16 ? P.cam_T[12 * (long long)gc + e - 4] : P.cam_isig[gc]);
    }
    for (int i = tid; i < T.n_kf; i += blockDim.x) rowTab[i] = P.tile_row[T.kf_off + i];
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic between lanes of ONE wave: order this wave's stores before its later loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__host__ __device__ inline size_t tile_tables_bytes(int n_kf) {
    // pose tables + camera tables + row table; a global-atomics tile may list up to 64 key-frames (> MAX_TILE_KF)
    size_t b = sizeof(double) * ((size_t)n_kf * POSE_TAB + MAX_WIN_CAM * 17) + sizeof(int) * (size_t)(n_kf > MAX_TILE_KF ? n_kf : MAX_TILE_KF);
    return (b + 15) & ~(size_t)15;
}

// First-round packets (DevPtrs::pre_lane / pre_kf): one launch per layout, behind its upload. What a lane of k_build / k_backsub needs
// to address the inputs of its tile's FIRST landmark round, laid out by (tile, 8-lane granule) so that those loads hang on blockIdx alone
// (132 KB for the 258 tiles of config 2; a packet per lane was 1 MB of extra traffic per launch).
__global__ __launch_bounds__(BUILD_THREADS) void k_pre_packets(const Tile* __restrict__ tiles, const int* __restrict__ lmk_ob, const int* __restrict__ lmk_oe,
                                                               const int* __restrict__ tile_kf, const int* __restrict__ kf_fidx, int4* __restrict__ pre_lane,
                                                               int2* __restrict__ pre_kf) {
    const Tile T = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < BUILD_THREADS / 8) {          // one packet per 8 lanes
        const int lane0 = tid * 8, wv = lane0 >> 6, ln = lane0 & 63;
        const int G = T.G, lpw = 64 / G, grp = ln / G, q0 = ln - grp * G, nl = T.lmk1 - T.lmk0;
        const bool valid = wv * lpw + grp < nl;
        const int gl = T.lmk0 + (valid ? wv * lpw + grp : 0);
        const int ob = nl > 0 ? lmk_ob[gl] : 0, nobs = (valid && nl > 0) ? lmk_oe[gl] - ob : 0;
        pre_lane[(long long)blockIdx.x * (BUILD_THREADS / 8) + tid] = make_int4(gl, ob, nobs | (valid ? 1 << 16 : 0), q0);
    }
    if (tid < PRE_KF) {
        int2 e = make_int2(-1, -1);
        if (tid < T.n_kf) { e.x = tile_kf[T.kf_off + tid]; e.y = kf_fidx[e.x]; }
        pre_kf[(long long)blockIdx.x * PRE_KF + tid] = e;
    }
}

// ---- K5: build the reduced system ----------------------------------------------------------------
// IMU = true (windows with IMU factors): the workgroups behind the tiles linearise one IMU factor pair each (imu_pair_lin_wg,
// one wave; its 500 registers leave one workgroup per CU, which is what a single window runs at anyway).
template <bool COST_ONLY> __device__ __forceinline__ void pose_factor_eval(const DevPtrs& P, int slot, int idx, int ln);   // below
template <int FACTOR, bool RARE, bool IMU>
__global__ __launch_bounds__(BUILD_THREADS, IMU ? 1 : 2) void k_build(DevPtrs P, int slot, int max_tile_kf, int strip_doubles, int Rp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (IMU && (int)blockIdx.x >= P.n_tiles) {
        pose_factor_eval<false>(P, slot, (int)blockIdx.x - P.n_tiles, threadIdx.x);
        return;
    }
    const Tile T = P.tiles[blockIdx.x];
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    // first-round packet of this lane (DevPtrs::pre_lane): addressed by blockIdx alone, in flight beside the tile record
    int4 pl = make_int4(0, -1, 0, 0);
    int2 pk = make_int2(-1, -1);
    if (P.pre_lane) {
        pl = P.pre_lane[(long long)blockIdx.x * (BUILD_THREADS / 8) + (tid >> 3)];   // one packet per 8 lanes: a landmark's group is >= 8 lanes wide
        pk = P.pre_kf[(long long)blockIdx.x * PRE_KF + tid / POSE_TAB];     // tid / POSE_TAB <= 6 < PRE_KF
    }
    SADVIO_TS(3, 32);
    __shared__ double s_part[BUILD_WAVES * 4];
    // LDS carve
    double* poseTab = (double*)smem;
    double* camTab = poseTab + (size_t)max_tile_kf * POSE_TAB;
    int* rowTab = (int*)(camTab + MAX_WIN_CAM * 17);
    double* stage = (double*)(smem + tile_tables_bytes(max_tile_kf));  // [BUILD_WAVES][strip_doubles] wave-private strips
    const int G = T.G, lpw = 64 / G;
    const int grp = ln / G, q = ln - grp * G;
    const int nl = T.lmk1 - T.lmk0;
    // ---- everything that does not depend on the LM decision of the previous slot is fetched BEFORE that decision is summed:
    //      the wave's first landmark round (CSR range, position, BOTH delta buffers, the lane's observation) and the pose
    //      tables of BOTH buffers (first POSE_TAB-chunk per thread); the decision then only selects. A single window spends
    //      ~2.6 us in the decision and ~2.3 us in these dependent loads: now they overlap. ----
    LmState st;
    IterAcc a;
    LmState prev;
    const bool own_decision = !(slot == 0 || P.decide_kernel);
    // few tiles: every workgroup recomputes the decision of the previous slot (cheaper than one more launch on the critical
    // path): totals = window part (k_solve) + the tiles' k_backsub partials. The partials are read by ALL threads, every
    // load issued before the first add (< 1 024 tiles in this mode: <= 4 per thread) — one HBM round trip instead of one per
    // 64 tiles on a single wave — and they are the FIRST loads of the kernel: the memory counter retires in order.
    double pc = 0.0, pm = 0.0, psn = 0.0, pcn = 0.0;
    // a single window: its index, tile range and state records are known without the tile record (one dependent round trip less).
    // A real branch, not a select: a select would wait for the tile record's scalar load before forming either address.
    auto decision_inputs = [&](int dec_w, int t0, int nt) {
        if (!own_decision) {
            // many tiles: the accept / reject decision of the previous slot was taken once per window by k_decide
            st = P.states[(long long)dec_w * P.state_stride + slot];
            return;
        }
        a = P.acc[(long long)dec_w * P.state_stride + slot - 1];
        prev = P.states[(long long)dec_w * P.state_stride + slot - 1];
        if (P.world > 1) {
            if (wv == 0) {
                const double* rs = P.rank_s + (long long)dec_w * P.world * 4;
                for (int r = ln; r < P.world; r += 64) { pc += rs[4 * r]; pm += rs[4 * r + 1]; psn += rs[4 * r + 2]; pcn += rs[4 * r + 3]; }
            }
        } else {
            const TileAcc* ta = P.tacc + (long long)((slot - 1) & 1) * P.n_tiles + t0;
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = tid + u * BUILD_THREADS;
                const bool in = t < nt;
                const TileAcc* e = ta + (in ? t : 0);
                v[u][0] = in ? e->cand_cost : 0.0; v[u][1] = in ? e->mcc : 0.0; v[u][2] = in ? e->step_norm2 : 0.0; v[u][3] = in ? e->cand_norm2 : 0.0;
            }
            pc = (v[0][0] + v[1][0]) + (v[2][0] + v[3][0]); pm = (v[0][1] + v[1][1]) + (v[2][1] + v[3][1]);
            psn = (v[0][2] + v[1][2]) + (v[2][2] + v[3][2]); pcn = (v[0][3] + v[1][3]) + (v[2][3] + v[3][3]);
        }
    };
    if (P.n_win == 1) decision_inputs(0, 0, P.n_tiles);
    else decision_inputs(T.w, T.win_tile0, T.win_ntiles);
    bool first_valid;
    int gl_first, pre_ob, pre_oe, pre_o;     // pre_o: the lane's own observation of the first round, -1 = none
    if (P.pre_lane) {
        first_valid = (pl.z >> 16) & 1; gl_first = pl.x; pre_ob = pl.y; pre_oe = pl.y + (pl.z & 0xff);
        pre_o = (pl.w + (tid & 7) < (pl.z & 0xff)) ? pl.y + pl.w + (tid & 7) : -1;
    } else {
        first_valid = wv * lpw + grp < nl;
        gl_first = T.lmk0 + (first_valid ? wv * lpw + grp : 0);
        pre_ob = P.lmk_ob[gl_first]; pre_oe = P.lmk_oe[gl_first];
        pre_o = (first_valid && q < pre_oe - pre_ob) ? pre_ob + q : -1;
    }
    const int pre_lcode = P.lmk_const ? P.lmk_const[gl_first] : 0;
    double pre_p[3], pre_x0[3], pre_x1[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        pre_p[i] = P.lmk_p[3 * (long long)gl_first + i];
        pre_x0[i] = P.xl[3 * (long long)gl_first + i];
        pre_x1[i] = P.xl[P.xl_stride + 3 * (long long)gl_first + i];
    }
    double pre_t0 = 0.0, pre_t1 = 0.0;
    if (P.pre_lane) {
        if (pk.x >= 0) {
            const long long src = (long long)pk.x * POSE_TAB + (tid - (tid / POSE_TAB) * POSE_TAB);
            pre_t0 = P.ptab[src]; pre_t1 = P.ptab[P.ptab_stride + src];
        }
    } else if (tid < T.n_kf * POSE_TAB) {
        const int k = tid / POSE_TAB, e = tid - k * POSE_TAB;
        const long long src = (long long)P.tile_kf[T.kf_off + k] * POSE_TAB + e;
        pre_t0 = P.ptab[src]; pre_t1 = P.ptab[P.ptab_stride + src];
    }
    for (int i = tid; i < T.n_cam * 17; i += blockDim.x) {
        const int c = i / 17, e = i - 17 * c;
        const int gc = T.cam_base + c;
        camTab[i] = e < 4 ? P.cam_K[4 * (long long)gc + e] : (e < 16 ? P.cam_T[12 * (long long)gc + e - 4] : P.cam_isig[gc]);
    }
    for (int i = tid; i < T.n_kf; i += blockDim.x) rowTab[i] = P.tile_row[T.kf_off + i];
    ObsPre pre_obs;
    pre_obs.slot = 0; pre_obs.craw = 0; pre_obs.m[0] = pre_obs.m[1] = pre_obs.m[2] = 0.0;
    if (pre_o >= 0) pre_obs = obs_prefetch<FACTOR>(P, pre_o);
    if (own_decision) {
        pc = wave_sum(pc); pm = wave_sum(pm); psn = wave_sum(psn); pcn = wave_sum(pcn);
        if (ln == 0) { s_part[wv * 4] = pc; s_part[wv * 4 + 1] = pm; s_part[wv * 4 + 2] = psn; s_part[wv * 4 + 3] = pcn; }
        __syncthreads();
        a.cand_cost += (s_part[0] + s_part[4]) + (s_part[8] + s_part[12]); a.mcc += (s_part[1] + s_part[5]) + (s_part[9] + s_part[13]);
        a.step_norm2 += (s_part[2] + s_part[6]) + (s_part[10] + s_part[14]); a.cand_norm2 += (s_part[3] + s_part[7]) + (s_part[11] + s_part[15]);
        st = lm_decide(prev, a, P.o);
        __syncthreads();  // s_part is reused below
        if (T.first_of_window && tid == 0) { P.states[(long long)T.w * P.state_stride + slot] = st; trace_write(P, T.w, slot - 1, prev, a, st); }
    }
    if (st.done) return;
    SADVIO_TS(3, 33);
    double* Stile = stage + BUILD_WAVES * strip_doubles;
    const int Nt = 6 * T.n_free;
    const int tri_n = Nt * (Nt + 1) / 2;
    double* gT = Stile + tri_n;
    double* gfT = gT + Nt;
    double* hdT = gfT + Nt;
    const bool lds_mode = T.lds_mode != 0;
    // pose tables of the buffer that holds x: the prefetched chunk, the rest (tiles with more than 6 key-frames) from HBM
    if (tid < T.n_kf * POSE_TAB) poseTab[tid] = st.cur ? pre_t1 : pre_t0;
    {
        const double* src = P.ptab + (long long)st.cur * P.ptab_stride;
        for (int i = tid + blockDim.x; i < T.n_kf * POSE_TAB; i += blockDim.x) {
            const int k = i / POSE_TAB, e = i - k * POSE_TAB;
            poseTab[i] = src[(long long)P.tile_kf[T.kf_off + k] * POSE_TAB + e];
        }
    }
    if (lds_mode) {
        const int nz = tri_n + 3 * Nt;
        for (int i = tid; i < nz; i += blockDim.x) Stile[i] = 0.0;
    }
    __syncthreads();

    SADVIO_TS(3, 34);
    double* Sg = P.S + T.S_off;
    double* gredg = P.gred + T.red_off;
    double* gfullg = P.gfull + T.red_off;
    double* hdg = P.hdiag + T.red_off;
    const int dpf = T.dpf, Np = T.Np;
    const double* xl = P.xl + (long long)st.cur * P.xl_stride;
    double* wstage = stage + wv * strip_doubles;
    const bool gemm_mode = T.lds_mode == 2;
    double cost_part = 0.0, fixed_part = 0.0, gmax_part = 0.0;
    // MFMA path: a tile may hold several rounds of landmarks per wave (batched windows: fewer, larger tiles amortise the
    // table staging, the merge and the flush). The Y E^T products accumulate in registers across the rounds; the
    // block-diagonal / gradient sums go to the tile with ds_add_f64 (one lane per entry and wave after the DPP sums).
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 accs[3] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};  // <= 2 x 2 lower tile pairs (Nt <= 32)
    for (int base = wv * lpw; base < nl; base += BUILD_WAVES * lpw) {
        const int lm = base + grp;
        const bool lmk_valid = lm < nl;
        const int gl = T.lmk0 + (lmk_valid ? lm : 0);
        const bool first = base == wv * lpw;          // this round was fetched at the top of the kernel
        const int ob = first ? pre_ob : P.lmk_ob[gl], oe = first ? pre_oe : P.lmk_oe[gl];
        const int nobs = lmk_valid ? oe - ob : 0;
        // 0 free (eliminated here), 1 constant, 2 kept in the reduced system by a dense prior: its pose-pose part
        // goes the usual way, the landmark rows / columns are added by k_build_kept
        const int lcode = first ? pre_lcode : (P.lmk_const ? P.lmk_const[gl] : 0);
        const bool lfree = lcode == 0;
        ObsLin L;
        L.valid = q < nobs;
        L.row = -1; L.slot = 0; L.counted = false;
        if (L.valid) {
            double pw[3];
            ObsPre obq;
            if (first) {
#pragma unroll
                for (int i = 0; i < 3; i++) pw[i] = pre_p[i] + (st.cur ? pre_x1[i] : pre_x0[i]);
                obq = pre_obs;
            } else {
#pragma unroll
                for (int i = 0; i < 3; i++) pw[i] = P.lmk_p[3 * (long long)gl + i] + xl[3 * (long long)gl + i];
                obq = obs_prefetch<FACTOR>(P, ob + q);
            }
            lane_linearize<FACTOR, RARE>(P, poseTab, camTab, rowTab, T.cam_base, obq, pw, lfree, lcode != 1, L);
            const double c = L.rho;
            if (L.counted) cost_part += c;
            else { fixed_part += c; L.r[0] = 0.0; L.r[1] = 0.0; }
        } else {
            L.r[0] = L.r[1] = 0.0;
#pragma unroll
            for (int i = 0; i < 12; i++) L.Jp[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) L.Jl[i] = 0.0;
        }
    SADVIO_TS(3, 35);
        double Mi[6], g[3];
        const bool active = group_eliminate(P, L, G, gl, lmk_valid, lfree, nobs, st.radius, slot == 0, q == 0, Mi, g);
        if (active && q == 0) gmax_part = fmax(gmax_part, fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))));
    SADVIO_TS(3, 36);
        // N = Jl Li^T (2x3: Jl M^-1 Jl^T = N N^T), gamma = Li g_l, reduced residual r~ = r - Jl M^-1 g_l = r - N gamma
        double N[6], gam[3];
        li_row(Mi, L.Jl[0], L.Jl[1], L.Jl[2], N);
        li_row(Mi, L.Jl[3], L.Jl[4], L.Jl[5], N + 3);
        li_vec(Mi, g, gam);
        if (gemm_mode) {
            // ---- Schur accumulation without LDS atomics ---------------------------------------------------
            //   S_tile = sum_a Jp_a^T Jp_a - sum_l Y_l Y_l^T,  Y_l[KF] = sum_{a in KF} Jp_a^T (Jl_a Li^T): E M^-1 E^T in the
            //   symmetric form of a landmark-first Cholesky (one strip is both MFMA operands).
            // (1) the (at most two, adjacent) observations of a landmark in one key-frame are pre-summed with DPP;
            // (2) Y / E go with plain stores into wave-private strips [row][k], k = 4 * landmark + c: distinct
            //     landmarks own distinct k, so nothing collides; sum_l Y_l E_l^T is then a K-contraction on the
            //     FP64 matrix cores (one v_mfma_f64_16x16x4_f64 chain per 16x16 block);
            // (3) the block-diagonal part and the gradients are summed across the wave's 8 landmarks with DPP /
            //     v_permlane swaps when all of them see the same key-frames in the same lanes (the common case for
            //     landmarks created together), else with ds_add_f64.
            constexpr int Kw = 32, KS = Kw + 2;   // G == 8 on this path: 8 landmarks per wave and round
            double* Yb = wstage;
            const double* Eb = Yb;
            {
                double2* z = (double2*)wstage;
                const double2 zero2 = make_double2(0.0, 0.0);
                for (int i = ln; i < Rp * KS / 2; i += 64) z[i] = zero2;  // Rp * KS doubles
            }
            const bool vrow = L.valid && L.row >= 0;
            const int myrow = vrow ? L.row : -2 - ln;           // unique negative: never equal to a neighbour's
            const int row_prev = dpp_i32<0x111>(myrow);         // lane - 1
            const int row_next = dpp_i32<0x101>(myrow);         // lane + 1
            const bool follower = vrow && q > 0 && row_prev == myrow;
            const bool has_follower = vrow && q + 1 < G && row_next == myrow;
            // RARE: a landmark held by a prior carries pseudo-observations on the prior's key-frame next to its real ones: runs of up
            // to four lanes on one key-frame, summed in two steps (lane + 1, then lane + 2)
            const int row_next2 = RARE ? dpp_i32<0x102>(myrow) : -1;   // lane + 2 (read by every lane, outside the branches)
            const bool has_follower2 = RARE && vrow && q + 2 < G && row_next2 == myrow;
            const bool head = vrow && !follower;
            const double rt0 = L.r[0] - (N[0] * gam[0] + N[1] * gam[1] + N[2] * gam[2]);
            const double rt1 = L.r[1] - (N[3] * gam[0] + N[4] * gam[1] + N[5] * gam[2]);
            wave_lds_fence();
    SADVIO_TS(3, 37);
            // Y rows of this lane's key-frame (pair-summed), plain 16-byte stores by the run heads
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double j0 = vrow ? L.Jp[i] : 0.0, j1 = vrow ? L.Jp[6 + i] : 0.0;
                double y[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    y[c] = j0 * N[c] + j1 * N[3 + c];
                    const double yn = dpp_f64<0x101>(y[c]);
                    if (has_follower) y[c] += yn;
                    if (RARE) {
                        const double yn2 = dpp_f64<0x102>(y[c]);
                        if (has_follower2) y[c] += yn2;
                    }
                }
                if (head) {
                    double2* yp = (double2*)(Yb + (myrow + i) * KS + 4 * grp);
                    yp[0] = make_double2(y[0], y[1]); yp[1] = make_double2(y[2], 0.0);
                }
            }
            wave_lds_fence();
    SADVIO_TS(3, 38);
            {
                const int lr = ln & 15, lk = ln >> 4;
                const int nt16 = (Nt + 15) >> 4;
#pragma unroll
                for (int pp = 0; pp < 3; pp++) {
                    const int tr = pp < 1 ? 0 : 1, tc = pp - tr;
                    if (tr < nt16) {
                        d4& loc = accs[pp];
                        double av[8], bv[8];
#pragma unroll
                        for (int kk = 0; kk < 8; kk++) {  // all operand loads first, then the MFMA chain
                            const int k = (4 * kk < Kw) ? 4 * kk + lk : lk;
                            av[kk] = Yb[(16 * tr + lr) * KS + k];
                            bv[kk] = Eb[(16 * tc + lr) * KS + k];
                        }
#pragma unroll
                        for (int kk = 0; kk < 8; kk++)
                            if (4 * kk < Kw) loc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], loc, 0, 0, 0);
                    }
                }
            }
            // block-diagonal part D = Jp^T Jp (21) + reduced / full gradient (6 + 6)
            {
                const int rowu = vrow ? L.row : -1;
                int mism = (dpp_i32<0x128>(rowu) != rowu) | (xor16_other(rowu) != rowu) | (xor32_other(rowu) != rowu);
                const bool uniform = (G == 8) && (__ballot(mism) == 0ull);
                const bool adder = head && (!uniform || grp == 0);   // uniform: the wave's 8 landmarks were summed with DPP
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double j0 = vrow ? L.Jp[i] : 0.0, j1 = vrow ? L.Jp[6 + i] : 0.0;
#pragma unroll
                    for (int j = 0; j <= i; j++) {
                        double v = j0 * L.Jp[j] + j1 * L.Jp[6 + j];
                        if (!vrow) v = 0.0;
                        const double vn = dpp_f64<0x101>(v);
                        if (has_follower) v += vn;
                        if (RARE) { const double vn2 = dpp_f64<0x102>(v); if (has_follower2) v += vn2; }
                        if (uniform) v = across_groups8_sum(v);
                        if (adder) { atomic_add_f64(&Stile[tri(myrow + i, myrow + j)], v); if (i == j) atomic_add_f64(&hdT[myrow + i], v); }
                    }
                }
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double j0 = vrow ? L.Jp[i] : 0.0, j1 = vrow ? L.Jp[6 + i] : 0.0;
                    double gr = j0 * rt0 + j1 * rt1, gf = j0 * L.r[0] + j1 * L.r[1];
                    const double grn = dpp_f64<0x101>(gr), gfn = dpp_f64<0x101>(gf);
                    if (has_follower) { gr += grn; gf += gfn; }
                    if (RARE) {
                        const double grn2 = dpp_f64<0x102>(gr), gfn2 = dpp_f64<0x102>(gf);
                        if (has_follower2) { gr += grn2; gf += gfn2; }
                    }
                    if (uniform) { gr = across_groups8_sum(gr); gf = across_groups8_sum(gf); }
                    if (adder) { atomic_add_f64(&gT[myrow + i], gr); atomic_add_f64(&gfT[myrow + i], gf); }
                }
            }
    SADVIO_TS(3, 39);
            wave_lds_fence();
        } else {
        // exchange Jp / Jl with the other lanes of the group through the wave's private LDS strip
#pragma unroll
        for (int i = 0; i < 12; i++) wstage[i * 64 + ln] = L.Jp[i];
#pragma unroll
        for (int i = 0; i < 6; i++) wstage[(12 + i) * 64 + ln] = L.Jl[i];
        wave_lds_fence();
        if (L.valid && L.row >= 0) {
            const double rt0 = L.r[0] - (N[0] * gam[0] + N[1] * gam[1] + N[2] * gam[2]);
            const double rt1 = L.r[1] - (N[3] * gam[0] + N[4] * gam[1] + N[5] * gam[2]);
            const int pa = L.row;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double j0 = L.Jp[i], j1 = L.Jp[6 + i];
                const double gr = j0 * rt0 + j1 * rt1;
                const double gf = j0 * L.r[0] + j1 * L.r[1];
                const double hd = j0 * j0 + j1 * j1;
                if (lds_mode) {
                    atomic_add_f64(&gT[pa + i], gr); atomic_add_f64(&gfT[pa + i], gf); atomic_add_f64(&hdT[pa + i], hd);
                } else {
                    const int row = (pa / 6) * dpf + i;
                    atomic_add_f64(&gredg[row], gr); atomic_add_f64(&gfullg[row], gf); atomic_add_f64(&hdg[row], hd);
                }
            }
        }
        // S blocks: lane a adds rows of Jp_a^T W_ab Jp_b for every partner b of its landmark whose block is
        // on or below the diagonal; W_ab = delta_ab I - N_a N_b^T, N_b = Jl_b Li^T (the group shares Li)
        {
            for (int b = 0; b < T.kmax; b++) {  // wave-uniform bound: the shuffle below is convergent
                const int lb = grp * G + b;  // partner lane
                const int pb = __shfl(L.row, lb, 64);
                if (!(L.valid && L.row >= 0) || b >= nobs || pb < 0 || pb > L.row) continue;
                double Jpb[12], Jlb[6];
#pragma unroll
                for (int i = 0; i < 12; i++) Jpb[i] = wstage[i * 64 + lb];
#pragma unroll
                for (int i = 0; i < 6; i++) Jlb[i] = wstage[(12 + i) * 64 + lb];
                double Nb[6];
                li_row(Mi, Jlb[0], Jlb[1], Jlb[2], Nb);
                li_row(Mi, Jlb[3], Jlb[4], Jlb[5], Nb + 3);
                double w00 = -(N[0] * Nb[0] + N[1] * Nb[1] + N[2] * Nb[2]);
                double w01 = -(N[0] * Nb[3] + N[1] * Nb[4] + N[2] * Nb[5]);
                double w10 = -(N[3] * Nb[0] + N[4] * Nb[1] + N[5] * Nb[2]);
                double w11 = -(N[3] * Nb[3] + N[4] * Nb[4] + N[5] * Nb[5]);
                if (b == q) { w00 += 1.0; w11 += 1.0; }
                const int pa = L.row;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double c0 = L.Jp[i] * w00 + L.Jp[6 + i] * w10;
                    const double c1 = L.Jp[i] * w01 + L.Jp[6 + i] * w11;
                    const int row = pa + i;
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        const int col = pb + j;
                        if (col > row) continue;
                        const double v = c0 * Jpb[j] + c1 * Jpb[6 + j];
                        if (lds_mode) atomic_add_f64(&Stile[tri(row, col)], v);
                        else atomic_add_f64(&Sg[s_index(T.ld, (pa / 6) * dpf + i, (pb / 6) * dpf + j)], v);
                    }
                }
            }
        }
        wave_lds_fence();  // the strip is reused by the next round
    }
        }
    if (gemm_mode && wv * lpw < nl) {
        // the wave's Y E^T sum goes to the tile (ds_add_f64: the 4 waves add to the same entries)
        const int lr = ln & 15, lk = ln >> 4;
        const int nt16 = (Nt + 15) >> 4;
#pragma unroll
        for (int pp = 0; pp < 3; pp++) {
            const int tr = pp < 1 ? 0 : 1, tc = pp - tr;
            if (tr < nt16) {
                const int col = 16 * tc + lr;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int row = 16 * tr + lk + 4 * rg;
                    if (row < Nt && col <= row) atomic_add_f64(&Stile[tri(row, col)], -accs[pp][rg]);
                }
            }
        }
    }
    SADVIO_TS(3, 40);
    // cost / gradient-max: per-tile partial, plain store (no same-address atomics across the chip)
    const double c = wave_sum(cost_part), f = wave_sum(fixed_part), gm = wave_max(gmax_part);
    if (ln == 0) { s_part[wv * 4] = c; s_part[wv * 4 + 1] = f; s_part[wv * 4 + 2] = gm; }
    __syncthreads();
    if (tid == 0) {
        double cs = 0.0, fs = 0.0, gs = 0.0;
        for (int k = 0; k < BUILD_WAVES; k++) { cs += s_part[k * 4]; fs += s_part[k * 4 + 1]; gs = fmax(gs, s_part[k * 4 + 2]); }
        TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + blockIdx.x;
        ta->lin_cost = cs; ta->fixed_cost = fs; ta->gmax = gs;
    }
    SADVIO_TS(3, 41);
    // the tile is complete: every wave's ds_add_f64 was issued before the barrier above
    if (lds_mode) {
        // flush non-zeros; in lds_mode rowTab rows are 6*rank with the list sorted by global index, so the
        // local lower triangle maps onto the global lower triangle. One wave per row, lanes along the row.
        int* growTab = (int*)stage;  // local row -> global row (stage strip is free now)
        for (int k = tid; k < T.n_kf; k += blockDim.x) {
            const int r = rowTab[k];
            if (r >= 0) {
                const int fi = P.kf_fidx[P.tile_kf[T.kf_off + k]];
#pragma unroll
                for (int i = 0; i < 6; i++) growTab[r + i] = fi * dpf + i;
            }
        }
        __syncthreads();
        if (T.ld) {   // full row-major S: one wave per row, lanes along the row (consecutive addresses)
            for (int row = wv; row < Nt; row += BUILD_WAVES) {
                const long long grow = (long long)growTab[row] * T.ld;
                const double* srow = Stile + tri(row, 0);
                for (int col = ln; col <= row; col += 64) {
                    const double v = srow[col];
                    if (v != 0.0) atomic_add_f64(&Sg[grow + growTab[col]], v);
                }
            }
        } else {      // tile-packed S (column-major inside a tile): one wave per column, lanes down the column
            for (int col = wv; col < Nt; col += BUILD_WAVES) {
                const int gc = growTab[col];
                for (int row = col + ln; row < Nt; row += 64) {
                    const double v = Stile[tri(row, col)];
                    if (v != 0.0) atomic_add_f64(&Sg[c16_index(growTab[row], gc)], v);
                }
            }
        }
        for (int i = tid; i < Nt; i += blockDim.x) {
            const int row = growTab[i];
            if (gT[i] != 0.0) atomic_add_f64(&gredg[row], gT[i]);
            if (gfT[i] != 0.0) atomic_add_f64(&gfullg[row], gfT[i]);
            if (hdT[i] != 0.0) atomic_add_f64(&hdg[row], hdT[i]);
        }
    }
    SADVIO_TS(3, 42);
}

// Blocked right-looking Cholesky of the packed lower-triangular matrix P ((N+1) rows: row N is the
// right-hand side, so the forward substitution comes for free) followed by block back-substitution.
// The scalar sqrt / divide chain of the pivots bounds the latency, so it gets a wave of its own:
//   * wave 0 ("pivot wave") works one block column AHEAD: while waves 1.. apply the trailing update of
//     block column k, it updates pivot k+1 (one lane per element, gathered with v_readlane), factors it
//     with a hand-rolled rsqrt (v_rsq_f64 + 2 Newton steps) and publishes the inverse factor;
//   * the panel L_rk = A_rk Linv^T is also written to a transposed strip LpT[c][row] that the trailing
//     update reads with unit stride; 4x2 register micro-tiles;
//   * two workgroup barriers per block column, one per back-substitution step.
// On return xs = S^-1 rhs. Returns false if a pivot is not positive.
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    // one third-order step (see c16_rsqrt, chol16.h): e = 1 - d y^2, y <- y (1 + e / 2 + 3 e^2 / 8); five dependent operations
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double p = __builtin_fma(0.375, e, 0.5);
    const double q = e * p;
    return __builtin_fma(y, q, y);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

constexpr int NBP = 6;  // rows of the transposed panel strip (max NB)

template <int NB>
__device__ __forceinline__ void factor_pivot(double (&Lk)[NB][NB], double (&Li)[NB][NB], bool& bad) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        double d = Lk[j][j];
#pragma unroll
        for (int q = 0; q < j; q++) d -= Lk[j][q] * Lk[j][q];
        if (!(d > 0.0) || !isfinite(d)) bad = true;
        double inv = rsqrt_nr(d);
        Lk[j][j] = d * inv;
        Li[j][j] = inv;
#pragma unroll
        for (int i = j + 1; i < NB; i++) {
            double t = Lk[i][j];
#pragma unroll
            for (int q = 0; q < j; q++) t -= Lk[i][q] * Lk[j][q];
            Lk[i][j] = t * inv;
        }
    }
#pragma unroll
    for (int j = 0; j < NB; j++)
#pragma unroll
        for (int i = j + 1; i < NB; i++) {
            double t = 0.0;
#pragma unroll
            for (int q = j; q < i; q++) t -= Lk[i][q] * Li[q][j];
            Li[i][j] = t * Li[i][i];
        }
}

// Pivot wave: element (i, j), i >= j, of the NB x NB block at row/col `s0`, optionally minus the rank-NB
// update from the panel strip; one lane per element, then every lane gets all elements (uniform).
template <int NB, bool UPDATE>
__device__ __forceinline__ void pivot_gather_factor(const double* P, const double* LpT, int ldp, int s0, int ln,
                                                    double* linv_out, bool& bad) {
    constexpr int NE = NB * (NB + 1) / 2;
    // lane -> (i, j)
    int i = 0, j = 0;
    {
        int e = ln < NE ? ln : 0, r = 0;
        while (e >= r + 1) { e -= r + 1; r++; }
        i = r; j = e;
    }
    double acc = P[tri(s0 + i, s0 + j)];
    if (UPDATE) {
#pragma unroll
        for (int c = 0; c < NB; c++) acc -= LpT[c * ldp + i] * LpT[c * ldp + j];
    }
    double Lk[NB][NB], Li[NB][NB];
    {
        int e = 0;
#pragma unroll
        for (int ii = 0; ii < NB; ii++)
#pragma unroll
            for (int jj = 0; jj <= ii; jj++) { Lk[ii][jj] = readlane_f64(acc, e); e++; }
    }
    factor_pivot<NB>(Lk, Li, bad);
    // publish the inverse factor: lane e stores element e (values are wave-uniform, select by lane)
    {
        double v = 0.0;
        int e = 0;
#pragma unroll
        for (int ii = 0; ii < NB; ii++)
#pragma unroll
            for (int jj = 0; jj <= ii; jj++) { if (ln == e) v = Li[ii][jj]; e++; }
        if (ln < NE) linv_out[i * NB + j] = v;
    }
}

// TIL 16x16 tiles (pair indices pp, pp + stride, ...) of the trailing update C -= Lp_i Lp_j^T on the FP64 matrix
// cores; every LDS load is unconditional (clamped address), masking happens on the loaded values.
template <int NB, int TIL, bool BAND = false>
__device__ __forceinline__ void trail_tiles(double* P, const double* LpT, int ldp, int s, int m, int pp0, int stride, int lr, int lk, int rmin, int rhs_row) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 c[TIL];
    double a0[TIL], b0[TIL], a1[TIL], b1[TIL];
    int addr[TIL][4];
    bool okr[TIL][4];
#pragma unroll
    for (int u = 0; u < TIL; u++) {
        int ta = 0, rem = pp0 + u * stride;
        while (rem >= ta + 1) { rem -= ta + 1; ta++; }
        const int i0 = ta << 4, j0 = rem << 4;
        const int ai = i0 + lr, bj = j0 + lr;
        const int aic = ai < m ? ai : m - 1, bjc = bj < m ? bj : m - 1;
        const int k1 = lk + 4 < NB ? lk + 4 : 0;
        a0[u] = LpT[lk * ldp + aic]; b0[u] = LpT[lk * ldp + bjc];
        a1[u] = LpT[k1 * ldp + aic]; b1[u] = LpT[k1 * ldp + bjc];
        const int col = j0 + lr;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = i0 + lk + 4 * rg;
            okr[u][rg] = row < m && row >= rmin && col <= row && col < m - 1;
            const int rowc = row < m ? row : m - 1;
            const int colc = col <= rowc ? col : rowc;
            addr[u][rg] = tri((BAND && rowc == m - 1) ? rhs_row : s + rowc, s + colc);  // local row m - 1 is the rhs row (banded windows skip the zero rows between)
            c[u][rg] = P[addr[u][rg]];
        }
        a0[u] = (ai < m && lk < NB) ? -a0[u] : 0.0; b0[u] = (bj < m && lk < NB) ? b0[u] : 0.0;
        a1[u] = (ai < m && lk + 4 < NB) ? -a1[u] : 0.0; b1[u] = (bj < m && lk + 4 < NB) ? b1[u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < TIL; u++) {
        c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], c[u], 0, 0, 0);
        c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], c[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < TIL; u++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
            if (okr[u][rg]) P[addr[u][rg]] = c[u][rg];
}

// xs = L^-T x for the packed factor P (off-diagonal blocks) + inverse pivot blocks linvTab; x is destroyed. One barrier
// per block column; every thread of the workgroup must call it.
template <int NB>
__device__ __forceinline__ void block_backsub(const double* P, int nblk, double* x, double* xs, const double* linvTab) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = nblk - 1; k >= 0; k--) {
        const int c0 = k * NB;
        double xk[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) {
            double t = 0.0;
#pragma unroll
            for (int q = c; q < NB; q++) t += linvTab[k * NB * NB + q * NB + c] * x[c0 + q];
            xk[c] = t;
        }
        if (tid < NB) {
            double v = xk[0];
#pragma unroll
            for (int c = 1; c < NB; c++) if (tid == c) v = xk[c];
            xs[c0 + tid] = v;
        }
        for (int j = tid; j < c0; j += nt) {
            double t = x[j];
#pragma unroll
            for (int c = 0; c < NB; c++) t -= P[tri(c0 + c, j)] * xk[c];
            x[j] = t;
        }
        __syncthreads();
    }
}

// PARTIAL: only the first `nsteps` block columns are eliminated (the rest of the matrix is left as the updated
// Schur complement, pivot blocks included) and no back-substitution is done: one window of the block-banded solver.
// `band` > 0 (banded window): the rows of a pivot block's panel beyond band - NB below it are structurally zero, so the
// panel and the trailing update cover only those rows plus the rhs row.
template <int NB, bool PARTIAL = false, bool BAND = false>
__device__ __forceinline__ bool chol_solve_packed(double* P, int N, double* x, double* xs, double* LpT,
                                                  double* linvTab, long long* ts, int nsteps = 0, int band = 0) {
    const int tid = threadIdx.x, nt = blockDim.x, ln = tid & 63;
    const int nblk = PARTIAL ? nsteps : N / NB;
    const int ldp = N + 2;
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < nblk * NB * NB; i += nt) linvTab[i] = 0.0;
    __syncthreads();
    if (tid < 64) {
        bool bad = false;
        pivot_gather_factor<NB, false>(P, LpT, ldp, 0, ln, linvTab, bad);
        if (bad && ln == 0) s_bad = 1;
    }
    __syncthreads();
    for (int k = 0; k < nblk; k++) {
        const int c0 = k * NB;
        const int s = c0 + NB;
        const int mfull = N + 1 - s;  // trailing rows s .. N (row N = rhs)
        const int mb = BAND ? min(mfull - 1, band - NB) : mfull - 1;  // matrix rows with a non-zero panel
        const int m = mb + 1;         // + the rhs row, local index mb
        if (ts && k == 2 && tid == 0) ts[9] = wall_clock64();
        // --- panel: L_rk = A_rk Linv^T for the rows below the pivot, also into the transposed strip;
        //     one thread per output element (row r, column c): short chains, all loads unconditional ---
        {
            const double* lt = linvTab + k * NB * NB;
            for (int e = tid; e < m * NB; e += nt) {
                const int r = e / NB, c = e - r * NB;
                const double* row = P + tri((!BAND || r < mb) ? s + r : N, c0);
                double a[NB], lc[NB];
#pragma unroll
                for (int q = 0; q < NB; q++) { a[q] = row[q]; lc[q] = lt[c * NB + q]; }  // lt is zero above the diagonal
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < NB; q++) t += a[q] * lc[q];
                LpT[c * ldp + r] = t;
            }
        }
        __syncthreads();
        if (ts && k == 2 && tid == 0) ts[10] = wall_clock64();
        if (tid < 64) {
            // --- look-ahead: the pivot wave updates + factors pivot k+1 ---
            if (k + 1 < nblk) {
                bool bad = false;
                pivot_gather_factor<NB, true>(P, LpT, ldp, s, ln, linvTab + (k + 1) * NB * NB, bad);
                if (bad && ln == 0) s_bad = 1;
            }
            if (ts && k == 2 && tid == 0) ts[11] = wall_clock64();
        } else {
            // --- trailing update by the other waves on the FP64 matrix cores: one 16x16 tile of
            //     C -= Lp_i Lp_j^T per v_mfma_f64_16x16x4_f64 pair (K = NB <= 8); pivot block k+1 excluded ---
            // file the panel (still in the strip) into the packed matrix, where the back-substitution reads it
            // (nobody touches these columns during the trailing update)
            for (int e = tid - 64; e < m * NB; e += nt - 64) {
                const int r = e / NB, c = e - r * NB;
                P[tri((!BAND || r < mb) ? s + r : N, c0 + c)] = LpT[c * ldp + r];
            }
            typedef double d4 __attribute__((ext_vector_type(4)));
            const int nbw = (nt >> 6) - 1;                              // bulk waves
            const int bw = __builtin_amdgcn_readfirstlane(tid >> 6) - 1;  // this wave's index among them
            const int nt16 = (m + 15) >> 4;
            const int npair = nt16 * (nt16 + 1) / 2;
            const int lr = ln & 15, lk = ln >> 4;
            // full groups of 4 tiles per wave are processed together (all LDS loads first, independent and back to
            // back, then the MFMAs, then the stores: the tiles of one step are disjoint and the strip is read-only
            // here); the remainder one tile at a time
            // the next pivot block is updated by the pivot wave (look-ahead); the last step of a partial
            // factorisation has no look-ahead, so the bulk waves update those rows as well
            const int rmin = (PARTIAL && k + 1 == nblk) ? 0 : NB;
            int pp = bw;
            for (; pp + 3 * nbw < npair; pp += 4 * nbw) trail_tiles<NB, 4, BAND>(P, LpT, ldp, s, m, pp, nbw, lr, lk, rmin, N);
            for (; pp < npair; pp += nbw) trail_tiles<NB, 1, BAND>(P, LpT, ldp, s, m, pp, nbw, lr, lk, rmin, N);
            if (ts && k == 2 && tid == 64) ts[12] = wall_clock64();
        }
        __syncthreads();
        if (ts && k == 2 && tid == 0) ts[13] = wall_clock64();
    }
    if (ts && tid == 0) ts[14] = wall_clock64();
    if (s_bad) return false;
    if (PARTIAL) return true;
    // --- block back-substitution: xs = L^-T y, y = row N of P; one barrier per block ---
    for (int i = tid; i < N; i += nt) x[i] = P[tri(N, i)];
    __syncthreads();
    for (int k = nblk - 1; k >= 0; k--) {   // (block_backsub, spelled out: the call costs this kernel ~1 us)
        const int c0 = k * NB;
        double xk[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) {
            double t = 0.0;
#pragma unroll
            for (int q = c; q < NB; q++) t += linvTab[k * NB * NB + q * NB + c] * x[c0 + q];
            xk[c] = t;
        }
        if (tid < NB) {
            double v = xk[0];
#pragma unroll
            for (int c = 1; c < NB; c++) if (tid == c) v = xk[c];
            xs[c0 + tid] = v;
        }
        for (int j = tid; j < c0; j += nt) {
            double t = x[j];
#pragma unroll
            for (int c = 0; c < NB; c++) t -= P[tri(c0 + c, j)] * xk[c];
            x[j] = t;
        }
        __syncthreads();
    }
    return true;
}

// reduced-vector column of Jacobian column `a` of an IMUFactor, blocks [pose_i 6|pose_j 6|dv_i 3|dv_j 3|dba_i 3|dbg_i 3]
__device__ __forceinline__ int imu_col(int a, int fi, int fj) {
    if (a < 6) return fi < 0 ? -1 : fi * 15 + a;
    if (a < 12) return fj < 0 ? -1 : fj * 15 + a - 6;
    if (a < 15) return fi < 0 ? -1 : fi * 15 + 6 + a - 12;
    if (a < 18) return fj < 0 ? -1 : fj * 15 + 6 + a - 15;
    if (a < 21) return fi < 0 ? -1 : fi * 15 + 9 + a - 18;
    return fi < 0 ? -1 : fi * 15 + 12 + a - 21;
}

// ---- sparse prior factors inside the reduced solve --------------------------------------------------------
// reduced-vector column of Jacobian column a (0..14) of sparse factor f; -1 = constant / unused
__device__ __forceinline__ int sparse_col(const SparseDev& f, int a, int fi, int dpf, int lr0, int lr1) {
    if (f.type == 4) return -1;  // rides the Schur elimination as pseudo-observations (k_build / k_backsub)
    if (f.type == 5) return a < 6 ? (fi < 0 ? -1 : fi * dpf + a) : (a < 12 ? (lr0 < 0 ? -1 : lr0 + a - 6) : -1);  // lr0 = column of key-frame b
    if (f.type == 0) return (fi >= 0 && a < dpf) ? fi * dpf + a : -1;
    if (f.type == 1) return a < 6 ? (fi < 0 ? -1 : fi * dpf + a) : (a < 9 ? (lr0 < 0 ? -1 : lr0 + a - 6) : -1);
    if (f.type == 2) return (a < 3 && lr0 >= 0) ? lr0 + a : -1;
    return a < 3 ? (lr0 < 0 ? -1 : lr0 + a) : (a < 6 ? (lr1 < 0 ? -1 : lr1 + a - 3) : -1);
}
__device__ __forceinline__ int sparse_rows(const SparseDev& f) { return f.type == 0 ? 15 : (f.type == 5 ? 6 : 3); }
// reduced column of the second parameter block of a factor: a landmark kept in the reduced system, or (Relative6DPose) key-frame b
__device__ __forceinline__ int sparse_lr0(const DevPtrs& P, const WinDev& W, const SparseDev& f) {
    if (f.type == 5) { const int fj = P.kf_fidx[f.kf2]; return fj < 0 ? -1 : fj * W.dpf; }
    return (f.lmk0 >= 0 && P.lmk_red) ? P.lmk_red[f.lmk0] : -1;
}

// Evaluate sparse factor f at the state (xp, xv, xba, xbg, xl) + optional step `y` (reduced vector, may be null).
// J (rows x 15) may be null. Returns false if every parameter block is constant.
template <bool INL>
__device__ __forceinline__ bool sparse_eval_t(const DevPtrs& P, const WinDev& W, const SparseDev& f, const double* xp, const double* xv,
                                              const double* xba, const double* xbg, const double* xl, const double* y,
                                              double* r, double* J) {
    const int fi = f.kf >= 0 ? P.kf_fidx[f.kf] : -1;
    const int lr0 = sparse_lr0(P, W, f);
    const int lr1 = (f.lmk1 >= 0 && P.lmk_red) ? P.lmk_red[f.lmk1] : -1;
    if (fi < 0 && lr0 < 0 && lr1 < 0) return false;
    const int dpf = W.dpf;
    if (f.type == 5) {
        double da[6], db[6];
        for (int q = 0; q < 6; q++) {
            da[q] = xp[6 * (long long)f.kf + q] + ((y && fi >= 0) ? y[fi * dpf + q] : 0.0);
            db[q] = xp[6 * (long long)f.kf2 + q] + ((y && lr0 >= 0) ? y[lr0 + q] : 0.0);
        }
        relative_pose_factor(P.kf_T0 + 12 * (long long)f.kf, P.kf_T0 + 12 * (long long)f.kf2, f.T_prior, f.W, da, db, r, J);
        return true;
    }
    if (f.type == 0) {
        double prm[15];
        const long long k = f.kf;
        for (int q = 0; q < 6; q++) prm[q] = xp[6 * k + q] + ((y && fi >= 0) ? y[fi * dpf + q] : 0.0);
        for (int q = 0; q < 3; q++) {
            const bool st = y && fi >= 0 && dpf == 15;
            prm[6 + q] = xv[3 * k + q] + (st ? y[fi * 15 + 6 + q] : 0.0);
            prm[9 + q] = xba[3 * k + q] + (st ? y[fi * 15 + 9 + q] : 0.0);
            prm[12 + q] = xbg[3 * k + q] + (st ? y[fi * 15 + 12 + q] : 0.0);
        }
        if (INL) imu_prior_factor_body(P.kf_T0 + 12 * k, P.kf_vel + 3 * k, P.kf_ba + 3 * k, P.kf_bg + 3 * k, f.T_prior, f.v_prior,
                                       f.ba_prior, f.bg_prior, f.W, prm, r, J);
        else imu_prior_factor(P.kf_T0 + 12 * k, P.kf_vel + 3 * k, P.kf_ba + 3 * k, P.kf_bg + 3 * k, f.T_prior, f.v_prior,
                              f.ba_prior, f.bg_prior, f.W, prm, r, J);
        return true;
    }
    double q0[3], q1[3] = {0.0, 0.0, 0.0};
    for (int a = 0; a < 3; a++) {
        q0[a] = P.lmk_p[3 * (long long)f.lmk0 + a] + xl[3 * (long long)f.lmk0 + a] + ((y && lr0 >= 0) ? y[lr0 + a] : 0.0);
        if (f.type == 3) q1[a] = P.lmk_p[3 * (long long)f.lmk1 + a] + xl[3 * (long long)f.lmk1 + a] + ((y && lr1 >= 0) ? y[lr1 + a] : 0.0);
    }
    if (f.type == 1) {
        double d6[6];
        for (int q = 0; q < 6; q++) d6[q] = xp[6 * (long long)f.kf + q] + ((y && fi >= 0) ? y[fi * dpf + q] : 0.0);
        pose_to_landmark_factor(P.kf_T0 + 12 * (long long)f.kf, q0, f.delta, f.W, d6, r, J);
        return true;
    }
    double e[3];
    for (int a = 0; a < 3; a++) e[a] = q0[a] - q1[a] - f.delta[a];  // type 2: q1 = 0, delta = the prior position
    m3_vec(f.W, e, r);
    if (J) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { J[i * 15 + j] = f.W[3 * i + j]; J[i * 15 + 3 + j] = -f.W[3 * i + j]; }
    }
    return true;
}

__device__ __noinline__ bool sparse_eval(const DevPtrs& P, const WinDev& W, const SparseDev& f, const double* xp, const double* xv,
                                        const double* xba, const double* xbg, const double* xl, const double* y,
                                        double* r, double* J) {
    return sparse_eval_t<false>(P, W, f, xp, xv, xba, xbg, xl, y, r, J);
}

// ---- K6: reduced solve, one workgroup per window -------------------------------------------------
// MODE 0: the whole step in LDS (Np <= MAX_LDS_NP): gather S, pose-only factors, damping, Cholesky, candidate.
// Windows whose reduced system does not fit LDS (W.ld != 0, S kept as a full row-major lower triangle in HBM)
// run the same front (MODE 1) and back (MODE 2) halves around a library factorisation of S in place.
// EXTRAS = false compiles the IMU / sparse-prior / dense-prior sections out: the plain visual window (config 2)
// then runs a kernel without their register and scratch footprint.

// PosePriordx at the deltas d6: its linearisation record (see DevPtrs::prior_lin). k_solve's front half only adds the record
// of the buffer that holds x — the evaluation itself (log, Jr^-1: ~0.8 us on one lane) is off the critical path: it was done
// behind the previous factorisation, on the candidate that became x.
__device__ __forceinline__ double prior_lin_record(const double* T0, const double* Tp, const double* inf, const double* d6, double* rec) {
    double r[6], J[36];
    pose_prior_factor(T0, Tp, inf, d6, r, J);
    double c = 0.0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        c += r[a] * r[a];
        double g = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q++) g += J[6 * q + a] * r[q];
        rec[a] = g;
#pragma unroll
        for (int b = 0; b <= a; b++) {
            double hh = 0.0;
#pragma unroll
            for (int q = 0; q < 6; q++) hh += J[6 * q + a] * J[6 * q + b];
            rec[6 + a * (a + 1) / 2 + b] = hh;
        }
    }
    rec[27] = c;
    return c;
}

// The two transcendental bodies of k_solve's back half (candidate pose tables: sin / cos; prior records at the candidate: log,
// acos, Jr^-1) as CALLS: inlined into the 512-thread kernel they lengthened live ranges across the factorisation and cost it 42
// spilled VGPRs; a handful of lanes run them, off the critical path of the other waves.
// (arguments in registers: six scalars + pointers; a by-value struct of 18 doubles travels through scratch)
__device__ __noinline__ void pose_table_store(const double* T0, double d0, double d1, double d2, double d3, double d4, double d5, double* dst) {
    double T0r[12], tab[POSE_TAB];
#pragma unroll
    for (int i = 0; i < 12; i++) T0r[i] = T0[i];
    const double d6[6] = {d0, d1, d2, d3, d4, d5};
    pose_table_entry(T0r, d6, tab);
#pragma unroll
    for (int i = 0; i < POSE_TAB; i++) dst[i] = tab[i];
}
__device__ __noinline__ double prior_lin_record_call(const double* T0, double d0, double d1, double d2, double d3, double d4, double d5, const PriorDev* pr,
                                                     double* rec) {
    double T0r[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T0r[i] = T0[i];
    const double d6[6] = {d0, d1, d2, d3, d4, d5};
    const PriorDev q = *pr;
    return prior_lin_record(T0r, q.T_prior, q.inf, d6, rec);
}

template <int MODE, bool EXTRAS>
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(DevPtrs P, int slot) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool BIG = MODE != 0;
    const int w = blockIdx.x;
    const WinDev W = P.win[w];
    if (BIG != (W.ld != 0)) return;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, nwv = blockDim.x >> 6;  // wv in an SGPR: row tests below are scalar branches
    LmState* stp = P.states + (long long)w * P.state_stride + slot;
    IterAcc* acc = P.acc + (long long)w * P.state_stride + slot;
    __shared__ int s_done;                      // the decision of this kernel's own finalisation step (thread 0)
    __shared__ double s_red[SOLVE_THREADS / 64 * 4];
    SADVIO_TS(3, 0);
#ifdef SADVIO_KERNEL_TS
    if ((P.debug & 4096) && blockIdx.x == 0 && tid == 0 && slot == 3) P.dbg_ts[20] = clock64();
#endif
    LmState st = *stp;                          // uniform address: every thread keeps its own copy (no LDS round trip, no barrier)
    if (MODE == 1 && tid == 0) P.big_info[w] = 0;
    if (st.done) return;
    const int Np = W.Np;
    const long long ld = W.ld;
    double* Sg = P.S + W.S_off;
    double* gredg = P.gred + W.red_off;
    double* gfullg = P.gfull + W.red_off;
    double* hdg = P.hdiag + W.red_off;
    // N_p <= MAX_LDS_NP: the reduced system lives in LDS as the 16 x 16 tile-packed image of chol16.h (matrix + right-hand
    // side row), which is also its layout in HBM: the hand-off is a linear copy
    const int nbt = c16_blocks(Np + 1);         // tile rows incl. the right-hand-side row
    const int img_n = c16_size(Np);
    double* A = BIG ? Sg : (double*)smem;       // tile-packed lower (LDS) | full row-major lower (HBM)
    double* y = BIG ? gredg : A + img_n;        // rhs -> work vector
    double* gf = BIG ? gfullg : y + Np;         // full gradient
    double* hd = BIG ? hdg : gf + Np;           // diag(H)
    double* xs = BIG ? gredg : hd + Np;         // solution (BIG: the library factorisation overwrites the right-hand side)
    double* pub = xs + Np + (Np & 1);           // chol16: exchange area (C16_WORK doubles), 16-byte aligned
    double* yv = pub + C16_WORK;                // chol16: back-substitution exchange (16 * nbt)
    // MODE 0: what the back half needs from HBM is fetched with the front half's loads and parked in LDS across the
    // factorisation: the first SOLVE_KFC key-frames (free index, x, T0)
    double* kfc = yv + 16 * nbt;                // [SOLVE_KFC][SOLVE_KFC_STRIDE]
    auto aidx = [&](int i, int j) -> long long { return BIG ? (long long)i * ld + j : (long long)c16_index(i, j); };  // i >= j
    const int cur = st.cur;
    const double* xp = P.xp + (long long)cur * P.xp_stride;
    const int n_imu = W.imu_end - W.imu_begin;
    double early_cost = 0.0, early_fixed = 0.0, early_gm = 0.0;
    // Front half. Every HBM read is issued before the first wait. In-LDS windows: the image of S goes straight into LDS
    // (global_load_lds_dwordx4: no staging registers, no ds_write pass); the pose priors only contribute their linearisation
    // records (DevPtrs::prior_lin) of the buffer that holds x.
    double cost_part = 0.0, fixed_part = 0.0;
    const int n_pri = W.prior_end - W.prior_begin;
    const double* plin = P.prior_lin + (long long)cur * P.prior_lin_stride + (long long)W.prior_begin * PRIOR_LIN;
    // item = (prior, entry of g | H | cost): the first item of this thread is loaded here, windows with more items loop below
    double pl_v = 0.0;
    int pl_fi = -1;
    if (MODE != 2 && tid < n_pri * 28) {
        const int p = tid / 28, e = tid - 28 * p;
        pl_v = plin[p * PRIOR_LIN + e];
        pl_fi = P.kf_fidx[P.priors[W.prior_begin + p].kf];
    }
    double* sp = P.s_pose + W.red_off;
    double c_sp = 1.0;
    // A window sharded over several GPUs must come out of this kernel BIT-IDENTICAL on every rank (the ranks solve the
    // all-reduced system redundantly and take the LM decisions independently): there the pose-only factors are added one
    // factor at a time, plain read-modify-writes of the factor's distinct entries between barriers, instead of LDS atomics
    // whose order depends on wave scheduling.
    const bool det = P.world > 1;
    if (MODE != 2) {
    // IMUFactor + IMUBiasFactor (K3): imu_pair_lin_wg (an extra workgroup of k_build per factor pair) left every entry the
    // pair adds to the reduced system, with its position, in the factor's scratch row (ba_types.h). Item = (factor, entry); the
    // first eight items of every thread are fetched here, with the image
    double im_ix[8], im_v[8];
    // (P.imu_direct: imu_pair_lin_wg added the pairs' entries to the accumulators in HBM itself; only their costs are picked up here)
    const int n_imu_items = (EXTRAS && !det && !P.imu_direct) ? n_imu * IMU_NE : 0;
    if (EXTRAS) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int it = tid + u * (int)blockDim.x;
            im_ix[u] = -1.0; im_v[u] = 0.0;
            if (it < n_imu_items) {
                const int k = it / IMU_NE, e = it - IMU_NE * k;
                const double* row = P.imu_scratch + (long long)cur * P.imu_scratch_stride + (long long)(W.imu_begin + k) * IMU_ROW;
                im_ix[u] = row[IMU_IX + e]; im_v[u] = row[IMU_H + e];
            }
        }
    }
    if (!BIG) {
        // the image: a linear, fully coalesced 16-byte copy (S is kept in HBM in the layout it has in LDS); a wave-instruction
        // moves 64 consecutive double2 to a wave-uniform LDS base + lane * 16
        typedef __attribute__((address_space(3))) void lds_void_t;
        typedef __attribute__((address_space(1))) const void glb_void_t;
        const double2* g2 = (const double2*)Sg;
        double2* A2 = (double2*)A;
        const int n2 = img_n >> 1;              // a whole number of 256-double tiles: a multiple of 64 double2
        const int stride = nwv * 64;
        for (int i = wv * 64; i < n2; i += stride)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(g2 + i + ln), (lds_void_t*)(A2 + i), 16, 0, 0);
    }
    // the column vectors (N_p <= MAX_LDS_NP < SOLVE_THREADS: one column per thread) and the Jacobi scale of iteration 0
    double c_y = 0.0, c_gf = 0.0, c_hd = 0.0;
    if (!BIG && tid < Np) {
        c_y = gredg[tid]; c_gf = gfullg[tid]; c_hd = hdg[tid];
        if (slot != 0) c_sp = sp[tid];
    }
    // candidate-pose pass (behind the factorisation): this thread's key-frame is fetched now and parked in LDS
    const bool kf_pre = MODE == 0 && tid < W.n_kf && tid < SOLVE_KFC;
    int kf_fi = -1;
    double kf_t6[6], kf_t12[12], kf_vbb[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (kf_pre) {
        const int g = W.kf_base + tid;
        kf_fi = P.kf_fidx[g];
#pragma unroll
        for (int i = 0; i < 6; i++) kf_t6[i] = xp[6 * (long long)g + i];
#pragma unroll
        for (int i = 0; i < 12; i++) kf_t12[i] = P.kf_T0[12 * (long long)g + i];
        if (W.dpf == 15) {
            const double* xs3[3] = {P.xv + (long long)cur * P.xv_stride, P.xba + (long long)cur * P.xv_stride, P.xbg + (long long)cur * P.xv_stride};
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int i = 0; i < 3; i++) kf_vbb[3 * q + i] = xs3[q][3 * (long long)g + i];
        }
    }
    // window totals of the linearisation (the tiles' k_build partials / the ranks' of a sharded window)
    if (P.world > 1) {
        const double* rb = P.rank_b + (long long)w * P.world * 4;
        for (int r = tid; r < P.world; r += blockDim.x) { early_cost += rb[4 * r]; early_fixed += rb[4 * r + 1]; early_gm = fmax(early_gm, rb[4 * r + 2]); }
    } else {
        const TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + W.tile_begin;
        for (int t = tid; t < W.tile_end - W.tile_begin; t += blockDim.x) {
            early_cost += ta[t].lin_cost; early_fixed += ta[t].fixed_cost; early_gm = fmax(early_gm, ta[t].gmax);
        }
    }
    if (!BIG) {
        if (tid < Np) {
            y[tid] = c_y; gf[tid] = c_gf; hd[tid] = c_hd;
            gredg[tid] = 0.0; gfullg[tid] = 0.0; hdg[tid] = 0.0;
        }
        // (the accumulator in HBM is re-zeroed by the tiles of k_backsub / k_lm_pass: zero_s_slice)
        if (kf_pre) {
            double* c = kfc + tid * SOLVE_KFC_STRIDE;
            c[0] = (double)kf_fi;
#pragma unroll
            for (int i = 0; i < 6; i++) c[1 + i] = kf_t6[i];
#pragma unroll
            for (int i = 0; i < 12; i++) c[7 + i] = kf_t12[i];
            if (W.dpf == 15) {
#pragma unroll
                for (int i = 0; i < 9; i++) c[19 + i] = kf_vbb[i];
            }
        }
    }
    __syncthreads();
    SADVIO_TS(3, 1);
    // pose-only factors at x: PosePriordx (K4) from their linearisation records: item = (prior, entry), one atomic each
    if (det) {
        for (int p = 0; p < n_pri; p++) {
            const int fi = P.kf_fidx[P.priors[W.prior_begin + p].kf];
            if (tid < 28) {
                const double v = plin[p * PRIOR_LIN + tid];
                if (tid == 27) { if (fi < 0) fixed_part += v; else cost_part += v; }
                else if (fi >= 0) {
                    const int base = fi * W.dpf;
                    if (tid < 6) { y[base + tid] += v; gf[base + tid] += v; }
                    else {
                        const int q = tid - 6;
                        const int a = (int)((__builtin_sqrtf((float)(8 * q + 1)) - 1.0f) * 0.5f);
                        const int b = q - ((a * (a + 1)) >> 1);
                        A[aidx(base + a, base + b)] += v;
                        if (a == b) hd[base + a] += v;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int it = tid; !det && it < n_pri * 28; it += blockDim.x) {
        const int p = it / 28, e = it - 28 * p;
        double v = pl_v;
        int fi = pl_fi;
        if (it != tid) { v = plin[p * PRIOR_LIN + e]; fi = P.kf_fidx[P.priors[W.prior_begin + p].kf]; }
        if (e == 27) { if (fi < 0) fixed_part += v; else cost_part += v; continue; }
        if (fi < 0) continue;
        const int base = fi * W.dpf;
        if (e < 6) { atomic_add_f64(&y[base + e], v); atomic_add_f64(&gf[base + e], v); }
        else {
            const int q = e - 6;
            const int a = (int)((__builtin_sqrtf((float)(8 * q + 1)) - 1.0f) * 0.5f);   // q = a (a + 1) / 2 + b, b <= a <= 5
            const int b = q - ((a * (a + 1)) >> 1);
            atomic_add_f64(&A[aidx(base + a, base + b)], v);
            if (a == b) atomic_add_f64(&hd[base + a], v);
        }
    }
    if (EXTRAS && n_imu > 0) {
        if (det || P.imu_direct) {
            for (int k = tid; k < n_imu; k += blockDim.x) {
                const double* row = P.imu_scratch + (long long)cur * P.imu_scratch_stride + (long long)(W.imu_begin + k) * IMU_ROW;
                if (row[IMU_IX + IMU_E_COST] == -2.0) fixed_part += row[IMU_H + IMU_E_COST]; else cost_part += row[IMU_H + IMU_E_COST];
            }
            for (int k = 0; det && k < n_imu; k++) {
                const double* row = P.imu_scratch + (long long)cur * P.imu_scratch_stride + (long long)(W.imu_begin + k) * IMU_ROW;
                if (tid < IMU_E_BH) {
                    const int ix = (int)row[IMU_IX + tid];
                    const double v = row[IMU_H + tid];
                    if (ix >= 0) {
                        const int ca = ix >> 16, cb = ix & 0xffff;
                        if (tid < IMU_E_G) { A[aidx(ca, cb)] += v; if (ca == cb) hd[ca] += v; }
                        else { y[ca] += v; gf[ca] += v; }
                    }
                }
                __syncthreads();
                if (tid >= IMU_E_BH && tid < IMU_E_COST) {     // bias random walk of this factor (30 distinct targets)
                    const int ix = (int)row[IMU_IX + tid];
                    const double v = row[IMU_H + tid];
                    if (ix >= 0) {
                        const int ca = ix >> 16, cb = ix & 0xffff;
                        if (tid < IMU_E_BG) { A[aidx(ca, cb)] += v; if (ca == cb) hd[ca] += v; }
                        else { y[ca] += v; gf[ca] += v; }
                    }
                }
                __syncthreads();
            }
        }
        for (int it0 = tid; it0 < n_imu_items; it0 += 8 * blockDim.x) {   // all loads of eight items in flight before the first LDS atomic
            if (it0 != tid) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int it = it0 + u * (int)blockDim.x;
                    im_ix[u] = -1.0; im_v[u] = 0.0;
                    if (it < n_imu_items) {
                        const int k = it / IMU_NE, e = it - IMU_NE * k;
                        const double* row = P.imu_scratch + (long long)cur * P.imu_scratch_stride + (long long)(W.imu_begin + k) * IMU_ROW;
                        im_ix[u] = row[IMU_IX + e]; im_v[u] = row[IMU_H + e];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int it = it0 + u * (int)blockDim.x;
                const int ix = (int)im_ix[u];
                if (ix == -1) continue;
                const int e = it % IMU_NE;
                const double v = im_v[u];
                if (e == IMU_E_COST) { if (ix == -2) fixed_part += v; else cost_part += v; continue; }
                const int ca = ix >> 16, cb = ix & 0xffff;
                if (e < IMU_E_G || (e >= IMU_E_BH && e < IMU_E_BG)) {
                    atomic_add_f64(&A[aidx(ca, cb)], v);
                    if (ca == cb) atomic_add_f64(&hd[ca], v);
                } else { atomic_add_f64(&y[ca], v); atomic_add_f64(&gf[ca], v); }
            }
        }
    }
    // sparse (NFR) prior factors evaluated outside the Schur elimination (a sparsified VIO prior is one IMUPriordx here + hundreds
    // of pose-to-landmark factors that ride the elimination as pseudo-observations): sparse_factor_eval<false> left every entry the
    // factor adds, with its position, in its scratch row: item = (listed factor, entry)
    const int n_spl = W.spl_end - W.spl_begin;
    if (EXTRAS && n_spl > 0) {
        if (det) {   // sharded window: one factor at a time with plain adds (its 135 targets are distinct)
            for (int kl = 0; kl < n_spl; kl++) {
                const double* row = P.sp_scratch + (long long)cur * P.sp_scratch_stride + (long long)P.sp_list[W.spl_begin + kl] * SPARSE_J;
                if (tid < SPARSE_NE) {
                    const int ix = (int)row[SPARSE_IX + tid];
                    const double v = row[SPARSE_H + tid];
                    if (tid == SPARSE_E_COST) { if (ix == -2) fixed_part += v; else cost_part += v; }
                    else if (ix >= 0) {
                        const int ca = ix >> 16, cb = ix & 0xffff;
                        if (tid < SPARSE_E_G) { A[aidx(ca, cb)] += v; if (ca == cb) hd[ca] += v; }
                        else { y[ca] += v; gf[ca] += v; }
                    }
                }
                __syncthreads();
            }
        }
        for (int it = tid; !det && it < n_spl * SPARSE_NE; it += blockDim.x) {
            const int kl = it / SPARSE_NE, e = it - SPARSE_NE * kl;
            const double* row = P.sp_scratch + (long long)cur * P.sp_scratch_stride + (long long)P.sp_list[W.spl_begin + kl] * SPARSE_J;
            const int ix = (int)row[SPARSE_IX + e];
            const double v = row[SPARSE_H + e];
            if (e == SPARSE_E_COST) { if (ix == -2) fixed_part += v; else cost_part += v; continue; }
            if (ix < 0) continue;
            const int ca = ix >> 16, cb = ix & 0xffff;
            if (e < SPARSE_E_G) {
                atomic_add_f64(&A[aidx(ca, cb)], v);
                if (ca == cb) atomic_add_f64(&hd[ca], v);
            } else { atomic_add_f64(&y[ca], v); atomic_add_f64(&gf[ca], v); }
        }
    }
    // linexd observations: r, J (rows x 12 over [key-frame | line]) from k_line_eval<true>; item = (observation, lower-triangle entry)
    const int n_lo = W.lobs_end - W.lobs_begin;
    if (EXTRAS && n_lo > 0) {
        const int rows = W.factor_type == 0 ? 4 : 2;
        for (int k = tid; k < n_lo; k += blockDim.x) {
            const double* sc = P.line_scratch + (long long)(W.lobs_begin + k) * LINE_ROW;
            if (sc[53] != 0.0) cost_part += sc[52]; else fixed_part += sc[52];
        }
        for (int it = tid; it < n_lo * 78; it += blockDim.x) {
            const int k = it / 78;
            int e = it - 78 * k, a = 0;
            while (e >= a + 1) { e -= a + 1; a++; }
            const int b = e;
            const LineObsDev& ob = P.lobs[W.lobs_begin + k];
            const int fi = P.kf_fidx[ob.kf], lc = P.lines[ob.line].col;
            const int ca = a < 6 ? (fi < 0 ? -1 : fi * W.dpf + a) : (lc < 0 ? -1 : lc + a - 6);
            const int cb = b < 6 ? (fi < 0 ? -1 : fi * W.dpf + b) : (lc < 0 ? -1 : lc + b - 6);
            if (ca < 0 || cb < 0) continue;
            const double* sc = P.line_scratch + (long long)(W.lobs_begin + k) * LINE_ROW;
            double h = 0.0;
            for (int q = 0; q < rows; q++) h += sc[q * 12 + a] * sc[q * 12 + b];
            atomic_add_f64(&A[ca >= cb ? aidx(ca, cb) : aidx(cb, ca)], h);
            if (a == b) {
                double g = 0.0;
                for (int q = 0; q < rows; q++) g += sc[q * 12 + a] * sc[48 + q];
                atomic_add_f64(&y[ca], g); atomic_add_f64(&gf[ca], g); atomic_add_f64(&hd[ca], h);
            }
        }
    }
    // dense marginalisation prior (K4 / a9): its gradient, diagonal and J^T J were added to gred / gfull / hdiag / S
    // by the wide kernels k_prior_r / k_prior_gh before this kernel; only its cost is picked up here
    if (EXTRAS && W.dp_n_full > 0 && tid == 0) {
        double* pc = P.dp_data + W.dp_off + 2 * (size_t)W.dp_n_full * W.dp_n + (size_t)W.dp_n * W.dp_n + W.dp_n_full + W.dp_n + W.dp_n_full;
        if (P.world > 1) {   // the row blocks' partial sums of k_prior_r, in index order
            const double* pp = pc + 2;   // (dp_ptr(P, W, 7), defined below)
            double c = 0.0;
            for (int b = 0; b < (W.dp_n_full + 3) / 4; b++) c += pp[b];
            cost_part += c;
        } else {
            cost_part += pc[0];
            pc[0] = 0.0;
        }
    }
    // window totals of the linearisation: tiles' k_build partials + the pose-only factors evaluated here
    double gm = early_gm;
    cost_part += early_cost; fixed_part += early_fixed;   // the tiles' partials were loaded at the top of the kernel
    __syncthreads();
    // gradient tolerance (TrustRegionMinimizer::GradientToleranceReached) on the gradient at x; in the same pass (LDS image):
    // Jacobi scaling (iteration 0), LM diagonal, right-hand side into row Np of the packed matrix — none of it depends on the
    // finalisation test below, which only ends the solve
    if (!BIG) {
        if (tid < Np) {
            gm = fmax(gm, fabs(gf[tid]));
            const double hdi = hd[tid];
            double s = c_sp;
            if (slot == 0) { s = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(hdi)) : 1.0; sp[tid] = s; }
            const double s2 = s * s;
            A[c16_index(tid, tid)] += fmin(fmax(s2 * hdi, P.o.min_lm_diagonal), P.o.max_lm_diagonal) / st.radius / s2;
            A[c16_index(Np, tid)] = y[tid];     // right-hand side = row Np (the rest of that row and the padding rows are zero: S is)
        }
    } else {
        for (int i = tid; i < Np; i += blockDim.x) gm = fmax(gm, fabs(gf[i]));
    }
    gm = wave_max(gm); cost_part = wave_sum(cost_part); fixed_part = wave_sum(fixed_part);
    if (ln == 0) { s_red[wv * 3] = gm; s_red[wv * 3 + 1] = cost_part; s_red[wv * 3 + 2] = fixed_part; }
    if (tid == 0) s_done = 0;
    __syncthreads();
    if (!BIG) c16_symmetrize(A, nbt);           // the assembly writes i >= j only; the diagonal tiles are used as full symmetric tiles
    if (tid == blockDim.x - 1) {   // the last thread has the least symmetrisation work above
        double g = 0, cs = 0, fs = 0;
        for (int k = 0; k < nwv; k++) { g = fmax(g, s_red[k * 3]); cs += s_red[k * 3 + 1]; fs += s_red[k * 3 + 2]; }
        acc->gmax_bits = (unsigned long long)__double_as_longlong(g);
        acc->lin_cost = cs;
        acc->fixed_cost = fs;
        acc->cand_cost = 0.0; acc->mcc = 0.0; acc->step_norm2 = 0.0; acc->cand_norm2 = 0.0;
        // FinalizeIterationAndCheckIfMinimizerCanContinue of the previous iteration (of iteration zero at slot 0), in Ceres'
        // order: solver time, then the gradient tolerance (the iteration limit was applied by lm_decide)
        const bool time_up = P.o.max_time_ticks > 0.0 && P.t_start && (double)(wall_clock64() - *P.t_start) >= P.o.max_time_ticks;
        acc->time_up = time_up ? 1 : 0;
        // (max_num_iterations = 0: Ceres evaluates the initial cost, tests the gradient tolerance and stops with NO_CONVERGENCE)
        const bool no_iterations = st.iter == 0 && P.o.max_num_iterations <= 0;
        if (time_up || g <= P.o.gradient_tolerance || no_iterations) {
            LmState e = st;
            e.done = 1; e.termination = (time_up || g > P.o.gradient_tolerance) ? 0 : 3;
            e.x_cost = 0.5 * cs;
            if (e.iter == 0) e.initial_cost = e.x_cost;
            *stp = e;
            s_done = 1;
        }
    }
    __syncthreads();
    if (s_done) return;
    SADVIO_TS(3, 2);
    if (BIG) {
        for (int i = tid; i < Np; i += blockDim.x) {
            double s;
            if (slot == 0) { s = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(hd[i])) : 1.0; sp[i] = s; }
            else s = sp[i];
            double s2 = s * s;
            A[aidx(i, i)] += fmin(fmax(s2 * hd[i], P.o.min_lm_diagonal), P.o.max_lm_diagonal) / st.radius / s2;
        }
        __syncthreads();
    }
    SADVIO_TS(3, 3);
    if (MODE == 1) return;  // the host enqueues potrf / potrs on S, gred next
    }  // MODE != 2
    long long* ts = SADVIO_TS_PTR((P.debug & 4096) && blockIdx.x == 0 && slot == 3);
    if (MODE == 0) {
        bool ok = true;  // Np == 0 (every key-frame constant, landmarkOptimization): nothing to factor
        if (Np > 0) ok = c16_solve<1>(A, Np, xs, pub, yv, ts ? ts + 64 : nullptr);   // SADVIO_KERNEL_TS builds: per-block-column cycle stamps
        if (!ok) {
            if (tid == 0) acc->chol_fail = 1;
            return;
        }
    } else {
        // potrf / potrs ran on S / gred in place; info > 0 = leading minor not positive definite
        y = P.delta + W.red_off;
        if (P.big_info[w] != 0) {
            for (int i = tid; i < Np; i += blockDim.x) { gredg[i] = 0.0; gfullg[i] = 0.0; hdg[i] = 0.0; }
            if (tid == 0) acc->chol_fail = 1;
            return;
        }
    }
    SADVIO_TS(3, 4);
    // delta = -y ; candidate poses ; norms ; pose-only model cost
    double* dl = P.delta + W.red_off;
    double sn = 0.0, cn = 0.0;
    bool bad = false;
    for (int i = tid; i < Np; i += blockDim.x) {
        double d = -xs[i];
        dl[i] = d;
        y[i] = d;
        sn += d * d;
        if (!isfinite(d)) bad = true;
        if (BIG) { gredg[i] = 0.0; gfullg[i] = 0.0; hdg[i] = 0.0; }
    }
    __syncthreads();
    SADVIO_TS(3, 5);
    double* xpc = P.xp + (long long)(1 - cur) * P.xp_stride;
    for (int k = tid; k < W.n_kf; k += blockDim.x) {
        int g = W.kf_base + k;
        const bool pre = MODE == 0 && k < SOLVE_KFC;  // parked in LDS by the front half
        const double* kc = kfc + (pre ? k : 0) * SOLVE_KFC_STRIDE;
        int fi = pre ? (int)kc[0] : P.kf_fidx[g];
        double d6[6];
        const double* T0r = pre ? kc + 7 : P.kf_T0 + 12 * (long long)g;   // LDS (parked by the front half) or HBM: a flat pointer either way
        double vbb[9];                          // v, ba, bg at x: fetched before the table is computed
        if (W.dpf == 15) {
            const double* xs3[3] = {P.xv + (long long)cur * P.xv_stride, P.xba + (long long)cur * P.xv_stride, P.xbg + (long long)cur * P.xv_stride};
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int i = 0; i < 3; i++) vbb[3 * q + i] = pre ? kc[19 + 3 * q + i] : xs3[q][3 * (long long)g + i];
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double v = (pre ? kc[1 + i] : xp[6 * (long long)g + i]) + (fi < 0 ? 0.0 : y[fi * W.dpf + i]);
            xpc[6 * (long long)g + i] = v;
            d6[i] = v;
            if (fi >= 0) cn += v * v;
        }
        // pose table of the candidate buffer (k_backsub reads it now, k_build reads it if the step is accepted)
        pose_table_store(T0r, d6[0], d6[1], d6[2], d6[3], d6[4], d6[5], P.ptab + (long long)(1 - cur) * P.ptab_stride + (long long)g * POSE_TAB);
        if (W.dpf == 15) {
            double* xs3[3] = {P.xv + (long long)(1 - cur) * P.xv_stride, P.xba + (long long)(1 - cur) * P.xv_stride, P.xbg + (long long)(1 - cur) * P.xv_stride};
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double v = vbb[3 * q + i] + (fi < 0 ? 0.0 : y[fi * 15 + 6 + 3 * q + i]);
                    xs3[q][3 * (long long)g + i] = v;
                    if (fi >= 0) cn += v * v;
                }
        }
    }
    if (EXTRAS && W.line_end > W.line_begin) {
        const double* xlc = P.xline + (long long)cur * P.xline_stride;
        double* xln = P.xline + (long long)(1 - cur) * P.xline_stride;
        for (int it = tid; it < (W.line_end - W.line_begin) * 6; it += blockDim.x) {
            const int l = W.line_begin + it / 6, q = it % 6;
            const int lc = P.lines[l].col;
            const double v = xlc[6 * (long long)l + q] + (lc < 0 ? 0.0 : y[lc + q]);
            xln[6 * (long long)l + q] = v;
            if (lc >= 0) cn += v * v;
        }
    }
    SADVIO_TS(3, 6);
    // priors: model cost change -(J d)^T (r + J d / 2) = -(d.g + d^T H d / 2) from the record at x, and the linearisation
    // record AT THE CANDIDATE (its cost is the candidate cost; the next front half adds it if the step is accepted). Priors
    // are dealt from the LAST thread downwards: a different wave than the pose tables above, so the two run side by side.
    double mcc = 0.0, cc = 0.0;
    {
        const double* plx = P.prior_lin + (long long)cur * P.prior_lin_stride + (long long)W.prior_begin * PRIOR_LIN;
        double* plc = P.prior_lin + (long long)(1 - cur) * P.prior_lin_stride + (long long)W.prior_begin * PRIOR_LIN;
        for (int p = (int)blockDim.x - 1 - tid; p < W.prior_end - W.prior_begin; p += blockDim.x) {
            const PriorDev* prp = P.priors + W.prior_begin + p;
            const int pkf = prp->kf;
            const int fi = P.kf_fidx[pkf];
            if (fi < 0) continue;
            double d[6], d6[6];
#pragma unroll
            for (int i = 0; i < 6; i++) { d[i] = y[fi * W.dpf + i]; d6[i] = xp[6 * (long long)pkf + i] + d[i]; }
            double lin = 0.0, quad = 0.0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                lin += d[a] * plx[p * PRIOR_LIN + a];
#pragma unroll
                for (int b = 0; b <= a; b++) quad += (a == b ? 0.5 : 1.0) * d[a] * d[b] * plx[p * PRIOR_LIN + 6 + a * (a + 1) / 2 + b];
            }
            mcc += -(lin + quad);
            cc += prior_lin_record_call(P.kf_T0 + 12 * (long long)pkf, d6[0], d6[1], d6[2], d6[3], d6[4], d6[5], prp, plc + p * PRIOR_LIN);
        }
    }
    if (EXTRAS && n_imu > 0) {
        const double* xv = P.xv + (long long)cur * P.xv_stride;
        const double* xba = P.xba + (long long)cur * P.xv_stride;
        const double* xbg = P.xbg + (long long)cur * P.xv_stride;
        // model cost change, item = (factor, residual row); y holds delta. Dealt from the second wave upwards: the first one is
        // busy with the candidate poses above, the last one with the priors. Everything an item needs is in the scratch row.
        const int t0 = blockDim.x > 128 ? 64 : 0;
        for (int it = tid - t0; it >= 0 && it < n_imu * 15; it += blockDim.x - t0) {
            const int k = it / 15, q = it - 15 * k;
            const double* sc = P.imu_scratch + (long long)cur * P.imu_scratch_stride + (long long)(W.imu_begin + k) * IMU_ROW;
            const int fi = (int)sc[IMU_META], fj = (int)sc[IMU_META + 1];
            if (fi < 0 && fj < 0) continue;
            double m = 0.0, r;
            if (q < 9) {
                double jr[24];
#pragma unroll
                for (int a = 0; a < 24; a++) jr[a] = sc[q * 24 + a];
                r = sc[9 * 24 + q];
#pragma unroll
                for (int a = 0; a < 24; a++) { const int ca = imu_col(a, fi, fj); if (ca >= 0) m += jr[a] * y[ca]; }
            } else {
                const int e = q - 9, ax = e % 3, gy = e / 3;
                const double sgm = sc[IMU_META + 2 + gy];
                r = sc[IMU_J + e];
                if (fi >= 0) m -= sgm * y[fi * 15 + 9 + 3 * gy + ax];
                if (fj >= 0) m += sgm * y[fj * 15 + 9 + 3 * gy + ax];
            }
            mcc += -m * (r + 0.5 * m);
        }
        // candidate cost of the IMU factors: imu_pair_cost (extra workgroups of k_backsub, after this kernel) adds it to acc->cand_cost
    }
    if (EXTRAS && W.spl_end > W.spl_begin) {
        // model cost change, item = (listed factor, residual row), dealt from the fifth wave upwards (see the IMU items); the row holds
        // the reduced column of every Jacobian column
        const int t0 = blockDim.x > 320 ? 256 : 0;
        for (int it = tid - t0; it >= 0 && it < (W.spl_end - W.spl_begin) * 15; it += blockDim.x - t0) {
            const int kl = it / 15, q = it - 15 * kl;
            const double* sc = P.sp_scratch + (long long)cur * P.sp_scratch_stride + (long long)P.sp_list[W.spl_begin + kl] * SPARSE_J;
            if (q >= (int)sc[SPARSE_COL + 15]) continue;
            double jr[15], cd[15];
#pragma unroll
            for (int a = 0; a < 15; a++) { jr[a] = sc[q * 15 + a]; cd[a] = sc[SPARSE_COL + a]; }
            const double r = sc[225 + q];
            double m = 0.0;
#pragma unroll
            for (int a = 0; a < 15; a++) { const int ca = (int)cd[a]; if (ca >= 0) m += jr[a] * y[ca]; }
            mcc += -m * (r + 0.5 * m);
        }
        // candidate cost of these factors: sparse_factor_eval<true> (extra workgroups of k_backsub) adds it to acc->cand_cost
    }
    if (EXTRAS && W.lobs_end > W.lobs_begin) {
        const int rows = W.factor_type == 0 ? 4 : 2;
        for (int it = tid; it < (W.lobs_end - W.lobs_begin) * 4; it += blockDim.x) {
            const int k = it >> 2, q = it & 3;
            if (q >= rows) continue;
            const LineObsDev& ob = P.lobs[W.lobs_begin + k];
            const int fi = P.kf_fidx[ob.kf], lc = P.lines[ob.line].col;
            if (fi < 0 && lc < 0) continue;
            const double* sc = P.line_scratch + (long long)(W.lobs_begin + k) * LINE_ROW;
            double m = 0.0;
            for (int a = 0; a < 6; a++) {
                if (fi >= 0) m += sc[q * 12 + a] * y[fi * W.dpf + a];
                if (lc >= 0) m += sc[q * 12 + 6 + a] * y[lc + a];
            }
            mcc += -m * (sc[48 + q] + 0.5 * m);
        }
        // candidate cost of the line factors: k_line_eval<false>, after this kernel
    }
    SADVIO_TS(3, 7);
    sn = wave_sum(sn); cn = wave_sum(cn); mcc = wave_sum(mcc); cc = wave_sum(cc);
    if (bad) acc->chol_fail = 1;
    __syncthreads();
    if (ln == 0) { s_red[wv * 4] = sn; s_red[wv * 4 + 1] = cn; s_red[wv * 4 + 2] = mcc; s_red[wv * 4 + 3] = cc; }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int k = 0; k < nwv; k++) { a0 += s_red[k * 4]; a1 += s_red[k * 4 + 1]; a2 += s_red[k * 4 + 2]; a3 += s_red[k * 4 + 3]; }
        acc->step_norm2 = a0; acc->cand_norm2 = a1; acc->mcc = a2; acc->cand_cost = a3;
    }
    SADVIO_TS(3, 8);
    if (ts && tid == 0) { ts[15] = wall_clock64(); ts[21] = clock64(); }
}


// The reduced system of an in-LDS window is an accumulator in HBM (k_build adds into it with atomics) that has to be zero when
// the next k_build starts. k_solve only READS it; the re-zeroing is spread over the tiles of the back-substitution kernel that
// runs in between — a slice of a few hundred bytes per workgroup — so that k_solve's HBM traffic is the image it needs and no more.
__device__ __forceinline__ void zero_s_slice(const DevPtrs& P, const Tile& T, int tile_index) {
    if (T.ld != 0) return;                       // out-of-LDS systems are re-zeroed by their own solver
    const int n2 = c16_size(T.Np) >> 1;          // double2 of the tile-packed image
    const int per = (n2 + T.win_ntiles - 1) / T.win_ntiles;
    const int i0 = (tile_index - T.win_tile0) * per, i1 = min(i0 + per, n2);
    double2* S2 = (double2*)(P.S + T.S_off);
    const double2 z2 = make_double2(0.0, 0.0);
    for (int i = i0 + (int)threadIdx.x; i < i1; i += blockDim.x) S2[i] = z2;
}

// ---- K7: back-substitution + candidate cost -------------------------------------------------------
// IMU = true (windows with IMU factors): the workgroups behind the tiles evaluate the cost of one IMU factor pair each at the
// candidate (imu_pair_cost: one wave, beside the tiles instead of behind them on the stream).
template <int FACTOR, bool RARE, bool IMU>
__global__ __launch_bounds__(BUILD_THREADS) void k_backsub(DevPtrs P, int slot, int max_tile_kf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (IMU && (int)blockIdx.x >= P.n_tiles) {
        if (threadIdx.x < 64) pose_factor_eval<true>(P, slot, (int)blockIdx.x - P.n_tiles, threadIdx.x);
        return;
    }
    const Tile T = P.tiles[blockIdx.x];
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    // first-round packet of this lane (DevPtrs::pre_lane, see k_build): with it every input of the opening is at most two dependent
    // round trips away (packet / tile record / state -> data) instead of four (tile -> key-frame list -> free index -> step)
    int4 pl = make_int4(0, -1, 0, 0);
    int2 pk = make_int2(-1, -1), pk6 = make_int2(-1, -1);
    if (P.pre_lane) {
        pl = P.pre_lane[(long long)blockIdx.x * (BUILD_THREADS / 8) + (tid >> 3)];   // one packet per 8 lanes: a landmark's group is >= 8 lanes wide
        pk = P.pre_kf[(long long)blockIdx.x * PRE_KF + tid / POSE_TAB];     // tid / POSE_TAB <= 6 < PRE_KF
        if (tid < PRE_KF * 6) pk6 = P.pre_kf[(long long)blockIdx.x * PRE_KF + tid / 6];
    }
    LmState st;
    int chol_fail;
    if (P.n_win == 1) { st = P.states[slot]; chol_fail = P.acc[slot].chol_fail; }      // (a real branch: see k_build)
    else { st = P.states[(long long)T.w * P.state_stride + slot]; chol_fail = P.acc[(long long)T.w * P.state_stride + slot].chol_fail; }
    zero_s_slice(P, T, blockIdx.x);
    double* poseTab = (double*)smem;
    double* camTab = poseTab + (size_t)max_tile_kf * POSE_TAB;
    int* rowTab = (int*)(camTab + MAX_WIN_CAM * 17);
    double* candTab = (double*)(smem + tile_tables_bytes(max_tile_kf));  // [n_kf][12] R|t at the candidate poses
    double* dpTab = candTab + (size_t)max_tile_kf * 12;                  // [n_kf][6] pose step of each listed key-frame
    const int G = T.G, lpw = 64 / G;
    const int grp = ln / G, q = ln - grp * G;
    const int nl = T.lmk1 - T.lmk0;
    // the wave's first landmark round is fetched before the tables are staged - and, with the packets, before the state record says
    // which delta buffer holds x (both are fetched, the record selects)
    bool first_valid;
    int gl_first, pre_ob, pre_oe, pre_o;
    if (P.pre_lane) {
        first_valid = (pl.z >> 16) & 1; gl_first = pl.x; pre_ob = pl.y; pre_oe = pl.y + (pl.z & 0xff);
        pre_o = (pl.w + (tid & 7) < (pl.z & 0xff)) ? pl.y + pl.w + (tid & 7) : -1;
    } else {
        first_valid = wv * lpw + grp < nl;
        gl_first = T.lmk0 + (first_valid ? wv * lpw + grp : 0);
        pre_ob = P.lmk_ob[gl_first]; pre_oe = P.lmk_oe[gl_first];
        pre_o = (first_valid && q < pre_oe - pre_ob) ? pre_ob + q : -1;
    }
    const int pre_lcode = P.lmk_const ? P.lmk_const[gl_first] : 0;
    double pre_p[3], pre_xa[3], pre_xb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        pre_p[i] = P.lmk_p[3 * (long long)gl_first + i];
        pre_xa[i] = P.xl[3 * (long long)gl_first + i];
        pre_xb[i] = P.xl[P.xl_stride + 3 * (long long)gl_first + i];
    }
    ObsPre pre_obs;
    pre_obs.slot = 0; pre_obs.craw = 0; pre_obs.m[0] = pre_obs.m[1] = pre_obs.m[2] = 0.0;
    if (pre_o >= 0) pre_obs = obs_prefetch<FACTOR>(P, pre_o);
    // pose tables of both buffers: this thread's entry of the first 256 (packets), selected below
    double pre_t0 = 0.0, pre_t1 = 0.0;
    const bool pre_tab = P.pre_lane != nullptr;
    if (pre_tab && pk.x >= 0) {
        const long long src = (long long)pk.x * POSE_TAB + (tid - (tid / POSE_TAB) * POSE_TAB);
        pre_t0 = P.ptab[src]; pre_t1 = P.ptab[P.ptab_stride + src];
    }
    if (st.done || chol_fail) {
        if (tid == 0) {
            TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + blockIdx.x;
            ta->cand_cost = 0.0; ta->mcc = 0.0; ta->step_norm2 = 0.0; ta->cand_norm2 = 0.0;
        }
        return;
    }
    const int cur = st.cur;
    const double* xl = P.xl + (long long)cur * P.xl_stride;
    double* xlc = P.xl + (long long)(1 - cur) * P.xl_stride;
    double pre_x[3];
#pragma unroll
    for (int i = 0; i < 3; i++) pre_x[i] = cur ? pre_xb[i] : pre_xa[i];
    if (pre_tab) {
        const double* src = P.ptab + (long long)cur * P.ptab_stride;
        const double* srcc = P.ptab + (long long)(1 - cur) * P.ptab_stride;
        if (tid < T.n_kf * POSE_TAB) {
            const int k = tid / POSE_TAB, e = tid - k * POSE_TAB;
            poseTab[tid] = cur ? pre_t1 : pre_t0;
            if (e < 12) candTab[k * 12 + e] = cur ? pre_t0 : pre_t1;
        }
        for (int i = tid + blockDim.x; i < T.n_kf * POSE_TAB; i += blockDim.x) {     // tiles with more than 6 key-frames
            const int k = i / POSE_TAB, e = i - k * POSE_TAB;
            const long long o = (long long)P.tile_kf[T.kf_off + k] * POSE_TAB + e;
            poseTab[i] = src[o];
            if (e < 12) candTab[k * 12 + e] = srcc[o];
        }
        for (int i = tid; i < T.n_cam * 17; i += blockDim.x) {
            const int c = i / 17, e = i - 17 * c;
            const int gc = T.cam_base + c;
            camTab[i] = e < 4 ? P.cam_K[4 * (long long)gc + e] : (e < 16 ? P.cam_T[12 * (long long)gc + e - 4] : P.cam_isig[gc]);
        }
        for (int i = tid; i < T.n_kf; i += blockDim.x) rowTab[i] = P.tile_row[T.kf_off + i];
        const double* dp = P.delta + T.red_off;
        for (int i = tid; i < T.n_kf * 6; i += blockDim.x) {
            const int k = i / 6, e = i - 6 * k;
            const int fi = (i < PRE_KF * 6) ? pk6.y : P.kf_fidx[P.tile_kf[T.kf_off + k]];
            dpTab[i] = fi < 0 ? 0.0 : dp[fi * T.dpf + e];
        }
    } else {
        stage_tables(P, T, cur, poseTab, camTab, rowTab);
        const double* src = P.ptab + (long long)(1 - cur) * P.ptab_stride;
        for (int i = tid; i < T.n_kf * 12; i += blockDim.x) {
            const int k = i / 12, e = i - 12 * k;
            candTab[i] = src[(long long)P.tile_kf[T.kf_off + k] * POSE_TAB + e];
        }
        const double* dp = P.delta + T.red_off;
        for (int i = tid; i < T.n_kf * 6; i += blockDim.x) {
            const int k = i / 6, e = i - 6 * k;
            const int fi = P.kf_fidx[P.tile_kf[T.kf_off + k]];
            dpTab[i] = fi < 0 ? 0.0 : dp[fi * T.dpf + e];
        }
    }
    __syncthreads();
    double sn = 0.0, cn = 0.0, mcc = 0.0, cc = 0.0;
    for (int base = wv * lpw; base < nl; base += BUILD_WAVES * lpw) {
        const int lm = base + grp;
        const bool lmk_valid = lm < nl;
        const int gl = T.lmk0 + (lmk_valid ? lm : 0);
        const bool first = base == wv * lpw;
        const int ob = first ? pre_ob : P.lmk_ob[gl], oe = first ? pre_oe : P.lmk_oe[gl];
        const int nobs = lmk_valid ? oe - ob : 0;
        const int lcode = first ? pre_lcode : (P.lmk_const ? P.lmk_const[gl] : 0);
        const bool lfree = lcode == 0;
        double p0[3], x0[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            p0[i] = first ? pre_p[i] : P.lmk_p[3 * (long long)gl + i];
            x0[i] = first ? pre_x[i] : xl[3 * (long long)gl + i];
        }
        ObsLin L;
        L.valid = q < nobs;
        L.row = -1; L.slot = 0; L.counted = false;
        ObsPre obq = pre_obs;
        if (L.valid) {
            const double pw[3] = {p0[0] + x0[0], p0[1] + x0[1], p0[2] + x0[2]};
            if (!first) obq = obs_prefetch<FACTOR>(P, ob + q);
            lane_linearize<FACTOR, RARE>(P, poseTab, camTab, rowTab, T.cam_base, obq, pw, lcode != 1, lcode != 1, L);
            if (!L.counted) { L.r[0] = 0.0; L.r[1] = 0.0; }
        } else {
            L.r[0] = L.r[1] = 0.0;
#pragma unroll
            for (int i = 0; i < 12; i++) L.Jp[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) L.Jl[i] = 0.0;
        }
        double Mi[6], g[3];
        const bool active = group_eliminate(P, L, G, gl, lmk_valid, lfree, nobs, st.radius, false, false, Mi, g);

        // predicted residual e = r + Jp dp of this lane's observation
        double e0 = L.r[0], e1 = L.r[1];
        const double* d = dpTab + L.slot * 6;
        if (L.valid && L.row >= 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) { e0 += L.Jp[i] * d[i]; e1 += L.Jp[6 + i] * d[i]; }
        }
        // delta_l = -M^-1 sum_a Jl_a^T e_a = -Li^T (Li t)   (every lane of the group)
        const double tt[3] = {group_sum(L.Jl[0] * e0 + L.Jl[3] * e1, G), group_sum(L.Jl[1] * e0 + L.Jl[4] * e1, G),
                              group_sum(L.Jl[2] * e0 + L.Jl[5] * e1, G)};
        double ut[3], vt3[3];
        li_vec(Mi, tt, ut);
        li_tvec(Mi, ut, vt3);
        double d0 = -vt3[0], d1 = -vt3[1], d2 = -vt3[2];
        if (RARE && lcode == 2) {  // kept landmark: its step is part of the reduced solution (|step|^2 is counted there)
            const double* dr = P.delta + T.red_off + P.lmk_red[gl];
            d0 = dr[0]; d1 = dr[1]; d2 = dr[2];
        }
        const double c0 = x0[0] + d0, c1 = x0[1] + d1, c2 = x0[2] + d2;
        if (lmk_valid && q == 0) {
            xlc[3 * (long long)gl] = c0; xlc[3 * (long long)gl + 1] = c1; xlc[3 * (long long)gl + 2] = c2;
            if (active) {
                sn += d0 * d0 + d1 * d1 + d2 * d2;
                cn += c0 * c0 + c1 * c1 + c2 * c2;
            } else if (RARE && lcode == 2 && (P.world == 1 || P.rank == 0)) cn += c0 * c0 + c1 * c1 + c2 * c2;   // (sharded: every rank carries the kept landmarks, rank 0 counts them)
        }
        if (L.valid && L.counted) {
            // model cost change of this residual block: -(J d)^T (r + J d / 2)
            const double m0 = (e0 - L.r[0]) + L.Jl[0] * d0 + L.Jl[1] * d1 + L.Jl[2] * d2;
            const double m1 = (e1 - L.r[1]) + L.Jl[3] * d0 + L.Jl[4] * d1 + L.Jl[5] * d2;
            mcc += -m0 * (L.r[0] + 0.5 * m0) - m1 * (L.r[1] + 0.5 * m1);
            // residual at the candidate point
            const int craw = obq.craw;
            const int cam = craw < 0 ? 0 : craw - T.cam_base;
            const double* ct = camTab + cam * 17;
            const double pw[3] = {p0[0] + c0, p0[1] + c1, p0[2] + c2};
            const double* ctab = candTab + L.slot * 12;
            double r[2];
            if (RARE && craw < 0) {
                const int code = -1 - craw;
                const SparseDev& f = P.sparse[code >> 1];
                p2l_pseudo_obs<false>(ctab, pw, f.delta, f.W, code & 1, r, nullptr, nullptr);
                cc += r[0] * r[0] + r[1] * r[1];
            } else {
            if (FACTOR == 0) {
                pixel_factor<false>(ctab, ct, ct + 4, pw, obq.m[0], obq.m[1], ct[16], r, nullptr, nullptr);
            } else {
                double bb[3] = {obq.m[0], obq.m[1], obq.m[2]};
                angular_factor<false>(ctab, ct + 4, pw, bb, ct[16], r, nullptr, nullptr);
            }
            double sc_unused;
            cc += RARE ? huber_rho(P.o.huber_a, r[0] * r[0] + r[1] * r[1], sc_unused) : r[0] * r[0] + r[1] * r[1];
            }
        }
    }
    sn = wave_sum(sn); cn = wave_sum(cn); mcc = wave_sum(mcc); cc = wave_sum(cc);
    __shared__ double s_part[BUILD_WAVES * 4];
    if (ln == 0) { s_part[wv * 4] = cc; s_part[wv * 4 + 1] = mcc; s_part[wv * 4 + 2] = sn; s_part[wv * 4 + 3] = cn; }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int k = 0; k < BUILD_WAVES; k++) { a0 += s_part[k * 4]; a1 += s_part[k * 4 + 1]; a2 += s_part[k * 4 + 2]; a3 += s_part[k * 4 + 3]; }
        TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + blockIdx.x;
        ta->cand_cost = a0; ta->mcc = a1; ta->step_norm2 = a2; ta->cand_norm2 = a3;
    }
}

// ---- dense marginalisation prior (MarginalizationFactor, marginalization.hpp:113-215) on many workgroups ----------
// J (n_full x n, several MB) is streamed by the whole chip instead of one workgroup: per LM step
//   k_prior_r   r = r0 + J dx (one wave per row), cost = sum r^2                      -> dp scratch, dp cost slot
//   k_prior_gh  g = J^T r into gred / gfull, diag(J^T J) into hdiag (one wave per column); J^T J into S (one thread
//               per element, plain read-modify-write: nothing else touches S between k_build and k_solve)
//   k_prior_m   after the step: m = J delta (one wave per row) -> model cost change / candidate cost of the slot
// dp_data layout per window: J | J^T | J^T J | r0 | dx | r | cost(1) | - | per-row-block partial sums: cost | mcc | candidate cost (3 x ceil(nf / 4): sharded windows).
constexpr int DP_LDS_N = 2048;   // prior variables whose gathered deltas are staged in LDS by k_prior_r / k_prior_m (16 KB); larger priors gather per row
__device__ __forceinline__ double* dp_ptr(const DevPtrs& P, const WinDev& W, int which) {
    const size_t nf = W.dp_n_full, n = W.dp_n;
    double* D = P.dp_data + W.dp_off;
    switch (which) {
        case 0: return D;                                  // J
        case 1: return D + nf * n;                         // J^T
        case 2: return D + 2 * nf * n;                     // H
        case 3: return D + 2 * nf * n + n * n;             // r0
        case 4: return D + 2 * nf * n + n * n + nf;        // dx
        case 5: return D + 2 * nf * n + n * n + nf + n;    // r
        case 6: return D + 2 * nf * n + n * n + 2 * nf + n;  // cost
        default: return D + 2 * nf * n + n * n + 2 * nf + n + 2;  // partial sums of a sharded window: [3][ceil(nf / 4)]
    }
}

__global__ __launch_bounds__(256) void k_prior_r(DevPtrs P, int slot) {
    const int w = blockIdx.y;
    const WinDev W = P.win[w];
    if (W.dp_n_full == 0) return;
    const LmState st = P.states[(long long)w * P.state_stride + slot];
    if (st.done) return;
    const int n = W.dp_n, nf = W.dp_n_full, ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    __shared__ double s_c[4];
    __shared__ double dxs[DP_LDS_N];   // the gathered dx of the window (kind -> index -> delta array: two dependent loads), once per workgroup
    double c = 0.0;
    const int* kind = P.dp_ints + W.dp_int_off;
    const int* index = kind + n;
    const int cur = st.cur;
    const double* srcs[5] = {P.xp + (long long)cur * P.xp_stride, P.xv + (long long)cur * P.xv_stride, P.xba + (long long)cur * P.xv_stride,
                             P.xbg + (long long)cur * P.xv_stride, P.xl + (long long)cur * P.xl_stride};
    const bool staged = n <= DP_LDS_N;
    if (staged) {
#pragma unroll 4
        for (int a = threadIdx.x; a < n; a += 256) {
            const int k = kind[a];
            dxs[a] = k < 0 ? 0.0 : srcs[max(k, 0)][index[a]];
        }
        __syncthreads();
    }
    if (i < nf) {
        const double* Jr = dp_ptr(P, W, 0) + (size_t)i * n;
        double s = 0.0;
        if (staged) {
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int a = ln;
            for (; a + 192 < n; a += 256) {   // four row loads in flight per lane
                const double j0 = Jr[a], j1 = Jr[a + 64], j2 = Jr[a + 128], j3 = Jr[a + 192];
                s += j0 * dxs[a]; s1 += j1 * dxs[a + 64]; s2 += j2 * dxs[a + 128]; s3 += j3 * dxs[a + 192];
            }
            for (; a < n; a += 64) s += Jr[a] * dxs[a];
            s = (s + s1) + (s2 + s3);
        } else {
            for (int a = ln; a < n; a += 64) {
                const int k = kind[a];
                const double dx = k < 0 ? 0.0 : srcs[k][index[a]];
                s += Jr[a] * dx;
            }
        }
        s = wave_sum(s);
        if (ln == 0) { s += dp_ptr(P, W, 3)[i]; dp_ptr(P, W, 5)[i] = s; c = s * s; }
    }
    if (ln == 0) s_c[wv] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        // a window sharded over several GPUs: every rank evaluates the prior (its variables - the kept frame and the kept landmarks - are
        // replicated) and must get the SAME BITS for the cost, or the ranks' LM decisions drift apart: per-block partials, summed in
        // index order by k_solve, instead of atomic adds in scheduling order
        // (grid.x is the batch maximum of ceil(nf / 4): a block beyond THIS window's rows has no slot — its store would land in the
        // other partial regions or in the next window's prior)
        if (P.world > 1) { if ((int)blockIdx.x < (nf + 3) / 4) dp_ptr(P, W, 7)[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3]; }
        else atomic_add_f64(dp_ptr(P, W, 6), s_c[0] + s_c[1] + s_c[2] + s_c[3]);
    }
}

__global__ __launch_bounds__(256) void k_prior_gh(DevPtrs P, int slot, int col_blocks) {
    const int w = blockIdx.y;
    const WinDev W = P.win[w];
    if (W.dp_n_full == 0) return;
    if (P.world > 1 && P.rank != 0) return;   // sharded window: J^T J and J^T r enter the all-reduced system once
    if (P.states[(long long)w * P.state_stride + slot].done) return;
    const int n = W.dp_n, nf = W.dp_n_full;
    const int* col = P.dp_ints + W.dp_int_off + 2 * n;
    const double* H = dp_ptr(P, W, 2);
    if ((int)blockIdx.x < col_blocks) {
        const int ln = threadIdx.x & 63, a = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (a >= n) return;
        const int ca = col[a];
        if (ca < 0) return;
        const double* Jt = dp_ptr(P, W, 1);
        const double* r = dp_ptr(P, W, 5);
        double g = 0.0;
        for (int i = ln; i < nf; i += 64) g += Jt[(size_t)a * nf + i] * r[i];
        g = wave_sum(g);
        if (ln == 0) {
            P.gred[W.red_off + ca] += g; P.gfull[W.red_off + ca] += g; P.hdiag[W.red_off + ca] += H[(size_t)a * n + a];
        }
        return;
    }
    const long long idx = (long long)(blockIdx.x - col_blocks) * 256 + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const int a = (int)(idx / n), b = (int)(idx - (long long)a * n);
    if (b > a) return;
    const int ca = col[a], cb = col[b];
    if (ca < 0 || cb < 0) return;
    double* Sg = P.S + W.S_off;
    Sg[ca >= cb ? s_index(W.ld, ca, cb) : s_index(W.ld, cb, ca)] += H[idx];
}

__global__ __launch_bounds__(256) void k_prior_m(DevPtrs P, int slot) {
    const int w = blockIdx.y;
    const WinDev W = P.win[w];
    if (W.dp_n_full == 0) return;
    IterAcc* acc = P.acc + (long long)w * P.state_stride + slot;
    if (P.states[(long long)w * P.state_stride + slot].done || acc->chol_fail) return;
    const int n = W.dp_n, nf = W.dp_n_full, ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    __shared__ double s_m[4], s_c[4];
    double mc = 0.0, cc = 0.0;
    __shared__ double dls[DP_LDS_N];   // the step of every prior variable (col -> delta), gathered once per workgroup
    const int* col = P.dp_ints + W.dp_int_off + 2 * n;
    const double* dl = P.delta + W.red_off;
    const bool staged = n <= DP_LDS_N;
    if (staged) {
#pragma unroll 4
        for (int a = threadIdx.x; a < n; a += 256) {
            const int ca = col[a];
            dls[a] = ca >= 0 ? dl[max(ca, 0)] : 0.0;
        }
        __syncthreads();
    }
    if (i < nf) {
        const double* Jr = dp_ptr(P, W, 0) + (size_t)i * n;
        double m = 0.0;
        if (staged) {
            double m1 = 0.0, m2 = 0.0, m3 = 0.0;
            int a = ln;
            for (; a + 192 < n; a += 256) {
                const double j0 = Jr[a], j1 = Jr[a + 64], j2 = Jr[a + 128], j3 = Jr[a + 192];
                m += j0 * dls[a]; m1 += j1 * dls[a + 64]; m2 += j2 * dls[a + 128]; m3 += j3 * dls[a + 192];
            }
            for (; a < n; a += 64) m += Jr[a] * dls[a];
            m = (m + m1) + (m2 + m3);
        } else {
            for (int a = ln; a < n; a += 64) {
                const int ca = col[a];
                if (ca >= 0) m += Jr[a] * dl[ca];
            }
        }
        m = wave_sum(m);
        if (ln == 0) { const double r = dp_ptr(P, W, 5)[i]; mc = -m * (r + 0.5 * m); cc = (r + m) * (r + m); }
    }
    if (ln == 0) { s_m[wv] = mc; s_c[wv] = cc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (P.world > 1) {   // summed in index order by k_rank_partials (see k_prior_r)
            double* pp = dp_ptr(P, W, 7);
            const int nb = (nf + 3) / 4;
            if ((int)blockIdx.x < nb) {   // blocks beyond this window's rows (grid.x = the batch maximum) have no slot
                pp[nb + blockIdx.x] = s_m[0] + s_m[1] + s_m[2] + s_m[3];
                pp[2 * nb + blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
            }
        } else {
            atomic_add_f64(&acc->mcc, s_m[0] + s_m[1] + s_m[2] + s_m[3]);
            atomic_add_f64(&acc->cand_cost, s_c[0] + s_c[1] + s_c[2] + s_c[3]);
        }
    }
}

// Rows / columns of the prior-kept landmarks in the reduced system: one thread per observation adds
// Jl^T Jp (3x6), Jl^T Jl (3x3, lower), Jl^T r and the diagonal; few hundred landmarks per window at most.
template <int FACTOR>
__global__ __launch_bounds__(128) void k_build_kept(DevPtrs P, int slot) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n_kept) return;
    const int o = P.kept_obs[3 * e], gl = P.kept_obs[3 * e + 1], w = P.kept_obs[3 * e + 2];
    const LmState st = P.states[(long long)w * P.state_stride + slot];
    if (st.done) return;
    const WinDev W = P.win[w];
    const int cur = st.cur;
    const int kf = P.obs_kf[o], cam = P.obs_cam[o];
    const double* tab = P.ptab + (long long)cur * P.ptab_stride + (long long)kf * POSE_TAB;
    const double* xl = P.xl + (long long)cur * P.xl_stride + 3 * (long long)gl;
    const double pw[3] = {P.lmk_p[3 * (long long)gl] + xl[0], P.lmk_p[3 * (long long)gl + 1] + xl[1], P.lmk_p[3 * (long long)gl + 2] + xl[2]};
    double r[2], Jp[12], Jl[6];
    if (FACTOR == 0) {
        const double* m = P.obs_meas + 2 * (long long)o;
        pixel_factor<true>(tab, P.cam_K + 4 * (long long)cam, P.cam_T + 12 * (long long)cam, pw, m[0], m[1], P.cam_isig[cam], r, Jp, Jl);
    } else {
        const double* m = P.obs_meas + 3 * (long long)o;
        double b[3] = {m[0], m[1], m[2]};
        angular_factor<true>(tab, P.cam_T + 12 * (long long)cam, pw, b, P.cam_isig[cam], r, Jp, Jl);
    }
    {
        double sc;
        (void)huber_rho(P.o.huber_a, r[0] * r[0] + r[1] * r[1], sc);
        if (sc != 1.0) {
            r[0] *= sc; r[1] *= sc;
            for (int i = 0; i < 12; i++) Jp[i] *= sc;
            for (int i = 0; i < 6; i++) Jl[i] *= sc;
        }
    }
    const int fi = P.kf_fidx[kf];
    const int lr = P.lmk_red[gl];
    double* Sg = P.S + W.S_off;
    for (int a = 0; a < 3; a++) {
        const int row = lr + a;
        if (fi >= 0)
            for (int i = 0; i < 6; i++) atomic_add_f64(&Sg[s_index(W.ld, row, fi * W.dpf + i)], Jl[a] * Jp[i] + Jl[3 + a] * Jp[6 + i]);
        for (int b = 0; b <= a; b++) atomic_add_f64(&Sg[s_index(W.ld, row, lr + b)], Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b]);
        const double g = Jl[a] * r[0] + Jl[3 + a] * r[1];
        atomic_add_f64(&P.gred[W.red_off + row], g);
        atomic_add_f64(&P.gfull[W.red_off + row], g);
        atomic_add_f64(&P.hdiag[W.red_off + row], Jl[a] * Jl[a] + Jl[3 + a] * Jl[3 + a]);
    }
}

// where a listed sparse factor parks its cost at the candidate on a sharded window: the cost entry of the OTHER buffer's row, which
// the linearisation at that candidate (next k_build) fills with the same number; the row of x stays intact for a rejected step
__device__ __forceinline__ double& sparse_cand_cost_slot(const DevPtrs& P, int buf, int k) {
    return P.sp_scratch[(long long)buf * P.sp_scratch_stride + (long long)k * SPARSE_J + SPARSE_H + SPARSE_E_COST];
}

// Sharded window: this rank's sums over its own tiles into its slot of rank_b (which = 0, after k_build) or
// rank_s (which = 1, after k_backsub); the other ranks' slots are zeroed so that the all-reduce gathers.
__global__ void k_rank_partials(DevPtrs P, int slot, int which) {
    const int w = blockIdx.x, ln = threadIdx.x;
    const WinDev W = P.win[w];
    const TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + W.tile_begin;
    const int nt = W.tile_end - W.tile_begin;
    double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
    for (int t = ln; t < nt; t += 64) {
        if (which == 0) { a += ta[t].lin_cost; b += ta[t].fixed_cost; c = fmax(c, ta[t].gmax); }
        else { a += ta[t].cand_cost; b += ta[t].mcc; c += ta[t].step_norm2; d += ta[t].cand_norm2; }
    }
    a = wave_sum(a); b = wave_sum(b); c = which == 0 ? wave_max(c) : wave_sum(c); d = wave_sum(d);
    double* dst = (which == 0 ? P.rank_b : P.rank_s) + (long long)w * P.world * 4;
    for (int i = ln; i < P.world * 4; i += 64) dst[i] = 0.0;
    __syncthreads();
    if (ln == 0) { double* m = dst + 4 * P.rank; m[0] = a; m[1] = b; m[2] = c; m[3] = d; }
    if (which == 1 && ln == 0 && P.world > 1) {
        // candidate cost of the replicated pose-only factors (IMU pairs, listed sparse factors): their waves left it in their rows;
        // ONE thread adds them in index order, so that every rank's total - hence rho, the radius and the accept / reject
        // decision - has the same bits whatever order the waves ran in (atomic adds here let the ranks drift apart)
        const long long so = (long long)w * P.state_stride + slot;
        const LmState st = P.states[so];
        if (!st.done) {
            const int cb = 1 - st.cur;
            double cc = 0.0;
            for (int k = W.imu_begin; k < W.imu_end; k++) {
                if (P.kf_fidx[P.imus[k].kf_i] < 0 && P.kf_fidx[P.imus[k].kf_j] < 0) continue;   // all constant: part of the fixed cost
                cc += P.imu_scratch[(long long)cb * P.imu_scratch_stride + (long long)k * IMU_ROW + IMU_H + IMU_E_COST];
            }
            for (int q = W.spl_begin; q < W.spl_end; q++) cc += sparse_cand_cost_slot(P, cb, P.sp_list[q]);
            P.acc[so].cand_cost += cc;
            if (W.dp_n_full > 0 && !P.acc[so].chol_fail) {   // the dense prior's model cost change / candidate cost (k_prior_m), row blocks in index order
                const double* pp = dp_ptr(P, W, 7);
                const int nb = (W.dp_n_full + 3) / 4;
                double m = 0.0, c2 = 0.0;
                for (int b = 0; b < nb; b++) { m += pp[nb + b]; c2 += pp[2 * nb + b]; }
                P.acc[so].mcc += m; P.acc[so].cand_cost += c2;
            }
        }
    }
}

// Start of a solve: zero the delta buffers and every accumulator, write the initial LM state of each window
// (one launch instead of eight memsets and a host-to-device copy).
__device__ __forceinline__ void init_tables_item(const DevPtrs& P, int g, int n_kf_tot);

// One launch opens a solve: blocks [0, reset_blocks) zero the deltas / accumulators and write the initial LM state, the
// blocks behind them (64 work items each) compute the pose tables and the pose priors' linearisation records at x = 0
// (the two were separate launches: one boundary + ~4 us per solve).
__global__ __launch_bounds__(256) void k_reset(DevPtrs P, int reset_blocks, int n_kf_tot) {
    if ((int)blockIdx.x >= reset_blocks) {
        if (threadIdx.x < 64) init_tables_item(P, ((int)blockIdx.x - reset_blocks) * 64 + threadIdx.x, n_kf_tot);
        return;
    }
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)reset_blocks * blockDim.x;
    if (t == 0 && P.t_start) *P.t_start = wall_clock64();
    for (long long i = t; i < P.n_xp; i += nt) P.xp[i] = 0.0;
    for (long long i = t; i < P.n_xv; i += nt) { P.xv[i] = 0.0; P.xba[i] = 0.0; P.xbg[i] = 0.0; }
    for (long long i = t; i < P.n_xl; i += nt) P.xl[i] = 0.0;
    for (long long i = t; i < 2 * P.xline_stride; i += nt) P.xline[i] = 0.0;
    unsigned long long* a = (unsigned long long*)P.acc;
    const long long na = (long long)P.n_win * P.state_stride * (long long)(sizeof(IterAcc) / 8);
    for (long long i = t; i < na; i += nt) a[i] = 0ull;
    unsigned long long* ta = (unsigned long long*)P.tacc;
    const long long nta = 2LL * P.n_tiles * (long long)(sizeof(TileAcc) / 8);
    for (long long i = t; i < nta; i += nt) ta[i] = 0ull;
    const long long ns = (long long)P.n_win * P.state_stride;
    if (P.trace) for (long long i = t; i < ns * 8; i += nt) P.trace[i] = 0.0;
    for (long long i = t; i < ns; i += nt) {
        LmState z{};
        if (i % P.state_stride == 0) { z.radius = P.o.initial_radius; z.decrease_factor = 2.0; }
        P.states[i] = z;
    }
}

// Pose tables of delta buffer 0 at the start of a solve (all deltas zero).
__device__ __forceinline__ void init_tables_item(const DevPtrs& P, int g, int n_kf_tot) {
    if (g < P.n_prior_tot) {   // linearisation records of the pose priors at x = 0 (buffer 0): see DevPtrs::prior_lin
        const PriorDev pr = P.priors[g];
        double d6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        prior_lin_record(P.kf_T0 + 12 * (long long)pr.kf, pr.T_prior, pr.inf, d6, P.prior_lin + (long long)g * PRIOR_LIN);
    }
    if (g >= n_kf_tot) return;
    double d6[6], tab[POSE_TAB];
    for (int i = 0; i < 6; i++) d6[i] = 0.0;
    pose_table_entry(P.kf_T0 + 12 * (long long)g, d6, tab);
    for (int i = 0; i < POSE_TAB; i++) P.ptab[(long long)g * POSE_TAB + i] = tab[i];
}

// Decision of slot `slot` (one wave per window): state[slot + 1] = decide(state[slot], totals[slot]). Launched after
// every k_backsub when the batch has many tiles (P.decide_kernel), and always once at the end of the solve
// (final = 1), where it also writes the read-back record.
__global__ void k_decide(DevPtrs P, int slot, int final) {
    const int w = blockIdx.x, ln = threadIdx.x;
    __shared__ double s4[4];
    const WinDev W = P.win[w];
    LmState f;
    const bool already = final && P.decide_kernel;  // the per-slot launch of the last slot did it
    if (!already) {
        if (P.lm_sacc) {
            // throughput path: the partial records of k_lm_pass, every thread a fixed share, waves in index order (one decider per
            // window: any fixed order will do)
            __shared__ double sq[16 * 4];
            const long long n = (long long)(W.tile_end - W.tile_begin) * P.lm_ksub;
            const double* sa = P.lm_sacc + ((long long)(slot & 1) * P.n_tiles + W.tile_begin) * P.lm_ksub * 4;
            double c = 0.0, m = 0.0, sn = 0.0, cn = 0.0;
            for (long long t = ln; t < n; t += blockDim.x) { c += sa[4 * t]; m += sa[4 * t + 1]; sn += sa[4 * t + 2]; cn += sa[4 * t + 3]; }
            c = wave_sum(c); m = wave_sum(m); sn = wave_sum(sn); cn = wave_sum(cn);
            if ((ln & 63) == 0) { double* q = sq + 4 * (ln >> 6); q[0] = c; q[1] = m; q[2] = sn; q[3] = cn; }
            __syncthreads();
            if (ln < 4) {
                double v = 0.0;
                for (int q = 0; q < (int)(blockDim.x >> 6); q++) v += sq[4 * q + ln];
                s4[ln] = v;
            }
        } else
        if (blockDim.x == 64) wave_sum_backsub_partials(P, slot & 1, w, W.tile_begin, W.tile_end - W.tile_begin, ln, s4);
        else {
            // thousands of tiles (configs 4 / 5, large batches): every wave of the workgroup sums a slice of the partials
            __shared__ double sw[16 * 4];
            const int wv = ln >> 6, nw = blockDim.x >> 6, nt_w = W.tile_end - W.tile_begin;
            const int per = (nt_w + nw - 1) / nw, t0 = min(wv * per, nt_w), t1 = min(t0 + per, nt_w);
            if (P.world > 1) { if (wv == 0) wave_sum_backsub_partials(P, slot & 1, w, W.tile_begin, nt_w, ln & 63, sw); }
            else wave_sum_backsub_partials(P, slot & 1, w, W.tile_begin + t0, t1 - t0, ln & 63, sw + 4 * wv);
            __syncthreads();
            if (ln < 4) {
                double v = 0.0;
                for (int q = 0; q < (P.world > 1 ? 1 : nw); q++) v += sw[4 * q + ln];
                s4[ln] = v;
            }
        }
        __syncthreads();
    }
    if (ln == 0) {
        if (already) f = P.states[(long long)w * P.state_stride + slot + 1];
        else {
            IterAcc a = P.acc[(long long)w * P.state_stride + slot];
            a.cand_cost += s4[0]; a.mcc += s4[1]; a.step_norm2 += s4[2]; a.cand_norm2 += s4[3];
            const LmState prev = P.states[(long long)w * P.state_stride + slot];
            f = lm_decide(prev, a, P.o);
            P.states[(long long)w * P.state_stride + slot + 1] = f;
            trace_write(P, w, slot, prev, a, f);
        }
        if (final) {
            FinalRec rec;
            rec.s = f;
            rec.fixed_cost = P.acc[(long long)w * P.state_stride].fixed_cost;
            P.final_out[w] = rec;
        }
    }
}

// IMUFactor + IMUBiasFactor of one key-frame pair (residuals.hpp:133-300). Inside k_solve (512 threads, 128 VGPRs) this code lived in
// scratch memory and cost 49 us + 21 us per LM step on a 12-KF window; it is evaluated by workgroups of its own instead.
// What the pair adds to the reduced system (r, the whitened 9x24 Jacobian, J^T J, J^T r, the bias random walk, the cost: the
// scratch row of ba_types.h) is kept per delta buffer, like the pose priors' records (DevPtrs::prior_lin). No kernel of its
// own and no second stream inside the LM loop - a fork / join through a side stream was measured at the price of running the
// evaluation serially (~28 us per step) - the pairs ride the two tile kernels as extra workgroups:
//   imu_pair_cost   (k_backsub, slot s): the pair's cost at the candidate x + delta, added to the slot's cand_cost; one wave
//   imu_pair_lin_wg (k_build, slot s)  : the linearisation at the candidate of slot s - 1 (buffer 1 - cur; at x = 0 for slot 0)
//                 BESIDE the tiles that take that slot's accept / reject decision: it is what enters the reduced system if the step
//                 was accepted; a rejected step leaves the row of x (the other buffer) untouched. The whole workgroup.
__device__ __forceinline__ void imu_pair_cost(const DevPtrs& P, int slot, int k, int ln) {
    // the factor's constants (1.2 KB) come in with one coalesced copy: lane 0 evaluating the factor from global memory spends its
    // time on ~200 dependent scalar loads
    __shared__ ImuDev f;
    {
        const unsigned long long* src = (const unsigned long long*)(P.imus + k);
        unsigned long long* dst = (unsigned long long*)&f;
        for (int i = ln; i < (int)(sizeof(ImuDev) / 8); i += 64) dst[i] = src[i];
    }
    wave_lds_fence();
    const long long so = (long long)f.win * P.state_stride + slot;
    const LmState st = P.states[so];
    if (st.done) return;
    const int i = f.kf_i, j = f.kf_j;
    if (P.kf_fidx[i] < 0 && P.kf_fidx[j] < 0) return;   // both key-frames constant: the pair's cost is part of the fixed cost
    const int buf = 1 - st.cur;
    const double* xp = P.xp + (long long)buf * P.xp_stride;
    const double* xv = P.xv + (long long)buf * P.xv_stride;
    const double* xba = P.xba + (long long)buf * P.xv_stride;
    const double* xbg = P.xbg + (long long)buf * P.xv_stride;
    if (ln != 0) return;
    double dpi[6], dpj[6], r[9];
    ImuMid mid;   // nothing reads it: the stores fold away
    for (int q = 0; q < 6; q++) { dpi[q] = xp[6 * (long long)i + q]; dpj[q] = xp[6 * (long long)j + q]; }
    imu_residual_part(f, P.kf_T0 + 12 * (long long)i, P.kf_T0 + 12 * (long long)j, P.kf_vel + 3 * (long long)i,
                      P.kf_vel + 3 * (long long)j, dpi, dpj, xv + 3 * (long long)i, xv + 3 * (long long)j,
                      xba + 3 * (long long)i, xbg + 3 * (long long)i, r, mid);
    double c = 0.0;
    for (int q = 0; q < 9; q++) c += r[q] * r[q];
    double rba[3], rbg[3];
    for (int q = 0; q < 3; q++) {
        rba[q] = f.sa * (P.kf_ba[3 * (long long)j + q] + xba[3 * (long long)j + q] - P.kf_ba[3 * (long long)i + q] - xba[3 * (long long)i + q]);
        rbg[q] = f.sg * (P.kf_bg[3 * (long long)j + q] + xbg[3 * (long long)j + q] - P.kf_bg[3 * (long long)i + q] - xbg[3 * (long long)i + q]);
    }
    for (int q = 0; q < 3; q++) c += rba[q] * rba[q];
    for (int q = 0; q < 3; q++) c += rbg[q] * rbg[q];
    // sharded window: every rank must leave the step with the same bits (the ranks take the LM decisions independently), and
    // the order of atomic adds depends on workgroup scheduling - the pair's cost goes to its row (the slot the
    // linearisation at this candidate will fill with the same value) and k_rank_partials sums the rows in index order
    if (P.world > 1) P.imu_scratch[(long long)buf * P.imu_scratch_stride + (long long)k * IMU_ROW + IMU_H + IMU_E_COST] = c;
    else atomic_add_f64(&P.acc[so].cand_cost, c);
}

// The LINEARISATION of one IMU factor pair by a whole 256-thread workgroup (an extra workgroup of k_build, or of k_pf_eval<false>): the
// same arithmetic as imu_residual_part / imu_jacobian_part (device_math.h), dealt over lane 0 of the four waves - four dependent-
// latency-bound streams on four SIMDs instead of one (9.7 us on one lane: the pair's workgroup outlasted every tile of k_build) -
// with the products that only need many independent dot products (W J, J^T J, J^T r) spread over all threads.
//   phase 1  wave 0: R_i, t_i | wave 1: R_j, t_j | wave 2: the bias-corrected DeltaR, Jr(J_dR_bg dbg) | wave 3: the LM decision of the
//            previous slot (P.imu_direct, see below), Jr(w_i), Jr(w_j), the velocity term, the bias random walk
//   phase 2  wave 0: r_dR, Jr(r_dR)^-1 | wave 1: r_dv | wave 2: r_dp | wave 3: the Jacobian blocks that need neither
//   phase 3  r = W e (9 lanes); the remaining Jacobian blocks, one chain of 3 x 3 products per wave
//   phase 4  J <- W J, one entry per thread; the pair's cost
//   phase 5  what the pair adds to the reduced system, entry by entry (the scratch row of ba_types.h), and - one device
//            (P.imu_direct) - the adds themselves: straight into the accumulators in HBM (S, gred, gfull, hdiag: the tiles, the kept
//            landmarks and the dense prior add into them as well) instead of through k_solve's item loop, whose 11 x 354 LDS atomics
//            by ONE workgroup were 2.3 us of every VIO step. Which row is added is the decision of the previous slot: accepted ->
//            the row just evaluated (the candidate's), rejected -> the row of x kept in the other buffer. The decision is re-derived
//            exactly as the tiles of this launch do (same bits: wave_sum_backsub_partials), or read if k_decide took it.
static_assert(BUILD_THREADS == 256, "imu_pair_lin_wg deals an IMU pair over exactly four waves (k_build's extra workgroups, k_pf_eval<false>)");
__device__ __forceinline__ void imu_pair_lin_wg(const DevPtrs& P, int slot, int k, int tid) {
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, nthr = blockDim.x;
    __shared__ ImuDev f;
    __shared__ double U[9 * 24], UW[9 * 24];
    __shared__ double rs[9], rbs[6], e9[9], s_cost;
    __shared__ ImuMid mid;
    __shared__ double s_ti[3], s_vi[3], s_DRc[9], s_Jrb[9], s_Jrw[18], s_Jr_ri[9], s4[4];
    __shared__ int s_add;
#ifdef SADVIO_KERNEL_TS
    if ((P.debug & 4096) && k == 0 && tid == 0 && slot == 3) P.dbg_ts[56] = wall_clock64();
#endif
    {
        const unsigned long long* src = (const unsigned long long*)(P.imus + k);
        unsigned long long* dst = (unsigned long long*)&f;
        for (int q = tid; q < (int)(sizeof(ImuDev) / 8); q += nthr) dst[q] = src[q];
        for (int q = tid; q < 9 * 24; q += nthr) U[q] = 0.0;
    }
    __syncthreads();
    const bool at_x = slot == 0;
    const long long so = (long long)f.win * P.state_stride + (slot > 0 ? slot - 1 : slot);
    const LmState st = P.states[so];
    if (st.done) return;
    const int i = f.kf_i, j = f.kf_j;
    const int fi = P.kf_fidx[i], fj = P.kf_fidx[j];
    const bool all_const = fi < 0 && fj < 0;
    const int buf = at_x ? st.cur : 1 - st.cur;
    const double* xp = P.xp + (long long)buf * P.xp_stride;
    const double* xv = P.xv + (long long)buf * P.xv_stride;
    const double* xba = P.xba + (long long)buf * P.xv_stride;
    const double* xbg = P.xbg + (long long)buf * P.xv_stride;
    double* sc = P.imu_scratch + (long long)buf * P.imu_scratch_stride + (long long)k * IMU_ROW;
    const double* Ti0 = P.kf_T0 + 12 * (long long)i;
    const double* Tj0 = P.kf_T0 + 12 * (long long)j;
    const WinDev& Wd = P.win[f.win];
    const bool direct = P.imu_direct != 0;
    const double G[3] = {0.0, 0.0, -9.81};  // IMU.h:8
    const double dt = f.dt;
    // ---- phase 1 ----
    if (wv == 3) {
        int add_buf = direct ? buf : -1;       // the buffer whose row goes into the reduced system of this slot (-1: none)
        if (direct && slot > 0) {
            LmState nst;
            if (P.decide_kernel) nst = P.states[so + 1];
            else {
                wave_sum_backsub_partials(P, (slot - 1) & 1, f.win, Wd.tile_begin, Wd.tile_end - Wd.tile_begin, ln, s4);
                wave_lds_fence();
                IterAcc a = P.acc[so];
                a.cand_cost += s4[0]; a.mcc += s4[1]; a.step_norm2 += s4[2]; a.cand_norm2 += s4[3];
                nst = lm_decide(st, a, P.o);
            }
            add_buf = nst.done ? -1 : nst.cur;
        }
        if (ln == 0) s_add = add_buf;
    }
    if (ln == 0) {
        if (wv == 0) {
            double dpi[6], dRi[9], Ri[9], ti[3];
            for (int q = 0; q < 6; q++) dpi[q] = xp[6 * (long long)i + q];
            so3_exp(dpi, dRi);
            m3_mul(Ti0, dRi, Ri);
            m3_vec(Ti0, dpi + 3, ti);
            for (int q = 0; q < 3; q++) s_ti[q] = ti[q] + Ti0[9 + q];
            for (int q = 0; q < 9; q++) mid.Ri[q] = Ri[q];
        } else if (wv == 1) {
            double dpj[6], dRj[9], Rj[9], tj[3];
            for (int q = 0; q < 6; q++) dpj[q] = xp[6 * (long long)j + q];
            so3_exp(dpj, dRj);
            m3_mul(Tj0, dRj, Rj);
            m3_vec(Tj0, dpj + 3, tj);
            for (int q = 0; q < 3; q++) mid.tj[q] = tj[q] + Tj0[9 + q];
            for (int q = 0; q < 9; q++) { mid.Rj[q] = Rj[q]; mid.dRj[q] = dRj[q]; }
        } else if (wv == 2) {
            double dbg[3], jb[3], Eb[9], DRc[9], Jrb[9];
            for (int q = 0; q < 3; q++) dbg[q] = xbg[3 * (long long)i + q];
            m3_vec(f.J_dR_bg, dbg, jb);
            so3_exp(jb, Eb);
            m3_mul(f.dR, Eb, DRc);
            so3_right_jacobian(jb, Jrb);
            for (int q = 0; q < 3; q++) mid.jb[q] = jb[q];
            for (int q = 0; q < 9; q++) { s_DRc[q] = DRc[q]; s_Jrb[q] = Jrb[q]; }
        } else {
            double wi[3], wj[3], Jrwi[9], Jrwj[9];
            for (int q = 0; q < 3; q++) { wi[q] = xp[6 * (long long)i + q]; wj[q] = xp[6 * (long long)j + q]; }
            so3_right_jacobian(wi, Jrwi);
            so3_right_jacobian(wj, Jrwj);
            for (int q = 0; q < 9; q++) { s_Jrw[q] = Jrwi[q]; s_Jrw[9 + q] = Jrwj[q]; }
            for (int q = 0; q < 3; q++) {
                const double vi = P.kf_vel[3 * (long long)i + q] + xv[3 * (long long)i + q];
                const double vj = P.kf_vel[3 * (long long)j + q] + xv[3 * (long long)j + q];
                s_vi[q] = vi;
                mid.a[q] = vj - vi - G[q] * dt;
            }
            for (int q = 0; q < 3; q++) {   // the bias random walk (IMUBiasFactor)
                const double rba = f.sa * (P.kf_ba[3 * (long long)j + q] + xba[3 * (long long)j + q] - P.kf_ba[3 * (long long)i + q] - xba[3 * (long long)i + q]);
                const double rbg = f.sg * (P.kf_bg[3 * (long long)j + q] + xbg[3 * (long long)j + q] - P.kf_bg[3 * (long long)i + q] - xbg[3 * (long long)i + q]);
                sc[IMU_J + q] = rba; sc[IMU_J + 3 + q] = rbg; rbs[q] = rba; rbs[3 + q] = rbg;
            }
        }
    }
    __syncthreads();
    // ---- phase 2 ----
    if (ln == 0) {
        if (wv == 0) {
            double RiRjT[9], dR[9], r_dr[3], Jr_r[9], Jr_ri[9];
            m3_mul_t(mid.Ri, mid.Rj, RiRjT);
            m3_tmul(s_DRc, RiRjT, dR);   // dR = (DeltaR exp(J_dR_bg dbg))^T R_i R_j^T  (:157-158)
            so3_log(dR, r_dr);
            so3_right_jacobian(r_dr, Jr_r);
            m3_inverse(Jr_r, Jr_ri);
            for (int q = 0; q < 9; q++) { mid.RiRjT[q] = RiRjT[q]; mid.dR[q] = dR[q]; s_Jr_ri[q] = Jr_ri[q]; }
            for (int q = 0; q < 3; q++) { mid.r_dr[q] = r_dr[q]; e9[q] = r_dr[q]; }
        } else if (wv == 1) {
            double dba[3], dbg[3], Ra[3], t1[3], t2[3];
            for (int q = 0; q < 3; q++) { dba[q] = xba[3 * (long long)i + q]; dbg[q] = xbg[3 * (long long)i + q]; }
            m3_vec(mid.Ri, mid.a, Ra);
            m3_vec(f.J_dv_bg, dbg, t1); m3_vec(f.J_dv_ba, dba, t2);
            for (int q = 0; q < 3; q++) e9[3 + q] = Ra[q] - (f.dv[q] + t1[q] + t2[q]);
        } else if (wv == 2) {
            // positions in world: p = -R^T t, T.inverse().translation() with Eigen::Affine3d semantics (the linear part inverted as a general 3x3)
            double dba[3], dbg[3], Rii[9], Rji[9], pi[3], pj[3], b[3], Rb[3], t1[3], t2[3];
            for (int q = 0; q < 3; q++) { dba[q] = xba[3 * (long long)i + q]; dbg[q] = xbg[3 * (long long)i + q]; }
            m3_inverse(mid.Ri, Rii); m3_inverse(mid.Rj, Rji);
            m3_vec(Rii, s_ti, pi); m3_vec(Rji, mid.tj, pj);
            for (int q = 0; q < 3; q++) { pi[q] = -pi[q]; pj[q] = -pj[q]; }
            for (int q = 0; q < 3; q++) b[q] = pj[q] - pi[q] - s_vi[q] * dt - 0.5 * G[q] * dt * dt;
            m3_vec(mid.Ri, b, Rb);
            m3_vec(f.J_dp_bg, dbg, t1); m3_vec(f.J_dp_ba, dba, t2);
            for (int q = 0; q < 3; q++) {
                e9[6 + q] = Rb[q] - (f.dp[q] + t1[q] + t2[q]);
                mid.cc[q] = pj[q] - s_vi[q] * dt - 0.5 * G[q] * dt * dt;
            }
        } else {
            double S[9], RS[9], B[9];
            so3_skew(mid.a, S); m3_mul(mid.Ri, S, RS); m3_mul(RS, s_Jrw, B);       // pose_i, rows 3..5 (:174-187)
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) U[(3 + a) * 24 + b] = -B[3 * a + b];
            m3_mul_t(mid.Ri, mid.dRj, B);                                          // pose_j translation: exp(w_j)^T as coded (:198)
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) U[(6 + a) * 24 + 9 + b] = -B[3 * a + b];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {             // dv_i, dv_j, dba, dbg (:203-237)
                U[(3 + a) * 24 + 12 + b] = -mid.Ri[3 * a + b];
                U[(6 + a) * 24 + 12 + b] = -mid.Ri[3 * a + b] * dt;
                U[(3 + a) * 24 + 15 + b] = mid.Ri[3 * a + b];
                U[(3 + a) * 24 + 18 + b] = -f.J_dv_ba[3 * a + b];
                U[(6 + a) * 24 + 18 + b] = -f.J_dp_ba[3 * a + b];
                U[(3 + a) * 24 + 21 + b] = -f.J_dv_bg[3 * a + b];
                U[(6 + a) * 24 + 21 + b] = -f.J_dp_bg[3 * a + b];
            }
        }
    }
    __syncthreads();
#ifdef SADVIO_KERNEL_TS
    if ((P.debug & 4096) && k == 0 && tid == 0 && slot == 3) P.dbg_ts[57] = wall_clock64();
#endif
    // ---- phase 3 ----
    if (wv == 0 && ln < 9) {
        double r = 0.0;
        for (int q = 0; q < 9; q++) r += f.W[9 * ln + q] * e9[q];
        rs[ln] = r; sc[9 * 24 + ln] = r;
    }
    if (ln == 0) {
        double A[9], B[9];
        if (wv == 0) {
            m3_mul(s_Jr_ri, mid.Rj, A); m3_mul(A, s_Jrw, B);                       // pose_i rotation (:174-176)
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) U[a * 24 + b] = B[3 * a + b];
        } else if (wv == 1) {
            double S[9], C1[9], C2[9];
            m3_mul(s_Jr_ri, mid.Rj, A); m3_mul(A, s_Jrw + 9, B);                   // pose_j (:190-200)
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) U[a * 24 + 6 + b] = -B[3 * a + b];
            so3_skew(mid.tj, S); m3_mul(mid.RiRjT, S, C1); m3_mul(C1, mid.Rj, C2); m3_mul(C2, s_Jrw + 9, B);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) U[(6 + a) * 24 + 6 + b] = -B[3 * a + b];
        } else if (wv == 2) {
            double S[9], RS[9];
            so3_skew(mid.cc, S); m3_mul(mid.Ri, S, RS); m3_mul(RS, s_Jrw, B);      // pose_i, rows 6..8: p_j instead of p_j - p_i, R_i0 (:181-185)
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { U[(6 + a) * 24 + b] = -B[3 * a + b]; U[(6 + a) * 24 + 3 + b] = Ti0[3 * a + b]; }
        } else {
            double D1[9], D2[9];
            m3_mul_t(s_Jr_ri, mid.dR, A);                                          // Jr^-1 dR^T
            m3_mul(A, s_Jrb, D1); m3_mul(D1, f.J_dR_bg, D2);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) U[a * 24 + 21 + b] = -D2[3 * a + b];
        }
    }
    __syncthreads();
#ifdef SADVIO_KERNEL_TS
    if ((P.debug & 4096) && k == 0 && tid == 0 && slot == 3) P.dbg_ts[58] = wall_clock64();
#endif
    // ---- phase 4: J <- W J (W is upper triangular: L^T), the pair's cost ----
    if (tid < 9 * 24) {
        const int q = tid / 24, c = tid - 24 * q;
        double v = 0.0;
        for (int kk = q; kk < 9; kk++) v += f.W[9 * q + kk] * U[kk * 24 + c];
        UW[tid] = v; sc[tid] = v;
    } else if (tid == 9 * 24) {
        double c = 0.0;
        for (int q = 0; q < 9; q++) c += rs[q] * rs[q];
        for (int q = 0; q < 3; q++) c += rbs[q] * rbs[q];
        for (int q = 0; q < 3; q++) c += rbs[3 + q] * rbs[3 + q];
        s_cost = c;
    }
    __syncthreads();
#ifdef SADVIO_KERNEL_TS
    if ((P.debug & 4096) && k == 0 && tid == 0 && slot == 3) P.dbg_ts[59] = wall_clock64();
#endif
    // ---- phase 5 ----
    const int add_buf = s_add;
    const bool fresh = add_buf == buf;
    auto emit = [&](int e, double v, int ix) {   // entry e of the pair (position ix) into the reduced system
        if (ix < 0) return;
        const int ca = ix >> 16, cb = ix & 0xffff;
        if (e < IMU_E_G || (e >= IMU_E_BH && e < IMU_E_BG)) {
            atomic_add_f64(P.S + Wd.S_off + s_index(Wd.ld, ca, cb), v);
            if (ca == cb) atomic_add_f64(P.hdiag + Wd.red_off + ca, v);
        } else {
            atomic_add_f64(P.gred + Wd.red_off + ca, v);
            atomic_add_f64(P.gfull + Wd.red_off + ca, v);
        }
    };
    if (!fresh && add_buf >= 0) {   // rejected step: the row of x (written by an earlier launch)
        const double* ro = P.imu_scratch + (long long)add_buf * P.imu_scratch_stride + (long long)k * IMU_ROW;
        for (int e = tid; e < IMU_E_COST; e += nthr) emit(e, ro[IMU_H + e], (int)ro[IMU_IX + e]);
    }
    // H = J^T J (lower triangle, 300), g = J^T r (24) and where each entry goes in the reduced system
    for (int e = tid; e < IMU_E_BH; e += nthr) {
        double v = 0.0;
        int ix = -1;
        if (e < IMU_E_G) {
            int a = 0, b = e;
            while (b >= a + 1) { b -= a + 1; a++; }
#pragma unroll
            for (int q = 0; q < 9; q++) v += UW[q * 24 + a] * UW[q * 24 + b];
            const int ca = imu_col(a, fi, fj), cb = imu_col(b, fi, fj);
            if (ca >= 0 && cb >= 0) ix = ca >= cb ? (ca << 16) | cb : (cb << 16) | ca;
        } else {
            const int a = e - IMU_E_G;
#pragma unroll
            for (int q = 0; q < 9; q++) v += UW[q * 24 + a] * rs[q];
            const int ca = imu_col(a, fi, fj);
            if (ca >= 0) ix = ca << 16;
        }
        sc[IMU_H + e] = v;
        sc[IMU_IX + e] = (double)ix;
        if (fresh) emit(e, v, ix);
    }
    // the bias random walk (Jacobians -/+ s I: entries s^2, -s^2 and -/+ s r), the pair's cost, what k_solve's model-cost pass reads:
    // the last wave (the first ones have a second round of entries above)
    const int t = tid - (nthr - 64);
    if (t >= 0 && t < 30) {
        const int e = t < 18 ? t : t - 18;                  // matrix entries 0..17 = (combo, kind 0..2); gradient 0..11 = (combo, i | j)
        const int combo = t < 18 ? e / 3 : e / 2, kind = t < 18 ? e % 3 : e % 2;
        const int ax = combo % 3, gy = combo / 3;
        const double sgm = gy ? f.sg : f.sa, s2 = sgm * sgm;
        const int ci = fi < 0 ? -1 : fi * 15 + 9 + 3 * gy + ax, cj = fj < 0 ? -1 : fj * 15 + 9 + 3 * gy + ax;
        double v; int ix = -1;
        if (t < 18) {
            if (kind == 0) { v = s2; if (ci >= 0) ix = (ci << 16) | ci; }
            else if (kind == 1) { v = s2; if (cj >= 0) ix = (cj << 16) | cj; }
            else { v = -s2; if (ci >= 0 && cj >= 0) ix = ci >= cj ? (ci << 16) | cj : (cj << 16) | ci; }
            sc[IMU_H + IMU_E_BH + e] = v; sc[IMU_IX + IMU_E_BH + e] = (double)ix;
            if (fresh) emit(IMU_E_BH + e, v, ix);
        } else {
            const double rb = rbs[3 * gy + ax];
            if (kind == 0) { v = -sgm * rb; if (ci >= 0) ix = ci << 16; }
            else { v = sgm * rb; if (cj >= 0) ix = cj << 16; }
            sc[IMU_H + IMU_E_BG + e] = v; sc[IMU_IX + IMU_E_BG + e] = (double)ix;
            if (fresh) emit(IMU_E_BG + e, v, ix);
        }
    } else if (t == 30) {
        sc[IMU_H + IMU_E_COST] = s_cost; sc[IMU_IX + IMU_E_COST] = all_const ? -2.0 : -3.0;
    } else if (t == 31) {
        sc[IMU_META] = (double)fi; sc[IMU_META + 1] = (double)fj; sc[IMU_META + 2] = f.sa; sc[IMU_META + 3] = f.sg;
    }
#ifdef SADVIO_KERNEL_TS
    if ((P.debug & 4096) && k == 0 && tid == 0 && slot == 3) P.dbg_ts[60] = wall_clock64();
#endif
}

// Sparse prior factors that the solve evaluates itself (IMUPriordx, landmark priors / chains), ONE WAVE per listed factor, the
// same split as imu_pair_cost / imu_pair_lin_wg (extra workgroups of the tile kernels, rows kept per delta buffer):
//   COST_ONLY = true  (k_backsub, slot s): cost at the candidate x + delta (delta = the reduced step k_solve left in P.delta; the
//                 landmark candidates are being written by the tiles of the same launch), added to the slot's cand_cost
//   COST_ONLY = false (k_build, slot s)  : r, J into the scratch row of the candidate buffer of slot s - 1 (of x = 0 at slot 0)
template <bool COST_ONLY>
__device__ __forceinline__ void sparse_factor_eval(const DevPtrs& P, int slot, int k, int ln) {
    constexpr bool LIN = !COST_ONLY;
    __shared__ SparseDev f;   // one coalesced copy of the factor's constants (1.9 KB, the 15 x 15 square-root information) instead of lane 0's scalar loads
    {
        const unsigned long long* src = (const unsigned long long*)(P.sparse + k);
        unsigned long long* dst = (unsigned long long*)&f;
        for (int i = ln; i < (int)(sizeof(SparseDev) / 8); i += 64) dst[i] = src[i];
    }
    wave_lds_fence();
    const WinDev& W = P.win[f.win];
    const long long so = (long long)f.win * P.state_stride + (LIN && slot > 0 ? slot - 1 : slot);
    const LmState st = P.states[so];
    if (st.done) return;
    const int cur = (LIN && slot > 0) ? 1 - st.cur : st.cur;   // LIN: the buffer that is evaluated; COST_ONLY: the buffer of x
    const double* xp = P.xp + (long long)cur * P.xp_stride;
    const double* xv = P.xv + (long long)cur * P.xv_stride;
    const double* xba = P.xba + (long long)cur * P.xv_stride;
    const double* xbg = P.xbg + (long long)cur * P.xv_stride;
    const double* xl = P.xl + (long long)cur * P.xl_stride;
    __shared__ double Js[225];
    __shared__ double rs[16];
    __shared__ int s_in;
    double* sc = P.sp_scratch + (long long)cur * P.sp_scratch_stride + (long long)k * SPARSE_J;
    const int rows = sparse_rows(f);
    if (f.type == 0) {
        // IMUPriordx (residuals.hpp:634-700): lane 0 evaluates the 6-row pose part, the 15 x 15 whitening (r = W e, the pose columns of
        // J = W[:, :6] J6) is one row per lane. Done by one lane the two products live in scratch memory: 43 us for one factor.
        __shared__ double e_s[15], J6_s[36];
        const int fi = P.kf_fidx[f.kf];
        const double* y = LIN ? nullptr : P.delta + W.red_off;
        if (ln == 0) {
            const long long kk = f.kf;
            double prm[15], e6[6], J6[36];
            const bool st15 = y && fi >= 0 && W.dpf == 15;
            for (int q = 0; q < 6; q++) prm[q] = xp[6 * kk + q] + ((y && fi >= 0) ? y[fi * W.dpf + q] : 0.0);
            for (int q = 0; q < 3; q++) {
                prm[6 + q] = xv[3 * kk + q] + (st15 ? y[fi * 15 + 6 + q] : 0.0);
                prm[9 + q] = xba[3 * kk + q] + (st15 ? y[fi * 15 + 9 + q] : 0.0);
                prm[12 + q] = xbg[3 * kk + q] + (st15 ? y[fi * 15 + 12 + q] : 0.0);
            }
            const double ones[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
            pose_prior_factor(P.kf_T0 + 12 * kk, f.T_prior, ones, prm, e6, LIN ? J6 : nullptr);
            for (int q = 0; q < 6; q++) e_s[q] = e6[q];
            for (int q = 0; q < 3; q++) {
                e_s[6 + q] = P.kf_vel[3 * kk + q] + prm[6 + q] - f.v_prior[q];
                e_s[9 + q] = P.kf_ba[3 * kk + q] + prm[9 + q] - f.ba_prior[q];
                e_s[12 + q] = P.kf_bg[3 * kk + q] + prm[12 + q] - f.bg_prior[q];
            }
            if (LIN) for (int q = 0; q < 36; q++) J6_s[q] = J6[q];
            s_in = fi >= 0 ? 1 : 0;
        }
        wave_lds_fence();
        if (ln < 15) {
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < 15; q++) r += f.W[ln * 15 + q] * e_s[q];
            rs[ln] = s_in ? r : 0.0;     // a factor on a constant key-frame is not in the program
            if (LIN) {
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < 6; q++) v += f.W[ln * 15 + q] * J6_s[q * 6 + a];
                    Js[ln * 15 + a] = v;
                }
#pragma unroll
                for (int a = 6; a < 15; a++) Js[ln * 15 + a] = (ln == a) ? 1.0 : 0.0;
            }
        }
        wave_lds_fence();
        if (!LIN && ln == 0) {
            double c = 0.0;
            for (int q = 0; q < 15; q++) c += rs[q] * rs[q];
            if (P.world > 1) sparse_cand_cost_slot(P, 1 - cur, k) = s_in ? c : 0.0;   // summed in index order by k_rank_partials (see imu_pair_cost)
            else if (s_in) atomic_add_f64(&P.acc[so].cand_cost, c);
        }
    } else
    if (ln == 0) {
        double r[15];
        for (int q = 0; q < 15; q++) r[q] = 0.0;
        const bool in_program = sparse_eval_t<true>(P, W, f, xp, xv, xba, xbg, xl, LIN ? nullptr : P.delta + W.red_off, r, LIN ? Js : nullptr);
        s_in = in_program ? 1 : 0;
        if (LIN) { for (int q = 0; q < 15; q++) rs[q] = r[q]; }
        else {
            double c = 0.0;
            for (int q = 0; q < rows; q++) c += r[q] * r[q];
            if (P.world > 1) sparse_cand_cost_slot(P, 1 - cur, k) = in_program ? c : 0.0;
            else if (in_program) atomic_add_f64(&P.acc[so].cand_cost, c);
        }
    }
    if (LIN) {
        wave_lds_fence();
        if (s_in) for (int e = ln; e < rows * 15; e += 64) sc[e] = Js[e];
        if (ln < 15) sc[225 + ln] = rs[ln];
        if (ln == 0) sc[240] = (double)s_in;
        // what the factor adds to the reduced system, entry by entry with its position (k_solve: one uniform pass)
        const int fi = f.kf >= 0 ? P.kf_fidx[f.kf] : -1;
        const int lr0 = sparse_lr0(P, W, f);
        const int lr1 = (f.lmk1 >= 0 && P.lmk_red) ? P.lmk_red[f.lmk1] : -1;
        for (int e = ln; e < SPARSE_NE; e += 64) {
            double v = 0.0;
            int ix = -1;
            if (e < SPARSE_E_G) {
                int a = 0, b = e;
                while (b >= a + 1) { b -= a + 1; a++; }
                const int ca = sparse_col(f, a, fi, W.dpf, lr0, lr1), cb = sparse_col(f, b, fi, W.dpf, lr0, lr1);
                if (s_in && ca >= 0 && cb >= 0) {
                    for (int q = 0; q < rows; q++) v += Js[q * 15 + a] * Js[q * 15 + b];
                    ix = ca >= cb ? (ca << 16) | cb : (cb << 16) | ca;
                }
            } else if (e < SPARSE_E_COST) {
                const int a = e - SPARSE_E_G;
                const int ca = sparse_col(f, a, fi, W.dpf, lr0, lr1);
                if (s_in && ca >= 0) {
                    for (int q = 0; q < rows; q++) v += Js[q * 15 + a] * rs[q];
                    ix = ca << 16;
                }
                sc[SPARSE_COL + a] = (double)(s_in ? ca : -1);
            } else {
                for (int q = 0; q < rows; q++) v += rs[q] * rs[q];
                ix = s_in ? -3 : -2;
                sc[SPARSE_COL + 15] = (double)rows;
            }
            sc[SPARSE_H + e] = v;
            sc[SPARSE_IX + e] = (double)ix;
        }
    }
}

// extra workgroup `idx` of k_build (COST_ONLY = false) / k_backsub (true): IMU factor pairs first, then the listed sparse-prior factors
// COST_ONLY: the first wave of the workgroup calls it (ln = lane); the linearisation is called by EVERY thread of a 256-thread workgroup
// (ln = thread index: an IMU pair is dealt over the four waves, imu_pair_lin_wg; a sparse factor takes the first wave)
template <bool COST_ONLY>
__device__ __forceinline__ void pose_factor_eval(const DevPtrs& P, int slot, int idx, int ln) {
    if (COST_ONLY) {
        if (idx < P.n_imu_tot) imu_pair_cost(P, slot, idx, ln);
        else sparse_factor_eval<true>(P, slot, P.sp_list[idx - P.n_imu_tot], ln);
    } else {
        if (idx < P.n_imu_tot) imu_pair_lin_wg(P, slot, idx, ln);
        else if (ln < 64) sparse_factor_eval<false>(P, slot, P.sp_list[idx - P.n_imu_tot], ln);
    }
}
// the same as a kernel of its own (one 64-lane workgroup per factor): large batches on the throughput kernels, whose tile kernels
// carry no extra workgroups
template <bool COST_ONLY>
__global__ __launch_bounds__(COST_ONLY ? 64 : BUILD_THREADS) void k_pf_eval(DevPtrs P, int slot) { pose_factor_eval<COST_ONLY>(P, slot, blockIdx.x, threadIdx.x); }

// linexd observations, one 64-lane workgroup per observation, kernels of their own on a side stream: LIN -> J (rows x 12:
// key-frame | line), r, loss-corrected cost and the in-program flag into the scratch row at x; !LIN -> cost at the candidate
// (k_solve left the candidate key-frame and line deltas in the other buffer). Lines are few (tens per window).
template <bool LIN>
__global__ __launch_bounds__(64) void k_line_eval(DevPtrs P, int slot, int own_decide) {
    const int k = blockIdx.x, ln = threadIdx.x;
    const LineObsDev& ob = P.lobs[k];
    const WinDev& W = P.win[ob.win];
    const long long so = (long long)ob.win * P.state_stride + slot;
    LmState st;
    if (LIN && own_decide && slot > 0 && !P.decide_kernel) {
        __shared__ double s4[4];
        wave_sum_backsub_partials(P, (slot - 1) & 1, ob.win, W.tile_begin, W.tile_end - W.tile_begin, ln, s4);
        __syncthreads();
        IterAcc a = P.acc[so - 1];
        a.cand_cost += s4[0]; a.mcc += s4[1]; a.step_norm2 += s4[2]; a.cand_norm2 += s4[3];
        st = lm_decide(P.states[so - 1], a, P.o);
    } else st = P.states[so];
    if (st.done || ln != 0) return;
    const LineDev& L = P.lines[ob.line];
    const int fi = P.kf_fidx[ob.kf];
    const bool in_program = fi >= 0 || L.col >= 0;
    if (!LIN && !in_program) return;
    const int buf = LIN ? st.cur : 1 - st.cur;
    const double* dp = P.xp + (long long)buf * P.xp_stride + 6 * (long long)ob.kf;
    const double* dl = P.xline + (long long)buf * P.xline_stride + 6 * (long long)ob.line;
    double d6[6], l6[6], r[4] = {0.0, 0.0, 0.0, 0.0}, Jf[24], Jl[24];
    for (int q = 0; q < 6; q++) { d6[q] = dp[q]; l6[q] = dl[q]; }
    const bool pixel = W.factor_type == 0;
    const int rows = pixel ? 4 : 2;
    const double* Tsf = P.cam_T + 12 * (long long)ob.cam;
    if (pixel) {
        double tab[POSE_TAB];
        pose_table_entry(P.kf_T0 + 12 * (long long)ob.kf, d6, tab);
        line_pixel_factor(tab, P.cam_K + 4 * (long long)ob.cam, Tsf, L.T, L.model, ob.meas, l6, r, LIN ? Jf : nullptr, Jl);
    } else {
        line_angular_factor(P.kf_T0 + 12 * (long long)ob.kf, Tsf, L.T, ob.meas, d6, l6, r, LIN ? Jf : nullptr, Jl);
    }
    double ssq = 0.0, sc = 1.0;
    for (int q = 0; q < rows; q++) ssq += r[q] * r[q];
    const double rho = huber_rho(P.o.huber_a, ssq, sc);   // the line blocks carry the caller's loss function (…Analytic.cpp:303-306)
    if (!LIN) { atomic_add_f64(&P.acc[so].cand_cost, rho); return; }
    double* row = P.line_scratch + (long long)k * LINE_ROW;
    for (int i = 0; i < rows; i++) {
        for (int q = 0; q < 6; q++) { row[i * 12 + q] = sc * Jf[i * 6 + q]; row[i * 12 + 6 + q] = sc * Jl[i * 6 + q]; }
        row[48 + i] = sc * r[i];
    }
    row[52] = rho;
    row[53] = in_program ? 1.0 : 0.0;
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
    if (cam < 0) return;  // pseudo-observation of a sparse prior factor: not part of the caller's observation list
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

// ALandmark::avgChi2err (ALandmark.cpp:98-128) per landmark, at the deltas held in buffer 0: one lane per landmark
// (front-end windows hold a few hundred landmarks with short tracks). out[2 l] = mean chi2, out[2 l + 1] = n_obs.
template <int FACTOR>
__global__ void k_lmk_chi2(DevPtrs P, int w, const double* wh, double inv_sigma_px, double* out) {
    const WinDev W = P.win[w];
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= W.n_lmk) return;
    const long long gl = W.lmk_base + l;
    const double pw[3] = {P.lmk_p[3 * gl] + P.xl[3 * gl], P.lmk_p[3 * gl + 1] + P.xl[3 * gl + 1], P.lmk_p[3 * gl + 2] + P.xl[3 * gl + 2]};
    double sum = 0.0;
    int n = 0;
    for (int o = P.lmk_ob[gl]; o < P.lmk_oe[gl]; o++) {
        const int kf = P.obs_kf[o], cam = P.obs_cam[o];
        if (cam < 0) continue;  // pseudo-observation of a sparse prior factor
        double d6[6], dR[9], R[9], pf[3], pc[3];
        for (int i = 0; i < 6; i++) d6[i] = P.xp[6 * (long long)kf + i];
        const double* T0 = P.kf_T0 + 12 * (long long)kf;
        so3_exp(d6, dR);
        m3_mul(T0, dR, R);
        double t[3];
        m3_vec(T0, d6 + 3, t);
        m3_vec(R, pw, pf);
        for (int a = 0; a < 3; a++) pf[a] += t[a] + T0[9 + a];
        const double* Ts = P.cam_T + 12 * (long long)cam;
        m3_vec(Ts, pf, pc);
        for (int a = 0; a < 3; a++) pc[a] += Ts[9 + a];
        const double* K = P.cam_K + 4 * (long long)cam;
        const double u = (K[0] * pc[0] + K[2] * pc[2]) / pc[2], v = (K[1] * pc[1] + K[3] * pc[2]) / pc[2];
        double mu, mv;
        if (FACTOR == 0) { mu = P.obs_meas[2 * (long long)o]; mv = P.obs_meas[2 * (long long)o + 1]; }
        else {
            const double* b = P.obs_meas + 3 * (long long)o;
            mu = K[0] * b[0] / b[2] + K[2]; mv = K[1] * b[1] / b[2] + K[3];
        }
        const bool ok = !(pc[2] < 0.1) && !(u < 0.0 || v < 0.0 || u > wh[2 * cam] || v > wh[2 * cam + 1]) && isfinite(u) && isfinite(v);
        const double is = inv_sigma_px > 0.0 ? inv_sigma_px : P.cam_isig[cam];
        const double e0 = (u - mu) * is, e1 = (v - mv) * is;
        sum += ok ? e0 * e0 + e1 * e1 : 1000.0;
        n++;
    }
    out[2 * l] = n ? sum / (double)n : 0.0;
    out[2 * l + 1] = (double)n;
}

}  // namespace sadvio
