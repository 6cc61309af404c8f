// ba_types.h — device-resident layout of a batch of flattened windows and the LM control state.
//
// HBM layout (all FP64 / int32, struct-of-arrays, windows concatenated):
//   kf_T0      [n_kf_tot][12]   T_f_w at the linearisation origin (frame.h:48-51)
//   kf_fidx    [n_kf_tot]       index among the window's free key-frames, -1 = constant (…Analytic.cpp:219)
//   xp/xv/xba/xbg  [2][...]     delta states, double-buffered: buffer `cur` = x, buffer `1-cur` = candidate
//   cam_K [n_cam_tot][4], cam_T [n_cam_tot][12], cam_isig [n_cam_tot] = 1/sigma
//   lmk_p [n_lmk_tot][3], xl [2][n_lmk_tot][3], s_lmk [n_lmk_tot][3] (Jacobi scale, iteration 0)
//   lmk_ob / lmk_oe [n_lmk_tot] CSR begin/end into the observation arrays (global indices)
//   obs_kf / obs_cam [n_obs_tot] global key-frame / camera index, obs_meas [n_obs_tot][2|3]
//   S [sum Np^2] reduced pose Hessian (lower triangle valid), gred/gfull/hdiag/delta/s_pose [sum Np]
// Observation records are read with unit stride by consecutive lanes (CSR order); the key-frame /
// camera tables are staged in LDS once per workgroup.
#pragma once
#include <stdint.h>

namespace sadvio {

struct SolveOpts {
    int max_num_iterations;
    int jacobi_scaling;
    int max_num_consecutive_invalid_steps;
    int pad;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius;
    double min_lm_diagonal, max_lm_diagonal, min_relative_decrease;
    double huber_a;  // ceres::HuberLoss(a) on the visual factors, 0 = none
    double max_time_ticks;  // max_solver_time_in_seconds in ticks of the 100 MHz wall clock, 0 = no limit
};

struct WinDev {
    int n_kf, n_cam, n_lmk, n_obs;
    int kf_base, cam_base, lmk_base, obs_base;
    int n_free_kf, dpf, Np, Npose;
    long long S_off;  // doubles
    int red_off;      // offset into the length-(sum Np) vectors
    int tile_begin, tile_end;
    int prior_begin, prior_end;
    int imu_begin, imu_end;
    int factor_type, has_imu;
    int ld;           // 0: packed S solved in LDS; Np: full row-major S in HBM (Np > MAX_LDS_NP)
    // dense marginalisation prior (MarginalizationFactor, marginalization.hpp:88-218): r = r0 + J dx
    int n_red;        // landmarks kept in the reduced system (the prior couples them): 3 columns each after the poses
    int dp_n_full, dp_n;   // rows / columns of J (0 = no prior)
    int dp_int_off;   // into dp_ints: kind[n] index[n] col[n]
    long long dp_off; // into dp_data: J[nf*n] Jt[n*nf] H[n*n] r0[nf] dx[n] r[nf]
    int kept_begin, kept_end;  // slice of kept_obs (observations of the reduced landmarks)
    int sp_begin, sp_end;      // slice of the sparse prior factors
    int spl_begin, spl_end;    // slice of sp_list: the factors evaluated inside the solve (not riding the Schur elimination)
    int line_begin, line_end;  // slice of the linexd landmarks (6 reduced columns each after the kept landmarks)
    int lobs_begin, lobs_end;  // slice of their observations
};

// One workgroup of k_build / k_backsub: a run of consecutive landmarks of one window. Each landmark is
// processed by a group of G consecutive lanes of one wave (lane q of the group owns the landmark's q-th
// observation), so per-landmark reductions are lane shuffles. The key-frames the tile touches are listed
// (sorted by global index) in tile_kf[kf_off .. kf_off + n_kf); obs_slot[o] indexes that list. The free
// ones among them span the tile's private LDS copy of the reduced system (6 rows each).
struct Tile {
    int w;
    int lmk0, lmk1;   // global landmark range
    int G;            // lanes per landmark: power of two, 8 .. 64
    int kf_off, n_kf; // slice of tile_kf / tile_row
    int n_free;       // free key-frames in the slice: LDS tile dimension = 6 * n_free
    int lds_mode;     // 2: LDS tile, MFMA contraction over landmarks; 1: LDS tile, per-pair ds_add_f64; 0: global atomics
    int dpf, Np;      // window's reduced layout
    int red_off;
    int cam_base, n_cam;
    int first_of_window;
    int kmax;         // max observations of one landmark in the tile (<= G)
    int win_tile0, win_ntiles;  // the window's tile range (for summing per-tile partials)
    int ld;           // 0: S packed lower triangle; else full row-major with this leading dimension
    long long S_off;
    int chunk0, chunk1;   // slice of the chunk tables (lm_kernels.h): <= 12 landmarks and <= 64 observations per chunk
};

struct PriorDev {
    int kf;  // global key-frame index
    int pad;
    double T_prior[12];
    double inf[6];
};

// IMUFactor + IMUBiasFactor constants of one consecutive key-frame pair (residuals.hpp:133-300). The 9x9
// square-root information W = L^T, L L^T = cov^-1 (residuals.hpp:151-154) is computed once on the host at
// set_imu_factors time instead of at every evaluation.
struct ImuDev {
    int kf_i, kf_j;  // global key-frame indices (i = older frame)
    double dt;
    double dR[9], dv[3], dp[3];
    double J_dR_bg[9], J_dv_ba[9], J_dv_bg[9], J_dp_ba[9], J_dp_bg[9];
    double W[81];
    double sa, sg;   // 1 / sqrt(dt * bacc_noise^2), 1 / sqrt(dt * bgyr_noise^2)
    int win, pad;    // window of the factor (filled when the per-window lists are concatenated)
};
// One factor of the sparsified marginalisation prior (sadvio_sparse_prior with global indices).
struct SparseDev {
    int type, kf, lmk0, lmk1;   // type 4: pose-to-landmark factor riding the Schur elimination; 5: Relative6DPose (kf, kf2)
    int win, kf2;   // window of the factor; second key-frame of a relative-pose factor
    double T_prior[12], v_prior[3], ba_prior[3], bg_prior[3], delta[3];
    double W[225];
};
// linexd landmark (a pose T_w_l whose x axis carries the two model points) and one of its observations
struct LineDev {
    double T[12], model[6];
    int col;   // first of its 6 columns in the window's reduced vector, -1 = constant
    int win;
};
struct LineObsDev {
    int line, kf, cam, win;   // global line / key-frame / camera indices
    double meas[6];           // pixel: the two end points (4); angular: the two bearing vectors
};
constexpr int LINE_ROW = 4 * 12 + 4 + 2;  // scratch row of one line observation: J (rows x 12) | r | rho | in-program flag
// scratch row of one listed sparse-prior factor, written by sparse_factor_eval<false>: J (rows x 15) | r 15 | in-program flag, - |
// SPARSE_NE entry values | SPARSE_NE entry targets | reduced column of each of the 15 Jacobian columns (-1: constant) | rows.
// Entries (as in the IMU rows): H = J^T J (lower, 120) | g = J^T r (15) | the factor's squared residual sum (1).
constexpr int SPARSE_NE = 136, SPARSE_E_G = 120, SPARSE_E_COST = 135;
constexpr int SPARSE_H = 15 * 15 + 15 + 2;
constexpr int SPARSE_IX = SPARSE_H + SPARSE_NE;
constexpr int SPARSE_COL = SPARSE_IX + SPARSE_NE;
constexpr int SPARSE_J = SPARSE_COL + 16;
constexpr int IMU_J = 9 * 24 + 9;  // whitened Jacobian + residual kept in HBM scratch between phases
// scratch row of one IMU factor pair, all written by imu_pair_lin_wg (kernels.h): J 216 | r 9 | bias residuals 6 | IMU_NE entry values | IMU_NE entry
// targets | fi, fj, sa, sg. The entries are what the factor pair (IMUFactor + IMUBiasFactor) adds to the window's reduced system:
// H = J^T J (lower, 300) | g = J^T r (24) | bias random walk: 18 matrix entries, 12 gradient entries | the pair's squared residual
// sum (1). Target of a matrix entry: (row << 16) | col; of a gradient entry: row << 16; -1 = not in the system (constant
// key-frame); cost entry: -2 = fixed cost (both key-frames constant), -3 = cost. Stored as doubles.
constexpr int IMU_NE = 355;
constexpr int IMU_E_G = 300, IMU_E_BH = 324, IMU_E_BG = 342, IMU_E_COST = 354;
constexpr int IMU_H = IMU_J + 6;
constexpr int IMU_IX = IMU_H + IMU_NE;
constexpr int IMU_META = IMU_IX + IMU_NE;
constexpr int IMU_ROW = IMU_META + 5;

// Levenberg-Marquardt control state at the beginning of a slot (one step attempt).
struct LmState {
    double radius, decrease_factor, x_cost, x_norm, initial_cost;
    int cur;  // which delta buffer holds x
    int done, termination, iter;
    int n_success, n_unsuccess, n_invalid, pad;
};

// What the host reads back after a solve (one record per window, written by k_final).
struct FinalRec {
    LmState s;
    double fixed_cost;
};

// Per-slot accumulators (zeroed before the solve; filled with atomics by the slot's kernels).
struct IterAcc {
    double lin_cost;    // sum r^2 at the linearisation point (all non-fixed residual blocks)
    double cand_cost;   // sum r^2 at the candidate point
    double mcc;         // model cost change  -(J d)^T (r + J d / 2)
    double step_norm2;  // |delta|^2
    double cand_norm2;  // |x + delta|^2
    double fixed_cost;  // sum r^2 of blocks whose parameters are all constant (slot 0 only)
    unsigned long long gmax_bits;  // max |gradient| as IEEE bits (non-negative doubles order like u64)
    int chol_fail;
    int time_up;   // the solver-time limit was exceeded when this slot's k_solve ran (it then ended the solve)
};

// Per-tile partial sums of one slot, written with plain stores by the tile's workgroup and summed by the
// consumers (same-address FP64 atomics from ~1000 waves cost ~12 ns each, i.e. tens of microseconds).
struct TileAcc {
    double lin_cost, fixed_cost, gmax;          // k_build
    double cand_cost, mcc, step_norm2, cand_norm2;  // k_backsub
    double pad;
};

constexpr int BUILD_THREADS = 256;  // 4 waves; each wave owns 64 / G landmarks per round
constexpr int BUILD_WAVES = BUILD_THREADS / 64;
constexpr int MAX_TILE_KF = 24;      // key-frames (free + constant) a tile may touch
constexpr int MAX_TILE_FREE_KF = 20; // free ones: LDS tile <= 120 x 121 / 2 doubles = 58 KB
constexpr int MAX_GEMM_FREE_KF = 5;  // tiles touching <= 8 free key-frames accumulate through the MFMA contraction
constexpr int MAX_WIN_CAM = 8;       // cameras per window staged in LDS
constexpr int STAGE_VALS = 18;       // Jp[12] Jl[6] exchanged between the lanes of a landmark group
constexpr int MAX_LMK_OBS = 64;      // observations per landmark (one lane each)
constexpr int SOLVE_THREADS = 512;
constexpr int SOLVE_KFC_STRIDE = 32; // doubles per parked key-frame: free index | x 6 | T0 12 | v, ba, bg at x 9
constexpr int SOLVE_KFC = 32;        // k_solve<0>: key-frames / priors whose back-half inputs are parked in LDS across the factorisation
constexpr int PRIOR_LIN = 28;        // doubles of a pose prior's linearisation record: g 6 | H lower 21 | |r|^2
constexpr int PRE_KF = 8;            // key-frames of a tile's list carried by its first-round packet (covers the first 256 / POSE_TAB = 6 tables)
constexpr int PRE_MAX_TILES = 1024;  // submissions with more tiles run without the packets (batches: the throughput regime)
constexpr int MAX_LDS_NP = 174;     // packed lower triangle incl. rhs row + panel strip: ~150 KB

}  // namespace sadvio
