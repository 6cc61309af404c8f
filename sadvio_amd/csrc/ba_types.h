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
};

struct Tile {
    int w;           // window
    int lmk0, lmk1;  // global landmark range
    int obs0, obs1;  // global observation range
    int kmax;        // max observations of one landmark in the tile
};

struct PriorDev {
    int kf;  // global key-frame index
    int pad;
    double T_prior[12];
    double inf[6];
};

// Levenberg-Marquardt control state at the beginning of a slot (one step attempt).
struct LmState {
    double radius, decrease_factor, x_cost, x_norm, initial_cost;
    int cur;  // which delta buffer holds x
    int done, termination, iter;
    int n_success, n_unsuccess, n_invalid, pad;
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
    int pad;
};

constexpr int BUILD_THREADS = 256;
constexpr int MAX_TILE_OBS = 256;  // one observation per thread
constexpr int MAX_TILE_LMK = 64;
constexpr int MAX_LDS_NPOSE = 120;  // 20 free key-frames x 6: lower triangle = 7260 doubles = 58 KB
constexpr int MAX_LDS_KF = 64;      // key-frame table staged in LDS
constexpr int OBS_STAGE = 26;       // Jp[12] Jl[6] r[2] N[6]
constexpr int LMK_STAGE = 12;       // Minv[6] gl[3] pad[3]
constexpr int SOLVE_THREADS = 256;
constexpr int MAX_LDS_NP = 192;     // packed lower triangle 18528 doubles = 148 KB

}  // namespace sadvio
