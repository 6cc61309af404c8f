/*
 * sadvio_ba.h — C ABI of the MI355X-native sliding-window bundle-adjustment backend.
 *
 * This is the drop-in boundary for SaDVIO's optimizer hot path. Every entry point names the
 * reference interface it replaces (paths relative to the reference checkout, `cpp/...`).
 * The reference has no FFI of its own (100 % C++); what a maintainer binds is
 * an `isae::AOptimizer` subclass that flattens the object graph into `sadvio_flat_window`,
 * calls these functions and applies the returned *deltas* with the reference's own
 * composition rules (AOptimizer.cpp:329-340,391-434). See INTEGRATION.md.
 *
 * Conventions
 *   - All floating point is IEEE FP64 (the reference is FP64 throughout).
 *   - A rigid transform is 12 doubles: R row-major (9) followed by t (3).
 *     `T_f_w` = world->frame (frame.h:48-51), `T_s_f` = frame->sensor (ASensor.h:43-44).
 *   - Optimisation variables are deltas initialised to zero and composed on the right:
 *     T_f_w = T_f_w0 * (exp(w), t)  (geometry.h:198-203; parametersBlock.hpp:34-37 — NOT the
 *     SE3 exponential), p = p0 + dl (geometry.h:205-210).
 *   - Return value 0 = OK, negative = SADVIO_E_*; nothing throws across the boundary.
 *   - A handle owns one HIP stream and all device mirrors; it is not re-entrant. Two handles
 *     (front-end / back-end optimizer instances, slamParameters.cpp:273-275) may be used
 *     concurrently from different host threads.
 *   - The library has NO CPU fallback: every compute entry point fails with
 *     SADVIO_E_NO_DEVICE / SADVIO_E_HIP when no gfx950 device is usable.
 */
#ifndef SADVIO_BA_H
#define SADVIO_BA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SADVIO_OK 0
#define SADVIO_E_INVALID_ARG (-1)
#define SADVIO_E_NOT_USABLE (-2) /* solver produced no usable solution (AOptimizer.cpp:259) */
#define SADVIO_E_HIP (-3)
#define SADVIO_E_RCCL (-4)
#define SADVIO_E_NO_DEVICE (-5)
#define SADVIO_E_STATE (-6)   /* call order violated (e.g. solve before set_window) */
#define SADVIO_E_REFUSED (-7) /* marginalisation refused: n < 4 (marginalization.cpp:215-216) */

/* Visual factor flavour of a window. */
#define SADVIO_FACTOR_PIXEL 0   /* ReprojectionErrCeres_pointxd_dx, BundleAdjustmentCERESAnalytic.h:41-98 */
#define SADVIO_FACTOR_ANGULAR 1 /* AngularErrCeres_pointxd_dx, AngularAdjustmentCERESAnalytic.h:45-120 */

/* Termination codes in sadvio_solve_summary.termination (Ceres TerminationType semantics). */
#define SADVIO_TERM_NO_CONVERGENCE 0 /* max_num_iterations reached */
#define SADVIO_TERM_FUNCTION_TOL 1
#define SADVIO_TERM_PARAMETER_TOL 2
#define SADVIO_TERM_GRADIENT_TOL 3
#define SADVIO_TERM_MIN_RADIUS 4
#define SADVIO_TERM_FAILURE 5 /* too many consecutive invalid steps */

typedef struct sadvio_ba_handle sadvio_ba_handle;

typedef struct sadvio_ba_config {
    int32_t device;          /* HIP device ordinal */
    int32_t profile_kernels; /* 1: bracket every kernel class with hipEvents (see get_kernel_times) */
    int32_t use_graph;       /* 1: replay the iteration sequence from a captured hipGraph */
    int32_t reserved;
} sadvio_ba_config;

/*
 * One sliding window, flattened (SoA, index based). Replaces the object-graph walk of
 * BundleAdjustmentCERESAnalytic::addResidualsLocalMap (BundleAdjustmentCERESAnalytic.cpp:197-314)
 * / AngularAdjustmentCERESAnalytic::addResidualsLocalMap (…Angular….cpp:212-339).
 * Inclusion rules the flattener must apply (the adapter side owns them):
 *   - key-frames in the order of LocalMap::getLastNFramesIn, newest first (amap.h:28-32);
 *     kf_const[i] = 1 iff i > n_kf - fixed - 1 (…Analytic.cpp:219);
 *   - landmark included iff isInitialized && !isOutlier (…Analytic.cpp:239);
 *   - observation included iff its sensor's frame is a key-frame of the window (…Analytic.cpp:256-258).
 * Observations are CSR by landmark. All arrays are caller-owned; set_window copies them.
 */
typedef struct sadvio_flat_window {
    int32_t n_kf;
    int32_t n_cam;
    int32_t n_lmk;
    int32_t n_obs;
    int32_t factor_type; /* SADVIO_FACTOR_* */
    int32_t has_imu;     /* 1: every key-frame carries (v, ba, bg) states (AOptimizer.cpp:29-53) */

    const int64_t *kf_id;    /* [n_kf] opaque ids, echoed unchanged (frame.h:21-22) */
    const double *kf_T_f_w;  /* [n_kf][12] */
    const uint8_t *kf_const; /* [n_kf] */
    const double *kf_vel;    /* [n_kf][3] or NULL (IMU.h:77-100) */
    const double *kf_ba;     /* [n_kf][3] or NULL */
    const double *kf_bg;     /* [n_kf][3] or NULL */

    const double *cam_K;     /* [n_cam][4] fx fy cx cy (ASensor.cpp:17) */
    const double *cam_T_s_f; /* [n_cam][12] (ASensor.h:44) */
    const double *cam_sigma; /* [n_cam] measurement sigma: 1.0 pixel (…Analytic.h:46);
                                1.5/f angular window BA (…Angular….cpp:283) */

    const int64_t *lmk_id;      /* [n_lmk] opaque ids, echoed unchanged (ALandmark.h:18-19) */
    const double *lmk_p;        /* [n_lmk][3] T_w_lmk.translation() (ALandmark.h:36-39) */
    const uint8_t *lmk_const;   /* [n_lmk] or NULL = all free */
    const int32_t *lmk_obs_ptr; /* [n_lmk+1] CSR offsets into obs_* */
    const int32_t *obs_kf;      /* [n_obs] index into kf_* */
    const int32_t *obs_cam;     /* [n_obs] index into cam_* */
    const double *obs_meas;     /* [n_obs][2] pixel uv (AFeature2D.h:21) | [n_obs][3] unit bearing (AFeature2D.h:80) */
} sadvio_flat_window;

/*
 * Line landmarks ("linexd": Line3D with ModelLine3D, Line3D.h:8-36, Model3D.h:45-51) of one window — SURVEY.md §8f rank 3.
 * A line is a 6-dof pose parameter block, T_w_l <- T_w_l (exp(w), t) (PoseParametersBlock; write-back AOptimizer.cpp:334-336),
 * observed as a 2-D segment. Residual blocks, added by addResidualsLocalMap for every feature whose frame is a key-frame of
 * the window (BundleAdjustmentCERESAnalytic.cpp:273-311 / AngularAdjustmentCERESAnalytic.cpp:293-333):
 *   pixel windows    ReprojectionErrCeres_linexd_dx (…Analytic.h:102-195): the two model points projected with
 *                    Camera::project against the two measured end points, 4 residuals, sigma 1. As coded the landmark is
 *                    moved by T_w_l (I, x[0..2]) — the first three entries of its 6-vector, used as a TRANSLATION — while the
 *                    Jacobian treats them as a rotation ([-R [pt]x | I]); reproduced. The cost-only branch of the reference
 *                    (:168-177) reads an uninitialised projection: the residual of the Jacobian branch is used for both.
 *   angular windows  AngularErrCeres_linexd_dx (…Angular….h:368-469): coplanarity of the observation plane (b0 x b1) with
 *                    the line, 2 residuals, weight 1 / sigma^2 with sigma 1.
 * Lines stay in the reduced system (6 columns each after the key-frames and the prior-kept landmarks): they are few.
 */
typedef struct sadvio_line_set {
    int32_t n_line, n_obs;
    const int64_t *line_id;      /* [n_line] opaque ids, echoed */
    const double *line_T_w_l;    /* [n_line][12] landmark->getPose() */
    const double *line_model;    /* [n_line][6] the two model points (ModelLine3D: (-0.5, 0, 0), (0.5, 0, 0)) */
    const uint8_t *line_const;   /* [n_line] or NULL */
    const int32_t *line_obs_ptr; /* [n_line + 1] CSR */
    const int32_t *obs_kf;       /* [n_obs] */
    const int32_t *obs_cam;      /* [n_obs] */
    const double *obs_meas;      /* pixel: [n_obs][4] end points (AFeature::getPoints) | angular: [n_obs][6] two bearings */
} sadvio_line_set;

/* Constants of one IMUFactor + IMUBiasFactor pair (residuals.hpp:133-300); pairing rule
 * AOptimizer::addIMUResiduals (AOptimizer.cpp:55-92): consecutive KFs, dt <= 1 s. kf_i is the
 * older frame (imu_j->getLastKF()). All 3x3 are row-major. */
typedef struct sadvio_imu_factor {
    int32_t kf_i, kf_j;
    double dt;            /* (ts_j - ts_i) * 1e-9 */
    double delta_R[9];    /* IMU::getDeltaR of imu_j */
    double delta_v[3];
    double delta_p[3];
    double J_dR_bg[9];    /* IMU.h:141-145 */
    double J_dv_ba[9];
    double J_dv_bg[9];
    double J_dp_ba[9];
    double J_dp_bg[9];
    double cov[81];       /* IMU::getCov of imu_j, 9x9 row-major */
    double bacc_noise;    /* imu_i->getbAccNoise() (residuals.hpp:259) */
    double bgyr_noise;    /* imu_i->getbGyrNoise() (residuals.hpp:261) */
} sadvio_imu_factor;

/* PosePriordx (residuals.hpp:601-632); sqrt_inf = diag(inf_diag) (…Analytic.cpp:226). */
typedef struct sadvio_pose_prior {
    int32_t kf;
    int32_t pad;
    double T_prior[12];
    double inf_diag[6];
} sadvio_pose_prior;

/* One factor of the sparsified (NFR) marginalisation prior, added by the sparse branch of
 * addMarginalizationResiduals (BundleAdjustmentCERESAnalytic.cpp:363-426). Every landmark such a factor touches
 * stays in the reduced system (it is coupled to other variables than its own observations).
 *   SADVIO_SPARSE_IMU_PRIOR      IMUPriordx               (residuals.hpp:649-695)  kf; T_prior, v/ba/bg_prior, sqrt_inf 15x15
 *   SADVIO_SPARSE_POSE_TO_LMK    PoseToLandmarkFactor     (residuals.hpp:570-595)  kf, lmk0; delta, sqrt_inf 3x3
 *   SADVIO_SPARSE_LMK_PRIOR      Landmark3DPrior          (residuals.hpp:512-522)  lmk0; delta = prior, sqrt_inf 3x3
 *   SADVIO_SPARSE_LMK_TO_LMK     LandmarkToLandmarkFactor (residuals.hpp:537-556)  lmk0, lmk1; delta, sqrt_inf 3x3
 * The linearisation-point values the reference copies into the factor (_T, _v, _ba, _bg, _lmk, _t_w_lmk) are the
 * window's own kf_* / lmk_p entries. sqrt_inf is row-major; 3x3 matrices use the first 9 entries. */
#define SADVIO_SPARSE_IMU_PRIOR 0
#define SADVIO_SPARSE_POSE_TO_LMK 1
#define SADVIO_SPARSE_LMK_PRIOR 2
#define SADVIO_SPARSE_LMK_TO_LMK 3
/*   SADVIO_SPARSE_RELATIVE_POSE  Relative6DPose           (residuals.hpp:70-131)   kf = a, kf_b = b; T_prior = T_a_b_prior,
 *                                sqrt_inf 6x6. The factor a pose graph over sparsified windows is made of (SURVEY.md §8f
 *                                rank 2; information from sadvio_ba_marginalize_relative). As coded it composes its deltas on
 *                                the FRAME-TO-WORLD transforms, T_w_a (exp(w), t): in a window that carries such factors the
 *                                kf_T_f_w slots of a and b are read as T_w_a / T_w_b (a pose-graph window passes frame-to-world
 *                                poses; mixing with visual factors, which read the slot as world-to-frame, is the caller's
 *                                business). */
#define SADVIO_SPARSE_RELATIVE_POSE 4
typedef struct sadvio_sparse_prior {
    int32_t type;
    int32_t kf;          /* key-frame index in the window, or -1 */
    int32_t lmk0, lmk1;  /* landmark indices in the window, or -1 */
    double T_prior[12];  /* IMUPriordx; Relative6DPose: T_a_b_prior */
    double v_prior[3], ba_prior[3], bg_prior[3];
    double delta[3];
    double sqrt_inf[225];
    int32_t kf_b;        /* Relative6DPose: the second key-frame */
    int32_t pad;
} sadvio_sparse_prior;

/* Solver options. sadvio_ba_default_options() fills the reference's hard-coded values
 * (AOptimizer.cpp:315-323) and, for everything the reference leaves unset, the defaults of
 * Ceres Solver 2.2.0 (docker/Dockerfile:50), the version the reference pins. */
typedef struct sadvio_solve_options {
    int32_t max_num_iterations;            /* 20 */
    int32_t jacobi_scaling;                /* 1 */
    int32_t max_num_consecutive_invalid_steps; /* 5 */
    int32_t reserved;
    double function_tolerance;             /* 1e-3 */
    double gradient_tolerance;             /* 1e-10 */
    double parameter_tolerance;            /* 1e-8 */
    double initial_trust_region_radius;    /* 1e4 */
    double max_trust_region_radius;        /* 1e16 */
    double min_trust_region_radius;        /* 1e-32 */
    double min_lm_diagonal;                /* 1e-6 */
    double max_lm_diagonal;                /* 1e32 */
    double min_relative_decrease;          /* 1e-3 */
    /* ceres::HuberLoss(a) on the visual factors (0 = no loss function, as localMapBA / singleFrameOptimization):
     * landmarkOptimization and singleFrameVIOptimization use a = sqrt(1.345) (AOptimizer.cpp:102,223). Applied
     * the way Ceres' Corrector does for rho'' <= 0: residual and Jacobian scaled by sqrt(rho'), cost = rho / 2. */
    double huber_a;                        /* 0 */
    /* Ceres' max_solver_time_in_seconds (singleFrameVIOptimization sets 0.005, AOptimizer.cpp:254): checked, as in
     * TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue, after every iteration — the solve ends with
     * NO_CONVERGENCE once the time since its start exceeds the limit. Measured on the device (constant 100 MHz clock) from
     * the first kernel of the solve. 0 (and the reference's localMapBA: Ceres' default 1e6 s) = no limit.
     * Deviations from Ceres: (i) the clock covers DEVICE time only — host-side flattening, the upload of set_windows and the
     * graph launch are outside it, whereas Ceres counts its preprocessor and wall time from Solve(); a cap therefore binds
     * later here; (ii) the limit is IGNORED on a window sharded over several GPUs (the ranks' clocks are not synchronised:
     * a rank-local decision would desynchronise the collective schedule); (iii) with a limit the iteration count depends on
     * timing, so parity runs keep it at 0. */
    double max_solver_time_in_seconds;     /* 0 */
} sadvio_solve_options;

typedef struct sadvio_solve_summary {
    int32_t iterations;             /* step attempts performed (Ceres iteration count, excl. iteration 0) */
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    int32_t termination;            /* SADVIO_TERM_* */
    double initial_cost;            /* 1/2 sum r^2 over the reduced program at x0 */
    double final_cost;
    double fixed_cost;              /* cost of residual blocks whose parameters are all constant */
    double final_radius;
} sadvio_solve_summary;

/* Number of usable gfx950 devices (0 when none: every compute call then fails). */
int sadvio_ba_device_count(void);

/* Fill `opts` with the reference's options for localMapBA (AOptimizer.cpp:315-323). */
void sadvio_ba_default_options(sadvio_solve_options *opts);

/* Create / destroy a backend instance. Replaces constructing one optimizer object
 * (slamParameters.cpp:263-282). */
int sadvio_ba_create(const sadvio_ba_config *cfg, sadvio_ba_handle **out);
void sadvio_ba_destroy(sadvio_ba_handle *h);

/* Upload `n_windows` independent windows (a batch: independent sub-windows solved
 * concurrently; n_windows = 1 is the reference's case). Replaces addResidualsLocalMap
 * (…Analytic.cpp:197-314). Clears any factors set by earlier set_* calls. */
int sadvio_ba_set_windows(sadvio_ba_handle *h, int32_t n_windows, const sadvio_flat_window *windows);
/* One layout build per key-frame: between begin_update and commit_update, set_windows and the factor setters that follow it
 * (set_pose_priors, set_imu_factors, set_sparse_priors, set_dense_prior, set_lines) only RECORD their arguments (validated and
 * copied as usual); commit_update builds the device layout once and uploads everything in one staged copy. Without the
 * bracket every setter rebuilds what it changes — set_sparse_priors the whole tiling, because eliminable pose-to-landmark
 * factors become pseudo-observations of their landmarks — which is what addResidualsLocalMap + addMarginalizationResiduals
 * (…Analytic.cpp:197-426) amount to when they are called back to back. Compute calls are refused inside the bracket. */
int sadvio_ba_begin_update(sadvio_ba_handle *h);
int sadvio_ba_commit_update(sadvio_ba_handle *h);
/* The single-window call the adapter of one optimizer instance uses: set_windows(h, 1, window). */
int sadvio_ba_set_window(sadvio_ba_handle *h, const sadvio_flat_window *window);

/* Line landmarks of window `w` (NULL or n_line = 0 clears them). Not supported on a window sharded over several GPUs. */
int sadvio_ba_set_lines(sadvio_ba_handle *h, int32_t w, const sadvio_line_set *lines);
/* Solved 6-vectors of the lines of window `w` ([n_line][6]), applied as T_w_l <- T_w_l (exp(w), t) (AOptimizer.cpp:334-336). */
int sadvio_ba_get_line_deltas(sadvio_ba_handle *h, int32_t w, double *line_delta6);

/* PosePriordx blocks of window `w` (…Analytic.cpp:224-228). */
int sadvio_ba_set_pose_priors(sadvio_ba_handle *h, int32_t w, int32_t n, const sadvio_pose_prior *priors);

/* IMUFactor + IMUBiasFactor blocks of window `w` (AOptimizer.cpp:55-92). */
int sadvio_ba_set_imu_factors(sadvio_ba_handle *h, int32_t w, int32_t n, const sadvio_imu_factor *factors);

/* Dense marginalisation prior of window `w` = MarginalizationFactor (marginalization.hpp:88-218,
 * added by addMarginalizationResiduals, …Analytic.cpp:316-360): r = r0 + J dx, J is
 * n_full x n row-major. `kf_keep` (or -1 in pure VO) is the key-frame whose 15 states occupy
 * columns [kf_col, kf_col+15) (pose6, v3, ba3, bg3); kept landmark `lmk_index[i]` occupies
 * columns [lmk_col[i], lmk_col[i]+3); lmk_col[i] = -1 means "skipped" (marginalization.hpp:138).
 * J == NULL with n_full == SADVIO_PRIOR_RESIDENT attaches THE HANDLE'S PRIOR — the (J, r0) the last sadvio_ba_marginalize
 * (or sadvio_ba_set_prior) left on the device, the way the reference keeps `_marginalization_last` inside the optimizer
 * (AOptimizer.h:88-90): nothing crosses PCIe; `n` is then ignored, `r0` must be NULL, and the caller only names the
 * variables of THIS window that the prior's columns refer to (it owns the id -> index bookkeeping, like the reference's
 * _map_frame_idx / _map_lmk_idx). */
#define SADVIO_PRIOR_RESIDENT (-1)
int sadvio_ba_set_dense_prior(sadvio_ba_handle *h, int32_t w, int32_t n_full, int32_t n,
                              const double *J, const double *r0, int32_t kf_keep, int32_t kf_col,
                              int32_t n_keep, const int32_t *lmk_index, const int32_t *lmk_col);

/* ---- marginalisation of the oldest key-frame into a dense prior (K8) --------------------------------------
 * Replaces BundleAdjustmentCERESAnalytic::marginalize / Marginalization::{computeInformationAndGradient,
 * computeSchurComplement, rankReveallingDecomposition, computeJacobiansAndResiduals}
 * (…Analytic.cpp:431-663, marginalization.cpp:145-265,318-342,516-530) on window `w` as uploaded by set_windows.
 * The selection of marginalised / kept landmarks (preMarginalize, marginalization.cpp:23-143) is object-graph
 * logic and stays with the caller, who passes the two index lists in the reference's order.
 * Column layout (marginalization.cpp:38-113): marginalised = [frame0 pose 6 (+ v, ba, bg 9 if marg_has_imu) |
 * lmk_marg 3 each]; kept = [frame1 15 states if kf_keep >= 0 | lmk_keep 3 each]. */
/* Eigenvalue cut of the pseudo-inverse of Amm and of the rank-revealing decomposition of Ak.
 *   SADVIO_EIG_CUT_REFERENCE   (0, the default of a zero-initialised request): the reference's arithmetic — keep
 *       lambda > 1e-12, ABSOLUTE (Marginalization::_eps, marginalization.hpp:58, applied at marginalization.cpp:237,322).
 *       In float64 that constant sits below the rounding noise of the sums it is applied to (|A| ~ 1e5..1e8 => noise
 *       ~ 1e-11..1e-8): exactly-null directions are kept or dropped by the sign of a rounding error, in the reference as here;
 *       the prior's INFORMATION (J^T J, J^T r0) is unaffected to rounding (a kept noise direction carries ~1e-11 of it),
 *       only n_full varies. Weakly observed directions (far, low-parallax depth: 1e-12 < lambda < n eps lambda_max) are
 *       KEPT, as the reference keeps them.
 *   SADVIO_EIG_CUT_NOISE_FLOOR (1): keep lambda > max(1e-12, n eps lambda_max) — the numerically meaningful rank; n_full
 *       is reproducible across implementations, at the price of dropping directions whose information is below the floor
 *       (what it drops from Ak itself is <= n eps lambda_max by construction; a direction v dropped from Amm^+ leaves
 *       (Arm v)(Arm v)^T / lambda in Ak. tests/test_gpu_margloop.py measures both modes on a low-parallax window — 300
 *       directions between the cuts: |Ak_ref - Ak_floor|_2 = 2e-10 lambda_max — and bounds the effect on the next solve).
 * Form of the prior handed back / kept on the device (both give the same MarginalizationFactor cost, gradient and
 * Gauss-Newton matrix: r0 + J dx enters the solve only through J^T J, J^T r0 and |r0|^2):
 *   SADVIO_PRIOR_FORM_EIGEN    (0): the reference's J = Lambda^1/2 U^T, r0 = -Lambda^-1/2 U^T bk, rows in ascending
 *       eigenvalue order (marginalization.cpp:318-342,516-530) — one-sided block Jacobi on the device, ~20 ms at n ~ 900.
 *   SADVIO_PRIOR_FORM_CHOLESKY (1): J = G, a Cholesky factor G^T G = Ak, r0 = -G^-T bk by carrying bk through the
 *       factorisation — no eigen-decomposition; sadvio_ba_sparsify then takes Sigma_k = Ak^-1 from the triangular inverse of
 *       G. Two routes (~1 ms / ~2 ms at n ~ 900): under SADVIO_EIG_CUT_REFERENCE with an earlier prior folded in (Ak is then
 *       normally of full rank) the factorisation is UNPIVOTED — G = L^T, upper triangular in the caller's column order,
 *       n_full = n — and every pivot is tested afterwards (positive, above the cut, above the rounding noise of its own
 *       elimination); a failed test, a first marginalisation and SADVIO_EIG_CUT_NOISE_FLOOR (whose point is a reliable
 *       numerical rank, which an unpivoted factorisation cannot give) take the rank-revealing route: diagonal pivoting with
 *       the cut applied to the pivots, G's rows in pivot order. Either way only J^T J, J^T r0 and |r0|^2 are defined by the
 *       form; the rows themselves are not the reference's. */
#define SADVIO_EIG_CUT_REFERENCE 0
#define SADVIO_EIG_CUT_NOISE_FLOOR 1
#define SADVIO_PRIOR_FORM_EIGEN 0
#define SADVIO_PRIOR_FORM_CHOLESKY 1

typedef struct sadvio_marg_request {
    int32_t kf_marg;                 /* frame0 */
    int32_t kf_keep;                 /* frame1 when it carries an IMU (15 columns), else -1 */
    int32_t marg_has_imu;            /* frame0->getIMU() */
    int32_t n_marg;
    const int32_t *lmk_marg;         /* window landmark indices, order of _lmk_to_marg */
    int32_t n_keep;
    const int32_t *lmk_keep;         /* order of _lmk_to_keep */
    const sadvio_imu_factor *imu;    /* IMUFactor + IMUBiasFactor(frame0, frame1) or NULL (…Analytic.cpp:451-508) */
    int32_t n_prior;                 /* PosePriordx blocks (<= 4), .kf = kf_marg or kf_keep (:605-617) */
    const sadvio_pose_prior *priors;
    int32_t last_n_full, last_n;     /* previous MarginalizationFactor (:574-603); last_n_full = 0 if none;
                                        last_n_full = SADVIO_PRIOR_RESIDENT: the handle's prior (last_J / last_r0 / last_n ignored) */
    const double *last_J, *last_r0;
    int32_t last_kf, last_kf_col;    /* window index of its kept frame (= kf_marg now) or -1, its first column */
    int32_t last_n_keep;
    const int32_t *last_lmk_index;   /* window landmark indices of its kept landmarks */
    const int32_t *last_lmk_col;     /* their columns (-1 = skipped) */
    int32_t eig_cut_mode;            /* SADVIO_EIG_CUT_* */
    int32_t prior_form;              /* SADVIO_PRIOR_FORM_* */
} sadvio_marg_request;

typedef struct sadvio_marg_result {
    int32_t m, n, n_full;
    int32_t kf_col;   /* column of kf_keep's 15 states in the new prior, -1 if none */
    int32_t sweeps_mm, sweeps_k;  /* Jacobi sweeps of the two eigen-decompositions (sweeps_k = 0 in the Cholesky form) */
} sadvio_marg_result;

/* Returns SADVIO_E_REFUSED when n < 4 (marginalization.cpp:215-216; the reference then clears its prior, and so does the
 * handle). Outputs (caller-allocated): lmk_col[n_keep] column of each kept landmark in the new prior.
 * The new prior STAYS ON THE DEVICE as the handle's prior (AOptimizer.h:88-90: `_marginalization_last`), replacing the
 * previous one: the next window attaches it with sadvio_ba_set_dense_prior(.., SADVIO_PRIOR_RESIDENT, ..) or sparsifies it
 * with sadvio_ba_sparsify(.., J = NULL, ..), the next marginalisation folds it in with last_n_full = SADVIO_PRIOR_RESIDENT.
 * J / r0 may be NULL (the default path: no read-back); when given, J[n*n] receives the n_full x n prior Jacobian
 * (row-major, packed) and r0[n] the n_full prior residuals.
 * ASYNCHRONOUS without a read-back: the call returns once the host has taken its route decisions; the tail kernels (packing of the
 * prior, the swap into the handle) are stream work like everything that reads the prior afterwards (all on the handle's one stream,
 * so the order is kept). A device fault in that tail therefore surfaces in the NEXT call of this handle that waits (solve,
 * get_deltas, a read-back); SADVIO_DEBUG != 0 or cfg.profile_kernels make marginalize wait itself so that it is attributed here. A
 * consumer on ANOTHER stream or handle must synchronise with this handle first (sadvio_ba_get_prior does).
 * Refused (SADVIO_E_INVALID_ARG) on a window sharded over several GPUs: each rank holds a landmark partition only. */
int sadvio_ba_marginalize(sadvio_ba_handle *h, int32_t w, const sadvio_marg_request *rq, sadvio_marg_result *res,
                          int32_t *lmk_col, double *J, double *r0);

/* Route counters of the Cholesky-form marginalisations of this handle since its creation: calls, calls that took the unpivoted
 * wide-panel factorisation (Ak of full rank, every pivot tested), calls that tried it and fell back to the rank-revealing (pivoted)
 * route. calls - unpivoted - fell_back = calls that pivoted without a try (no earlier prior, or the noise-floor cut). */
int sadvio_ba_marg_stats(sadvio_ba_handle *h, int32_t *calls, int32_t *unpivoted, int32_t *fell_back);

/* The handle's prior: read-back on request (any pointer may be NULL; J has room for n_full * n, r0 for n_full doubles),
 * upload of a prior kept elsewhere (e.g. restored from a dump; n_full = 0 clears it), and its shape. */
typedef struct sadvio_prior_info {
    int32_t valid, n_full, n, form;
} sadvio_prior_info;
int sadvio_ba_get_prior(sadvio_ba_handle *h, sadvio_prior_info *info, double *J, double *r0);
int sadvio_ba_set_prior(sadvio_ba_handle *h, int32_t n_full, int32_t n, int32_t form, const double *J, const double *r0);

/* NFR sparsification of a dense prior (Marginalization::sparsifyVIO / sparsifyVO, marginalization.cpp:362-514) into
 * the factor list of the sparse branch of addMarginalizationResiduals (…Analytic.cpp:363-426): vio != 0 -> one
 * IMUPriordx on kf_keep + one PoseToLandmarkFactor per kept landmark; vio == 0 -> greedy landmark chain ordered by
 * |tr Lambda_ij|, a Landmark3DPrior on the minimum-entropy landmark + LandmarkToLandmarkFactor links. The prior is
 * the (J, column map) pair produced by sadvio_ba_marginalize / passed to sadvio_ba_set_dense_prior; J == NULL = the
 * handle's prior (n_full / n are then taken from it and the arguments ignored; nothing is uploaded). A host J must be in
 * the EIGEN form (orthogonal rows); the handle's prior may be in either form. Linearisation values (T_f_w, v, ba, bg,
 * landmark positions) are those of window `w`. `out` has room for n_keep + 1 factors. Refused on a sharded window. */
int sadvio_ba_sparsify(sadvio_ba_handle *h, int32_t w, int32_t vio, int32_t n_full, int32_t n, const double *J,
                       int32_t kf_keep, int32_t kf_col, int32_t n_keep, const int32_t *lmk_index, const int32_t *lmk_col,
                       int32_t *n_out, sadvio_sparse_prior *out);

/* ---- relative-pose information between two key-frames (NFR), SURVEY.md §8f rank 2 --------------------------------
 * Replaces BundleAdjustmentCERESAnalytic::marginalizeRelative (…Analytic.cpp:665-809) with
 * Marginalization::preMarginalizeRelative (marginalization.cpp:532-588) on window `w`: every landmark of kf_a that also
 * has a feature in kf_b is marginalised (reprojection factors of its features in the two frames), the two poses are
 * kept (n = 12), Ak -> Sigma_k = pseudo-inverse through the rank-revealing decomposition, and the information of a
 * Relative6DPose(T_w_a, T_w_b, T_a_b = T_a_w T_w_b, sqrt_inf = I) factor is recovered: inf = (J Sigma_k J^T)^-1.
 * As coded: a landmark enters the list once PER feature it has in kf_b, and its factors are added once per entry (stereo
 * landmarks count twice). Frames with IMU states are refused (the reference indexes their velocity / bias columns
 * outside of its own layout, :705-737). Returns SADVIO_E_REFUSED when no landmark is shared, SADVIO_E_INVALID_ARG on a
 * window sharded over several GPUs (each rank only holds its landmark partition: the sum would be partial).
 * Eigenvalue cuts (`eig_cut_mode`, SADVIO_EIG_CUT_* above): REFERENCE drops eigenvalues <= 1e-12 absolutely
 * (Marginalization::_eps) on the 3m x 3m landmark block and on Ak; NOISE_FLOOR uses max(1e-12, m eps lambda_max) on the
 * landmark block and max(1e-12, 12 eps lambda_max (2 + n_items)) on Ak (in float64 the absolute cut sits
 * below the rounding noise of the sums it is applied to — on the reference's own test fixture the null eigenvalue computes
 * to +-1e-11, and the gauge null space of a summed Schur complement to 1e-7 .. 4e-6). With thousands of shared landmarks
 * that floor reaches 1e-8 .. 1e-6 relative to lambda_max ~ 1e7..1e9: directions whose information is below it (far,
 * low-parallax depth) are treated as unobserved, where the reference mode inverts them (and, for an exactly-null
 * direction, inverts its rounding noise: the recovered information is then only meaningful if the pair is well posed).
 * inf36: 6x6 row-major (rotation 3 | translation 3); Ak144 (may be NULL): the 12x12 reduced information. */
int sadvio_ba_marginalize_relative(sadvio_ba_handle *h, int32_t w, int32_t kf_a, int32_t kf_b, int32_t eig_cut_mode, double *inf36,
                                   double *Ak144);

/* ---- one window spanning several GPUs (no reference counterpart: the reference is one process) ----
 * The landmarks of a window (with all their observations) are partitioned over `world` processes, one GPU each;
 * key-frames, cameras, pose priors and IMU factors are replicated. A marginalisation prior rides a sharded window in its
 * SPARSIFIED form (sadvio_ba_sparsify): the IMUPriordx factor is replicated (every rank passes it), each
 * PoseToLandmarkFactor goes to the rank that owns its landmark, with the landmark index of that rank's window — or as the DENSE
 * prior (sadvio_ba_set_dense_prior with host J / r0; round 5): every rank passes the whole prior and carries ALL of its kept landmarks
 * as variables of its window, with their observations on rank 0 only (the others list them without observations;
 * sadvio_amd/sharding.py builds such shards); rank 0 adds J^T J / J^T r to the all-reduced system, every rank evaluates the prior's
 * cost from row-block partials summed in index order (same bits on every rank). HARD PRECONDITION, checked on ranks != 0: a kept
 * landmark has observations on rank 0 only (SADVIO_E_INVALID_ARG otherwise — they would be counted twice in the all-reduced system). Factors that hold landmarks in the reduced system
 * without a dense prior (Landmark3DPrior, landmark chains), the handle-resident prior (SADVIO_PRIOR_RESIDENT: marginalize is not
 * available on a sharded window), line landmarks and marginalize_relative are refused on a sharded window. Every rank calls set_windows with ITS landmarks, then solve(); per LM step the library all-reduces [S | g | diag | per-rank cost partials] once and the
 * step's candidate-cost partials once; every rank then solves the identical reduced system redundantly
 * (no broadcast) and back-substitutes its own landmarks. Pose deltas and summaries are identical on all ranks.
 *
 * set_collective installs a caller-provided in-place sum all-reduce over device memory, enqueued on
 * `hip_stream` (returns 0 on success); comm_init_rccl installs the built-in one (ncclAllReduce, ncclDouble,
 * ncclSum over RCCL / xGMI) from a 128-byte ncclUniqueId created by rank 0 with rccl_unique_id and
 * distributed by the caller. Both must be called before set_windows. */
typedef int (*sadvio_allreduce_fn)(void *ctx, double *device_buf, int64_t count, void *hip_stream);
int sadvio_ba_set_collective(sadvio_ba_handle *h, int32_t rank, int32_t world, sadvio_allreduce_fn fn, void *ctx);
#define SADVIO_RCCL_ID_BYTES 128
int sadvio_ba_rccl_unique_id(void *id128);
int sadvio_ba_comm_init_rccl(sadvio_ba_handle *h, int32_t rank, int32_t world, const void *id128);
/* What the handle's collective really spans: with the built-in RCCL collective the numbers come from the communicator itself
 * (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), otherwise from set_collective. Any pointer may be NULL. */
int sadvio_ba_comm_info(sadvio_ba_handle *h, int32_t *nranks, int32_t *rank, int32_t *device, int32_t *is_rccl);

/* Sparse (NFR) prior factors of window `w`; replaces the previous list (n = 0 clears it). */
int sadvio_ba_set_sparse_priors(sadvio_ba_handle *h, int32_t w, int32_t n, const sadvio_sparse_prior *factors);

/* Run the Levenberg-Marquardt solve of every uploaded window: replaces the body of
 * AOptimizer::localMapBA / localMapVIOptimization from ceres::Solve on (AOptimizer.cpp:326,388).
 * `summaries` has n_windows entries (may be NULL). */
int sadvio_ba_solve(sadvio_ba_handle *h, const sadvio_solve_options *opts, sadvio_solve_summary *summaries);

/* Read back the solved deltas of window `w`; arrays are index-aligned with the window's input
 * arrays (landmark / key-frame order is never permuted). Any pointer may be NULL.
 * The adapter applies them as AOptimizer.cpp:329-340 (poses, landmarks) and :391-418 (v, ba, bg). */
int sadvio_ba_get_deltas(sadvio_ba_handle *h, int32_t w, double *pose_delta6, double *lmk_delta3,
                         double *dv3, double *dba3, double *dbg3);

/* Per-iteration log of the last solve of window `w`: what ceres::Solver::Summary::iterations holds for the reference
 * (AOptimizer.cpp:325-327 keeps the summary; FullReport prints it). Row i (8 doubles) = the state after i step attempts:
 *   [0] cost  [1] cost_change  [2] trust_region_radius  [3] step_norm  [4] relative_decrease  [5] step_is_successful
 *   [6] gradient_max_norm at that state (-1 for the state after the last attempt, which is not linearised again)
 *   [7] model_cost_change of the attempt.
 * Row 0 is the starting point. `n_rows` receives iterations + 1; at most `cap_rows` rows are written. */
int sadvio_ba_get_trace(sadvio_ba_handle *h, int32_t w, int32_t cap_rows, double *rows8, int32_t *n_rows);

/* Echo of the opaque ids of window `w` in output order (bit-exact identity check). */
int sadvio_ba_get_ids(sadvio_ba_handle *h, int32_t w, int64_t *kf_id, int64_t *lmk_id);

/* Linearise window `w` at zero deltas and return per-observation residuals/Jacobians
 * (row-major r[2], J_pose[2x6], J_lmk[2x3]): the GPU counterpart of calling
 * CostFunction::Evaluate on every block (…Analytic.h:52-90 / …Angular….h:55-111). Parity probe. */
int sadvio_ba_linearize(sadvio_ba_handle *h, int32_t w, const double *pose_delta6, const double *lmk_delta3,
                        double *r2, double *J_pose12, double *J_lmk6);

/* ALandmark::sanityCheck / avgChi2err / chi2err (ALandmark.cpp:98-146) for every landmark of window `w`, the gate
 * AOptimizer::landmarkOptimization applies before it writes a landmark back (AOptimizer.cpp:124-141):
 *   avg_chi2[l] = mean over the landmark's observations of |proj - meas|^2 / sigma^2, where an observation whose
 *                 projection fails the tests of Camera::project (depth < 0.1, pixel outside [0,width]x[0,height],
 *                 non-finite; Camera.cpp:26-52) counts 1000;
 *   inlier[l]   = (n_obs >= 2 && avg_chi2 <= 2)   (95 % chi2 test on a 2-D detection).
 * `image_wh` = n_cam x (width, height) of the caller's cameras (NULL: (2 cx, 2 cy), the bound the factors use).
 * Evaluated at the deltas given (NULL = zeros = the linearisation origin; the reference tests the landmark's pose
 * BEFORE the solved delta is applied). sigma is the FEATURE's pixel sigma (AFeature::getSigma, 1.0 everywhere in the
 * reference, AFeature2D.h:18): `pixel_sigma` > 0 sets it; <= 0 selects the window's cam_sigma for pixel windows and
 * 1.0 for angular windows (whose cam_sigma is an angle). Angular windows: the measured pixel is recovered from the
 * stored bearing through K. Either output may be NULL. */
int sadvio_ba_landmark_chi2(sadvio_ba_handle *h, int32_t w, const double *pose_delta6, const double *lmk_delta3,
                            const double *image_wh, double pixel_sigma, double *avg_chi2, int32_t *inlier);

/* ---- visual-inertial initialisation: AOptimizer::VIInit (AOptimizer.cpp:448-581) ----
 * Unknowns: the gravity direction r_wi (2 parameters, R_w_i = exp_so3((r0, r1, 0)), :464-466, :537), one velocity
 * delta per frame that an IMUFactorInit touches (:459-463), the log-scale lambda (:478-481; constant unless
 * optim_scale) and the bias deltas dba, dbg (constant in VIInit, :469-476; `optim_bias` frees them, which is how the
 * reference's own factor test drives IMUFactorInit, imu_test.cpp:505-545). Residuals: IMUFactorInit
 * (residuals.hpp:302-410) per pair, + the two bias priors Landmark3DPrior(0, 0, I / sigma) (:503-516) when the biases
 * are free. Solved with the same Ceres-2.2 LM rules as the window solves (the reference: 50 iterations, f_tol 1e-3). */
typedef struct sadvio_viinit_problem {
    int32_t n_frames;
    int32_t n_factors;
    const double *T_f_w;              /* [n_frames][12] */
    const double *vel;                /* [n_frames][3] IMU::getVelocity of each frame */
    const sadvio_imu_factor *factors; /* IMUFactorInit(imu_i, imu_j): kf_i / kf_j index the frames (pairing rule :485-500:
                                         i = imu_j->getLastKF(), i != j); dt, delta_*, J_*, cov are read */
    int32_t optim_scale;
    int32_t optim_bias;
    double sigma_dba, sigma_dbg;      /* sqrt(dt_window) * b{acc,gyr}_noise (:504-512); read only when optim_bias */
} sadvio_viinit_problem;

typedef struct sadvio_viinit_result {
    double r_wi[2];
    double lambda;
    double dba[3], dbg[3];
    double R_w_i[9]; /* exp_so3((r_wi, 0)) row-major: what VIInit hands back through its R_w_i argument */
    double scale;    /* exp(lambda): VIInit's return value (:575) */
} sadvio_viinit_result;

/* dv3: [n_frames][3] velocity deltas (zero for frames no factor touches). The adapter applies the result as
 * AOptimizer.cpp:526-562 (velocities, T_f_w.translation() *= scale, T_f_w = T_f_w * T_w_i, landmarks, priors).
 * At most 48 frames (the whole solve runs inside one workgroup). */
int sadvio_ba_vi_init(sadvio_ba_handle *h, const sadvio_viinit_problem *prob, const sadvio_solve_options *opts,
                      sadvio_solve_summary *summary, sadvio_viinit_result *res, double *dv3);

/* Average device time in microseconds per kernel class since the last set_windows, measured
 * with hipEvents on the handle's stream (cfg.profile_kernels = 1). `names` receives pointers
 * to static strings. Returns the number of classes written (<= cap). */
int sadvio_ba_get_kernel_times(sadvio_ba_handle *h, int32_t cap, const char **names, double *avg_us,
                               int64_t *launches);

const char *sadvio_ba_last_error(sadvio_ba_handle *h);
const char *sadvio_ba_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SADVIO_BA_H */
