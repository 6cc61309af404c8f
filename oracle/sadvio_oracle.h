/*
 * oracle/sadvio_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, FP64) of the reference's BA hot path, used ONLY as the checker in
 * tests/, __graft_entry__.smoke() and the `cpu_baseline` leg of bench.py. Nothing in the product
 * path (sadvio_amd/, include/) may import, link or call anything under oracle/.
 *
 * Pinning status:
 *   - factor arithmetic, IMU pre-integration, marginalisation bookkeeping: pinned against the
 *     known answers of the reference's own tests (imu_test.cpp, marginalization_test.cpp,
 *     residual_test.cpp acceptance criteria) — see tests/test_oracle_*.py;
 *   - the Levenberg-Marquardt loop restates the PUBLISHED algorithm of Ceres Solver 2.2.0
 *     (third-party, absent from /root/reference; pinned by docker/Dockerfile:50). No Ceres build
 *     or Ceres output is available here, so end-to-end `localMapBA` iterates are
 *     "parity unpinned" beyond the reference's solver-level acceptance tests
 *     (imu_test.cpp:481-487,562-567), which the oracle passes.
 */
#ifndef SADVIO_ORACLE_H
#define SADVIO_ORACLE_H
#include "../include/sadvio_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_problem {
    const sadvio_flat_window *win;
    int32_t n_prior;
    const sadvio_pose_prior *priors;
    int32_t n_imu;
    const sadvio_imu_factor *imus;
    /* dense marginalisation prior (MarginalizationFactor) — n_full = 0 when absent */
    int32_t dp_n_full, dp_n;
    const double *dp_J;  /* n_full x n row-major */
    const double *dp_r0; /* n_full */
    int32_t dp_kf_keep, dp_kf_col, dp_n_keep;
    const int32_t *dp_lmk_index;
    const int32_t *dp_lmk_col;
    int32_t n_threads; /* landmark-parallel evaluation threads (reference: num_threads = 4) */
    /* sparse (NFR) prior factors, addMarginalizationResiduals sparse branch (…Analytic.cpp:363-426) */
    int32_t n_sparse;
    const sadvio_sparse_prior *sparse;
    /* line landmarks; NULL = none. Their solved 6-vectors go to line_delta6 ([n_line][6], may be NULL) */
    const sadvio_line_set *lines;
    double *line_delta6;
} oracle_problem;

/* Per-observation linearisation at the given deltas (NULL = zeros). Any output may be NULL. */
int oracle_linearize(const sadvio_flat_window *win, const double *pose_delta6, const double *lmk_delta3,
                     double *r2, double *J_pose12, double *J_lmk6, int32_t *valid);

/* Ceres-2.2-style LM over an explicit Schur complement (exact-arithmetic equivalent of the
 * reference's SPARSE_NORMAL_CHOLESKY solve). iter_log (may be NULL) receives 8 doubles per
 * iteration: [cost, cost_change, radius, step_norm, relative_decrease, successful, gradient_max, model_cost_change]. */
int oracle_solve(const oracle_problem *prob, const sadvio_solve_options *opts, sadvio_solve_summary *summary,
                 double *pose_delta6, double *lmk_delta3, double *dv3, double *dba3, double *dbg3,
                 double *iter_log, int32_t iter_log_cap);

/* marginalizeRelative (…Analytic.cpp:665-809): 6x6 information of the relative pose of two VO key-frames. */
int oracle_marginalize_relative(const sadvio_flat_window *win, int32_t kf_a, int32_t kf_b, int32_t eig_cut_mode, double *inf36, double *Ak144,
                                int32_t *m_out);

/* ALandmark::sanityCheck per landmark (ALandmark.cpp:98-146): mean chi2 of the landmark's observations (failed
 * projection = 1000) and the 95 % gate (n_obs >= 2 && mean <= 2). Either output may be NULL. */
int oracle_landmark_chi2(const sadvio_flat_window *win, const double *pose_delta6, const double *lmk_delta3,
                         const double *image_wh, double pixel_sigma, double *avg_chi2, int32_t *inlier);

/* One LM step computed two ways on the SAME linearisation (zero deltas): via the Schur
 * complement, and returns H (dense, full un-reduced) for an independent check in the tests. */
int oracle_first_step(const oracle_problem *prob, const sadvio_solve_options *opts, double *delta_pose6,
                      double *delta_lmk3, double *H_full, double *g_full, int32_t n_full_dim);

/* Factor probes for the known-answer tests. */
void oracle_factor_pixel(const double *T0, const double *K, const double *Tsf, const double *p0, const double *uv,
                         double sigma, const double *dpose, const double *dl, double *r, double *Jp, double *Jl,
                         int32_t *valid);
void oracle_factor_angular(const double *T0, const double *Tsf, const double *p0, const double *bearing,
                           double sigma, const double *dpose, const double *dl, double *r, double *Jp, double *Jl);
void oracle_factor_pose_prior(const double *T0, const double *Tprior, const double *inf_diag, const double *dpose,
                              double *r, double *J);
int oracle_factor_imu(const sadvio_imu_factor *f, const double *Ti0, const double *Tj0, const double *vi0,
                      const double *vj0, const double *params /*24: pi6 pj6 vi3 vj3 ba3 bg3*/, double *r9,
                      double *J /*9x24 row-major, blocks side by side*/);
void oracle_factor_imu_bias(const sadvio_imu_factor *f, const double *bai, const double *bgi, const double *baj,
                            const double *bgj, const double *params /*12*/, double *r6, double *J /*6x12*/);

/* Probe of one sparse prior factor (oracle/solver.c). */
int oracle_sparse_factor(const sadvio_flat_window *w, const sadvio_sparse_prior *s, const double *xp, const double *xv,
                         const double *xba, const double *xbg, const double *xl, double *r, double *J);

/* SO3 probes */
void oracle_so3_exp(const double *w, double *R);
void oracle_so3_log(const double *R, double *w);
void oracle_so3_right_jacobian(const double *w, double *J);

/* IMU pre-integration producer (IMU::processIMU, IMU.cpp:5-91). State struct is flat doubles. */
typedef struct oracle_imu_state {
    double acc[3], gyr[3];     /* measurement attached to this sample */
    double ba[3], bg[3], v[3];
    double T_f_w[12];          /* pose of the frame carrying this sample */
    double delta_R[9], delta_v[3], delta_p[3];
    double cov[81];
    double J_dR_bg[9], J_dv_ba[9], J_dv_bg[9], J_dp_ba[9], J_dp_bg[9];
    double ts_ns;              /* timestamp (ns) */
    int32_t is_keyframe;
    int32_t pad;
} oracle_imu_state;
/* cur <- processIMU(last, last_kf's biases); noise = (gyr_noise, acc_noise, rate_hz). Returns 1 on success. */
int oracle_imu_process(oracle_imu_state *cur, const oracle_imu_state *last, const double *kf_ba,
                       const double *kf_bg, double gyr_noise, double acc_noise, double rate_hz);
/* IMU::biasDeltaCorrection (IMU.cpp:104-108) */
void oracle_imu_bias_correction(oracle_imu_state *s, const double *d_ba, const double *d_bg);

/* ---- marginalisation (Marginalization::computeSchurComplement & co, marginalization.cpp) ---- */
typedef struct oracle_marg_request {
    const sadvio_flat_window *win; /* factor_type / cam_sigma as the reference's marginalize() uses them */
    int32_t kf_marg;               /* frame0 */
    int32_t kf_keep;               /* frame1 when it carries an IMU (15 columns), else -1 */
    int32_t marg_has_imu;          /* frame0->getIMU(): +9 marginalised columns (marginalization.cpp:44-47) */
    int32_t n_marg;
    const int32_t *lmk_marg;       /* window landmark indices, order of _lmk_to_marg */
    int32_t n_keep;
    const int32_t *lmk_keep;       /* order of _lmk_to_keep (resurrected landmarks last, :116-139) */
    const sadvio_imu_factor *imu;  /* IMUFactor+IMUBiasFactor(frame0, frame1) or NULL */
    int32_t n_prior;
    const sadvio_pose_prior *priors; /* PosePriordx blocks; .kf must be kf_marg or kf_keep */
    /* previous dense prior (MarginalizationFactor(_marginalization_last)); last_n_full = 0 if none */
    int32_t last_n_full, last_n;
    const double *last_J, *last_r0;
    int32_t last_kf;     /* window index of the previous prior's kept frame (== kf_marg) or -1 */
    int32_t last_kf_col;
    int32_t last_n_keep;
    const int32_t *last_lmk_index; /* window landmark indices of the previous prior's kept landmarks */
    const int32_t *last_lmk_col;
    int32_t eig_cut_mode; /* SADVIO_EIG_CUT_* (sadvio_ba.h): 0 = the reference's absolute 1e-12, 1 = with the noise floor */
    int32_t pad;
} oracle_marg_request;

typedef struct oracle_marg_result {
    int32_t m, n, n_full;
    int32_t kf_col; /* column of kf_keep in the reduced prior (after the -m shift), -1 if none */
} oracle_marg_result;

/* Returns SADVIO_E_REFUSED when n < 4 (marginalization.cpp:215-216). Output buffers (any may be NULL):
 * lmk_col[n_keep], A_full[(m+n)^2], b_full[m+n], Ak[n*n], bk[n], U[n*n] (n x n_full packed row-major),
 * Lambda[n], J[n*n] (n_full x n packed row-major), r0[n]. */
int oracle_marginalize(const oracle_marg_request *rq, oracle_marg_result *res, int32_t *lmk_col, double *A_full,
                       double *b_full, double *Ak, double *bk, double *U, double *Lambda, double *J, double *r0);

/* NFR sparsification of a dense prior into the sparse-branch factor list (sparsifyVIO / sparsifyVO,
 * marginalization.cpp:362-514). out has room for n_keep + 1 factors. */
int oracle_sparsify(const sadvio_flat_window *w, int32_t vio, int32_t n_full, int32_t n, const double *J, int32_t kf_keep,
                    int32_t kf_col, int32_t n_keep, const int32_t *lmk_index, const int32_t *lmk_col, int32_t *n_out,
                    sadvio_sparse_prior *out);

/* ---- VI initialisation (AOptimizer::VIInit, AOptimizer.cpp:448-581; IMUFactorInit, residuals.hpp:302-410) ---- */
/* params15 = r_wi[2] dv_i[3] dv_j[3] dba[3] dbg[3] lambda[1]; J = 9 x 15 row-major (whitened, blocks side by side). */
int oracle_factor_imu_init(const sadvio_imu_factor *f, const double *Ti, const double *Tj, const double *vi, const double *vj,
                           const double *params15, double *r9, double *J);
int oracle_viinit(const sadvio_viinit_problem *prob, const sadvio_solve_options *opts, sadvio_solve_summary *summary,
                  sadvio_viinit_result *res, double *dv3);

/* Symmetric eigen-decomposition (cyclic Jacobi), eigenvalues ascending, V column-eigenvectors row-major. */
void oracle_sym_eig(const double *A, int32_t n, double *evals, double *V);

#ifdef __cplusplus
}
#endif
#endif
