/*
 * oracle/marg.c — TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Restatement of the dense marginalisation prior of the reference:
 *   Marginalization::preMarginalize index layout      (marginalization.cpp:38-113)
 *   computeInformationAndGradient  A = sum J^T J, b = + sum J^T r   (:145-211, quirk B.7)
 *   computeSchurComplement  Amm symmetrised, eigen pseudo-inverse with 1e-12 cut,
 *                           Ak = Arr - Arm Amm^+ Arm^T, bk = brr - Arm Amm^+ bmm   (:213-265)
 *   rankReveallingDecomposition  Ak = U Lambda U^T keeping lambda > 1e-12   (:318-342)
 *   computeJacobiansAndResiduals  J = Lambda^{1/2} U^T, r0 = -Lambda^{-1/2} U^T bk   (:516-530)
 * and of the block list built by BundleAdjustmentCERESAnalytic::marginalize (…Analytic.cpp:431-617)
 * / AngularAdjustmentCERESAnalytic::marginalize (…Angular….cpp:488-693): IMUFactor + IMUBiasFactor,
 * reprojection factors of kept and marginalised landmarks seen from frame0, the previous
 * MarginalizationFactor, PosePriordx blocks. Every block is evaluated at zero deltas.
 * Eigen's SelfAdjointEigenSolver is replaced by a cyclic Jacobi eigen-solver: the prior (J^T J,
 * J^T r0) is invariant to the eigenvector sign / ordering conventions.
 */
#include <stdio.h>
#include <stdlib.h>
#include "factors.h"
#include "sadvio_oracle.h"

#define MARG_EPS 1e-12 /* marginalization.hpp:56 */
static void small_inverse(const double *A, int n, double *Ai);

/* Eigenvalue cut of the pseudo-inverse / rank-revealing decomposition (sadvio_ba.h: SADVIO_EIG_CUT_*).
 *   SADVIO_EIG_CUT_REFERENCE (0): the reference's arithmetic — keep lambda > 1e-12, absolute (Marginalization::_eps,
 *     marginalization.hpp:58, applied at marginalization.cpp:237,322). On the reference's own test fixture
 *     (marginalization_test.cpp, |Amm| ~ 1e5) the null eigenvalue computes to +-1e-11 — rounding noise ABOVE that cut — so
 *     whether an exactly-null direction is kept depends on the sign of a rounding error, there as here; n_full is therefore
 *     not reproducible across eigen-solvers in this mode, the prior's information (J^T J, J^T r0) is, to rounding.
 *   SADVIO_EIG_CUT_NOISE_FLOOR (1): the reference's constant with the standard noise floor n*eps*lambda_max, i.e. what the
 *     cut is meant to do (drop the null space) — reproducible n_full, at the price of dropping directions whose information
 *     lies between 1e-12 and the floor (far, low-parallax depth), which the reference keeps. */
static double marg_cut(const double *ev, int n, int mode) {
    if (mode != SADVIO_EIG_CUT_NOISE_FLOOR) return MARG_EPS;
    double mx = 0;
    for (int i = 0; i < n; i++) mx = fmax(mx, fabs(ev[i]));
    return fmax(MARG_EPS, (double)n * 2.220446049250313e-16 * mx);
}

void oracle_sym_eig(const double *Ain, int32_t n, double *evals, double *V) {
    double *A = (double *)malloc(sizeof(double) * (size_t)n * n);
    memcpy(A, Ain, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) {
            diag += A[(size_t)i * n + i] * A[(size_t)i * n + i];
            for (int j = i + 1; j < n; j++) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) evals[i] = A[(size_t)i * n + i];
    /* ascending order, like Eigen::SelfAdjointEigenSolver */
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (evals[j] < evals[m]) m = j;
        if (m != i) {
            double t = evals[i]; evals[i] = evals[m]; evals[m] = t;
            for (int k = 0; k < n; k++) {
                double v = V[(size_t)k * n + i]; V[(size_t)k * n + i] = V[(size_t)k * n + m]; V[(size_t)k * n + m] = v;
            }
        }
    }
    free(A);
}

/* A += J^T J over index-mapped blocks, b += J^T r  (marginalization.cpp:153-190) */
static void accumulate(double *A, double *b, int N, const double *J, const double *r, int rows, int ncols,
                       const int *col) {
    for (int a = 0; a < ncols; a++) {
        if (col[a] < 0) continue;
        double g = 0;
        for (int q = 0; q < rows; q++) g += J[q * ncols + a] * r[q];
        b[col[a]] += g;
        for (int c2 = 0; c2 < ncols; c2++) {
            if (col[c2] < 0) continue;
            double h = 0;
            for (int q = 0; q < rows; q++) h += J[q * ncols + a] * J[q * ncols + c2];
            A[(size_t)col[a] * N + col[c2]] += h;
        }
    }
}

int oracle_marginalize(const oracle_marg_request *rq, oracle_marg_result *res, int32_t *lmk_col_out, double *A_full,
                       double *b_full, double *Ak_out, double *bk_out, double *U_out, double *Lambda_out, double *J_out,
                       double *r0_out) {
    const sadvio_flat_window *w = rq->win;
    /* index layout, marginalization.cpp:38-113 */
    int m = 6 + (rq->marg_has_imu ? 9 : 0) + 3 * rq->n_marg;
    int n = (rq->kf_keep >= 0 ? 15 : 0) + 3 * rq->n_keep;
    int N = m + n;
    int *lcol = (int *)malloc(sizeof(int) * (size_t)(w->n_lmk > 0 ? w->n_lmk : 1));
    for (int l = 0; l < w->n_lmk; l++) lcol[l] = -1;
    int idx = 6 + (rq->marg_has_imu ? 9 : 0);
    for (int k = 0; k < rq->n_marg; k++) { lcol[rq->lmk_marg[k]] = idx; idx += 3; }
    int kf_keep_col = -1;
    if (rq->kf_keep >= 0) { kf_keep_col = idx; idx += 15; }
    for (int k = 0; k < rq->n_keep; k++) { lcol[rq->lmk_keep[k]] = idx; idx += 3; }
    if (res) { res->m = m; res->n = n; res->n_full = 0; res->kf_col = kf_keep_col >= 0 ? kf_keep_col - m : -1; }
    if (lmk_col_out)
        for (int k = 0; k < rq->n_keep; k++) lmk_col_out[k] = lcol[rq->lmk_keep[k]] - m;
    if (n < 4) { free(lcol); return SADVIO_E_REFUSED; } /* :215-216 */

    double *A = (double *)calloc((size_t)N * N, sizeof(double));
    double *b = (double *)calloc((size_t)N, sizeof(double));
    static const double z[24] = {0};

    /* IMUFactor + IMUBiasFactor (frame0, frame1), …Analytic.cpp:451-508 */
    if (rq->imu && rq->kf_keep >= 0 && rq->marg_has_imu) {
        const sadvio_imu_factor *f = rq->imu;
        int i = rq->kf_marg, j = rq->kf_keep;
        double W[81], r[9], Jpi[54], Jpj[54], Jvi[27], Jvj[27], Jba[27], Jbg[27], J[9 * 24];
        imu_sqrt_information(f->cov, W);
        imu_consts ic = {f->dt, f->delta_R, f->delta_v, f->delta_p, f->J_dR_bg, f->J_dv_ba, f->J_dv_bg, f->J_dp_ba,
                         f->J_dp_bg, W};
        const double *vi = w->kf_vel ? w->kf_vel + 3 * i : z, *vj = w->kf_vel ? w->kf_vel + 3 * j : z;
        factor_imu(&ic, w->kf_T_f_w + 12 * i, w->kf_T_f_w + 12 * j, vi, vj, z, z, z, z, z, z, r, Jpi, Jpj, Jvi, Jvj,
                   Jba, Jbg);
        int col[24];
        for (int q = 0; q < 9; q++) {
            for (int a = 0; a < 6; a++) { J[q * 24 + a] = Jpi[q * 6 + a]; J[q * 24 + 6 + a] = Jpj[q * 6 + a]; }
            for (int a = 0; a < 3; a++) {
                J[q * 24 + 12 + a] = Jvi[q * 3 + a]; J[q * 24 + 15 + a] = Jvj[q * 3 + a];
                J[q * 24 + 18 + a] = Jba[q * 3 + a]; J[q * 24 + 21 + a] = Jbg[q * 3 + a];
            }
        }
        for (int a = 0; a < 6; a++) { col[a] = a; col[6 + a] = kf_keep_col + a; }
        for (int a = 0; a < 3; a++) {
            col[12 + a] = 6 + a; col[15 + a] = kf_keep_col + 6 + a; col[18 + a] = 9 + a; col[21 + a] = 12 + a;
        }
        accumulate(A, b, N, J, r, 9, 24, col);
        double rb[6], Jb[72], sa, sg;
        const double *bai = w->kf_ba ? w->kf_ba + 3 * i : z, *bgi = w->kf_bg ? w->kf_bg + 3 * i : z;
        const double *baj = w->kf_ba ? w->kf_ba + 3 * j : z, *bgj = w->kf_bg ? w->kf_bg + 3 * j : z;
        factor_imu_bias(f->dt, f->bacc_noise, f->bgyr_noise, bai, bgi, baj, bgj, z, z, z, z, rb, &sa, &sg);
        memset(Jb, 0, sizeof(Jb));
        int colb[12];
        for (int a = 0; a < 3; a++) {
            Jb[a * 12 + a] = -sa; Jb[(3 + a) * 12 + 3 + a] = -sg; Jb[a * 12 + 6 + a] = sa; Jb[(3 + a) * 12 + 9 + a] = sg;
            colb[a] = 9 + a; colb[3 + a] = 12 + a; colb[6 + a] = kf_keep_col + 9 + a; colb[9 + a] = kf_keep_col + 12 + a;
        }
        accumulate(A, b, N, Jb, rb, 6, 12, colb);
    }

    /* reprojection factors of kept then marginalised landmarks seen from frame0, …Analytic.cpp:510-572 */
    for (int pass = 0; pass < 2; pass++) {
        int cnt = pass == 0 ? rq->n_keep : rq->n_marg;
        const int32_t *list = pass == 0 ? rq->lmk_keep : rq->lmk_marg;
        for (int k = 0; k < cnt; k++) {
            int l = list[k];
            for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
                if (w->obs_kf[o] != rq->kf_marg) continue;
                int cam = w->obs_cam[o];
                double r[2], Jp[12], Jl[6], J[18];
                double sigma = w->cam_sigma ? w->cam_sigma[cam] : 1.0;
                if (w->factor_type == SADVIO_FACTOR_PIXEL)
                    factor_pixel(w->kf_T_f_w + 12 * rq->kf_marg, w->cam_K + 4 * cam, w->cam_T_s_f + 12 * cam,
                                 w->lmk_p + 3 * l, w->obs_meas + 2 * o, sigma, z, z, r, Jp, Jl);
                else
                    factor_angular(w->kf_T_f_w + 12 * rq->kf_marg, w->cam_T_s_f + 12 * cam, w->lmk_p + 3 * l,
                                   w->obs_meas + 3 * o, sigma, z, z, r, Jp, Jl);
                int col[9];
                for (int q = 0; q < 2; q++) {
                    for (int a = 0; a < 6; a++) J[q * 9 + a] = Jp[q * 6 + a];
                    for (int a = 0; a < 3; a++) J[q * 9 + 6 + a] = Jl[q * 3 + a];
                }
                for (int a = 0; a < 6; a++) col[a] = a;
                for (int a = 0; a < 3; a++) col[6 + a] = lcol[l] + a;
                accumulate(A, b, N, J, r, 2, 9, col);
            }
        }
    }

    /* previous prior, …Analytic.cpp:574-603: evaluated at zero deltas => r = r0, J = column slices */
    if (rq->last_n_full > 0) {
        int nl = rq->last_n, nf = rq->last_n_full;
        int *col = (int *)malloc(sizeof(int) * (size_t)nl);
        for (int a = 0; a < nl; a++) col[a] = -1;
        if (rq->last_kf >= 0) {
            int base = (rq->last_kf == rq->kf_marg) ? 0 : ((rq->last_kf == rq->kf_keep) ? kf_keep_col : -1);
            int width = (rq->last_kf == rq->kf_marg) ? (rq->marg_has_imu ? 15 : 6) : 15;
            if (base >= 0)
                for (int a = 0; a < width; a++) col[rq->last_kf_col + a] = base + a;
        }
        for (int k = 0; k < rq->last_n_keep; k++) {
            if (rq->last_lmk_col[k] < 0) continue;
            int lc = lcol[rq->last_lmk_index[k]];
            if (lc < 0) continue;
            for (int a = 0; a < 3; a++) col[rq->last_lmk_col[k] + a] = lc + a;
        }
        accumulate(A, b, N, rq->last_J, rq->last_r0, nf, nl, col);
        free(col);
    }

    /* PosePriordx blocks, …Analytic.cpp:605-617 (+ frame1's in the Angular variant, …Angular….cpp:682-693) */
    for (int k = 0; k < rq->n_prior; k++) {
        const sadvio_pose_prior *pr = rq->priors + k;
        int base = pr->kf == rq->kf_marg ? 0 : (pr->kf == rq->kf_keep ? kf_keep_col : -1);
        if (base < 0) continue;
        double r[6], J[36];
        int col[6];
        factor_pose_prior(w->kf_T_f_w + 12 * pr->kf, pr->T_prior, pr->inf_diag, z, r, J);
        for (int a = 0; a < 6; a++) col[a] = base + a;
        accumulate(A, b, N, J, r, 6, 6, col);
    }
    if (A_full) memcpy(A_full, A, sizeof(double) * (size_t)N * N);
    if (b_full) memcpy(b_full, b, sizeof(double) * (size_t)N);

    /* Schur complement with eigen pseudo-inverse, :234-248 */
    double *Amm = (double *)malloc(sizeof(double) * (size_t)m * m);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) Amm[(size_t)i * m + j] = 0.5 * (A[(size_t)i * N + j] + A[(size_t)j * N + i]);
    double *ev = (double *)malloc(sizeof(double) * (size_t)m), *V = (double *)malloc(sizeof(double) * (size_t)m * m);
    oracle_sym_eig(Amm, m, ev, V);
    double *Ainv = (double *)calloc((size_t)m * m, sizeof(double));
    double cut_m = marg_cut(ev, m, rq->eig_cut_mode);
    for (int k = 0; k < m; k++) {
        if (!(ev[k] > cut_m)) continue;
        double iv = 1.0 / ev[k];
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) Ainv[(size_t)i * m + j] += V[(size_t)i * m + k] * iv * V[(size_t)j * m + k];
    }
    /* T = Arm * Ainv (n x m) */
    double *T = (double *)calloc((size_t)n * m, sizeof(double));
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++) {
            double a = A[(size_t)(m + i) * N + k];
            if (a == 0.0) continue;
            for (int j = 0; j < m; j++) T[(size_t)i * m + j] += a * Ainv[(size_t)k * m + j];
        }
    double *Ak = (double *)malloc(sizeof(double) * (size_t)n * n), *bk = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) {
        double s = b[m + i];
        for (int k = 0; k < m; k++) s -= T[(size_t)i * m + k] * b[k];
        bk[i] = s;
        for (int j = 0; j < n; j++) {
            double h = A[(size_t)(m + i) * N + m + j];
            for (int k = 0; k < m; k++) h -= T[(size_t)i * m + k] * A[(size_t)(m + j) * N + k]; /* Arm^T */
            Ak[(size_t)i * n + j] = h;
        }
    }
    if (Ak_out) memcpy(Ak_out, Ak, sizeof(double) * (size_t)n * n);
    if (bk_out) memcpy(bk_out, bk, sizeof(double) * (size_t)n);

    /* rank revealing decomposition, :318-342. Eigen reads the lower triangle of Ak. */
    double *Aks = (double *)malloc(sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) Aks[(size_t)i * n + j] = Aks[(size_t)j * n + i] = Ak[(size_t)i * n + j];
    double *ev2 = (double *)malloc(sizeof(double) * (size_t)n), *V2 = (double *)malloc(sizeof(double) * (size_t)n * n);
    oracle_sym_eig(Aks, n, ev2, V2);
    int nf = 0;
    double cut_n = marg_cut(ev2, n, rq->eig_cut_mode);
    for (int k = 0; k < n; k++)
        if (ev2[k] > cut_n) nf++;
    if (res) res->n_full = nf;
    int c = 0;
    for (int k = 0; k < n; k++) {
        if (!(ev2[k] > cut_n)) continue;
        double lam = ev2[k];
        if (Lambda_out) Lambda_out[c] = lam;
        double dot = 0;
        for (int i = 0; i < n; i++) {
            if (U_out) U_out[(size_t)i * nf + c] = V2[(size_t)i * n + k];
            dot += V2[(size_t)i * n + k] * bk[i];
            if (J_out) J_out[(size_t)c * n + i] = sqrt(lam) * V2[(size_t)i * n + k]; /* :525 */
        }
        if (r0_out) r0_out[c] = -sqrt(1.0 / lam) * dot;                             /* :526-527 */
        c++;
    }
    free(A); free(b); free(lcol); free(Amm); free(ev); free(V); free(Ainv); free(T); free(Ak); free(bk);
    free(Aks); free(ev2); free(V2);
    return SADVIO_OK;
}

/* ---- relative-pose information (NFR) between two key-frames ----------------------------------------------------
 * BundleAdjustmentCERESAnalytic::marginalizeRelative (…Analytic.cpp:665-809) + Marginalization::preMarginalizeRelative
 * (marginalization.cpp:532-588), VO frames. As coded: the list of landmarks to marginalise gets one entry per feature a
 * landmark of frame a has in frame b (:548-559: a stereo landmark is entered twice, its column index is the first
 * entry's, the second entry's 3 columns stay empty) and the reprojection factors of a landmark (its features in a and
 * b) are added once per entry (:741-770). Kept: pose a at m, pose b at m + 6. The pseudo-inverse of Amm goes through
 * its eigen-decomposition in the reference; Amm is block diagonal here (3x3 per landmark, empty blocks for the repeated
 * entries), so the decomposition is done block by block — the same eigenvalues, the same cut, the same pseudo-inverse.
 * Then Sigma_k = U diag(1 / lambda) U^T (:255-262) and inf = (J Sigma_k J^T)^-1 with J = [Ja Jb] of
 * Relative6DPose(T_w_a, T_w_b, T_a_b, I) at zero deltas (:784-807). */
int oracle_marginalize_relative(const sadvio_flat_window *w, int32_t kf_a, int32_t kf_b, int32_t eig_cut_mode, double *inf36,
                                double *Ak_out /*144 or NULL*/, int32_t *m_out) {
    static const double z[6] = {0, 0, 0, 0, 0, 0};
    if (w->has_imu) return SADVIO_E_INVALID_ARG;
    /* entries: (landmark, first index) in the order of preMarginalizeRelative */
    int cap = 16, ne = 0;
    int *el = (int *)malloc(sizeof(int) * cap), *ec = (int *)malloc(sizeof(int) * cap);
    int *first = (int *)malloc(sizeof(int) * (size_t)(w->n_lmk > 0 ? w->n_lmk : 1));
    for (int l = 0; l < w->n_lmk; l++) first[l] = -1;
    int last = 0;
    for (int l = 0; l < w->n_lmk; l++) {   /* frame a's landmarks, window order */
        int in_a = 0;
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) in_a |= w->obs_kf[o] == kf_a;
        if (!in_a) continue;
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            if (w->obs_kf[o] != kf_b) continue;
            if (ne == cap) { cap *= 2; el = (int *)realloc(el, sizeof(int) * cap); ec = (int *)realloc(ec, sizeof(int) * cap); }
            if (first[l] < 0) first[l] = last;
            el[ne] = l; ec[ne] = first[l]; ne++;
            last += 3;
        }
    }
    const int m = last, n = 12, N = m + n;
    if (m_out) *m_out = m;
    if (ne == 0) { free(el); free(ec); free(first); return SADVIO_E_REFUSED; }
    /* Amm blocks (3x3 per column index), Arm (12 x m), Arr (12 x 12); b is not needed for the information */
    double *Amm = (double *)calloc((size_t)m * 3, sizeof(double));      /* block l0: rows l0..l0+2 x 3 */
    double *Arm = (double *)calloc((size_t)n * m, sizeof(double));
    double Arr[144];
    memset(Arr, 0, sizeof(Arr));
    for (int e = 0; e < ne; e++) {
        const int l = el[e], lc = ec[e];
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            const int kf = w->obs_kf[o];
            if (kf != kf_a && kf != kf_b) continue;
            const int cam = w->obs_cam[o], pc = kf == kf_a ? 0 : 6;
            double r[2], Jp[12], Jl[6];
            const double sigma = w->cam_sigma ? w->cam_sigma[cam] : 1.0;
            if (w->factor_type == SADVIO_FACTOR_PIXEL)
                factor_pixel(w->kf_T_f_w + 12 * kf, w->cam_K + 4 * cam, w->cam_T_s_f + 12 * cam, w->lmk_p + 3 * l, w->obs_meas + 2 * o,
                             sigma, z, z, r, Jp, Jl);
            else
                factor_angular(w->kf_T_f_w + 12 * kf, w->cam_T_s_f + 12 * cam, w->lmk_p + 3 * l, w->obs_meas + 3 * o, sigma, z, z, r, Jp, Jl);
            for (int q = 0; q < 2; q++) {
                for (int a = 0; a < 3; a++) {
                    for (int b2 = 0; b2 < 3; b2++) Amm[(size_t)(lc + a) * 3 + b2] += Jl[q * 3 + a] * Jl[q * 3 + b2];
                    for (int p = 0; p < 6; p++) Arm[(size_t)(pc + p) * m + lc + a] += Jp[q * 6 + p] * Jl[q * 3 + a];
                }
                for (int p = 0; p < 6; p++) for (int p2 = 0; p2 < 6; p2++) Arr[(pc + p) * 12 + pc + p2] += Jp[q * 6 + p] * Jp[q * 6 + p2];
            }
        }
    }
    /* Ak = Arr - Arm Amm^+ Arm^T, Amm^+ block by block; the cut of the whole matrix (marg_cut over ALL eigenvalues) */
    double *ev = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1)), *V = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1) * 3);
    for (int c0 = 0; c0 < m; c0 += 3) {
        double B[9], e3[3], V3[9];
        for (int a = 0; a < 3; a++) for (int b2 = 0; b2 < 3; b2++) B[3 * a + b2] = 0.5 * (Amm[(size_t)(c0 + a) * 3 + b2] + Amm[(size_t)(c0 + b2) * 3 + a]);
        oracle_sym_eig(B, 3, e3, V3);
        for (int k = 0; k < 3; k++) { ev[c0 + k] = e3[k]; for (int a = 0; a < 3; a++) V[(size_t)(c0 + a) * 3 + k] = V3[3 * a + k]; }
    }
    const double cut = marg_cut(ev, m, eig_cut_mode);
    double Ak[144];
    memcpy(Ak, Arr, sizeof(Ak));
    for (int c0 = 0; c0 < m; c0 += 3) {
        double Pi[9];
        memset(Pi, 0, sizeof(Pi));
        for (int k = 0; k < 3; k++) {
            if (!(ev[c0 + k] > cut)) continue;
            const double iv = 1.0 / ev[c0 + k];
            for (int a = 0; a < 3; a++) for (int b2 = 0; b2 < 3; b2++) Pi[3 * a + b2] += V[(size_t)(c0 + a) * 3 + k] * iv * V[(size_t)(c0 + b2) * 3 + k];
        }
        for (int i = 0; i < 12; i++) {
            double t[3];
            for (int b2 = 0; b2 < 3; b2++) t[b2] = Arm[(size_t)i * m + c0] * Pi[b2] + Arm[(size_t)i * m + c0 + 1] * Pi[3 + b2] + Arm[(size_t)i * m + c0 + 2] * Pi[6 + b2];
            for (int j = 0; j < 12; j++) Ak[i * 12 + j] -= t[0] * Arm[(size_t)j * m + c0] + t[1] * Arm[(size_t)j * m + c0 + 1] + t[2] * Arm[(size_t)j * m + c0 + 2];
        }
    }
    if (Ak_out) memcpy(Ak_out, Ak, sizeof(Ak));
    /* rank revealing decomposition (Eigen reads the lower triangle), Sigma_k = U diag(1 / lambda) U^T */
    double Aks[144], ev2[12], V2[144], Sk[144];
    for (int i = 0; i < 12; i++) for (int j = 0; j <= i; j++) Aks[i * 12 + j] = Aks[j * 12 + i] = Ak[i * 12 + j];
    oracle_sym_eig(Aks, 12, ev2, V2);
    /* Ak has an exact 6-dimensional null space (the gauge: only the relative pose is observed), whose eigenvalues
     * compute to the rounding noise of the Schur complement — a sum over the marginalised landmarks of terms as large as
     * lambda_max, i.e. ~ eps * lambda_max * (number of terms), 1e-7 .. 4e-6 on the test windows against 1e3 for the
     * smallest real eigenvalue. The reference's absolute 1e-12 (marginalization.cpp:322) keeps whichever of them come out
     * positive (1 / lambda ~ 1e6: its information matrix is then noise); the cut here is the noise floor of that sum, which
     * returns the exact-arithmetic value of the reference's formula (cf. marg_cut) — SADVIO_EIG_CUT_NOISE_FLOOR; SADVIO_EIG_CUT_REFERENCE
     * applies the absolute constant as coded. */
    double cut_n = marg_cut(ev2, 12, eig_cut_mode);
    if (eig_cut_mode == SADVIO_EIG_CUT_NOISE_FLOOR) {
        int n_l = 0;
        for (int l = 0; l < w->n_lmk; l++) n_l += first[l] >= 0;
        cut_n = fmax(cut_n, cut_n * (2.0 + n_l));
    }
    memset(Sk, 0, sizeof(Sk));
    for (int k = 0; k < 12; k++) {
        if (!(ev2[k] > cut_n)) continue;
        const double iv = 1.0 / ev2[k];
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) Sk[i * 12 + j] += V2[i * 12 + k] * iv * V2[j * 12 + k];
    }
    /* Relative6DPose(T_w_a, T_w_b, T_a_b = T_a_w T_w_b, I) at zero deltas */
    double Twa[12], Twb[12], Tab[12], W[36], r6[6], Ja[36], Jb[36], J[72];
    se3_inverse(w->kf_T_f_w + 12 * kf_a, Twa); se3_inverse(w->kf_T_f_w + 12 * kf_b, Twb);
    se3_mul(w->kf_T_f_w + 12 * kf_a, Twb, Tab);
    memset(W, 0, sizeof(W));
    for (int i = 0; i < 6; i++) W[7 * i] = 1.0;
    factor_relative_pose(Twa, Twb, Tab, W, z, z, r6, Ja, Jb);
    for (int i = 0; i < 6; i++) for (int q = 0; q < 6; q++) { J[i * 12 + q] = Ja[i * 6 + q]; J[i * 12 + 6 + q] = Jb[i * 6 + q]; }
    double JS[72], cov[36];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 12; j++) { double s2 = 0; for (int k = 0; k < 12; k++) s2 += J[i * 12 + k] * Sk[k * 12 + j]; JS[i * 12 + j] = s2; }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { double s2 = 0; for (int k = 0; k < 12; k++) s2 += JS[i * 12 + k] * J[j * 12 + k]; cov[i * 6 + j] = s2; }
    small_inverse(cov, 6, inf36);
    free(el); free(ec); free(first); free(Amm); free(Arm); free(ev); free(V);
    return SADVIO_OK;
}

/* ---- NFR sparsification (Marginalization::sparsifyVIO :362-408, sparsifyVO :410-514) -------------------------
 * Input: the dense prior J = Lambda^1/2 U^T (n_full x n) of oracle_marginalize; U and Sigma = Lambda^-1 are
 * recovered from it (lambda_c = |J_c|^2, U[:,c] = J_c / sqrt(lambda_c)), then the reference's formulas are applied
 * as coded: J~ = J_f U, cov = J~ Sigma J~^T, symmetric square root of the information through a 3x3 / 15x15
 * eigen-decomposition with the eigenvalue cut 1e-12. Output: the factor list the sparse branch of
 * addMarginalizationResiduals builds (…Analytic.cpp:363-426). */
static void small_inverse(const double *A, int n, double *Ai) {
    double M[15 * 30];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { M[i * 2 * n + j] = A[i * n + j]; M[i * 2 * n + n + j] = (i == j); }
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (fabs(M[r * 2 * n + c]) > fabs(M[p * 2 * n + c])) p = r;
        if (p != c) for (int j = 0; j < 2 * n; j++) { double t = M[c * 2 * n + j]; M[c * 2 * n + j] = M[p * 2 * n + j]; M[p * 2 * n + j] = t; }
        double d = 1.0 / M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] *= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f != 0.0) for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Ai[i * n + j] = M[i * 2 * n + n + j];
}

/* sqrt-information of a factor: S = cov (rows x rows). invert_first: VIO path (inf = cov^-1, keep eig > eps, sqrt);
 * else VO path (eig of cov, 1/eig for eig > eps, sqrt). W row-major rows x rows. */
static void nfr_sqrt_info(const double *S, int rows, int invert_first, double *W) {
    double M[225], ev[15], V[225];
    if (invert_first) small_inverse(S, rows, M); else memcpy(M, S, sizeof(double) * (size_t)rows * rows);
    for (int i = 0; i < rows; i++) for (int j = 0; j < i; j++) { double s = 0.5 * (M[i * rows + j] + M[j * rows + i]); M[i * rows + j] = M[j * rows + i] = s; }
    oracle_sym_eig(M, rows, ev, V);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < rows; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) {
                double e = ev[k] > MARG_EPS ? (invert_first ? ev[k] : 1.0 / ev[k]) : 0.0;
                s += V[i * rows + k] * sqrt(e) * V[j * rows + k];
            }
            W[i * rows + j] = s;
        }
}

/* cov = (Jsel U[cidx,:]) Sigma (Jsel U[cidx,:])^T */
static void nfr_cov(const double *J, int nf, int n, const double *lam, const double *Jsel, int rows, int cols, const int *cidx,
                    double *S) {
    memset(S, 0, sizeof(double) * (size_t)rows * rows);
    for (int c = 0; c < nf; c++) {
        double jt[15];
        for (int a = 0; a < rows; a++) {
            double s = 0;
            for (int k = 0; k < cols; k++) s += Jsel[a * cols + k] * (J[(size_t)c * n + cidx[k]] / sqrt(lam[c]));  /* J_f U */
            jt[a] = s;
        }
        for (int a = 0; a < rows; a++) for (int b = 0; b < rows; b++) S[a * rows + b] += jt[a] * (1.0 / lam[c]) * jt[b];
    }
}

int oracle_sparsify(const sadvio_flat_window *w, int32_t vio, int32_t nf, int32_t n, const double *J, int32_t kf_keep,
                    int32_t kf_col, int32_t n_keep, const int32_t *lmk_index, const int32_t *lmk_col, int32_t *n_out,
                    sadvio_sparse_prior *out) {
    *n_out = 0;
    if (n == 0 || nf == 0) return SADVIO_E_REFUSED;
    double *lam = (double *)malloc(sizeof(double) * (size_t)nf);
    for (int c = 0; c < nf; c++) { double s = 0; for (int i = 0; i < n; i++) s += J[(size_t)c * n + i] * J[(size_t)c * n + i]; lam[c] = s; }
    int cnt = 0;
    if (vio) {
        if (kf_keep < 0) { free(lam); return SADVIO_E_INVALID_ARG; }
        const double *T = w->kf_T_f_w + 12 * kf_keep; /* R row-major 9, t 3 */
        double tsk[9], Rt[9];
        so3_skew(T + 9, tsk);
        m3_mul(T, tsk, Rt);
        /* absolute factor of the kept frame first (…Analytic.cpp:373-386 adds it first) */
        {
            double Jsel[225], S[225];
            int cidx[15];
            memset(Jsel, 0, sizeof(Jsel));
            for (int a = 0; a < 15; a++) { Jsel[a * 15 + a] = 1.0; cidx[a] = kf_col + a; }
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    Jsel[i * 15 + j] = T[3 * i + j];            /* block(0, f, 3, 3) = R */
                    Jsel[i * 15 + 3 + j] = T[3 * i + j];        /* block(0, f+3, 3, 3) = R */
                    Jsel[(3 + i) * 15 + 3 + j] = T[3 * i + j];  /* block(3, f+3, 3, 3) = R */
                }
            nfr_cov(J, nf, n, lam, Jsel, 15, 15, cidx, S);
            sadvio_sparse_prior *o = out + cnt++;
            memset(o, 0, sizeof(*o));
            o->type = SADVIO_SPARSE_IMU_PRIOR; o->kf = kf_keep; o->lmk0 = o->lmk1 = -1;
            memcpy(o->T_prior, T, 96);
            for (int a = 0; a < 3; a++) {
                o->v_prior[a] = w->kf_vel ? w->kf_vel[3 * kf_keep + a] : 0.0;
                o->ba_prior[a] = w->kf_ba ? w->kf_ba[3 * kf_keep + a] : 0.0;
                o->bg_prior[a] = w->kf_bg ? w->kf_bg[3 * kf_keep + a] : 0.0;
            }
            nfr_sqrt_info(S, 15, 1, o->sqrt_inf);
        }
        for (int k = 0; k < n_keep; k++) {
            if (lmk_col[k] < 0) continue;
            double Jsel[27], S[9];
            int cidx[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    Jsel[i * 9 + j] = T[3 * i + j];          /* landmark block: R */
                    Jsel[i * 9 + 3 + j] = -Rt[3 * i + j];    /* frame rotation block: -R [t]x */
                    Jsel[i * 9 + 6 + j] = T[3 * i + j];      /* frame translation block: R */
                }
            for (int a = 0; a < 3; a++) { cidx[a] = lmk_col[k] + a; cidx[3 + a] = kf_col + a; cidx[6 + a] = kf_col + 3 + a; }
            nfr_cov(J, nf, n, lam, Jsel, 3, 9, cidx, S);
            sadvio_sparse_prior *o = out + cnt++;
            memset(o, 0, sizeof(*o));
            o->type = SADVIO_SPARSE_POSE_TO_LMK; o->kf = kf_keep; o->lmk0 = lmk_index[k]; o->lmk1 = -1;
            se3_apply(T, w->lmk_p + 3 * lmk_index[k], o->delta);
            nfr_sqrt_info(S, 3, 1, o->sqrt_inf);
        }
    } else {
        /* greedy chain ordering on |trace(Ak_ij)|, Ak = J^T J (:417-456) */
        int K = 0;
        int *idx = (int *)malloc(sizeof(int) * (size_t)(n_keep > 0 ? n_keep : 1));
        for (int k = 0; k < n_keep; k++) if (lmk_col[k] >= 0) idx[K++] = k;
        if (K < 2) { free(idx); free(lam); return SADVIO_E_REFUSED; }
        double *mi = (double *)calloc((size_t)K * K, sizeof(double));
        for (int a = 0; a < K; a++)
            for (int b = a + 1; b < K; b++) {
                double tr = 0;
                for (int c = 0; c < nf; c++)
                    for (int q = 0; q < 3; q++) tr += J[(size_t)c * n + lmk_col[idx[a]] + q] * J[(size_t)c * n + lmk_col[idx[b]] + q];
                mi[a * K + b] = mi[b * K + a] = fabs(tr);
            }
        int *order = (int *)malloc(sizeof(int) * (size_t)K), no = 0;
        int mr = 0, mc = 0;
        double best = -1;
        for (int j = 0; j < K; j++) for (int i = 0; i < K; i++) if (mi[i * K + j] > best) { best = mi[i * K + j]; mr = i; mc = j; } /* column-major visit */
        order[no++] = mr; order[no++] = mc;
        for (int i = 0; i < K; i++) { mi[i * K + mr] = 0; mi[mr * K + i] = 0; mi[i * K + mc] = 0; }
        int cur = mc;
        for (;;) {
            int bc = 0; double bv = mi[cur * K];
            for (int j = 1; j < K; j++) if (mi[cur * K + j] > bv) { bv = mi[cur * K + j]; bc = j; }
            if (bv == 0) break;
            order[no++] = bc;
            for (int j = 0; j < K; j++) mi[cur * K + j] = 0;
            for (int i = 0; i < K; i++) mi[i * K + bc] = 0;
            cur = bc;
        }
        /* landmark with the prior: minimum entropy = minimum det of its 3x3 covariance block (:458-466) */
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        int root = 0; double best_det = 0;
        for (int k = 0; k < no; k++) {
            int cidx[3] = {lmk_col[idx[order[k]]], lmk_col[idx[order[k]]] + 1, lmk_col[idx[order[k]]] + 2};
            double S[9];
            nfr_cov(J, nf, n, lam, I3, 3, 3, cidx, S);
            double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
            if (k == 0 || det < best_det) { best_det = det; root = k; }
        }
        {
            int l = idx[order[root]];
            int cidx[3] = {lmk_col[l], lmk_col[l] + 1, lmk_col[l] + 2};
            double S[9];
            nfr_cov(J, nf, n, lam, I3, 3, 3, cidx, S);
            sadvio_sparse_prior *o = out + cnt++;
            memset(o, 0, sizeof(*o));
            o->type = SADVIO_SPARSE_LMK_PRIOR; o->kf = -1; o->lmk0 = lmk_index[l]; o->lmk1 = -1;
            memcpy(o->delta, w->lmk_p + 3 * lmk_index[l], 24);
            nfr_sqrt_info(S, 3, 0, o->sqrt_inf);
        }
        for (int k = 0; k + 1 < no; k++) {
            int la = idx[order[k]], lb = idx[order[k + 1]];
            double Jsel[18] = {1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1}, S[9];
            int cidx[6];
            for (int a = 0; a < 3; a++) { cidx[a] = lmk_col[la] + a; cidx[3 + a] = lmk_col[lb] + a; }
            nfr_cov(J, nf, n, lam, Jsel, 3, 6, cidx, S);
            sadvio_sparse_prior *o = out + cnt++;
            memset(o, 0, sizeof(*o));
            o->type = SADVIO_SPARSE_LMK_TO_LMK; o->kf = -1; o->lmk0 = lmk_index[la]; o->lmk1 = lmk_index[lb];
            for (int a = 0; a < 3; a++) o->delta[a] = w->lmk_p[3 * lmk_index[la] + a] - w->lmk_p[3 * lmk_index[lb] + a];
            nfr_sqrt_info(S, 3, 0, o->sqrt_inf);
        }
        free(idx); free(mi); free(order);
    }
    free(lam);
    *n_out = cnt;
    return SADVIO_OK;
}
