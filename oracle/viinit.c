/*
 * oracle/viinit.c — TEST INFRASTRUCTURE ONLY (see sadvio_oracle.h).
 *
 * CPU restatement of AOptimizer::VIInit (AOptimizer.cpp:448-581): IMUFactorInit (residuals.hpp:302-410) over the
 * gravity direction (2), one velocity delta per frame, the bias deltas and the log scale, solved by the same
 * Ceres-2.2 Levenberg-Marquardt rules as oracle_solve (solver.c) on a dense normal matrix.
 * Pinned by the reference's own factor test: imu_test.cpp:489-545 (scale 0.5 recovered to 1e-2 from one factor with
 * every block free) and its Jacobian criterion (analytic vs numeric, sum of differences <= 1e-5).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "factors.h"
#include "sadvio_oracle.h"

static const double G_W[3] = {0.0, 0.0, -9.81}; /* IMU.h:8 */

/* params: r_wi[2] dvi[3] dvj[3] dba[3] dbg[3] lambda[1] (15). J: 9 x 15 row-major, whitened. */
static void factor_imu_init(const sadvio_imu_factor *f, const double *W, const double *Ti, const double *Tj, const double *vi0,
                            const double *vj0, const double *p, double *r9, double *J) {
    const double w[3] = {p[0], p[1], 0.0};
    double Rwi[9];
    so3_exp(w, Rwi); /* :309-310 */
    double vi[3], vj[3];
    for (int a = 0; a < 3; a++) { vi[a] = vi0[a] + p[2 + a]; vj[a] = vj0[a] + p[5 + a]; }
    const double *dba = p + 8, *dbg = p + 11;
    const double lambda = p[14], dt = f->dt;
    /* dR = (DeltaR exp(J_dR_bg dbg))^T R_i R_j^T  (:326-327) */
    double jb[3], E[9], DR[9], RiRjT[9], dR[9], DRt[9];
    m3_vec(f->J_dR_bg, dbg, jb);
    so3_exp(jb, E);
    m3_mul(f->delta_R, E, DR);
    m3_transpose(DR, DRt);
    m3_mul_t(Ti, Tj, RiRjT);
    m3_mul(DRt, RiRjT, dR);
    double e[9];
    so3_log(dR, e); /* r_dr */
    double RiRwi[9];
    m3_mul(Ti, Rwi, RiRwi);
    /* positions of the frames in the world: T^-1 translation = -R^T t */
    double pi[3], pj[3];
    for (int a = 0; a < 3; a++) {
        pi[a] = -(Ti[a] * Ti[9] + Ti[3 + a] * Ti[10] + Ti[6 + a] * Ti[11]);
        pj[a] = -(Tj[a] * Tj[9] + Tj[3 + a] * Tj[10] + Tj[6 + a] * Tj[11]);
    }
    double av[3], ap[3], t3[3];
    const double es = exp(lambda);
    for (int a = 0; a < 3; a++) {
        av[a] = (vj[a] - vi[a]) - G_W[a] * dt;                                        /* :329 */
        ap[a] = es * (pj[a] - pi[a]) - vi[a] * dt - 0.5 * G_W[a] * dt * dt;            /* :333-334 */
    }
    double bv[3], bp[3], t[3];
    m3_vec(f->J_dv_bg, dbg, bv); m3_vec(f->J_dv_ba, dba, t);
    for (int a = 0; a < 3; a++) bv[a] += t[a] + f->delta_v[a];
    m3_vec(f->J_dp_bg, dbg, bp); m3_vec(f->J_dp_ba, dba, t);
    for (int a = 0; a < 3; a++) bp[a] += t[a] + f->delta_p[a];
    m3_vec(RiRwi, av, t3);
    for (int a = 0; a < 3; a++) e[3 + a] = t3[a] - bv[a];
    m3_vec(RiRwi, ap, t3);
    for (int a = 0; a < 3; a++) e[6 + a] = t3[a] - bp[a];
    for (int i = 0; i < 9; i++) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += W[i * 9 + k] * e[k];
        r9[i] = s;
    }
    if (!J) return;
    double U[9 * 15];
    memset(U, 0, sizeof(U));
    /* gravity direction (:346-357): -R_i R_wi [a]x Jr(w)[:, 0:2] */
    double Jrw[9], S[9], M[9], N[9];
    so3_right_jacobian(w, Jrw);
    so3_skew(av, S); m3_mul(RiRwi, S, M); m3_mul(M, Jrw, N);
    for (int i = 0; i < 3; i++) for (int c = 0; c < 2; c++) U[(3 + i) * 15 + c] = -N[3 * i + c];
    so3_skew(ap, S); m3_mul(RiRwi, S, M); m3_mul(M, Jrw, N);
    for (int i = 0; i < 3; i++) for (int c = 0; c < 2; c++) U[(6 + i) * 15 + c] = -N[3 * i + c];
    for (int i = 0; i < 3; i++)
        for (int c = 0; c < 3; c++) {
            U[(3 + i) * 15 + 2 + c] = -RiRwi[3 * i + c];           /* v_i (:360-366) */
            U[(6 + i) * 15 + 2 + c] = -RiRwi[3 * i + c] * dt;
            U[(3 + i) * 15 + 5 + c] = RiRwi[3 * i + c];            /* v_j (:369-374) */
            U[(3 + i) * 15 + 8 + c] = -f->J_dv_ba[3 * i + c];      /* dba (:377-383) */
            U[(6 + i) * 15 + 8 + c] = -f->J_dp_ba[3 * i + c];
            U[(3 + i) * 15 + 11 + c] = -f->J_dv_bg[3 * i + c];     /* dbg (:386-395) */
            U[(6 + i) * 15 + 11 + c] = -f->J_dp_bg[3 * i + c];
        }
    /* d r_dr / d dbg = -Jr(r_dr)^-1 dR^T Jr(J_dR_bg dbg) J_dR_bg */
    double Jre[9], Jrei[9], dRt[9], Jrb[9], A[9], B[9], C9[9];
    so3_right_jacobian(e, Jre);
    m3_inverse(Jre, Jrei);
    m3_transpose(dR, dRt);
    so3_right_jacobian(jb, Jrb);
    m3_mul(Jrei, dRt, A); m3_mul(A, Jrb, B); m3_mul(B, f->J_dR_bg, C9);
    for (int i = 0; i < 3; i++) for (int c = 0; c < 3; c++) U[i * 15 + 11 + c] = -C9[3 * i + c];
    /* scale (:398-405): R_i R_wi (p_j - p_i)  — as coded, without the exp(lambda) factor of the true derivative */
    double dpv[3] = {pj[0] - pi[0], pj[1] - pi[1], pj[2] - pi[2]};
    m3_vec(RiRwi, dpv, t3);
    for (int i = 0; i < 3; i++) U[(6 + i) * 15 + 14] = t3[i];
    for (int i = 0; i < 9; i++)
        for (int c = 0; c < 15; c++) {
            double s = 0;
            for (int k = 0; k < 9; k++) s += W[i * 9 + k] * U[k * 15 + c];
            J[i * 15 + c] = s;
        }
}

int oracle_factor_imu_init(const sadvio_imu_factor *f, const double *Ti, const double *Tj, const double *vi, const double *vj,
                           const double *params15, double *r9, double *J) {
    double W[81];
    int rc = imu_sqrt_information(f->cov, W);
    if (rc) return rc;
    factor_imu_init(f, W, Ti, Tj, vi, vj, params15, r9, J);
    return 0;
}

typedef struct {
    const sadvio_viinit_problem *P;
    double *W;       /* n_factors x 81 */
    int *vcol;       /* column of each frame's velocity delta, -1 = not in the program */
    int c_ba, c_bg, c_l, D, rows;
} vctx;

static void unpack(const vctx *c, const double *x, int fi, double *p15, int *col15) {
    const sadvio_imu_factor *f = c->P->factors + fi;
    p15[0] = x[0]; p15[1] = x[1]; col15[0] = 0; col15[1] = 1;
    for (int a = 0; a < 3; a++) {
        col15[2 + a] = c->vcol[f->kf_i] + a; col15[5 + a] = c->vcol[f->kf_j] + a;
        p15[2 + a] = x[col15[2 + a]]; p15[5 + a] = x[col15[5 + a]];
        col15[8 + a] = c->c_ba >= 0 ? c->c_ba + a : -1; col15[11 + a] = c->c_bg >= 0 ? c->c_bg + a : -1;
        p15[8 + a] = c->c_ba >= 0 ? x[c->c_ba + a] : 0.0; p15[11 + a] = c->c_bg >= 0 ? x[c->c_bg + a] : 0.0;
    }
    col15[14] = c->c_l; p15[14] = c->c_l >= 0 ? x[c->c_l] : 0.0;
}

/* cost = 1/2 sum r^2; when H != NULL also accumulates H = J^T J (dense D x D), g = J^T r and keeps r / J rows for
 * the model cost change (Jrows: rows x D). */
static double evaluate(const vctx *c, const double *x, double *H, double *g, double *Jrows, double *rrows) {
    const sadvio_viinit_problem *P = c->P;
    const int D = c->D;
    double cost = 0;
    if (H) { memset(H, 0, sizeof(double) * (size_t)D * D); memset(g, 0, sizeof(double) * D); memset(Jrows, 0, sizeof(double) * (size_t)c->rows * D); }
    for (int fi = 0; fi < P->n_factors; fi++) {
        const sadvio_imu_factor *f = P->factors + fi;
        double p15[15], r9[9], J[135];
        int col[15];
        unpack(c, x, fi, p15, col);
        factor_imu_init(f, c->W + 81 * fi, P->T_f_w + 12 * f->kf_i, P->T_f_w + 12 * f->kf_j, P->vel + 3 * f->kf_i, P->vel + 3 * f->kf_j,
                        p15, r9, H ? J : NULL);
        for (int i = 0; i < 9; i++) cost += r9[i] * r9[i];
        if (H)
            for (int i = 0; i < 9; i++) {
                rrows[9 * fi + i] = r9[i];
                for (int a = 0; a < 15; a++)
                    if (col[a] >= 0) Jrows[(size_t)(9 * fi + i) * D + col[a]] += J[i * 15 + a];
            }
    }
    if (P->optim_bias) { /* Landmark3DPrior(0, 0, I / sigma): r = x / sigma (residuals.hpp:512-522) */
        for (int k = 0; k < 2; k++) {
            const int c0 = k ? c->c_bg : c->c_ba;
            const double is = 1.0 / (k ? P->sigma_dbg : P->sigma_dba);
            for (int a = 0; a < 3; a++) {
                const double r = is * x[c0 + a];
                cost += r * r;
                if (H) { const int row = 9 * P->n_factors + 3 * k + a; rrows[row] = r; Jrows[(size_t)row * D + c0 + a] = is; }
            }
        }
    }
    if (H)
        for (int row = 0; row < c->rows; row++) {
            const double *Jr = Jrows + (size_t)row * D;
            for (int i = 0; i < D; i++) {
                if (Jr[i] == 0.0) continue;
                g[i] += Jr[i] * rrows[row];
                for (int j = 0; j < D; j++) H[(size_t)i * D + j] += Jr[i] * Jr[j];
            }
        }
    return 0.5 * cost;
}

static int chol_solve_d(double *A, double *b, int n) {
    for (int j = 0; j < n; j++) {
        double s = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(s > 0.0) || !isfinite(s)) return 1;
        double d = sqrt(s);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double t = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = t / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double t = b[i];
        for (int k = 0; k < i; k++) t -= A[(size_t)i * n + k] * b[k];
        b[i] = t / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double t = b[i];
        for (int k = i + 1; k < n; k++) t -= A[(size_t)k * n + i] * b[k];
        b[i] = t / A[(size_t)i * n + i];
    }
    return 0;
}

int oracle_viinit(const sadvio_viinit_problem *P, const sadvio_solve_options *o, sadvio_solve_summary *sum,
                  sadvio_viinit_result *res, double *dv3) {
    vctx c;
    memset(&c, 0, sizeof(c));
    c.P = P;
    c.vcol = (int *)malloc(sizeof(int) * (size_t)(P->n_frames > 0 ? P->n_frames : 1));
    for (int i = 0; i < P->n_frames; i++) c.vcol[i] = -1;
    int D = 2;
    for (int fi = 0; fi < P->n_factors; fi++) {
        const sadvio_imu_factor *f = P->factors + fi;
        if (f->kf_i < 0 || f->kf_i >= P->n_frames || f->kf_j < 0 || f->kf_j >= P->n_frames || f->kf_i == f->kf_j) { free(c.vcol); return SADVIO_E_INVALID_ARG; }
    }
    for (int i = 0; i < P->n_frames; i++) { /* columns in frame order */
        int used = 0;
        for (int fi = 0; fi < P->n_factors; fi++) used |= (P->factors[fi].kf_i == i || P->factors[fi].kf_j == i);
        if (used) { c.vcol[i] = D; D += 3; }
    }
    c.c_ba = c.c_bg = c.c_l = -1;
    if (P->optim_bias) { c.c_ba = D; c.c_bg = D + 3; D += 6; }
    if (P->optim_scale) { c.c_l = D; D += 1; }
    c.D = D;
    c.rows = 9 * P->n_factors + (P->optim_bias ? 6 : 0);
    sadvio_solve_summary S;
    memset(&S, 0, sizeof(S));
    double *x = (double *)calloc((size_t)D, 8), *cand = (double *)calloc((size_t)D, 8), *delta = (double *)calloc((size_t)D, 8);
    int rc = SADVIO_OK;
    if (P->n_factors == 0) { /* nothing to optimise: Ceres returns at once on an empty program */
        S.termination = SADVIO_TERM_GRADIENT_TOL;
        goto finish;
    }
    c.W = (double *)malloc(sizeof(double) * 81 * (size_t)P->n_factors);
    for (int fi = 0; fi < P->n_factors; fi++)
        if (imu_sqrt_information(P->factors[fi].cov, c.W + 81 * fi)) { rc = SADVIO_E_INVALID_ARG; goto finish; }
    {
        double *H = (double *)malloc(sizeof(double) * (size_t)D * D), *g = (double *)malloc(sizeof(double) * D);
        double *A = (double *)malloc(sizeof(double) * (size_t)D * D), *rhs = (double *)malloc(sizeof(double) * D);
        double *Jrows = (double *)malloc(sizeof(double) * (size_t)c.rows * D), *rrows = (double *)malloc(sizeof(double) * c.rows);
        double *s = (double *)malloc(sizeof(double) * D);
        double x_cost = evaluate(&c, x, H, g, Jrows, rrows);
        for (int i = 0; i < D; i++) s[i] = o->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[(size_t)i * D + i])) : 1.0;
        S.initial_cost = x_cost;
        double radius = o->initial_trust_region_radius, decrease_factor = 2.0, x_norm = 0.0;
        int iter = 0, n_invalid = 0, term = SADVIO_TERM_NO_CONVERGENCE, done = 0;
        double gmax = 0;
        for (int i = 0; i < D; i++) gmax = fmax(gmax, fabs(g[i]));
        if (iter >= o->max_num_iterations) done = 1;
        else if (gmax <= o->gradient_tolerance) { done = 1; term = SADVIO_TERM_GRADIENT_TOL; }
        else if (radius <= o->min_trust_region_radius) { done = 1; term = SADVIO_TERM_MIN_RADIUS; }
        while (!done) {
            iter++;
            memcpy(A, H, sizeof(double) * (size_t)D * D);
            memcpy(rhs, g, sizeof(double) * D);
            for (int i = 0; i < D; i++) {
                double s2 = s[i] * s[i];
                double d = fmin(fmax(s2 * H[(size_t)i * D + i], o->min_lm_diagonal), o->max_lm_diagonal);
                A[(size_t)i * D + i] += d / radius / s2;
            }
            int fail = chol_solve_d(A, rhs, D);
            double mcc = 0;
            if (!fail) {
                for (int i = 0; i < D; i++) { delta[i] = -rhs[i]; if (!isfinite(delta[i])) fail = 1; }
                /* model cost change = -(J d)^T (r + J d / 2) */
                for (int row = 0; row < c.rows && !fail; row++) {
                    double jd = 0;
                    for (int i = 0; i < D; i++) jd += Jrows[(size_t)row * D + i] * delta[i];
                    mcc -= jd * (rrows[row] + 0.5 * jd);
                }
            }
            if (fail || !(mcc > 0.0)) {
                n_invalid++;
                S.num_unsuccessful_steps++;
                if (n_invalid >= o->max_num_consecutive_invalid_steps) { term = SADVIO_TERM_FAILURE; break; }
                radius *= 0.5;
            } else {
                n_invalid = 0;
                double sn2 = 0;
                for (int i = 0; i < D; i++) { cand[i] = x[i] + delta[i]; sn2 += delta[i] * delta[i]; }
                double cand_cost = evaluate(&c, cand, NULL, NULL, NULL, NULL);
                if (sqrt(sn2) <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) { term = SADVIO_TERM_PARAMETER_TOL; break; }
                double cost_change = x_cost - cand_cost;
                if (fabs(cost_change) <= o->function_tolerance * x_cost) { term = SADVIO_TERM_FUNCTION_TOL; break; }
                double rel_dec = (cand_cost >= DBL_MAX) ? -DBL_MAX : cost_change / mcc;
                if (rel_dec > o->min_relative_decrease) {
                    memcpy(x, cand, sizeof(double) * D);
                    double n2 = 0;
                    for (int i = 0; i < D; i++) n2 += x[i] * x[i];
                    x_norm = sqrt(n2);
                    x_cost = evaluate(&c, x, H, g, Jrows, rrows);
                    gmax = 0;
                    for (int i = 0; i < D; i++) gmax = fmax(gmax, fabs(g[i]));
                    S.num_successful_steps++;
                    double t = 2.0 * rel_dec - 1.0;
                    radius = fmin(o->max_trust_region_radius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
                    decrease_factor = 2.0;
                } else {
                    radius /= decrease_factor;
                    decrease_factor *= 2.0;
                    S.num_unsuccessful_steps++;
                }
            }
            if (iter >= o->max_num_iterations) { term = SADVIO_TERM_NO_CONVERGENCE; break; }
            if (gmax <= o->gradient_tolerance) { term = SADVIO_TERM_GRADIENT_TOL; break; }
            if (radius <= o->min_trust_region_radius) { term = SADVIO_TERM_MIN_RADIUS; break; }
        }
        S.iterations = iter; S.termination = term; S.final_cost = x_cost; S.final_radius = radius;
        if (term == SADVIO_TERM_FAILURE) rc = SADVIO_E_NOT_USABLE;
        free(H); free(g); free(A); free(rhs); free(Jrows); free(rrows); free(s);
    }
finish:
    if (sum) *sum = S;
    if (res) {
        memset(res, 0, sizeof(*res));
        res->r_wi[0] = x[0]; res->r_wi[1] = x[1];
        res->lambda = c.c_l >= 0 ? x[c.c_l] : 0.0;
        for (int a = 0; a < 3; a++) { res->dba[a] = c.c_ba >= 0 ? x[c.c_ba + a] : 0.0; res->dbg[a] = c.c_bg >= 0 ? x[c.c_bg + a] : 0.0; }
        const double w[3] = {x[0], x[1], 0.0};
        so3_exp(w, res->R_w_i);
        res->scale = exp(res->lambda);
    }
    if (dv3)
        for (int i = 0; i < P->n_frames; i++)
            for (int a = 0; a < 3; a++) dv3[3 * i + a] = c.vcol[i] >= 0 ? x[c.vcol[i] + a] : 0.0;
    free(x); free(cand); free(delta); free(c.vcol); free(c.W);
    return rc;
}
