/*
 * oracle/solver.c — TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Restates, in plain C / FP64:
 *   - the graph the reference hands to Ceres in AOptimizer::localMapBA / localMapVIOptimization
 *     (AOptimizer.cpp:299-446) from a flattened window;
 *   - the trust-region loop of Ceres Solver 2.2.0 (TrustRegionMinimizer + LevenbergMarquardtStrategy,
 *     third-party, pinned by docker/Dockerfile:50; algorithm restated from its published sources:
 *     Jacobi scaling computed once at iteration 0 as 1/(1+||col||), LM diagonal
 *     clamp(||col_scaled||^2, min_lm_diagonal, max_lm_diagonal)/radius, step quality
 *     rho = (cost - cost_new) / model_cost_change, radius /= max(1/3, 1 - (2 rho - 1)^3) on
 *     success, radius /= decrease_factor (2, 4, 8 …) on failure, function / parameter / gradient
 *     tolerance tests in Ceres' order; the step that triggers function tolerance is NOT applied);
 *   - the linear solve (JtJ + D^2) y = Jt r, which the reference does with SPARSE_NORMAL_CHOLESKY /
 *     CHOLMOD on the un-reduced system, here by explicit landmark elimination (Schur complement)
 *     + dense Cholesky: identical in exact arithmetic; tests/test_oracle_solver.py checks the
 *     equivalence against a direct solve of the full normal equations.
 */
#include <float.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "factors.h"
#include "sadvio_oracle.h"

typedef struct {
    int rows, ncols;
    int col[27];       /* reduced-vector column of each Jacobian column, -1 = constant */
    double J[15 * 27]; /* rows x ncols row-major */
    double r[15];
} small_factor;

typedef struct {
    const oracle_problem *P;
    const sadvio_flat_window *w;
    double huber_a;   /* ceres::HuberLoss(a) on the visual factors, 0 = none */
    int dpf;          /* per-KF reduced block: 6 (VO) or 15 (VIO) */
    int Nr;           /* reduced dimension */
    int *kf_off;      /* [n_kf] offset in reduced vector or -1 */
    int *lmk_red;     /* [n_lmk] offset in reduced vector (dense-prior landmarks) or -1 */
    int *lmk_elim;    /* [n_lmk] 1 = eliminated by Schur */
    int *lmk_active;  /* [n_lmk] 1 = has a parameter block in the reduced program */
    int *line_off;    /* [n_line] offset of a line landmark's 6 columns in the reduced vector, -1 = constant / unused */
    /* linearisation storage */
    double *r, *Jp, *Jl; /* per obs */
    double *E;           /* per obs 6x3 = Jp^T Jl */
    double *Hll, *gl;    /* per lmk 9 / 3 */
    double *Hred, *gred; /* Nr*Nr / Nr */
    small_factor *sf;
    int n_sf, cap_sf;
    double *W_imu;       /* per imu factor 81 */
    /* dense prior */
    double *dp_JtJ;      /* mapped Nr x Nr contribution precomputed (J constant) */
    int *dp_colmap;      /* [dp_n] -> reduced column or -1 */
    double *dp_res;      /* n_full residual at current x */
    /* scaling */
    double *s_red, *s_lmk; /* jacobi scale, Nr and 3*n_lmk */
} ctx_t;

static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

static const double *vec3_or_zero(const double *base, int i) {
    static const double z[3] = {0, 0, 0};
    return base ? base + 3 * i : z;
}

/* ----- evaluation of all residual blocks at x ----- */
typedef struct {
    const double *xp, *xl, *xv, *xba, *xbg;
    const double *xline;   /* [n_line][6] or NULL */
} state_t;

/* One line observation as a small block over [key-frame 6 | line 6] (…Analytic.cpp:296-310 / Angular….cpp:316-333).
 * Returns 0 when both parameter blocks are constant (fixed cost). */
static int line_small(const ctx_t *c, const state_t *x, int l, int o, small_factor *f, int want_J, double *rho) {
    static const double z6[6] = {0, 0, 0, 0, 0, 0};
    const sadvio_flat_window *w = c->w;
    const sadvio_line_set *L = c->P->lines;
    const int kf = L->obs_kf[o], cam = L->obs_cam[o], po = c->kf_off[kf], lo = c->line_off[l];
    const double *dp = x && x->xp ? x->xp + 6 * kf : z6, *dl = x && x->xline ? x->xline + 6 * l : z6;
    double Jf[24], Jl[24];
    if (w->factor_type == SADVIO_FACTOR_PIXEL) {
        f->rows = 4;
        factor_line_pixel(w->kf_T_f_w + 12 * kf, w->cam_K + 4 * cam, w->cam_T_s_f + 12 * cam, L->line_T_w_l + 12 * l, L->line_model + 6 * l,
                          L->obs_meas + 4 * o, 1.0, dp, dl, f->r, want_J ? Jf : NULL, want_J ? Jl : NULL);
    } else {
        f->rows = 2;
        factor_line_angular(w->kf_T_f_w + 12 * kf, w->cam_T_s_f + 12 * cam, L->line_T_w_l + 12 * l, L->obs_meas + 6 * o, 1.0, dp, dl, f->r,
                            want_J ? Jf : NULL, want_J ? Jl : NULL);
    }
    f->ncols = 12;
    for (int q = 0; q < 6; q++) { f->col[q] = po < 0 ? -1 : po + q; f->col[6 + q] = lo < 0 ? -1 : lo + q; }
    /* the line blocks are added with the caller's loss function like the point blocks (…Analytic.cpp:303-306): ceres::HuberLoss(a)
     * + Corrector over the whole 4 (2) row block */
    double ssq = 0, sc = 1.0;
    for (int q = 0; q < f->rows; q++) ssq += f->r[q] * f->r[q];
    *rho = ssq;
    if (c->huber_a > 0.0 && ssq > c->huber_a * c->huber_a) {
        const double rr = sqrt(ssq);
        double rho1 = c->huber_a / rr;
        if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
        sc = sqrt(rho1);
        *rho = 2.0 * c->huber_a * rr - c->huber_a * c->huber_a;
        for (int q = 0; q < f->rows; q++) f->r[q] *= sc;
    }
    if (want_J) for (int i = 0; i < f->rows; i++) for (int q = 0; q < 6; q++) { f->J[i * 12 + q] = sc * Jf[i * 6 + q]; f->J[i * 12 + 6 + q] = sc * Jl[i * 6 + q]; }
    return po >= 0 || lo >= 0;
}

static void push_sf(ctx_t *c, const small_factor *f) {
    if (c->n_sf == c->cap_sf) {
        c->cap_sf = c->cap_sf ? 2 * c->cap_sf : 64;
        c->sf = (small_factor *)realloc(c->sf, (size_t)c->cap_sf * sizeof(small_factor));
    }
    c->sf[c->n_sf++] = *f;
}

static void eval_obs(const sadvio_flat_window *w, int l, int o, const double *xp, const double *xl, double *r,
                     double *Jp, double *Jl, int *valid) {
    int kf = w->obs_kf[o], cam = w->obs_cam[o];
    const double *T0 = w->kf_T_f_w + 12 * kf;
    const double *Tsf = w->cam_T_s_f + 12 * cam;
    double sigma = w->cam_sigma ? w->cam_sigma[cam] : 1.0;
    const double *p0 = w->lmk_p + 3 * l;
    static const double z6[6] = {0, 0, 0, 0, 0, 0};
    const double *dp = xp ? xp + 6 * kf : z6;
    const double *dl = xl ? xl + 3 * l : z6;
    int v = 1;
    if (w->factor_type == SADVIO_FACTOR_PIXEL)
        v = factor_pixel(T0, w->cam_K + 4 * cam, Tsf, p0, w->obs_meas + 2 * o, sigma, dp, dl, r, Jp, Jl);
    else
        factor_angular(T0, Tsf, p0, w->obs_meas + 3 * o, sigma, dp, dl, r, Jp, Jl);
    if (valid) *valid = v;
}

/* One sparse prior factor as a small dense block over the reduced vector. Returns 0 when every parameter block
 * of the factor is constant (the block then leaves the program: fixed cost). */
static int sparse_small(const ctx_t *c, const state_t *x, const sadvio_sparse_prior *s, small_factor *f, int want_J) {
    const sadvio_flat_window *w = c->w;
    static const double z3[3] = {0, 0, 0};
    memset(f->col, 0xff, sizeof(f->col));
    if (s->type == SADVIO_SPARSE_IMU_PRIOR) {
        int k = s->kf, po = c->kf_off[k];
        if (po < 0) return 0;
        double params[15], J[225];
        for (int q = 0; q < 6; q++) params[q] = x->xp[6 * k + q];
        for (int q = 0; q < 3; q++) { params[6 + q] = x->xv[3 * k + q]; params[9 + q] = x->xba[3 * k + q]; params[12 + q] = x->xbg[3 * k + q]; }
        factor_imu_prior(w->kf_T_f_w + 12 * k, vec3_or_zero(w->kf_vel, k), vec3_or_zero(w->kf_ba, k), vec3_or_zero(w->kf_bg, k),
                         s->T_prior, s->v_prior, s->ba_prior, s->bg_prior, s->sqrt_inf, params, f->r, want_J ? J : NULL);
        f->rows = 15; f->ncols = 15;
        for (int q = 0; q < 15; q++) f->col[q] = q < c->dpf ? po + q : -1;
        if (want_J) for (int i = 0; i < 15; i++) for (int a = 0; a < 15; a++) f->J[i * 15 + a] = J[i * 15 + a];
        return 1;
    }
    if (s->type == SADVIO_SPARSE_RELATIVE_POSE) {
        int a = s->kf, b = s->kf_b, pa = c->kf_off[a], pb = c->kf_off[b];
        if (pa < 0 && pb < 0) return 0;
        double Ja[36], Jb[36];
        factor_relative_pose(w->kf_T_f_w + 12 * a, w->kf_T_f_w + 12 * b, s->T_prior, s->sqrt_inf, x->xp + 6 * a, x->xp + 6 * b,
                             f->r, want_J ? Ja : NULL, want_J ? Jb : NULL);
        f->rows = 6; f->ncols = 12;
        for (int q = 0; q < 6; q++) { f->col[q] = pa < 0 ? -1 : pa + q; f->col[6 + q] = pb < 0 ? -1 : pb + q; }
        if (want_J) for (int i = 0; i < 6; i++) for (int q = 0; q < 6; q++) { f->J[i * 12 + q] = Ja[i * 6 + q]; f->J[i * 12 + 6 + q] = Jb[i * 6 + q]; }
        return 1;
    }
    f->rows = 3;
    if (s->type == SADVIO_SPARSE_POSE_TO_LMK) {
        int k = s->kf, l = s->lmk0, po = c->kf_off[k], lo = c->lmk_red[l];
        if (po < 0 && lo < 0) return 0;
        double Jp[18], Jl[9];
        factor_pose_to_landmark(w->kf_T_f_w + 12 * k, w->lmk_p + 3 * l, s->delta, s->sqrt_inf, x->xp + 6 * k, x->xl + 3 * l,
                                f->r, want_J ? Jp : NULL, want_J ? Jl : NULL);
        f->ncols = 9;
        for (int q = 0; q < 6; q++) f->col[q] = po < 0 ? -1 : po + q;
        for (int q = 0; q < 3; q++) f->col[6 + q] = lo < 0 ? -1 : lo + q;
        if (want_J) for (int i = 0; i < 3; i++) {
            for (int a = 0; a < 6; a++) f->J[i * 9 + a] = Jp[i * 6 + a];
            for (int a = 0; a < 3; a++) f->J[i * 9 + 6 + a] = Jl[i * 3 + a];
        }
        return 1;
    }
    if (s->type == SADVIO_SPARSE_LMK_PRIOR) {
        int l = s->lmk0, lo = c->lmk_red[l];
        if (lo < 0) return 0;
        factor_landmark_prior(w->lmk_p + 3 * l, s->delta, s->sqrt_inf, x->xl + 3 * l, f->r, want_J ? f->J : NULL);
        f->ncols = 3;
        for (int q = 0; q < 3; q++) f->col[q] = lo + q;
        return 1;
    }
    {
        int l0 = s->lmk0, l1 = s->lmk1, o0 = c->lmk_red[l0], o1 = c->lmk_red[l1];
        if (o0 < 0 && o1 < 0) return 0;
        double J0[9], J1[9];
        factor_landmark_to_landmark(w->lmk_p + 3 * l0, w->lmk_p + 3 * l1, s->delta, s->sqrt_inf, x->xl + 3 * l0, x->xl + 3 * l1,
                                    f->r, J0, J1);
        (void)z3;
        f->ncols = 6;
        for (int q = 0; q < 3; q++) { f->col[q] = o0 < 0 ? -1 : o0 + q; f->col[3 + q] = o1 < 0 ? -1 : o1 + q; }
        if (want_J) for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) { f->J[i * 6 + a] = J0[i * 3 + a]; f->J[i * 6 + 3 + a] = J1[i * 3 + a]; }
        return 1;
    }
}

/* ceres::HuberLoss(a)::Evaluate + Corrector (public Ceres 2.2 semantics: rho'' <= 0 => residual and Jacobian
 * scaled by sqrt(rho'), cost = rho / 2). a = 0: no loss function. Returns rho(|r|^2). */
static double apply_loss(double a, double *r, double *Jp, double *Jl) {
    double s = r[0] * r[0] + r[1] * r[1];
    if (!(a > 0.0) || s <= a * a) return s;
    double rr = sqrt(s);
    double rho1 = a / rr;
    if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
    double sc = sqrt(rho1);
    r[0] *= sc; r[1] *= sc;
    if (Jp) for (int i = 0; i < 12; i++) Jp[i] *= sc;
    if (Jl) for (int i = 0; i < 6; i++) Jl[i] *= sc;
    return 2.0 * a * rr - a * a;
}


/* ---- multi-threaded accumulation (n_threads > 1 only: the cpu_baseline leg of bench.py and the offline fixture generators) ----
 * The dense reduced system is summed into per-thread private copies over a static partition of the landmarks and reduced in
 * thread order afterwards: deterministic for a given thread count; with one thread the code below is the serial loop itself
 * (same buffers, same order), so every parity test sees the arithmetic it always saw. */
static int par_threads(const oracle_problem *P, size_t doubles_per_thread) {
    int T = P->n_threads > 1 ? P->n_threads : 1;
#ifndef _OPENMP
    T = 1;
#endif
    while (T > 1 && (size_t)T * doubles_per_thread > ((size_t)1 << 26)) T /= 2;   /* <= 512 MB of private copies */
    return T;
}
static int par_tid(void) {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}

/* cost only (candidate evaluation). Returns 1/2 sum r^2 over the reduced program. */
static double eval_cost(ctx_t *c, const state_t *x) {
    const sadvio_flat_window *w = c->w;
    double cost = 0;
#pragma omp parallel for schedule(static) reduction(+ : cost) if (c->P->n_threads > 1) num_threads(c->P->n_threads > 1 ? c->P->n_threads : 1)
    for (int l = 0; l < w->n_lmk; l++) {
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            int kf = w->obs_kf[o];
            if (c->kf_off[kf] < 0 && !c->lmk_active[l]) continue; /* constant block: fixed cost */
            double r[2];
            eval_obs(w, l, o, x->xp, x->xl, r, NULL, NULL, NULL);
            cost += apply_loss(c->huber_a, r, NULL, NULL);
        }
    }
    const oracle_problem *P = c->P;
    for (int k = 0; k < P->n_prior; k++) {
        const sadvio_pose_prior *pr = P->priors + k;
        if (c->kf_off[pr->kf] < 0) continue;
        double r[6];
        factor_pose_prior(w->kf_T_f_w + 12 * pr->kf, pr->T_prior, pr->inf_diag, x->xp + 6 * pr->kf, r, NULL);
        for (int i = 0; i < 6; i++) cost += r[i] * r[i];
    }
    for (int k = 0; k < P->n_sparse; k++) {
        small_factor sf;
        if (!sparse_small(c, x, P->sparse + k, &sf, 0)) continue;
        for (int q = 0; q < sf.rows; q++) cost += sf.r[q] * sf.r[q];
    }
    if (P->lines)
        for (int l = 0; l < P->lines->n_line; l++)
            for (int o = P->lines->line_obs_ptr[l]; o < P->lines->line_obs_ptr[l + 1]; o++) {
                small_factor sf;
                double rho;
                if (!line_small(c, x, l, o, &sf, 0, &rho)) continue;
                cost += rho;
            }
    for (int k = 0; k < P->n_imu; k++) {
        const sadvio_imu_factor *f = P->imus + k;
        int i = f->kf_i, j = f->kf_j;
        if (c->kf_off[i] < 0 && c->kf_off[j] < 0) continue;
        imu_consts ic = {f->dt, f->delta_R, f->delta_v, f->delta_p, f->J_dR_bg, f->J_dv_ba, f->J_dv_bg,
                         f->J_dp_ba, f->J_dp_bg, c->W_imu + 81 * k};
        double r[9];
        factor_imu(&ic, w->kf_T_f_w + 12 * i, w->kf_T_f_w + 12 * j, vec3_or_zero(w->kf_vel, i),
                   vec3_or_zero(w->kf_vel, j), x->xp + 6 * i, x->xp + 6 * j, x->xv + 3 * i, x->xv + 3 * j,
                   x->xba + 3 * i, x->xbg + 3 * i, r, NULL, NULL, NULL, NULL, NULL, NULL);
        for (int q = 0; q < 9; q++) cost += r[q] * r[q];
        double rb[6], sa, sg;
        factor_imu_bias(f->dt, f->bacc_noise, f->bgyr_noise, vec3_or_zero(w->kf_ba, i), vec3_or_zero(w->kf_bg, i),
                        vec3_or_zero(w->kf_ba, j), vec3_or_zero(w->kf_bg, j), x->xba + 3 * i, x->xbg + 3 * i,
                        x->xba + 3 * j, x->xbg + 3 * j, rb, &sa, &sg);
        for (int q = 0; q < 6; q++) cost += rb[q] * rb[q];
    }
    if (P->dp_n_full > 0) {
        /* MarginalizationFactor::Evaluate, marginalization.hpp:113-145: r = r0 + J dx */
        int n = P->dp_n, nf = P->dp_n_full;
        double *dx = (double *)xcalloc((size_t)n, sizeof(double));
        if (P->dp_kf_keep >= 0) {
            int k = P->dp_kf_keep;
            for (int q = 0; q < 6; q++) dx[P->dp_kf_col + q] = x->xp[6 * k + q];
            for (int q = 0; q < 3; q++) {
                dx[P->dp_kf_col + 6 + q] = x->xv[3 * k + q];
                dx[P->dp_kf_col + 9 + q] = x->xba[3 * k + q];
                dx[P->dp_kf_col + 12 + q] = x->xbg[3 * k + q];
            }
        }
        for (int q = 0; q < P->dp_n_keep; q++) {
            if (P->dp_lmk_col[q] < 0) continue;
            for (int a = 0; a < 3; a++) dx[P->dp_lmk_col[q] + a] = x->xl[3 * P->dp_lmk_index[q] + a];
        }
        for (int i = 0; i < nf; i++) {   /* cost only: c->dp_res (the residual at the linearisation point, used by
                                            model_cost_change of the following attempts) must survive a rejected candidate */
            double s = P->dp_r0[i];
            for (int j = 0; j < n; j++) s += P->dp_J[(size_t)i * n + j] * dx[j];
            cost += s * s;
        }
        free(dx);
    }
    return 0.5 * cost;
}

static void accum_small(ctx_t *c, const small_factor *f) {
    int Nr = c->Nr;
    for (int a = 0; a < f->ncols; a++) {
        int ca = f->col[a];
        if (ca < 0) continue;
        double g = 0;
        for (int q = 0; q < f->rows; q++) g += f->J[q * f->ncols + a] * f->r[q];
        c->gred[ca] += g;
        for (int b = 0; b < f->ncols; b++) {
            int cb = f->col[b];
            if (cb < 0) continue;
            double h = 0;
            for (int q = 0; q < f->rows; q++) h += f->J[q * f->ncols + a] * f->J[q * f->ncols + b];
            c->Hred[(size_t)ca * Nr + cb] += h;
        }
    }
}

/* residuals + Jacobians + normal-equation blocks at x. Returns cost. */
static double eval_full(ctx_t *c, const state_t *x) {
    const sadvio_flat_window *w = c->w;
    const oracle_problem *P = c->P;
    int Nr = c->Nr;
    memset(c->Hred, 0, sizeof(double) * (size_t)Nr * Nr);
    memset(c->gred, 0, sizeof(double) * (size_t)Nr);
    memset(c->Hll, 0, sizeof(double) * 9 * (size_t)w->n_lmk);
    memset(c->gl, 0, sizeof(double) * 3 * (size_t)w->n_lmk);
    c->n_sf = 0;
    double cost = 0;
#pragma omp parallel for schedule(static) reduction(+ : cost) if (P->n_threads > 1) num_threads(P->n_threads > 1 ? P->n_threads : 1)
    for (int l = 0; l < w->n_lmk; l++) {
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            int kf = w->obs_kf[o];
            double *r = c->r + 2 * o, *Jp = c->Jp + 12 * o, *Jl = c->Jl + 6 * o, *E = c->E + 18 * o;
            if (c->kf_off[kf] < 0 && !c->lmk_active[l]) {
                r[0] = r[1] = 0;
                memset(Jp, 0, 96); memset(Jl, 0, 48); memset(E, 0, 144);
                continue;
            }
            eval_obs(w, l, o, x->xp, x->xl, r, Jp, Jl, NULL);
            cost += apply_loss(c->huber_a, r, Jp, Jl);
            if (c->kf_off[kf] < 0) memset(Jp, 0, 96);
            if (!c->lmk_active[l]) memset(Jl, 0, 48);
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 3; b++) E[a * 3 + b] = Jp[a] * Jl[b] + Jp[6 + a] * Jl[3 + b];
            if (c->lmk_elim[l]) {
                double *H = c->Hll + 9 * l, *g = c->gl + 3 * l;
                for (int a = 0; a < 3; a++) {
                    g[a] += Jl[a] * r[0] + Jl[3 + a] * r[1];
                    for (int b = 0; b < 3; b++) H[3 * a + b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
                }
            }
        }
    }
    /* pose (and reduced-landmark) blocks: accumulation into the dense reduced system (serial with one thread) */
    {
    const size_t nH = (size_t)Nr * Nr + (size_t)Nr;
    const int T = par_threads(P, nH);
    double *priv = T > 1 ? (double *)xcalloc((size_t)T * nH, sizeof(double)) : NULL;
#pragma omp parallel num_threads(T) if (T > 1)
    {
    double *Hred = T > 1 ? priv + (size_t)par_tid() * nH : c->Hred;
    double *gred = T > 1 ? Hred + (size_t)Nr * Nr : c->gred;
#pragma omp for schedule(static)
    for (int l = 0; l < w->n_lmk; l++) {
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            int kf = w->obs_kf[o];
            int po = c->kf_off[kf];
            const double *r = c->r + 2 * o, *Jp = c->Jp + 12 * o, *Jl = c->Jl + 6 * o, *E = c->E + 18 * o;
            if (po >= 0) {
                for (int a = 0; a < 6; a++) {
                    gred[po + a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
                    for (int b = 0; b < 6; b++)
                        Hred[(size_t)(po + a) * Nr + po + b] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
                }
            }
            int lo = c->lmk_red[l];
            if (lo >= 0) {
                for (int a = 0; a < 3; a++) {
                    gred[lo + a] += Jl[a] * r[0] + Jl[3 + a] * r[1];
                    for (int b = 0; b < 3; b++)
                        Hred[(size_t)(lo + a) * Nr + lo + b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
                }
                if (po >= 0)
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 3; b++) {
                            Hred[(size_t)(po + a) * Nr + lo + b] += E[a * 3 + b];
                            Hred[(size_t)(lo + b) * Nr + po + a] += E[a * 3 + b];
                        }
            }
        }
    }
    }
    if (T > 1) {
        for (int t = 0; t < T; t++) for (int i = 0; i < Nr; i++) c->gred[i] += priv[(size_t)t * nH + (size_t)Nr * Nr + i];
        /* Hred rows: reduce with the private layout [T][Nr*Nr + Nr] */
#pragma omp parallel for schedule(static) num_threads(T)
        for (long long i = 0; i < (long long)Nr * Nr; i++) {
            double a = c->Hred[i];
            for (int t = 0; t < T; t++) a += priv[(size_t)t * nH + i];
            c->Hred[i] = a;
        }
        free(priv);
    }
    }
    for (int k = 0; k < P->n_prior; k++) {
        const sadvio_pose_prior *pr = P->priors + k;
        int po = c->kf_off[pr->kf];
        if (po < 0) continue;
        small_factor f;
        f.rows = 6; f.ncols = 6;
        for (int q = 0; q < 6; q++) f.col[q] = po + q;
        factor_pose_prior(w->kf_T_f_w + 12 * pr->kf, pr->T_prior, pr->inf_diag, x->xp + 6 * pr->kf, f.r, f.J);
        for (int q = 0; q < 6; q++) cost += f.r[q] * f.r[q];
        push_sf(c, &f);
    }
    for (int k = 0; k < P->n_sparse; k++) {
        small_factor f;
        if (!sparse_small(c, x, P->sparse + k, &f, 1)) continue;
        for (int q = 0; q < f.rows; q++) cost += f.r[q] * f.r[q];
        push_sf(c, &f);
    }
    if (P->lines)
        for (int l = 0; l < P->lines->n_line; l++)
            for (int o = P->lines->line_obs_ptr[l]; o < P->lines->line_obs_ptr[l + 1]; o++) {
                small_factor f;
                double rho;
                if (!line_small(c, x, l, o, &f, 1, &rho)) continue;
                cost += rho;
                push_sf(c, &f);
            }
    for (int k = 0; k < P->n_imu; k++) {
        const sadvio_imu_factor *fi = P->imus + k;
        int i = fi->kf_i, j = fi->kf_j;
        int oi = c->kf_off[i], oj = c->kf_off[j];
        if (oi < 0 && oj < 0) continue;
        imu_consts ic = {fi->dt, fi->delta_R, fi->delta_v, fi->delta_p, fi->J_dR_bg, fi->J_dv_ba, fi->J_dv_bg,
                         fi->J_dp_ba, fi->J_dp_bg, c->W_imu + 81 * k};
        double Jpi[54], Jpj[54], Jvi[27], Jvj[27], Jba[27], Jbg[27];
        small_factor f;
        f.rows = 9; f.ncols = 24;
        factor_imu(&ic, w->kf_T_f_w + 12 * i, w->kf_T_f_w + 12 * j, vec3_or_zero(w->kf_vel, i),
                   vec3_or_zero(w->kf_vel, j), x->xp + 6 * i, x->xp + 6 * j, x->xv + 3 * i, x->xv + 3 * j,
                   x->xba + 3 * i, x->xbg + 3 * i, f.r, Jpi, Jpj, Jvi, Jvj, Jba, Jbg);
        for (int q = 0; q < 9; q++) {
            for (int a = 0; a < 6; a++) { f.J[q * 24 + a] = Jpi[q * 6 + a]; f.J[q * 24 + 6 + a] = Jpj[q * 6 + a]; }
            for (int a = 0; a < 3; a++) {
                f.J[q * 24 + 12 + a] = Jvi[q * 3 + a];
                f.J[q * 24 + 15 + a] = Jvj[q * 3 + a];
                f.J[q * 24 + 18 + a] = Jba[q * 3 + a];
                f.J[q * 24 + 21 + a] = Jbg[q * 3 + a];
            }
            cost += f.r[q] * f.r[q];
        }
        for (int a = 0; a < 6; a++) { f.col[a] = oi < 0 ? -1 : oi + a; f.col[6 + a] = oj < 0 ? -1 : oj + a; }
        for (int a = 0; a < 3; a++) {
            f.col[12 + a] = oi < 0 ? -1 : oi + 6 + a;
            f.col[15 + a] = oj < 0 ? -1 : oj + 6 + a;
            f.col[18 + a] = oi < 0 ? -1 : oi + 9 + a;
            f.col[21 + a] = oi < 0 ? -1 : oi + 12 + a;
        }
        push_sf(c, &f);
        /* bias random walk, blocks [dba_i, dbg_i, dba_j, dbg_j] */
        small_factor b;
        b.rows = 6; b.ncols = 12;
        double sa, sg;
        factor_imu_bias(fi->dt, fi->bacc_noise, fi->bgyr_noise, vec3_or_zero(w->kf_ba, i), vec3_or_zero(w->kf_bg, i),
                        vec3_or_zero(w->kf_ba, j), vec3_or_zero(w->kf_bg, j), x->xba + 3 * i, x->xbg + 3 * i,
                        x->xba + 3 * j, x->xbg + 3 * j, b.r, &sa, &sg);
        memset(b.J, 0, sizeof(double) * 72);
        for (int a = 0; a < 3; a++) {
            b.J[a * 12 + a] = -sa;           /* d/d dba_i */
            b.J[(3 + a) * 12 + 3 + a] = -sg; /* d/d dbg_i */
            b.J[a * 12 + 6 + a] = sa;        /* d/d dba_j */
            b.J[(3 + a) * 12 + 9 + a] = sg;  /* d/d dbg_j */
            b.col[a] = oi < 0 ? -1 : oi + 9 + a;
            b.col[3 + a] = oi < 0 ? -1 : oi + 12 + a;
            b.col[6 + a] = oj < 0 ? -1 : oj + 9 + a;
            b.col[9 + a] = oj < 0 ? -1 : oj + 12 + a;
        }
        for (int q = 0; q < 6; q++) cost += b.r[q] * b.r[q];
        push_sf(c, &b);
    }
    for (int k = 0; k < c->n_sf; k++) accum_small(c, c->sf + k);
    if (P->dp_n_full > 0) {
        state_t xs = *x;
        /* residual of the prior at x */
        int n = P->dp_n, nf = P->dp_n_full;
        double *dx = (double *)xcalloc((size_t)n, sizeof(double));
        if (P->dp_kf_keep >= 0) {
            int k = P->dp_kf_keep;
            for (int q = 0; q < 6; q++) dx[P->dp_kf_col + q] = xs.xp[6 * k + q];
            for (int q = 0; q < 3; q++) {
                dx[P->dp_kf_col + 6 + q] = xs.xv[3 * k + q];
                dx[P->dp_kf_col + 9 + q] = xs.xba[3 * k + q];
                dx[P->dp_kf_col + 12 + q] = xs.xbg[3 * k + q];
            }
        }
        for (int q = 0; q < P->dp_n_keep; q++) {
            if (P->dp_lmk_col[q] < 0) continue;
            for (int a = 0; a < 3; a++) dx[P->dp_lmk_col[q] + a] = xs.xl[3 * P->dp_lmk_index[q] + a];
        }
        for (int i = 0; i < nf; i++) {
            double s = P->dp_r0[i];
            for (int j = 0; j < n; j++) s += P->dp_J[(size_t)i * n + j] * dx[j];
            c->dp_res[i] = s;
            cost += s * s;
        }
        free(dx);
        for (int a = 0; a < n; a++) {
            int ca = c->dp_colmap[a];
            if (ca < 0) continue;
            double g = 0;
            for (int i = 0; i < nf; i++) g += P->dp_J[(size_t)i * n + a] * c->dp_res[i];
            c->gred[ca] += g;
            for (int b = 0; b < n; b++) {
                int cb = c->dp_colmap[b];
                if (cb < 0) continue;
                c->Hred[(size_t)ca * Nr + cb] += c->dp_JtJ[(size_t)a * n + b];
            }
        }
    }
    return 0.5 * cost;
}

/* dense Cholesky solve A x = b, A (n x n, row-major, symmetric) is overwritten. Returns 0 if PD. */
static int chol_solve(double *A, double *b, int n) {
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

static double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

/* LevenbergMarquardtStrategy::ComputeStep on the current linearisation: returns 0 on success and
 * fills dred (Nr) / dlmk (3 n_lmk) with delta (already negated and un-scaled). */
static int compute_step(ctx_t *c, const sadvio_solve_options *o, double radius, double *dred, double *dlmk) {
    const sadvio_flat_window *w = c->w;
    int Nr = c->Nr;
    double *S = (double *)xcalloc((size_t)Nr * Nr, sizeof(double));
    double *rhs = (double *)xcalloc((size_t)Nr, sizeof(double));
    double *Minv = (double *)xcalloc((size_t)w->n_lmk * 9, sizeof(double));
    memcpy(S, c->Hred, sizeof(double) * (size_t)Nr * Nr);
    memcpy(rhs, c->gred, sizeof(double) * (size_t)Nr);
    for (int i = 0; i < Nr; i++) {
        double s2 = c->s_red[i] * c->s_red[i];
        double d = clampd(s2 * c->Hred[(size_t)i * Nr + i], o->min_lm_diagonal, o->max_lm_diagonal);
        S[(size_t)i * Nr + i] += d / radius / s2;
    }
    int fail = 0;
    {
    const size_t nS = (size_t)Nr * Nr + (size_t)Nr;
    const int T = par_threads(c->P, nS);
    double *priv = T > 1 ? (double *)xcalloc((size_t)T * nS, sizeof(double)) : NULL;
#pragma omp parallel num_threads(T) if (T > 1)
    {
    double *St = T > 1 ? priv + (size_t)par_tid() * nS : S;
    double *rhst = T > 1 ? St + (size_t)Nr * Nr : rhs;
    /* `fail` is a reduction: each thread skips its remaining landmarks once ITS copy is set (the serial semantics at one
     * thread), the copies are OR-ed at the end of the loop — no unsynchronised write / read of a shared flag */
#pragma omp for schedule(static) reduction(| : fail)
    for (int l = 0; l < w->n_lmk; l++) {
        if (!c->lmk_elim[l] || fail) continue;
        double M[9];
        memcpy(M, c->Hll + 9 * l, sizeof(M));
        for (int a = 0; a < 3; a++) {
            double s2 = c->s_lmk[3 * l + a] * c->s_lmk[3 * l + a];
            double d = clampd(s2 * M[4 * a], o->min_lm_diagonal, o->max_lm_diagonal);
            M[4 * a] += d / radius / s2;
        }
        /* Landmark elimination in CHOLESKY form (round 5): M = L L^T, Li = L^-1, W_a = E_a Li^T, S -= W_a W_b^T — the block step of a
         * landmark-first Cholesky of the un-reduced system, which is what ceres::SPARSE_NORMAL_CHOLESKY / CHOLMOD computes under its
         * fill-reducing ordering (AOptimizer.cpp:315-323). The adjugate inverse + explicitly formed (E M^-1) E^T used before lands
         * 4e-6 .. 7e-5 from a long-double solve on the two ill-conditioned sweep windows where this form lands 4e-7 .. 4e-8
         * (scripts/elim_numerics.py). Minv[9 l ..] holds Li (lower, row-major). */
        double *Li = Minv + 9 * l;
        {
            double l00 = sqrt(M[0]), l10 = M[3] / l00, l20 = M[6] / l00;
            double l11 = sqrt(M[4] - l10 * l10);
            double l21 = (M[7] - l20 * l10) / l11;
            double l22 = sqrt(M[8] - l20 * l20 - l21 * l21);
            if (!(M[0] > 0) || !(l11 > 0) || !(l22 > 0) || !isfinite(l00) || !isfinite(l11) || !isfinite(l22)) { fail = 1; continue; }
            double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
            double i10 = -l10 * i00 * i11;
            double i21 = -l21 * i11 * i22;
            double i20 = -(l20 * i00 + l21 * i10) * i22;
            Li[0] = i00; Li[1] = 0; Li[2] = 0; Li[3] = i10; Li[4] = i11; Li[5] = 0; Li[6] = i20; Li[7] = i21; Li[8] = i22;
        }
        const double *g = c->gl + 3 * l;
        double Lg[3];   /* Li g_l */
        m3_vec(Li, g, Lg);
        int o0 = w->lmk_obs_ptr[l], o1 = w->lmk_obs_ptr[l + 1];
        for (int a = o0; a < o1; a++) {
            int pa = c->kf_off[w->obs_kf[a]];
            if (pa < 0) continue;
            const double *Ea = c->E + 18 * a;
            double Y[18]; /* 6x3 = Ea * Li^T */
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 3; j++)
                    Y[i * 3 + j] = Ea[i * 3] * Li[3 * j] + Ea[i * 3 + 1] * Li[3 * j + 1] + Ea[i * 3 + 2] * Li[3 * j + 2];
            for (int i = 0; i < 6; i++) rhst[pa + i] -= Y[i * 3] * Lg[0] + Y[i * 3 + 1] * Lg[1] + Y[i * 3 + 2] * Lg[2];
            for (int b = o0; b < o1; b++) {
                int pb = c->kf_off[w->obs_kf[b]];
                if (pb < 0) continue;
                const double *Eb = c->E + 18 * b;
                double Yb[18];
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 3; j++)
                        Yb[i * 3 + j] = Eb[i * 3] * Li[3 * j] + Eb[i * 3 + 1] * Li[3 * j + 1] + Eb[i * 3 + 2] * Li[3 * j + 2];
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++)
                        St[(size_t)(pa + i) * Nr + pb + j] -=
                            Y[i * 3] * Yb[j * 3] + Y[i * 3 + 1] * Yb[j * 3 + 1] + Y[i * 3 + 2] * Yb[j * 3 + 2];
            }
        }
    }
    }
    if (T > 1) {
        for (int t = 0; t < T; t++) for (int i = 0; i < Nr; i++) rhs[i] += priv[(size_t)t * nS + (size_t)Nr * Nr + i];
#pragma omp parallel for schedule(static) num_threads(T)
        for (long long i = 0; i < (long long)Nr * Nr; i++) {
            double a = S[i];
            for (int t = 0; t < T; t++) a += priv[(size_t)t * nS + i];
            S[i] = a;
        }
        free(priv);
    }
    }
    if (!fail && Nr > 0) fail = chol_solve(S, rhs, Nr);
    if (!fail) {
        for (int i = 0; i < Nr; i++) {
            dred[i] = -rhs[i];
            if (!isfinite(dred[i])) fail = 1;
        }
#pragma omp parallel for schedule(static) reduction(| : fail) if (c->P->n_threads > 1) num_threads(c->P->n_threads > 1 ? c->P->n_threads : 1)
        for (int l = 0; l < w->n_lmk; l++) {
            dlmk[3 * l] = dlmk[3 * l + 1] = dlmk[3 * l + 2] = 0;
            if (!c->lmk_elim[l]) continue;
            /* y_l = M^-1 (g_l - E^T y_p); delta_l = -y_l, with y_p = -dred */
            double t[3] = {c->gl[3 * l], c->gl[3 * l + 1], c->gl[3 * l + 2]};
            for (int a = w->lmk_obs_ptr[l]; a < w->lmk_obs_ptr[l + 1]; a++) {
                int pa = c->kf_off[w->obs_kf[a]];
                if (pa < 0) continue;
                const double *Ea = c->E + 18 * a;
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 3; j++) t[j] += Ea[i * 3 + j] * dred[pa + i];
            }
            double u[3], y[3];
            const double *Li = Minv + 9 * l;   /* y = M^-1 t = Li^T (Li t) */
            m3_vec(Li, t, u);
            y[0] = Li[0] * u[0] + Li[3] * u[1] + Li[6] * u[2];
            y[1] = Li[4] * u[1] + Li[7] * u[2];
            y[2] = Li[8] * u[2];
            dlmk[3 * l] = -y[0]; dlmk[3 * l + 1] = -y[1]; dlmk[3 * l + 2] = -y[2];
            if (!isfinite(y[0]) || !isfinite(y[1]) || !isfinite(y[2])) fail = 1;
        }
    }
    free(S); free(rhs); free(Minv);
    return fail;
}

/* model_cost_change = -(J delta)^T (r + J delta / 2) summed over every residual block
 * (TrustRegionMinimizer::ComputeTrustRegionStep). */
static double model_cost_change(ctx_t *c, const double *dred, const double *dlmk) {
    const sadvio_flat_window *w = c->w;
    const oracle_problem *P = c->P;
    double acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc) if (P->n_threads > 1) num_threads(P->n_threads > 1 ? P->n_threads : 1)
    for (int l = 0; l < w->n_lmk; l++) {
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            int po = c->kf_off[w->obs_kf[o]];
            const double *Jp = c->Jp + 12 * o, *Jl = c->Jl + 6 * o, *r = c->r + 2 * o;
            const double *dl = c->lmk_red[l] >= 0 ? dred + c->lmk_red[l] : dlmk + 3 * l;
            for (int q = 0; q < 2; q++) {
                double m = 0;
                if (po >= 0) for (int a = 0; a < 6; a++) m += Jp[q * 6 + a] * dred[po + a];
                if (c->lmk_active[l]) for (int a = 0; a < 3; a++) m += Jl[q * 3 + a] * dl[a];
                acc += -m * (r[q] + m / 2.0);
            }
        }
    }
    for (int k = 0; k < c->n_sf; k++) {
        const small_factor *f = c->sf + k;
        for (int q = 0; q < f->rows; q++) {
            double m = 0;
            for (int a = 0; a < f->ncols; a++)
                if (f->col[a] >= 0) m += f->J[q * f->ncols + a] * dred[f->col[a]];
            acc += -m * (f->r[q] + m / 2.0);
        }
    }
    if (P->dp_n_full > 0) {
        int n = P->dp_n, nf = P->dp_n_full;
        for (int i = 0; i < nf; i++) {
            double m = 0;
            for (int a = 0; a < n; a++)
                if (c->dp_colmap[a] >= 0) m += P->dp_J[(size_t)i * n + a] * dred[c->dp_colmap[a]];
            acc += -m * (c->dp_res[i] + m / 2.0);
        }
    }
    return acc;
}

static void ctx_init(ctx_t *c, const oracle_problem *P) {
    memset(c, 0, sizeof(*c));
    const sadvio_flat_window *w = P->win;
    c->P = P; c->w = w;
    c->dpf = w->has_imu ? 15 : 6;
    c->kf_off = (int *)xcalloc((size_t)w->n_kf, sizeof(int));
    c->lmk_red = (int *)xcalloc((size_t)w->n_lmk, sizeof(int));
    c->lmk_elim = (int *)xcalloc((size_t)w->n_lmk, sizeof(int));
    c->lmk_active = (int *)xcalloc((size_t)w->n_lmk, sizeof(int));
    int off = 0;
    for (int i = 0; i < w->n_kf; i++) {
        if (w->kf_const && w->kf_const[i]) c->kf_off[i] = -1;
        else { c->kf_off[i] = off; off += c->dpf; }
    }
    for (int l = 0; l < w->n_lmk; l++) {
        c->lmk_red[l] = -1;
        int is_const = w->lmk_const && w->lmk_const[l];
        int n_obs = w->lmk_obs_ptr[l + 1] - w->lmk_obs_ptr[l];
        c->lmk_active[l] = !is_const && n_obs > 0; /* Ceres drops blocks with no residuals */
        c->lmk_elim[l] = c->lmk_active[l];
    }
    for (int q = 0; q < P->dp_n_keep && P->dp_n_full > 0; q++) {
        int l = P->dp_lmk_index[q];
        if (P->dp_lmk_col[q] < 0) continue;
        if (w->lmk_const && w->lmk_const[l]) continue;
        c->lmk_red[l] = off; off += 3;
        c->lmk_active[l] = 1;
        c->lmk_elim[l] = 0;
    }
    for (int k = 0; k < P->n_sparse; k++) {
        const sadvio_sparse_prior *s = P->sparse + k;
        int ls[2] = {(s->type == SADVIO_SPARSE_IMU_PRIOR || s->type == SADVIO_SPARSE_RELATIVE_POSE) ? -1 : s->lmk0, s->type == SADVIO_SPARSE_LMK_TO_LMK ? s->lmk1 : -1};
        for (int q = 0; q < 2; q++) {
            int l = ls[q];
            if (l < 0 || c->lmk_red[l] >= 0) continue;
            if (w->lmk_const && w->lmk_const[l]) continue;
            c->lmk_red[l] = off; off += 3;
            c->lmk_active[l] = 1;
            c->lmk_elim[l] = 0;
        }
    }
    if (P->lines) {
        const sadvio_line_set *L = P->lines;
        c->line_off = (int *)xcalloc((size_t)L->n_line, sizeof(int));
        for (int l = 0; l < L->n_line; l++) {
            int used = 0;     /* Ceres drops parameter blocks no residual block of the program uses */
            for (int o = L->line_obs_ptr[l]; o < L->line_obs_ptr[l + 1]; o++) used = 1;
            if ((L->line_const && L->line_const[l]) || !used) c->line_off[l] = -1;
            else { c->line_off[l] = off; off += 6; }
        }
    }
    c->Nr = off;
    int Nr = off;
    c->r = (double *)xcalloc((size_t)w->n_obs * 2, sizeof(double));
    c->Jp = (double *)xcalloc((size_t)w->n_obs * 12, sizeof(double));
    c->Jl = (double *)xcalloc((size_t)w->n_obs * 6, sizeof(double));
    c->E = (double *)xcalloc((size_t)w->n_obs * 18, sizeof(double));
    c->Hll = (double *)xcalloc((size_t)w->n_lmk * 9, sizeof(double));
    c->gl = (double *)xcalloc((size_t)w->n_lmk * 3, sizeof(double));
    c->Hred = (double *)xcalloc((size_t)Nr * Nr, sizeof(double));
    c->gred = (double *)xcalloc((size_t)Nr, sizeof(double));
    c->s_red = (double *)xcalloc((size_t)Nr, sizeof(double));
    c->s_lmk = (double *)xcalloc((size_t)w->n_lmk * 3, sizeof(double));
    for (int i = 0; i < Nr; i++) c->s_red[i] = 1.0;
    for (int i = 0; i < 3 * w->n_lmk; i++) c->s_lmk[i] = 1.0;
    c->W_imu = (double *)xcalloc((size_t)P->n_imu * 81, sizeof(double));
    for (int k = 0; k < P->n_imu; k++) imu_sqrt_information(P->imus[k].cov, c->W_imu + 81 * k);
    if (P->dp_n_full > 0) {
        int n = P->dp_n, nf = P->dp_n_full;
        c->dp_colmap = (int *)xcalloc((size_t)n, sizeof(int));
        for (int a = 0; a < n; a++) c->dp_colmap[a] = -1;
        if (P->dp_kf_keep >= 0 && c->kf_off[P->dp_kf_keep] >= 0)
            for (int q = 0; q < 15 && q < c->dpf; q++) c->dp_colmap[P->dp_kf_col + q] = c->kf_off[P->dp_kf_keep] + q;
        for (int q = 0; q < P->dp_n_keep; q++) {
            int l = P->dp_lmk_index[q];
            if (P->dp_lmk_col[q] < 0 || c->lmk_red[l] < 0) continue;
            for (int a = 0; a < 3; a++) c->dp_colmap[P->dp_lmk_col[q] + a] = c->lmk_red[l] + a;
        }
        c->dp_res = (double *)xcalloc((size_t)nf, sizeof(double));
        c->dp_JtJ = (double *)xcalloc((size_t)n * n, sizeof(double));
        for (int i = 0; i < nf; i++)
            for (int a = 0; a < n; a++) {
                double ja = P->dp_J[(size_t)i * n + a];
                if (ja == 0.0) continue;
                for (int b = 0; b < n; b++) c->dp_JtJ[(size_t)a * n + b] += ja * P->dp_J[(size_t)i * n + b];
            }
    }
}

static void ctx_free(ctx_t *c) {
    free(c->kf_off); free(c->lmk_red); free(c->lmk_elim); free(c->lmk_active);
    free(c->r); free(c->Jp); free(c->Jl); free(c->E); free(c->Hll); free(c->gl);
    free(c->Hred); free(c->gred); free(c->s_red); free(c->s_lmk); free(c->W_imu);
    free(c->line_off); free(c->sf); free(c->dp_colmap); free(c->dp_res); free(c->dp_JtJ);
}

/* cost of residual blocks whose parameter blocks are all constant (Ceres: Summary::fixed_cost) */
static double fixed_cost(ctx_t *c) {
    const sadvio_flat_window *w = c->w;
    double cost = 0;
    for (int l = 0; l < w->n_lmk; l++)
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            if (c->kf_off[w->obs_kf[o]] >= 0 || c->lmk_active[l]) continue;
            double r[2];
            eval_obs(w, l, o, NULL, NULL, r, NULL, NULL, NULL);
            cost += r[0] * r[0] + r[1] * r[1];
        }
    if (c->P->lines)
        for (int l = 0; l < c->P->lines->n_line; l++)
            for (int o = c->P->lines->line_obs_ptr[l]; o < c->P->lines->line_obs_ptr[l + 1]; o++) {
                small_factor sf;
                double rho;
                if (line_small(c, NULL, l, o, &sf, 0, &rho)) continue;
                cost += rho;
            }
    static const double z6[6] = {0};
    for (int k = 0; k < c->P->n_prior; k++) {
        const sadvio_pose_prior *pr = c->P->priors + k;
        if (c->kf_off[pr->kf] >= 0) continue;
        double r[6];
        factor_pose_prior(w->kf_T_f_w + 12 * pr->kf, pr->T_prior, pr->inf_diag, z6, r, NULL);
        for (int i = 0; i < 6; i++) cost += r[i] * r[i];
    }
    return 0.5 * cost;
}

/* gather / scatter between the reduced vector and the per-type delta arrays */
static void apply_delta(const ctx_t *c, const double *dred, const double *dlmk, const double *xp, const double *xl,
                        const double *xv, const double *xba, const double *xbg, double *cp, double *cl, double *cv,
                        double *cba, double *cbg) {
    const sadvio_flat_window *w = c->w;
    for (int i = 0; i < w->n_kf; i++) {
        int o = c->kf_off[i];
        for (int q = 0; q < 6; q++) cp[6 * i + q] = xp[6 * i + q] + (o >= 0 ? dred[o + q] : 0.0);
        for (int q = 0; q < 3; q++) {
            int vio = (o >= 0 && c->dpf == 15);
            cv[3 * i + q] = xv[3 * i + q] + (vio ? dred[o + 6 + q] : 0.0);
            cba[3 * i + q] = xba[3 * i + q] + (vio ? dred[o + 9 + q] : 0.0);
            cbg[3 * i + q] = xbg[3 * i + q] + (vio ? dred[o + 12 + q] : 0.0);
        }
    }
    for (int l = 0; l < w->n_lmk; l++) {
        const double *d = c->lmk_red[l] >= 0 ? dred + c->lmk_red[l] : dlmk + 3 * l;
        for (int q = 0; q < 3; q++) cl[3 * l + q] = xl[3 * l + q] + (c->lmk_active[l] ? d[q] : 0.0);
    }
}

static void jacobi_scaling(ctx_t *c) {
    int Nr = c->Nr;
    for (int i = 0; i < Nr; i++) c->s_red[i] = 1.0 / (1.0 + sqrt(c->Hred[(size_t)i * Nr + i]));
    for (int l = 0; l < c->w->n_lmk; l++)
        for (int a = 0; a < 3; a++)
            c->s_lmk[3 * l + a] = c->lmk_elim[l] ? 1.0 / (1.0 + sqrt(c->Hll[9 * l + 4 * a])) : 1.0;
}

static double gradient_max_norm(const ctx_t *c) {
    double m = 0;
    for (int i = 0; i < c->Nr; i++) m = fmax(m, fabs(c->gred[i]));
    for (int l = 0; l < c->w->n_lmk; l++)
        if (c->lmk_elim[l])
            for (int a = 0; a < 3; a++) m = fmax(m, fabs(c->gl[3 * l + a]));
    return m;
}

static double vec_norm2(const double *a, size_t n) {
    double s = 0;
    for (size_t i = 0; i < n; i++) s += a[i] * a[i];
    return s;
}

void sadvio_oracle_default_options(sadvio_solve_options *o) {
    memset(o, 0, sizeof(*o));
    o->max_num_iterations = 20;   /* AOptimizer.cpp:319 */
    o->function_tolerance = 1e-3; /* AOptimizer.cpp:322 */
    o->jacobi_scaling = 1;
    o->max_num_consecutive_invalid_steps = 5;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->min_relative_decrease = 1e-3;
}

int oracle_solve(const oracle_problem *P, const sadvio_solve_options *o, sadvio_solve_summary *sum, double *pose_delta6,
                 double *lmk_delta3, double *dv3, double *dba3, double *dbg3, double *iter_log, int32_t iter_log_cap) {
    const sadvio_flat_window *w = P->win;
    ctx_t c;
    ctx_init(&c, P);
    c.huber_a = o ? o->huber_a : 0.0;
    size_t np = (size_t)w->n_kf * 6, nl = (size_t)w->n_lmk * 3, nv = (size_t)w->n_kf * 3;
    double *xp = (double *)xcalloc(np, 8), *xl = (double *)xcalloc(nl, 8), *xv = (double *)xcalloc(nv, 8),
           *xba = (double *)xcalloc(nv, 8), *xbg = (double *)xcalloc(nv, 8);
    double *cp = (double *)xcalloc(np, 8), *cl = (double *)xcalloc(nl, 8), *cv = (double *)xcalloc(nv, 8),
           *cba = (double *)xcalloc(nv, 8), *cbg = (double *)xcalloc(nv, 8);
    double *dred = (double *)xcalloc((size_t)c.Nr, 8), *dlmk = (double *)xcalloc(nl, 8);
    const size_t nln = P->lines ? (size_t)P->lines->n_line * 6 : 0;
    double *xline = (double *)xcalloc(nln, 8), *cline = (double *)xcalloc(nln, 8);
    state_t X = {xp, xl, xv, xba, xbg, xline};
    state_t C = {cp, cl, cv, cba, cbg, cline};

    sadvio_solve_summary S;
    memset(&S, 0, sizeof(S));
    S.fixed_cost = fixed_cost(&c);

    /* IterationZero */
    double x_cost = eval_full(&c, &X);
    if (o->jacobi_scaling) jacobi_scaling(&c);
    S.initial_cost = x_cost;
    double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
    double x_norm = 0.0;
    int iter = 0, n_invalid = 0;
    int term = SADVIO_TERM_NO_CONVERGENCE;
    double gmax = gradient_max_norm(&c);
    if (iter_log && iter_log_cap > 0) {
        double *L = iter_log; L[0] = x_cost; L[1] = 0; L[2] = radius; L[3] = 0; L[4] = 0; L[5] = 1; L[6] = gmax; L[7] = 0;
    }
    int done = 0;
    /* FinalizeIterationAndCheckIfMinimizerCanContinue after iteration 0: solver time (options.max_solver_time_in_seconds,
     * 0 = none), iterations, gradient, radius — in Ceres' order */
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
#define ORACLE_TIME_UP() (o->max_solver_time_in_seconds > 0.0 && (clock_gettime(CLOCK_MONOTONIC, &ts1), \
                          (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec) >= o->max_solver_time_in_seconds))
    if (ORACLE_TIME_UP()) { done = 1; term = SADVIO_TERM_NO_CONVERGENCE; }
    else if (iter >= o->max_num_iterations) { done = 1; term = SADVIO_TERM_NO_CONVERGENCE; }
    else if (gmax <= o->gradient_tolerance) { done = 1; term = SADVIO_TERM_GRADIENT_TOL; }
    else if (radius <= o->min_trust_region_radius) { done = 1; term = SADVIO_TERM_MIN_RADIUS; }

    while (!done) {
        iter++;
        double cost_change = 0, step_norm = 0, rel_dec = 0, mcc = 0;
        int successful = 0;
        int fail = compute_step(&c, o, radius, dred, dlmk);
        int step_valid = 0;
        if (!fail) {
            mcc = model_cost_change(&c, dred, dlmk);
            step_valid = mcc > 0.0;
        }
        if (!step_valid) {
            /* HandleInvalidStep */
            n_invalid++;
            if (n_invalid >= o->max_num_consecutive_invalid_steps) { term = SADVIO_TERM_FAILURE; S.num_unsuccessful_steps++; break; }
            radius *= 0.5; /* LevenbergMarquardtStrategy::StepIsInvalid */
            S.num_unsuccessful_steps++;
        } else {
            n_invalid = 0;
            apply_delta(&c, dred, dlmk, xp, xl, xv, xba, xbg, cp, cl, cv, cba, cbg);
            for (size_t i = 0; i < nln; i++) { const int lo = c.line_off[i / 6]; cline[i] = xline[i] + (lo >= 0 ? dred[lo + (int)(i % 6)] : 0.0); }
            double cand_cost = eval_cost(&c, &C);
            /* ParameterToleranceReached: norm over the reduced program's parameters */
            double sn2 = 0;
            for (int i = 0; i < c.Nr; i++) sn2 += dred[i] * dred[i];
            for (int l = 0; l < w->n_lmk; l++)
                if (c.lmk_elim[l]) sn2 += vec_norm2(dlmk + 3 * l, 3);
            step_norm = sqrt(sn2);
            double step_size_tol = o->parameter_tolerance * (x_norm + o->parameter_tolerance);
            if (step_norm <= step_size_tol) {
                term = SADVIO_TERM_PARAMETER_TOL;
                if (iter_log && iter < iter_log_cap) { double *L = iter_log + 8 * iter; L[0] = x_cost; L[1] = x_cost - cand_cost; L[2] = radius; L[3] = step_norm; L[4] = 0; L[5] = 0; L[6] = gmax; L[7] = mcc; }
                break;
            }
            /* FunctionToleranceReached */
            cost_change = x_cost - cand_cost;
            if (fabs(cost_change) <= o->function_tolerance * x_cost) {
                term = SADVIO_TERM_FUNCTION_TOL;
                if (iter_log && iter < iter_log_cap) { double *L = iter_log + 8 * iter; L[0] = x_cost; L[1] = cost_change; L[2] = radius; L[3] = step_norm; L[4] = 0; L[5] = 0; L[6] = gmax; L[7] = mcc; }
                break;
            }
            /* IsStepSuccessful (monotonic step evaluator) */
            rel_dec = (cand_cost >= DBL_MAX) ? -DBL_MAX : cost_change / mcc;
            if (rel_dec > o->min_relative_decrease) {
                /* HandleSuccessfulStep */
                memcpy(xp, cp, np * 8); memcpy(xl, cl, nl * 8); memcpy(xv, cv, nv * 8);
                memcpy(xba, cba, nv * 8); memcpy(xbg, cbg, nv * 8);
                memcpy(xline, cline, nln * 8);
                double n2 = 0;
                for (size_t i = 0; i < nln; i++) if (c.line_off[i / 6] >= 0) n2 += xline[i] * xline[i];
                for (int i = 0; i < w->n_kf; i++) {
                    if (c.kf_off[i] < 0) continue;
                    n2 += vec_norm2(xp + 6 * i, 6);
                    if (c.dpf == 15) n2 += vec_norm2(xv + 3 * i, 3) + vec_norm2(xba + 3 * i, 3) + vec_norm2(xbg + 3 * i, 3);
                }
                for (int l = 0; l < w->n_lmk; l++)
                    if (c.lmk_active[l]) n2 += vec_norm2(xl + 3 * l, 3);
                x_norm = sqrt(n2);
                x_cost = eval_full(&c, &X);
                gmax = gradient_max_norm(&c);
                successful = 1;
                S.num_successful_steps++;
                /* LevenbergMarquardtStrategy::StepAccepted */
                double t = 2.0 * rel_dec - 1.0;
                radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
                radius = fmin(o->max_trust_region_radius, radius);
                decrease_factor = 2.0;
            } else {
                /* StepRejected */
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
                S.num_unsuccessful_steps++;
            }
        }
        if (iter_log && iter < iter_log_cap) {
            double *L = iter_log + 8 * iter;
            L[0] = x_cost; L[1] = cost_change; L[2] = radius; L[3] = step_norm; L[4] = rel_dec; L[5] = successful; L[6] = gmax; L[7] = mcc;
        }
        if (ORACLE_TIME_UP()) { term = SADVIO_TERM_NO_CONVERGENCE; break; }
        if (iter >= o->max_num_iterations) { term = SADVIO_TERM_NO_CONVERGENCE; break; }
        if (gmax <= o->gradient_tolerance) { term = SADVIO_TERM_GRADIENT_TOL; break; }
        if (radius <= o->min_trust_region_radius) { term = SADVIO_TERM_MIN_RADIUS; break; }
    }
    S.iterations = iter;
    S.termination = term;
    S.final_cost = x_cost;
    S.final_radius = radius;
    if (sum) *sum = S;
    if (pose_delta6) memcpy(pose_delta6, xp, np * 8);
    if (P->line_delta6 && nln) memcpy(P->line_delta6, xline, nln * 8);
    if (lmk_delta3) memcpy(lmk_delta3, xl, nl * 8);
    if (dv3) memcpy(dv3, xv, nv * 8);
    if (dba3) memcpy(dba3, xba, nv * 8);
    if (dbg3) memcpy(dbg3, xbg, nv * 8);
    free(xp); free(xl); free(xv); free(xba); free(xbg);
    free(cp); free(cl); free(cv); free(cba); free(cbg);
    free(dred); free(dlmk); free(xline); free(cline);
    ctx_free(&c);
    return term == SADVIO_TERM_FAILURE ? SADVIO_E_NOT_USABLE : SADVIO_OK;
}

int oracle_first_step(const oracle_problem *P, const sadvio_solve_options *o, double *delta_pose6, double *delta_lmk3,
                      double *H_full, double *g_full, int32_t n_full_dim) {
    /* Un-reduced normal equations at x = 0, ordering [free KF blocks (dpf each) | landmarks (3 each, active)].
     * Returned so a test can solve (H + D^2) y = g directly (what SPARSE_NORMAL_CHOLESKY does) and
     * compare with the Schur path. */
    const sadvio_flat_window *w = P->win;
    ctx_t c;
    ctx_init(&c, P);
    c.huber_a = o ? o->huber_a : 0.0;
    size_t np = (size_t)w->n_kf * 6, nl = (size_t)w->n_lmk * 3, nv = (size_t)w->n_kf * 3;
    double *z = (double *)xcalloc(np + nl + 3 * nv, 8);
    state_t X = {z, z + np, z + np + nl, z + np + nl + nv, z + np + nl + 2 * nv};
    eval_full(&c, &X);
    if (o->jacobi_scaling) jacobi_scaling(&c);
    int Nr = c.Nr;
    int n_act = 0;
    int *lcol = (int *)xcalloc((size_t)w->n_lmk, sizeof(int));
    for (int l = 0; l < w->n_lmk; l++) lcol[l] = c.lmk_elim[l] ? Nr + 3 * n_act++ : -1;
    int N = Nr + 3 * n_act;
    int rc = 0;
    if (H_full && g_full) {
        if (n_full_dim != N) rc = N; /* tell the caller the right size */
        else {
            memset(H_full, 0, sizeof(double) * (size_t)N * N);
            memset(g_full, 0, sizeof(double) * (size_t)N);
            for (int i = 0; i < Nr; i++) {
                g_full[i] = c.gred[i];
                for (int j = 0; j < Nr; j++) H_full[(size_t)i * N + j] = c.Hred[(size_t)i * Nr + j];
            }
            for (int l = 0; l < w->n_lmk; l++) {
                if (lcol[l] < 0) continue;
                for (int a = 0; a < 3; a++) {
                    g_full[lcol[l] + a] = c.gl[3 * l + a];
                    for (int b = 0; b < 3; b++) H_full[(size_t)(lcol[l] + a) * N + lcol[l] + b] = c.Hll[9 * l + 3 * a + b];
                }
                for (int ob = w->lmk_obs_ptr[l]; ob < w->lmk_obs_ptr[l + 1]; ob++) {
                    int po = c.kf_off[w->obs_kf[ob]];
                    if (po < 0) continue;
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 3; b++) {
                            H_full[(size_t)(po + a) * N + lcol[l] + b] += c.E[18 * ob + 3 * a + b];
                            H_full[(size_t)(lcol[l] + b) * N + po + a] += c.E[18 * ob + 3 * a + b];
                        }
                }
            }
        }
    }
    if (rc == 0 && delta_pose6 && delta_lmk3) {
        double *dred = (double *)xcalloc((size_t)Nr, 8), *dlmk = (double *)xcalloc(nl, 8);
        int fail = compute_step(&c, o, o->initial_trust_region_radius, dred, dlmk);
        if (fail) rc = -1;
        memset(delta_pose6, 0, np * 8);
        for (int i = 0; i < w->n_kf; i++)
            if (c.kf_off[i] >= 0)
                for (int q = 0; q < 6; q++) delta_pose6[6 * i + q] = dred[c.kf_off[i] + q];
        memcpy(delta_lmk3, dlmk, nl * 8);
        free(dred); free(dlmk);
    }
    free(lcol); free(z);
    ctx_free(&c);
    return rc;
}

int oracle_linearize(const sadvio_flat_window *w, const double *pose_delta6, const double *lmk_delta3, double *r2,
                     double *J_pose12, double *J_lmk6, int32_t *valid) {
    for (int l = 0; l < w->n_lmk; l++)
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++) {
            double r[2], Jp[12], Jl[6];
            int v;
            eval_obs(w, l, o, pose_delta6, lmk_delta3, r, Jp, Jl, &v);
            if (r2) memcpy(r2 + 2 * o, r, 16);
            if (J_pose12) memcpy(J_pose12 + 12 * o, Jp, 96);
            if (J_lmk6) memcpy(J_lmk6 + 6 * o, Jl, 48);
            if (valid) valid[o] = v;
        }
    return 0;
}

/* ALandmark::chi2err / avgChi2err / sanityCheck (ALandmark.cpp:98-146) with the projection tests of
 * Camera::project(T_w_lmk, model, scale, p2ds) (Camera.cpp:26-52): depth < 0.1, outside [0,cols]x[0,rows] or
 * non-finite => the feature counts 1000. image_wh NULL: (2 cx, 2 cy). Angular windows: pixel recovered through K
 * from the stored bearing (the reference reads the feature's pixel, AFeature2D.h:21). */
int oracle_landmark_chi2(const sadvio_flat_window *w, const double *pose_delta6, const double *lmk_delta3,
                         const double *image_wh, double pixel_sigma, double *avg_chi2, int32_t *inlier) {
    static const double z6[6] = {0, 0, 0, 0, 0, 0};
    for (int l = 0; l < w->n_lmk; l++) {
        const double *dl = lmk_delta3 ? lmk_delta3 + 3 * l : z6;
        double pw[3] = {w->lmk_p[3 * l] + dl[0], w->lmk_p[3 * l + 1] + dl[1], w->lmk_p[3 * l + 2] + dl[2]};
        double sum = 0.0;
        int n = 0;
        for (int o = w->lmk_obs_ptr[l]; o < w->lmk_obs_ptr[l + 1]; o++, n++) {
            int kf = w->obs_kf[o], cam = w->obs_cam[o];
            const double *K = w->cam_K + 4 * cam;
            double dT[12], Tfw[12], Tsw[12], tc[3];
            se3_from_delta6(pose_delta6 ? pose_delta6 + 6 * kf : z6, dT);
            se3_mul(w->kf_T_f_w + 12 * kf, dT, Tfw);
            se3_mul(w->cam_T_s_f + 12 * cam, Tfw, Tsw); /* getWorld2SensorTransform() */
            se3_apply(Tsw, pw, tc);
            double pt[3] = {K[0] * tc[0] + K[2] * tc[2], K[1] * tc[1] + K[3] * tc[2], tc[2]};
            double u = pt[0] / pt[2], v = pt[1] / pt[2];
            double cols = image_wh ? image_wh[2 * cam] : 2.0 * K[2], rows = image_wh ? image_wh[2 * cam + 1] : 2.0 * K[3];
            double mu, mv;
            if (w->factor_type == SADVIO_FACTOR_PIXEL) { mu = w->obs_meas[2 * o]; mv = w->obs_meas[2 * o + 1]; }
            else {
                const double *b = w->obs_meas + 3 * o;
                mu = K[0] * b[0] / b[2] + K[2]; mv = K[1] * b[1] / b[2] + K[3];
            }
            if (tc[2] < 0.1 || u < 0 || v < 0 || u > cols || v > rows || !isfinite(u) || !isfinite(v)) { sum += 1000.0; continue; }
            /* f->getSigma(): the feature's PIXEL sigma, 1.0 in the reference (AFeature2D.h:18) */
            double sg = pixel_sigma > 0 ? pixel_sigma : (w->factor_type == SADVIO_FACTOR_PIXEL && w->cam_sigma ? w->cam_sigma[cam] : 1.0);
            double e0 = (u - mu) / sg, e1 = (v - mv) / sg;
            sum += e0 * e0 + e1 * e1; /* one point per feature: the mean over getPoints() is the value itself */
        }
        double avg = n ? sum / n : 0.0;
        if (avg_chi2) avg_chi2[l] = avg;
        if (inlier) inlier[l] = (n >= 2 && !(avg > 2.0)) ? 1 : 0;
    }
    return 0;
}

/* ---- factor / geometry probes ---- */
void oracle_factor_pixel(const double *T0, const double *K, const double *Tsf, const double *p0, const double *uv,
                         double sigma, const double *dpose, const double *dl, double *r, double *Jp, double *Jl,
                         int32_t *valid) {
    int v = factor_pixel(T0, K, Tsf, p0, uv, sigma, dpose, dl, r, Jp, Jl);
    if (valid) *valid = v;
}
void oracle_factor_angular(const double *T0, const double *Tsf, const double *p0, const double *bearing, double sigma,
                           const double *dpose, const double *dl, double *r, double *Jp, double *Jl) {
    factor_angular(T0, Tsf, p0, bearing, sigma, dpose, dl, r, Jp, Jl);
}
void oracle_factor_pose_prior(const double *T0, const double *Tprior, const double *inf_diag, const double *dpose,
                              double *r, double *J) {
    factor_pose_prior(T0, Tprior, inf_diag, dpose, r, J);
}
int oracle_factor_imu(const sadvio_imu_factor *f, const double *Ti0, const double *Tj0, const double *vi0,
                      const double *vj0, const double *p, double *r9, double *J) {
    double W[81];
    int rc = imu_sqrt_information(f->cov, W);
    if (rc) return rc;
    imu_consts ic = {f->dt, f->delta_R, f->delta_v, f->delta_p, f->J_dR_bg, f->J_dv_ba, f->J_dv_bg, f->J_dp_ba, f->J_dp_bg, W};
    double Jpi[54], Jpj[54], Jvi[27], Jvj[27], Jba[27], Jbg[27];
    factor_imu(&ic, Ti0, Tj0, vi0, vj0, p, p + 6, p + 12, p + 15, p + 18, p + 21, r9, J ? Jpi : NULL, J ? Jpj : NULL,
               J ? Jvi : NULL, J ? Jvj : NULL, J ? Jba : NULL, J ? Jbg : NULL);
    if (J)
        for (int q = 0; q < 9; q++) {
            for (int a = 0; a < 6; a++) { J[q * 24 + a] = Jpi[q * 6 + a]; J[q * 24 + 6 + a] = Jpj[q * 6 + a]; }
            for (int a = 0; a < 3; a++) {
                J[q * 24 + 12 + a] = Jvi[q * 3 + a]; J[q * 24 + 15 + a] = Jvj[q * 3 + a];
                J[q * 24 + 18 + a] = Jba[q * 3 + a]; J[q * 24 + 21 + a] = Jbg[q * 3 + a];
            }
        }
    return 0;
}
void oracle_factor_imu_bias(const sadvio_imu_factor *f, const double *bai, const double *bgi, const double *baj,
                            const double *bgj, const double *p, double *r6, double *J) {
    double sa, sg;
    factor_imu_bias(f->dt, f->bacc_noise, f->bgyr_noise, bai, bgi, baj, bgj, p, p + 3, p + 6, p + 9, r6, &sa, &sg);
    if (J) {
        memset(J, 0, sizeof(double) * 72);
        for (int a = 0; a < 3; a++) {
            J[a * 12 + a] = -sa; J[(3 + a) * 12 + 3 + a] = -sg;
            J[a * 12 + 6 + a] = sa; J[(3 + a) * 12 + 9 + a] = sg;
        }
    }
}
void oracle_so3_exp(const double *w, double *R) { so3_exp(w, R); }
void oracle_so3_log(const double *R, double *w) { so3_log(R, w); }
void oracle_so3_right_jacobian(const double *w, double *J) { so3_right_jacobian(w, J); }

/* Probe of one sparse prior factor at the given deltas (arrays indexed like the window; NULL = zeros):
 * r[rows], J[rows x 15] in the factor's own column order (type 0: pose6 v3 ba3 bg3; 1: pose6 lmk3; 2: lmk3;
 * 3: lmk0 3, lmk1 3). Returns the number of residual rows. */
/* probe: one line observation at the given deltas (NULL = zeros); J rows x 12 = [key-frame | line]; returns rows */
int oracle_line_factor(const sadvio_flat_window *w, const sadvio_line_set *L, int32_t l, int32_t o, const double *xp, const double *xline,
                       double *r, double *J) {
    static const double z6[6] = {0, 0, 0, 0, 0, 0};
    const int kf = L->obs_kf[o], cam = L->obs_cam[o];
    const double *dp = xp ? xp + 6 * kf : z6, *dl = xline ? xline + 6 * l : z6;
    double Jf[24], Jl[24];
    int rows;
    if (w->factor_type == SADVIO_FACTOR_PIXEL) {
        rows = 4;
        factor_line_pixel(w->kf_T_f_w + 12 * kf, w->cam_K + 4 * cam, w->cam_T_s_f + 12 * cam, L->line_T_w_l + 12 * l, L->line_model + 6 * l,
                          L->obs_meas + 4 * o, 1.0, dp, dl, r, Jf, Jl);
    } else {
        rows = 2;
        factor_line_angular(w->kf_T_f_w + 12 * kf, w->cam_T_s_f + 12 * cam, L->line_T_w_l + 12 * l, L->obs_meas + 6 * o, 1.0, dp, dl, r, Jf, Jl);
    }
    for (int i = 0; i < rows; i++) for (int q = 0; q < 6; q++) { J[i * 12 + q] = Jf[i * 6 + q]; J[i * 12 + 6 + q] = Jl[i * 6 + q]; }
    return rows;
}

int oracle_sparse_factor(const sadvio_flat_window *w, const sadvio_sparse_prior *s, const double *xp, const double *xv,
                         const double *xba, const double *xbg, const double *xl, double *r, double *J) {
    static const double z[15] = {0};
    if (s->type == SADVIO_SPARSE_IMU_PRIOR) {
        int k = s->kf;
        double params[15], Jf[225];
        for (int q = 0; q < 6; q++) params[q] = xp ? xp[6 * k + q] : 0.0;
        for (int q = 0; q < 3; q++) {
            params[6 + q] = xv ? xv[3 * k + q] : 0.0; params[9 + q] = xba ? xba[3 * k + q] : 0.0; params[12 + q] = xbg ? xbg[3 * k + q] : 0.0;
        }
        factor_imu_prior(w->kf_T_f_w + 12 * k, vec3_or_zero(w->kf_vel, k), vec3_or_zero(w->kf_ba, k), vec3_or_zero(w->kf_bg, k),
                         s->T_prior, s->v_prior, s->ba_prior, s->bg_prior, s->sqrt_inf, params, r, J ? Jf : NULL);
        if (J) memcpy(J, Jf, sizeof(Jf));
        return 15;
    }
    if (s->type == SADVIO_SPARSE_RELATIVE_POSE) {
        double Ja[36], Jb[36];
        factor_relative_pose(w->kf_T_f_w + 12 * s->kf, w->kf_T_f_w + 12 * s->kf_b, s->T_prior, s->sqrt_inf, xp ? xp + 6 * s->kf : z,
                             xp ? xp + 6 * s->kf_b : z, r, J ? Ja : NULL, J ? Jb : NULL);
        if (J) { memset(J, 0, sizeof(double) * 90); for (int i = 0; i < 6; i++) for (int q = 0; q < 6; q++) { J[i * 15 + q] = Ja[i * 6 + q]; J[i * 15 + 6 + q] = Jb[i * 6 + q]; } }
        return 6;
    }
    const double *d0 = xl ? xl + 3 * s->lmk0 : z;
    if (J) memset(J, 0, sizeof(double) * 45);
    if (s->type == SADVIO_SPARSE_POSE_TO_LMK) {
        double Jp[18], Jl[9];
        factor_pose_to_landmark(w->kf_T_f_w + 12 * s->kf, w->lmk_p + 3 * s->lmk0, s->delta, s->sqrt_inf,
                                xp ? xp + 6 * s->kf : z, d0, r, J ? Jp : NULL, J ? Jl : NULL);
        if (J) for (int i = 0; i < 3; i++) { for (int a = 0; a < 6; a++) J[i * 15 + a] = Jp[i * 6 + a]; for (int a = 0; a < 3; a++) J[i * 15 + 6 + a] = Jl[i * 3 + a]; }
        return 3;
    }
    if (s->type == SADVIO_SPARSE_LMK_PRIOR) {
        double Jl[9];
        factor_landmark_prior(w->lmk_p + 3 * s->lmk0, s->delta, s->sqrt_inf, d0, r, J ? Jl : NULL);
        if (J) for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) J[i * 15 + a] = Jl[i * 3 + a];
        return 3;
    }
    {
        const double *d1 = xl ? xl + 3 * s->lmk1 : z;
        double J0[9], J1[9];
        factor_landmark_to_landmark(w->lmk_p + 3 * s->lmk0, w->lmk_p + 3 * s->lmk1, s->delta, s->sqrt_inf, d0, d1, r, J0, J1);
        if (J) for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) { J[i * 15 + a] = J0[i * 3 + a]; J[i * 15 + 3 + a] = J1[i * 3 + a]; }
        return 3;
    }
}
