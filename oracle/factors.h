/*
 * oracle/factors.h — TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Plain-C restatement of the reference's cost functions on the BA hot path. Each evaluator
 * follows the reference's order of operations, including its quirks.
 * Jacobians are row-major `rows x block_size`, like Ceres. A Jacobian pointer may be NULL.
 */
#ifndef SADVIO_ORACLE_FACTORS_H
#define SADVIO_ORACLE_FACTORS_H
#include "so3.h"

static const double ORACLE_G[3] = {0.0, 0.0, -9.81}; /* IMU.h:8 */

/* small dense helpers, row-major */
static inline void mat_mul(const double *A, const double *B, double *C, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}

/*
 * Pixel reprojection factor.
 * Reference: ReprojectionErrCeres_pointxd_dx::Evaluate (BundleAdjustmentCERESAnalytic.h:52-90)
 *            + Camera::project (Camera.cpp:84-139).
 * T0 = T_f_w of the observing key-frame, K = (fx,fy,cx,cy), Tsf = frame->sensor,
 * p0 = landmark position, uv = measurement, dpose[6]/dl[3] = current deltas.
 * Quirk B.1: invalid projection => r = 0 while the Jacobians stay as computed.
 * Returns 1 if the projection was valid.
 */
static inline int factor_pixel(const double *T0, const double *K, const double *Tsf, const double *p0,
                               const double *uv, double sigma, const double *dpose, const double *dl,
                               double *r, double *Jp /*2x6*/, double *Jl /*2x3*/) {
    double dT[12], Tfw[12];
    se3_from_delta6(dpose, dT);
    se3_mul(T0, dT, Tfw); /* …Analytic.h:54-55 */
    double pw[3] = {p0[0] + dl[0], p0[1] + dl[1], p0[2] + dl[2]}; /* :58 */
    double w_inv = 1.0 / sigma;                                   /* info_sqrt_ = (1/sigma) I, :48 */

    double Tsw[12], tc[3];
    se3_mul(Tsf, Tfw, Tsw);
    se3_apply(Tsw, pw, tc); /* Camera.cpp:91-92 */
    double pt[3] = {K[0] * tc[0] + K[2] * tc[2], K[1] * tc[1] + K[3] * tc[2], tc[2]}; /* :95 */
    double Jh[6] = {1 / pt[2], 0.0, -pt[0] / (pt[2] * pt[2]), 0.0, 1 / pt[2], -pt[1] / (pt[2] * pt[2])}; /* :97-99 */
    double p2d[2] = {pt[0] / pt[2], pt[1] / pt[2]};

    if (Jp || Jl) {
        /* A = sqrt_info * J_h * K * R_s_f  (2x3) */
        double Kmat[9] = {K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1};
        double JhK[6], A[6];
        mat_mul(Jh, Kmat, JhK, 2, 3, 3);
        mat_mul(JhK, Tsf, A, 2, 3, 3);
        if (Jp) {
            /* Camera.cpp:104-115: J_int = [-R_fw [p_w]x Jr(log R_fw) | I] */
            double S[9], lw[3], Jr[9], RS[9], RSJ[9], Jint[18];
            so3_skew(pw, S);
            so3_log(Tfw, lw);
            so3_right_jacobian(lw, Jr);
            m3_mul(Tfw, S, RS);
            m3_mul(RS, Jr, RSJ);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    Jint[i * 6 + j] = -RSJ[3 * i + j];
                    Jint[i * 6 + 3 + j] = (i == j) ? 1.0 : 0.0;
                }
            double Jf[12];
            mat_mul(A, Jint, Jf, 2, 3, 6);
            /* …Analytic.h:71-79: J_proj_f *= blkdiag(Jr(log R_fw)^-1 Jr(w), R_f_w0) */
            double Jrinv[9], Jrw[9], B[9];
            m3_inverse(Jr, Jrinv);
            so3_right_jacobian(dpose, Jrw);
            m3_mul(Jrinv, Jrw, B);
            for (int i = 0; i < 2; i++) {
                for (int j = 0; j < 3; j++) {
                    double a = 0, b = 0;
                    for (int l = 0; l < 3; l++) {
                        a += Jf[i * 6 + l] * B[3 * l + j];
                        b += Jf[i * 6 + 3 + l] * T0[3 * l + j];
                    }
                    Jp[i * 6 + j] = w_inv * a;
                    Jp[i * 6 + 3 + j] = w_inv * b;
                }
            }
        }
        if (Jl) {
            /* Camera.cpp:117-126: J_lmk = sqrt_info J_h K R_s_f R_f_w */
            double Jt[6];
            mat_mul(A, Tfw, Jt, 2, 3, 3);
            for (int i = 0; i < 6; i++) Jl[i] = w_inv * Jt[i];
        }
    }
    int valid = 1;
    if (tc[2] < 0.1) valid = 0;                                                          /* Camera.cpp:128 */
    if (p2d[0] < 0 || p2d[1] < 0 || p2d[0] > 2 * K[2] || p2d[1] > 2 * K[3]) valid = 0;   /* :131-133 */
    if (!isfinite(p2d[0]) || !isfinite(p2d[1])) valid = 0;                               /* :135 */
    if (valid) {
        r[0] = w_inv * (p2d[0] - uv[0]);
        r[1] = w_inv * (p2d[1] - uv[1]);
    } else {
        r[0] = 0; r[1] = 0; /* …Analytic.h:63-65 */
    }
    return valid;
}

/*
 * Angular (bearing) reprojection factor.
 * Reference: AngularErrCeres_pointxd_dx::Evaluate (AngularAdjustmentCERESAnalytic.h:55-111).
 */
static inline void factor_angular(const double *T0, const double *Tsf, const double *p0, const double *b /*unit*/,
                                  double sigma, const double *dpose, const double *dl, double *r, double *Jp,
                                  double *Jl) {
    double dT[12];
    se3_from_delta6(dpose, dT); /* :57 */
    double weight = 1 / sigma;
    double pw[3] = {p0[0] + dl[0], p0[1] + dl[1], p0[2] + dl[2]};
    /* t_s_lmk = T_s_f * T_f_w * dT * (p0 + dl), :62 */
    double T1[12], T2[12], ts[3];
    se3_mul(Tsf, T0, T1);
    se3_mul(T1, dT, T2);
    se3_apply(T2, pw, ts);
    double nrm = v3_norm(ts);
    double bs[3] = {ts[0] / nrm, ts[1] / nrm, ts[2] / nrm};

    /* tangent basis, :66-77 */
    double d[3] = {b[0] - 1, b[1], b[2]};
    double b1[3];
    if (v3_norm(d) > 1e-5) { /* b x (1,0,0) */
        b1[0] = 0; b1[1] = b[2]; b1[2] = -b[1];
    } else { /* b x (0,0,1) */
        b1[0] = b[1]; b1[1] = -b[0]; b1[2] = 0;
    }
    double n1 = v3_norm(b1);
    b1[0] /= n1; b1[1] /= n1; b1[2] /= n1;
    double b2[3] = {b1[1] * b[2] - b1[2] * b[1], b1[2] * b[0] - b1[0] * b[2], b1[0] * b[1] - b1[1] * b[0]};
    double n2 = v3_norm(b2);
    b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
    double Pt[6] = {b1[0], b1[1], b1[2], b2[0], b2[1], b2[2]};
    double e[3] = {bs[0] - b[0], bs[1] - b[1], bs[2] - b[2]};
    r[0] = weight * (Pt[0] * e[0] + Pt[1] * e[1] + Pt[2] * e[2]);
    r[1] = weight * (Pt[3] * e[0] + Pt[4] * e[1] + Pt[5] * e[2]);

    if (Jp || Jl) {
        /* J_e_lmk = Pt (I - bs bs^T) R_s_f R_f_w / |t_s_lmk|, :89-91 */
        double M[9], PtM[6], Rsw[9], Je[6];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) M[3 * i + j] = ((i == j) ? 1.0 : 0.0) - bs[i] * bs[j];
        mat_mul(Pt, M, PtM, 2, 3, 3);
        m3_mul(Tsf, T0, Rsw);
        mat_mul(PtM, Rsw, Je, 2, 3, 3);
        for (int i = 0; i < 6; i++) Je[i] /= nrm;
        if (Jp) {
            /* J_bear_frame = [-dR [p]x Jr(log dR) | I], :94-98 */
            double S[9], lw[3], Jr[9], RS[9], RSJ[9], Jb[18];
            so3_skew(pw, S);
            so3_log(dT, lw);
            so3_right_jacobian(lw, Jr);
            m3_mul(dT, S, RS);
            m3_mul(RS, Jr, RSJ);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    Jb[i * 6 + j] = -RSJ[3 * i + j];
                    Jb[i * 6 + 3 + j] = (i == j) ? 1.0 : 0.0;
                }
            double Jf[12];
            mat_mul(Je, Jb, Jf, 2, 3, 6);
            for (int i = 0; i < 12; i++) Jp[i] = weight * Jf[i];
        }
        if (Jl) {
            double Jt[6];
            mat_mul(Je, dT, Jt, 2, 3, 3);
            for (int i = 0; i < 6; i++) Jl[i] = weight * Jt[i];
        }
    }
}

/*
 * PosePriordx (residuals.hpp:601-632). sqrt_inf = diag(inf_diag) (…Analytic.cpp:226).
 */
static inline void factor_pose_prior(const double *T0, const double *Tprior, const double *inf_diag,
                                     const double *dpose, double *r /*6*/, double *J /*6x6*/) {
    double dT[12], T[12], Tpi[12], E[12];
    se3_from_delta6(dpose, dT);
    se3_mul(T0, dT, T);
    se3_inverse(Tprior, Tpi);
    se3_mul(T, Tpi, E);
    double w[3];
    so3_log(E, w);
    r[0] = inf_diag[0] * w[0]; r[1] = inf_diag[1] * w[1]; r[2] = inf_diag[2] * w[2];
    r[3] = inf_diag[3] * E[9]; r[4] = inf_diag[4] * E[10]; r[5] = inf_diag[5] * E[11];
    if (J) {
        double Jr_w[9], Jr_w_inv[9], Jr_dw[9], RpT[9], ww[3], RRp[9];
        m3_mul_t(T, Tprior, RRp);
        so3_log(RRp, ww); /* :617 */
        so3_right_jacobian(ww, Jr_w);
        m3_inverse(Jr_w, Jr_w_inv);
        so3_right_jacobian(dpose, Jr_dw);
        double A[9], B00[9];
        m3_mul(Jr_w_inv, Tprior, A);
        m3_mul(A, Jr_dw, B00); /* :618-619 */
        double v[3], S[9], RS[9], B10[9];
        m3_transpose(Tprior, RpT);
        m3_vec(RpT, Tprior + 9, v);
        so3_skew(v, S);
        m3_mul(T, S, RS);
        m3_mul(RS, Jr_dw, B10); /* :620-622 */
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                J[i * 6 + j] = inf_diag[i] * B00[3 * i + j];
                J[i * 6 + 3 + j] = 0.0;
                J[(3 + i) * 6 + j] = inf_diag[3 + i] * B10[3 * i + j];
                J[(3 + i) * 6 + 3 + j] = inf_diag[3 + i] * T0[3 * i + j]; /* :623 */
            }
    }
}

/*
 * ReprojectionErrCeres_linexd_dx::Evaluate, Jacobian branch (BundleAdjustmentCERESAnalytic.h:116-166). Twl: landmark pose,
 * model: the two model points, uv4: the two measured end points, dline: the landmark's 6-vector. As coded:
 *   T_w_lmk = T_w_lmk_ * se3_doubleVec3dtoRT(parameters[1])   -> a translation by dline[0..2]   (:121)
 *   J_lmk block i = jac1 * [-R_w_l [pt_i]x | I]                                                  (:156-160)
 * r4, Jf (4x6), Jl (4x6) row-major. Returns the number of valid projections (an invalid one has r = 0, Jacobian kept).
 */
static inline int factor_line_pixel(const double *T0, const double *K, const double *Tsf, const double *Twl, const double *model,
                                    const double *uv4, double sigma, const double *dpose, const double *dline, double *r4,
                                    double *Jf, double *Jl) {
    static const double z3[3] = {0, 0, 0};
    int nv = 0;
    for (int i = 0; i < 2; i++) {
        const double *pt = model + 3 * i;
        double q[3] = {pt[0] + dline[0], pt[1] + dline[1], pt[2] + dline[2]}, pw[3];
        se3_apply(Twl, q, pw);                         /* T_w_lmk * Tpt, :127-131 */
        double Jp[12], J3[6];
        nv += factor_pixel(T0, K, Tsf, pw, uv4 + 2 * i, sigma, dpose, z3, r4 + 2 * i, (Jf || Jl) ? Jp : NULL, (Jf || Jl) ? J3 : NULL);
        if (Jf) for (int a = 0; a < 12; a++) Jf[12 * i + a] = Jp[a];      /* jac0 * J_lf_dlf, :144-152 */
        if (Jl) {
            double S[9], RS[9];
            so3_skew(pt, S);
            m3_mul(Twl, S, RS);
            for (int q2 = 0; q2 < 2; q2++)
                for (int a = 0; a < 3; a++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s -= J3[3 * q2 + k] * RS[3 * k + a];
                    Jl[(2 * i + q2) * 6 + a] = s;
                    Jl[(2 * i + q2) * 6 + 3 + a] = J3[3 * q2 + a];
                }
        }
    }
    return nv;
}

/*
 * AngularErrCeres_linexd_dx::Evaluate (AngularAdjustmentCERESAnalytic.h:378-459). b6: the two bearing vectors of the
 * feature; weight = 1 / sigma^2 (:382). r2, Jf (2x6), Jl (2x6) row-major. J_normalization is (I - X X^T) / |X| with the
 * UN-normalised X, as coded (geometry.h:332-334).
 */
static inline void factor_line_angular(const double *T0, const double *Tsf, const double *Twl, const double *b6, double sigma,
                                       const double *dpose, const double *dline, double *r2, double *Jf, double *Jl) {
    double dT[12], dTl[12], A[12], Bm[12], Cm[12], Tsl[12];
    se3_from_delta6(dpose, dT); se3_from_delta6(dline, dTl);
    const double weight = 1.0 / (sigma * sigma);
    se3_mul(Tsf, T0, A); se3_mul(A, dT, Bm); se3_mul(Bm, Twl, Cm); se3_mul(Cm, dTl, Tsl);   /* :384 */
    double n_obs[3] = {b6[1] * b6[5] - b6[2] * b6[4], b6[2] * b6[3] - b6[0] * b6[5], b6[0] * b6[4] - b6[1] * b6[3]};
    double nn = sqrt(n_obs[0] * n_obs[0] + n_obs[1] * n_obs[1] + n_obs[2] * n_obs[2]);
    for (int i = 0; i < 3; i++) n_obs[i] /= nn;                                            /* :388-389 */
    const double *t = Tsl + 9;
    const double tn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    double bl[3] = {t[0] / tn, t[1] / tn, t[2] / tn};                                       /* :392 */
    double dir[3] = {Tsl[0], Tsl[3], Tsl[6]};                                               /* R e_x, normalised (geometry.h:125-129) */
    { double dn = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]); for (int i = 0; i < 3; i++) dir[i] /= dn; }
    double nl[3] = {bl[1] * dir[2] - bl[2] * dir[1], bl[2] * dir[0] - bl[0] * dir[2], bl[0] * dir[1] - bl[1] * dir[0]};
    const double nln = sqrt(nl[0] * nl[0] + nl[1] * nl[1] + nl[2] * nl[2]);
    double nlh[3] = {nl[0] / nln, nl[1] / nln, nl[2] / nln};                                /* :393-394 */
    double cx[3] = {n_obs[1] * nlh[2] - n_obs[2] * nlh[1], n_obs[2] * nlh[0] - n_obs[0] * nlh[2], n_obs[0] * nlh[1] - n_obs[1] * nlh[0]};
    const double cxn = sqrt(cx[0] * cx[0] + cx[1] * cx[1] + cx[2] * cx[2]);
    r2[0] = weight * cxn;                                                                   /* :400 */
    r2[1] = weight * (n_obs[0] * bl[0] + n_obs[1] * bl[1] + n_obs[2] * bl[2]);              /* :401 */
    if (!Jf && !Jl) return;
    /* J_e0_n_lmk = J_norm(cx) * J_AcrossX(n_obs) * J_normalization(n_ldmk)   (:406-408) */
    double Sn[9], Jn_nl[9], M1[9], e0[3], e1[3];
    so3_skew(n_obs, Sn);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jn_nl[3 * i + j] = ((i == j ? 1.0 : 0.0) - nl[i] * nl[j]) / nln;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s -= Sn[3 * i + k] * Jn_nl[3 * k + j]; M1[3 * i + j] = s; }
    for (int j = 0; j < 3; j++) e0[j] = (cx[0] * M1[j] + cx[1] * M1[3 + j] + cx[2] * M1[6 + j]) / cxn;
    /* J_e1_t_lmk = n_obs^T J_normalization(t)   (:410-411) */
    double Jn_t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jn_t[3 * i + j] = ((i == j ? 1.0 : 0.0) - t[i] * t[j]) / tn;
    for (int j = 0; j < 3; j++) e1[j] = n_obs[0] * Jn_t[j] + n_obs[1] * Jn_t[3 + j] + n_obs[2] * Jn_t[6 + j];
    /* blocks of :413-436 */
    double Rsw[9], RswdR[9], lw[3], Jr_dT[9], Jr_dTl[9], v[3], S1[9], S2[9], P1[9], P2[9], Jt_dT[18], JR_dT[18], Jt_dL[18], JR_dL[18];
    m3_mul(Tsf, T0, Rsw);                      /* (T_s_f T_f_w).rotation() */
    m3_mul(Rsw, dT, RswdR);
    so3_log(dT, lw); so3_right_jacobian(lw, Jr_dT);
    so3_log(dTl, lw); so3_right_jacobian(lw, Jr_dTl);
    memset(Jt_dT, 0, sizeof(Jt_dT)); memset(JR_dT, 0, sizeof(JR_dT)); memset(Jt_dL, 0, sizeof(Jt_dL)); memset(JR_dL, 0, sizeof(JR_dL));
    m3_vec(Twl, dTl + 9, v); so3_skew(v, S1);                                  /* [R_w_l t_dl]x */
    m3_mul(RswdR, S1, P1); m3_mul(P1, Jr_dT, P2);
    so3_skew(Twl + 9, S2); m3_mul(RswdR, S2, P1);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Jt_dT[6 * i + j] = -P2[3 * i + j] - P1[3 * i + j]; Jt_dT[6 * i + 3 + j] = Rsw[3 * i + j]; }
    double Rwl_dRl[9], ex[3];
    m3_mul(Twl, dTl, Rwl_dRl);
    ex[0] = Rwl_dRl[0]; ex[1] = Rwl_dRl[3]; ex[2] = Rwl_dRl[6];              /* R_w_l dR_l e_x */
    so3_skew(ex, S1); m3_mul(RswdR, S1, P1); m3_mul(P1, Jr_dT, P2);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) JR_dT[6 * i + j] = -P2[3 * i + j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jt_dL[6 * i + 3 + j] = Cm[3 * i + j];   /* (T_s_f T_f_w dT T_w_l).rotation() */
    { const double e_x[3] = {1, 0, 0}; so3_skew(e_x, S1); }
    m3_mul(Cm, dTl, P1); m3_mul(P1, S1, P2); m3_mul(P2, Jr_dTl, P1);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) JR_dL[6 * i + j] = -P1[3 * i + j];
    /* row 0: e0 * ( [R_s_l e_x]x^T (J_normalization(t) J_t) + [n_ldmk_normed]x J_R ), row 1: e1 * J_t   (:441-457) */
    double Sd[9], Snl[9];
    { const double rx[3] = {Tsl[0], Tsl[3], Tsl[6]}; so3_skew(rx, Sd); }
    so3_skew(nlh, Snl);
    for (int which = 0; which < 2; which++) {
        const double *Jt = which ? Jt_dL : Jt_dT, *JR = which ? JR_dL : JR_dT;
        double *out = which ? Jl : Jf;
        if (!out) continue;
        for (int c = 0; c < 6; c++) {
            double a[3], b[3], g[3];
            for (int i = 0; i < 3; i++) { a[i] = Jn_t[3 * i] * Jt[c] + Jn_t[3 * i + 1] * Jt[6 + c] + Jn_t[3 * i + 2] * Jt[12 + c]; }
            for (int i = 0; i < 3; i++) b[i] = Sd[i] * a[0] + Sd[3 + i] * a[1] + Sd[6 + i] * a[2];       /* Sd^T a */
            for (int i = 0; i < 3; i++) g[i] = b[i] + Snl[3 * i] * JR[c] + Snl[3 * i + 1] * JR[6 + c] + Snl[3 * i + 2] * JR[12 + c];
            out[c] = weight * (e0[0] * g[0] + e0[1] * g[1] + e0[2] * g[2]);
            out[6 + c] = weight * (e1[0] * Jt[c] + e1[1] * Jt[6 + c] + e1[2] * Jt[12 + c]);
        }
    }
}

/*
 * Relative6DPose::Evaluate (residuals.hpp:70-131). Ta / Tb: the transforms the factor composes its deltas on (the reference
 * passes frame-to-world poses T_w_a, T_w_b, …Analytic.cpp:787-790); Tab: T_a_b_prior; W: 6x6 sqrt information, row-major.
 * r = W [log(R); t] of T = T_a_b_prior^-1 (Ta dTa)^-1 (Tb dTb). Ja, Jb 6x6 row-major (may be NULL).
 */
static inline void factor_relative_pose(const double *Ta, const double *Tb, const double *Tab, const double *W, const double *da,
                                        const double *db, double *r /*6*/, double *Ja /*6x6*/, double *Jb /*6x6*/) {
    double dTa[12], dTb[12], Tau[12], Tbu[12], Tba[12], Taui[12], M[12], T[12], e[6], w[3];
    se3_from_delta6(da, dTa); se3_from_delta6(db, dTb);
    se3_mul(Ta, dTa, Tau); se3_mul(Tb, dTb, Tbu);              /* :80-81 */
    se3_inverse(Tab, Tba);                                     /* :82 */
    se3_inverse(Tau, Taui);
    se3_mul(Tba, Taui, M); se3_mul(M, Tbu, T);                 /* :83 */
    so3_log(T, w);
    e[0] = w[0]; e[1] = w[1]; e[2] = w[2]; e[3] = T[9]; e[4] = T[10]; e[5] = T[11];   /* se3_RTtoVec6d, :84 */
    for (int i = 0; i < 6; i++) { double s = 0; for (int j = 0; j < 6; j++) s += W[6 * i + j] * e[j]; r[i] = s; }
    if (!Ja && !Jb) return;
    double Jrw[9], Jrwi[9];
    so3_right_jacobian(w, Jrw);                                /* w = log_so3(T.rotation()), :90 */
    m3_inverse(Jrw, Jrwi);
    double J[36];
    if (Ja) {
        double Jrd[9], A[9], B[9], C[9], tba[3], S[9];
        so3_right_jacobian(da, Jrd);
        memset(J, 0, sizeof(J));
        m3_tmul(Tbu, Tau, A);                                  /* R_b^T R_a */
        m3_mul(Jrwi, A, B); m3_mul(B, Jrd, C);                 /* :98-99, negated below */
        for (int i = 0; i < 3; i++) tba[i] = Tbu[9 + i] - Tau[9 + i];
        so3_skew(tba, S);
        double D1[9], D2[9], D3[9], D4[9];
        m3_mul_t(Tba, Tau, D1);                                /* R_ba_prior R_a^T */
        m3_mul(D1, S, D2); m3_mul(D2, Tau, D3); m3_mul(D3, Jrd, D4);   /* :102-104 */
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                J[i * 6 + j] = -C[3 * i + j];
                J[(3 + i) * 6 + j] = D4[3 * i + j];
                J[(3 + i) * 6 + 3 + j] = -Tba[3 * i + j];      /* :107 */
            }
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { double s = 0; for (int k = 0; k < 6; k++) s += W[6 * i + k] * J[6 * k + j]; Ja[6 * i + j] = s; }
    }
    if (Jb) {
        double Jrd[9], B[9], D1[9], D2[9];
        so3_right_jacobian(db, Jrd);
        memset(J, 0, sizeof(J));
        m3_mul(Jrwi, Jrd, B);                                  /* :118 */
        m3_mul_t(Tba, Tau, D1); m3_mul(D1, Tbu, D2);           /* :121 */
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { J[i * 6 + j] = B[3 * i + j]; J[(3 + i) * 6 + 3 + j] = D2[3 * i + j]; }
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { double s = 0; for (int k = 0; k < 6; k++) s += W[6 * i + k] * J[6 * k + j]; Jb[6 * i + j] = s; }
    }
}

/* IMUPriordx::Evaluate (residuals.hpp:649-695). params: pose6 | dv3 | dba3 | dbg3. r = W e (15), e = [log(R Rp^-1);
 * trans(T Tp^-1); v + dv - vp; ba + dba - bap; bg + dbg - bgp]. Jacobian 15x15 row-major [pose6|v3|ba3|bg3]: the pose
 * block is W * [J6; 0] (:676), the v / ba / bg blocks are plain identities at rows 6 / 9 / 12 -- NOT multiplied by
 * the sqrt information (:679-693, reproduced as coded). */
static inline void factor_imu_prior(const double *T0, const double *v0, const double *ba0, const double *bg0,
                                    const double *Tp, const double *vp, const double *bap, const double *bgp,
                                    const double *W /*15x15*/, const double *params /*15*/, double *r /*15*/,
                                    double *J /*15x15 or NULL*/) {
    static const double ones[6] = {1, 1, 1, 1, 1, 1};
    double e[15], J6[36];
    factor_pose_prior(T0, Tp, ones, params, e, J ? J6 : NULL);
    for (int a = 0; a < 3; a++) {
        e[6 + a] = v0[a] + params[6 + a] - vp[a];
        e[9 + a] = ba0[a] + params[9 + a] - bap[a];
        e[12 + a] = bg0[a] + params[12 + a] - bgp[a];
    }
    for (int i = 0; i < 15; i++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += W[i * 15 + k] * e[k];
        r[i] = s;
    }
    if (J) {
        memset(J, 0, sizeof(double) * 225);
        for (int i = 0; i < 15; i++)
            for (int a = 0; a < 6; a++) {
                double s = 0;
                for (int k = 0; k < 6; k++) s += W[i * 15 + k] * J6[k * 6 + a];
                J[i * 15 + a] = s;
            }
        for (int a = 0; a < 9; a++) J[(6 + a) * 15 + 6 + a] = 1.0;
    }
}

/* PoseToLandmarkFactor::Evaluate (residuals.hpp:570-595): r = W (T_f_w (exp w, t) (p + dl) - delta). */
static inline void factor_pose_to_landmark(const double *T0, const double *p0, const double *delta, const double *W /*3x3*/,
                                           const double *dpose, const double *dl, double *r, double *Jp /*3x6*/,
                                           double *Jl /*3x3*/) {
    double dT[12], T[12], q[3], Tq[3], e[3];
    se3_from_delta6(dpose, dT);
    se3_mul(T0, dT, T);
    for (int a = 0; a < 3; a++) q[a] = p0[a] + dl[a];
    se3_apply(T, q, Tq);
    for (int a = 0; a < 3; a++) e[a] = Tq[a] - delta[a];
    m3_vec(W, e, r);
    if (Jp) {
        double Sq[9], Jr[9], A[9], B[9], C[9], WR0[9];
        so3_skew(q, Sq);
        so3_right_jacobian(dpose, Jr);
        m3_mul(dT, Sq, A);      /* dR [q]x */
        m3_mul(A, Jr, B);       /* dR [q]x Jr(w) */
        m3_mul(W, T0, WR0);     /* W R0 */
        m3_mul(WR0, B, C);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { Jp[i * 6 + j] = -C[3 * i + j]; Jp[i * 6 + 3 + j] = WR0[3 * i + j]; }
    }
    if (Jl) m3_mul(W, T, Jl);   /* W (R0 dR) */
}

/* Landmark3DPrior (residuals.hpp:512-522): r = W (l + dl - prior), J = W. */
static inline void factor_landmark_prior(const double *p0, const double *prior, const double *W, const double *dl,
                                         double *r, double *J) {
    double e[3] = {p0[0] + dl[0] - prior[0], p0[1] + dl[1] - prior[1], p0[2] + dl[2] - prior[2]};
    m3_vec(W, e, r);
    if (J) memcpy(J, W, 72);
}

/* LandmarkToLandmarkFactor (residuals.hpp:537-556): r = W ((l0 + d0) - (l1 + d1) - delta), J0 = W, J1 = -W. */
static inline void factor_landmark_to_landmark(const double *p0, const double *p1, const double *delta, const double *W,
                                               const double *d0, const double *d1, double *r, double *J0, double *J1) {
    double e[3];
    for (int a = 0; a < 3; a++) e[a] = (p0[a] + d0[a]) - (p1[a] + d1[a]) - delta[a];
    m3_vec(W, e, r);
    if (J0) memcpy(J0, W, 72);
    if (J1) for (int a = 0; a < 9; a++) J1[a] = -W[a];
}

/* Cholesky-based sqrt information of a 9x9 covariance: W = L^T with L L^T = cov^-1
 * (residuals.hpp:151-154). Dense Gauss-Jordan inverse then LLT. Returns 0 on success. */
static inline int imu_sqrt_information(const double *cov, double *W /*9x9 row-major, upper*/) {
    double A[81], I[81];
    memcpy(A, cov, sizeof(A));
    memset(I, 0, sizeof(I));
    for (int i = 0; i < 9; i++) I[i * 9 + i] = 1.0;
    /* Gauss-Jordan with partial pivoting (Eigen's dynamic inverse is PartialPivLU) */
    for (int c = 0; c < 9; c++) {
        int piv = c;
        double best = fabs(A[c * 9 + c]);
        for (int rr = c + 1; rr < 9; rr++)
            if (fabs(A[rr * 9 + c]) > best) { best = fabs(A[rr * 9 + c]); piv = rr; }
        if (best == 0.0) return 1;
        if (piv != c)
            for (int j = 0; j < 9; j++) {
                double t = A[c * 9 + j]; A[c * 9 + j] = A[piv * 9 + j]; A[piv * 9 + j] = t;
                t = I[c * 9 + j]; I[c * 9 + j] = I[piv * 9 + j]; I[piv * 9 + j] = t;
            }
        double d = 1.0 / A[c * 9 + c];
        for (int j = 0; j < 9; j++) { A[c * 9 + j] *= d; I[c * 9 + j] *= d; }
        for (int rr = 0; rr < 9; rr++) {
            if (rr == c) continue;
            double f = A[rr * 9 + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 9; j++) { A[rr * 9 + j] -= f * A[c * 9 + j]; I[rr * 9 + j] -= f * I[c * 9 + j]; }
        }
    }
    /* LLT of the (symmetrised) information; W = L^T */
    double L[81];
    memset(L, 0, sizeof(L));
    for (int j = 0; j < 9; j++) {
        double s = 0.5 * (I[j * 9 + j] + I[j * 9 + j]);
        for (int k = 0; k < j; k++) s -= L[j * 9 + k] * L[j * 9 + k];
        if (!(s > 0.0)) return 2;
        double d = sqrt(s);
        L[j * 9 + j] = d;
        for (int i = j + 1; i < 9; i++) {
            double t = I[i * 9 + j]; /* Eigen LLT reads the lower triangle */
            for (int k = 0; k < j; k++) t -= L[i * 9 + k] * L[j * 9 + k];
            L[i * 9 + j] = t / d;
        }
    }
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) W[i * 9 + j] = L[j * 9 + i];
    return 0;
}

typedef struct {
    double dt;
    const double *dR, *dv, *dp;                 /* preintegrated deltas of imu_j */
    const double *J_dR_bg, *J_dv_ba, *J_dv_bg, *J_dp_ba, *J_dp_bg;
    const double *W;                            /* 9x9 sqrt information (imu_sqrt_information) */
} imu_consts;

/*
 * IMUFactor (residuals.hpp:133-245). Blocks [pose_i 6, pose_j 6, dv_i 3, dv_j 3, dba_i 3, dbg_i 3].
 * Ti0/Tj0 = T_f_w of frames i/j, vi0/vj0 their velocities. Jacobians J[k] row-major 9 x size_k.
 */
static inline void factor_imu(const imu_consts *c, const double *Ti0, const double *Tj0, const double *vi0,
                              const double *vj0, const double *dposei, const double *dposej, const double *dvi,
                              const double *dvj, const double *dba, const double *dbg, double *r /*9*/,
                              double *J_pi, double *J_pj, double *J_vi, double *J_vj, double *J_ba, double *J_bg) {
    double dTi[12], dTj[12], Ti[12], Tj[12];
    se3_from_delta6(dposei, dTi);
    se3_from_delta6(dposej, dTj);
    se3_mul(Ti0, dTi, Ti);
    se3_mul(Tj0, dTj, Tj);
    double vi[3], vj[3];
    for (int k = 0; k < 3; k++) { vi[k] = vi0[k] + dvi[k]; vj[k] = vj0[k] + dvj[k]; }
    double dt = c->dt;

    /* dR = (DeltaR exp(J_dR_bg dbg))^T R_i R_j^T, :157-158 */
    double jb[3], Eb[9], DRc[9], RiRjT[9], dR[9];
    m3_vec(c->J_dR_bg, dbg, jb);
    so3_exp(jb, Eb);
    m3_mul(c->dR, Eb, DRc);
    m3_mul_t(Ti, Tj, RiRjT);
    m3_tmul(DRc, RiRjT, dR);
    double r_dr[3];
    so3_log(dR, r_dr);
    /* r_dv, :160-161 */
    double a[3] = {vj[0] - vi[0] - ORACLE_G[0] * dt, vj[1] - vi[1] - ORACLE_G[1] * dt, vj[2] - vi[2] - ORACLE_G[2] * dt};
    double Ra[3], t1[3], t2[3];
    m3_vec(Ti, a, Ra);
    m3_vec(c->J_dv_bg, dbg, t1);
    m3_vec(c->J_dv_ba, dba, t2);
    double r_dv[3] = {Ra[0] - (c->dv[0] + t1[0] + t2[0]), Ra[1] - (c->dv[1] + t1[1] + t2[1]),
                      Ra[2] - (c->dv[2] + t1[2] + t2[2])};
    /* r_dp, :162-164 */
    double Tii[12], Tji[12];
    se3_inverse(Ti, Tii);
    se3_inverse(Tj, Tji);
    double bvec[3];
    for (int k = 0; k < 3; k++) bvec[k] = Tji[9 + k] - Tii[9 + k] - vi[k] * dt - 0.5 * ORACLE_G[k] * dt * dt;
    double Rb[3];
    m3_vec(Ti, bvec, Rb);
    m3_vec(c->J_dp_bg, dbg, t1);
    m3_vec(c->J_dp_ba, dba, t2);
    double r_dp[3] = {Rb[0] - (c->dp[0] + t1[0] + t2[0]), Rb[1] - (c->dp[1] + t1[1] + t2[1]),
                      Rb[2] - (c->dp[2] + t1[2] + t2[2])};
    double e[9] = {r_dr[0], r_dr[1], r_dr[2], r_dv[0], r_dv[1], r_dv[2], r_dp[0], r_dp[1], r_dp[2]};
    mat_mul(c->W, e, r, 9, 9, 1);

    double Jr_rdr[9], Jr_rdr_inv[9];
    so3_right_jacobian(r_dr, Jr_rdr);
    m3_inverse(Jr_rdr, Jr_rdr_inv);
    double tmp[9 * 6];

    if (J_pi) { /* :174-187 */
        double Jrw[9], A[9], B[9], S[9], RS[9];
        so3_right_jacobian(dposei, Jrw);
        memset(tmp, 0, sizeof(double) * 54);
        m3_mul(Jr_rdr_inv, Tj, A);
        m3_mul(A, Jrw, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[i * 6 + j] = B[3 * i + j];
        so3_skew(a, S);
        m3_mul(Ti, S, RS);
        m3_mul(RS, Jrw, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[(3 + i) * 6 + j] = -B[3 * i + j];
        double cc[3];
        for (int k = 0; k < 3; k++) cc[k] = Tji[9 + k] - vi[k] * dt - 0.5 * ORACLE_G[k] * dt * dt; /* p_j, not p_j - p_i */
        so3_skew(cc, S);
        m3_mul(Ti, S, RS);
        m3_mul(RS, Jrw, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[(6 + i) * 6 + j] = -B[3 * i + j];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[(6 + i) * 6 + 3 + j] = Ti0[3 * i + j]; /* UNperturbed, :185 */
        mat_mul(c->W, tmp, J_pi, 9, 9, 6);
    }
    if (J_pj) { /* :190-200 */
        double Jrw[9], A[9], B[9], S[9], C1[9], C2[9];
        so3_right_jacobian(dposej, Jrw);
        memset(tmp, 0, sizeof(double) * 54);
        m3_mul(Jr_rdr_inv, Tj, A);
        m3_mul(A, Jrw, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[i * 6 + j] = -B[3 * i + j];
        /* -R_i R_j^T [t_j]x R_j Jr(w_j) */
        so3_skew(Tj + 9, S);
        m3_mul(RiRjT, S, C1);
        m3_mul(C1, Tj, C2);
        m3_mul(C2, Jrw, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[(6 + i) * 6 + j] = -B[3 * i + j];
        /* -R_i exp(w_j)^T, :198 */
        m3_mul_t(Ti, dTj, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[(6 + i) * 6 + 3 + j] = -B[3 * i + j];
        mat_mul(c->W, tmp, J_pj, 9, 9, 6);
    }
    if (J_vi) { /* :203-209 */
        memset(tmp, 0, sizeof(double) * 27);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            tmp[(3 + i) * 3 + j] = -Ti[3 * i + j];
            tmp[(6 + i) * 3 + j] = -Ti[3 * i + j] * dt;
        }
        mat_mul(c->W, tmp, J_vi, 9, 9, 3);
    }
    if (J_vj) { /* :212-217 */
        memset(tmp, 0, sizeof(double) * 27);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[(3 + i) * 3 + j] = Ti[3 * i + j];
        mat_mul(c->W, tmp, J_vj, 9, 9, 3);
    }
    if (J_ba) { /* :220-226 */
        memset(tmp, 0, sizeof(double) * 27);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            tmp[(3 + i) * 3 + j] = -c->J_dv_ba[3 * i + j];
            tmp[(6 + i) * 3 + j] = -c->J_dp_ba[3 * i + j];
        }
        mat_mul(c->W, tmp, J_ba, 9, 9, 3);
    }
    if (J_bg) { /* :229-237 */
        memset(tmp, 0, sizeof(double) * 27);
        double Jrb[9], A[9], B[9], C[9];
        so3_right_jacobian(jb, Jrb);
        m3_mul_t(Jr_rdr_inv, dR, A); /* Jr^-1 dR^T */
        m3_mul(A, Jrb, B);
        m3_mul(B, c->J_dR_bg, C);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            tmp[i * 3 + j] = -C[3 * i + j];
            tmp[(3 + i) * 3 + j] = -c->J_dv_bg[3 * i + j];
            tmp[(6 + i) * 3 + j] = -c->J_dp_bg[3 * i + j];
        }
        mat_mul(c->W, tmp, J_bg, 9, 9, 3);
    }
}

/*
 * IMUBiasFactor (residuals.hpp:247-300). Blocks [dba_i, dbg_i, dba_j, dbg_j]; the Jacobians are
 * -/+ sa*I in rows 0..2 (ba) and -/+ sg*I in rows 3..5 (bg); returned as the two scalars.
 */
static inline void factor_imu_bias(double dt, double bacc_noise, double bgyr_noise, const double *bai,
                                   const double *bgi, const double *baj, const double *bgj, const double *dbai,
                                   const double *dbgi, const double *dbaj, const double *dbgj, double *r /*6*/,
                                   double *sa, double *sg) {
    double s2a = dt * bacc_noise * bacc_noise;
    double s2g = dt * bgyr_noise * bgyr_noise;
    *sa = 1 / sqrt(s2a);
    *sg = 1 / sqrt(s2g);
    for (int k = 0; k < 3; k++) {
        r[k] = *sa * (baj[k] + dbaj[k] - bai[k] - dbai[k]);
        r[3 + k] = *sg * (bgj[k] + dbgj[k] - bgi[k] - dbgi[k]);
    }
}
#endif
