// device_math.h — gfx950 device-side SO3 / factor arithmetic of the BA hot path (FP64).
//
// Restates the reference's closed forms with the SAME small-angle branches
// (cpp/include/utilities/geometry.h:17-23,30-37,131-147,149-166) and the reference's factor
// algebra (BundleAdjustmentCERESAnalytic.h:52-90 + Camera.cpp:84-139;
// AngularAdjustmentCERESAnalytic.h:55-111), simplified where the simplification is an identity:
//   * the pixel factor's J_pose rotation block is A * (-R [p]x Jr(log R)) * Jr(log R)^-1 * Jr(w)
//     in the reference (Camera.cpp:108-109 x …Analytic.h:73-77); Jr Jr^-1 = I, so the device
//     evaluates -J_lmk [p]x Jr(w) directly (no log / inverse per observation);
//   * J_h * K = (1/z) [[fx, 0, -fx x/z], [0, fy, -fy y/z]].
// Matrices are row-major double[9]; a rigid transform is R(9) | t(3).
#pragma once
#include <hip/hip_runtime.h>

namespace sadvio {

struct V3 {
    double x, y, z;
};

__device__ __forceinline__ void m3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void m3_vec(const double* A, const double* v, double* o) {
    double x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ void m3_tvec(const double* A, const double* v, double* o) {
    double x = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double y = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double z = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ void m3_mul_t(const double* A, const double* B, double* C) {  // C = A B^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
__device__ __forceinline__ void m3_tmul(const double* A, const double* B, double* C) {  // C = A^T B
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

__device__ __forceinline__ double v3_norm(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// sin / cos of a double in registers. The ROCm device-library sin()/cos() cost ~20 us per call site on
// gfx950 in this kernel mix (measured: 79 us for one pose table), far above the whole reduced solve, so the
// closed forms are evaluated with the classic fdlibm kernels: Cody-Waite reduction by pi/2 (exact for
// |x| < ~1e5, BA deltas are << 1 rad) + the degree-13/14 minimax polynomials on [-pi/4, pi/4] (< 1 ulp).
__device__ __forceinline__ void sincos_f64(double x, double* sn, double* cs) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double fn = rint(x * invpio2);
    double t = x - fn * pio2_1;
    double w = fn * pio2_2;
    double r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    (void)pio2_1t;
    const double z = y0 * y0;
    // kernel sin
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * y0;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    // kernel cos
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    const int q = ((int)fn) & 3;
    const double s0 = (q & 1) ? kc : ks;
    const double c0 = (q & 1) ? ks : kc;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

// geometry.h:17-23
__device__ __forceinline__ void so3_skew(const double* w, double* S) {
    S[0] = 0; S[1] = -w[2]; S[2] = w[1];
    S[3] = w[2]; S[4] = 0; S[5] = -w[0];
    S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
// geometry.h:30-37 (exactly I below 1e-5)
__device__ __forceinline__ void so3_right_jacobian(const double* w, double* J) {
    double n = v3_norm(w);
    J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 0; J[4] = 1; J[5] = 0; J[6] = 0; J[7] = 0; J[8] = 1;
    if (n < 1e-5) return;
    double S[9], S2[9];
    so3_skew(w, S);
    m3_mul(S, S, S2);
    double sn, cs;
    sincos_f64(n, &sn, &cs);
    double a = (1 - cs) / (n * n);
    double b = (n - sn) / (n * n * n);
#pragma unroll
    for (int i = 0; i < 9; i++) J[i] = J[i] - a * S[i] + b * S2[i];
}
// geometry.h:131-147 (first order below 1e-9)
__device__ __forceinline__ void so3_exp(const double* v, double* R) {
    double angle = v3_norm(v);
    double S[9];
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    if (angle < 1e-9) {
        so3_skew(v, S);
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] += S[i];
        return;
    }
    double axis[3] = {v[0] / angle, v[1] / angle, v[2] / angle};
    double S2[9];
    so3_skew(axis, S);
    m3_mul(S, S, S2);
    double sn_, cs_;
    sincos_f64(angle, &sn_, &cs_);
    double c = 1. - cs_, s = sn_;
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] += c * S2[i] + s * S[i];
}
// geometry.h:149-166 (first order if |sin| < 1e-9 or angle < 1e-9 — also near pi)
__device__ __forceinline__ void so3_log(const double* M, double* phi) {
    double cos_angle = 0.5 * (M[0] + M[4] + M[8]) - 0.5;
    cos_angle = fmin(fmax(cos_angle, -1.), 1.);
    double angle = acos(cos_angle);
    double sa, ca;
    sincos_f64(angle, &sa, &ca);
    double k = (fabs(sa) < 1e-9 || angle < 1e-9) ? 0.5 : 0.5 * angle / sa;
    phi[0] = k * (M[7] - M[5]);
    phi[1] = k * (M[2] - M[6]);
    phi[2] = k * (M[3] - M[1]);
}
// Inverse of a symmetric positive-definite 3x3 given as (a00,a01,a02,a11,a12,a22); returns det.
__device__ __forceinline__ double sym3_inverse(const double* a, double* inv) {
    double c00 = a[3] * a[5] - a[4] * a[4];
    double c01 = a[2] * a[4] - a[1] * a[5];
    double c02 = a[1] * a[4] - a[2] * a[3];
    double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    double id = 1.0 / det;
    inv[0] = c00 * id;
    inv[1] = c01 * id;
    inv[2] = c02 * id;
    inv[3] = (a[0] * a[5] - a[2] * a[2]) * id;
    inv[4] = (a[1] * a[2] - a[0] * a[4]) * id;
    inv[5] = (a[0] * a[3] - a[1] * a[1]) * id;
    return det;
}
// Landmark elimination in Cholesky form (round 5): M = L L^T for the damped symmetric block (a00,a01,a02,a11,a12,a22), returns
// Li = L^-1 as (i00, i10, i11, i20, i21, i22). With W = E Li^T the Schur term is E M^-1 E^T = W W^T and M^-1 t = Li^T (Li t): the
// block step of a landmark-first Cholesky of the un-reduced normal equations — what CHOLMOD computes for the reference
// (AOptimizer.cpp:315-323). The adjugate inverse used through round 4 lands 4e-6 .. 7e-5 from a long-double solve on the two
// ill-conditioned windows of the sweep where this form lands 4e-7 .. 4e-8 (scripts/elim_numerics.py). A block that is not positive
// definite yields NaNs (sqrt of a negative pivot), which the step's finiteness test turns into an invalid step, as 1 / det did.
__device__ __forceinline__ void sym3_chol_inverse(const double* a, double* Li) {
    const double l00 = sqrt(a[0]);
    const double i00 = 1.0 / l00;
    const double l10 = a[1] * i00, l20 = a[2] * i00;
    const double l11 = sqrt(a[3] - l10 * l10);
    const double i11 = 1.0 / l11;
    const double l21 = (a[4] - l20 * l10) * i11;
    const double l22 = sqrt(a[5] - l20 * l20 - l21 * l21);
    const double i22 = 1.0 / l22;
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    Li[0] = i00; Li[1] = i10; Li[2] = i11; Li[3] = i20; Li[4] = i21; Li[5] = i22;
}
// N = Jl Li^T (rows of a 2 x 3 Jacobian), u = Li t and v = Li^T u for the packed lower Li of sym3_chol_inverse
__device__ __forceinline__ void li_row(const double* Li, double j0, double j1, double j2, double* n) {
    n[0] = j0 * Li[0];
    n[1] = j0 * Li[1] + j1 * Li[2];
    n[2] = j0 * Li[3] + j1 * Li[4] + j2 * Li[5];
}
__device__ __forceinline__ void li_vec(const double* Li, const double* t, double* u) {
    u[0] = Li[0] * t[0];
    u[1] = Li[1] * t[0] + Li[2] * t[1];
    u[2] = Li[3] * t[0] + Li[4] * t[1] + Li[5] * t[2];
}
__device__ __forceinline__ void li_tvec(const double* Li, const double* u, double* v) {
    v[0] = Li[0] * u[0] + Li[1] * u[1] + Li[3] * u[2];
    v[1] = Li[2] * u[1] + Li[4] * u[2];
    v[2] = Li[5] * u[2];
}
__device__ __forceinline__ void m3_inverse(const double* A, double* I) {
    double c00 = A[4] * A[8] - A[5] * A[7];
    double c01 = A[5] * A[6] - A[3] * A[8];
    double c02 = A[3] * A[7] - A[4] * A[6];
    double id = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
    I[0] = c00 * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c01 * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c02 * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// Per-key-frame quantities at the current deltas, staged once per workgroup.
//   R = R0 exp(w), t = R0 td + t0   (T_f_w = T_f_w0 * (exp(w), td), geometry.h:198-203)
//   Jr = so3_rightJacobian(w); dR = exp(w)
// Relative6DPose::Evaluate (residuals.hpp:70-131): r = W [log R; t] of T = Tab^-1 (Ta dTa)^-1 (Tb dTb); J (6 x 15 row-major,
// columns [a 6 | b 6 | -]) may be null. Ta / Tb are whatever the window's key-frame slots hold (frame-to-world poses in the
// pose-graph use of the reference, …Analytic.cpp:787-790).
__device__ __forceinline__ void relative_pose_factor(const double* Ta, const double* Tb, const double* Tab, const double* W,
                                                     const double* da, const double* db, double* r, double* J) {
    double Ea[9], Eb[9], Rau[9], Rbu[9], tau[3], tbu[3], v[3];
    so3_exp(da, Ea); so3_exp(db, Eb);
    m3_mul(Ta, Ea, Rau); m3_mul(Tb, Eb, Rbu);
    m3_vec(Ta, da + 3, v); for (int i = 0; i < 3; i++) tau[i] = v[i] + Ta[9 + i];
    m3_vec(Tb, db + 3, v); for (int i = 0; i < 3; i++) tbu[i] = v[i] + Tb[9 + i];
    // R = Rp^T Rau^T Rbu ; t = Rp^T (Rau^T (tbu - tau) - tp)
    double RaT_Rb[9], R[9], d[3], u[3], t[3], w[3];
    m3_tmul(Rau, Rbu, RaT_Rb);
    m3_tmul(Tab, RaT_Rb, R);
    for (int i = 0; i < 3; i++) d[i] = tbu[i] - tau[i];
    m3_tvec(Rau, d, u);
    for (int i = 0; i < 3; i++) u[i] -= Tab[9 + i];
    m3_tvec(Tab, u, t);
    so3_log(R, w);
    const double e[6] = {w[0], w[1], w[2], t[0], t[1], t[2]};
    for (int i = 0; i < 6; i++) { double s = 0.0; for (int j = 0; j < 6; j++) s += W[6 * i + j] * e[j]; r[i] = s; }
    if (!J) return;
    double Jrw[9], Jrwi[9], Jra[9], Jrb[9], A[9], B[9], C[9], S[9], D1[9], D2[9], D3[9], D4[9], E2[9], F[9];
    so3_right_jacobian(w, Jrw); m3_inverse(Jrw, Jrwi);
    so3_right_jacobian(da, Jra); so3_right_jacobian(db, Jrb);
    m3_tmul(Rbu, Rau, A); m3_mul(Jrwi, A, B); m3_mul(B, Jra, C);                      // :98-99 (negated below)
    so3_skew(d, S);
    // T_b_a_prior.rotation() * R_a^T = Rp^T Rau^T
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0.0; for (int k = 0; k < 3; k++) s += Tab[3 * k + i] * Rau[3 * j + k]; D1[3 * i + j] = s; }
    m3_mul(D1, S, D2); m3_mul(D2, Rau, D3); m3_mul(D3, Jra, D4);                     // :102-104
    m3_mul(Jrwi, Jrb, E2);                                                            // :118
    m3_mul(D1, Rbu, F);                                                               // :121
    double Jf[72];                                                                    // 6 x 12 before the weighting
    for (int i = 0; i < 72; i++) Jf[i] = 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Jf[i * 12 + j] = -C[3 * i + j];
            Jf[(3 + i) * 12 + j] = D4[3 * i + j];
            Jf[(3 + i) * 12 + 3 + j] = -Tab[3 * j + i];                               // -Rp^T  (:107)
            Jf[i * 12 + 6 + j] = E2[3 * i + j];
            Jf[(3 + i) * 12 + 9 + j] = F[3 * i + j];
        }
    for (int i = 0; i < 6; i++)
        for (int c = 0; c < 15; c++) {
            double s = 0.0;
            if (c < 12) for (int k = 0; k < 6; k++) s += W[6 * i + k] * Jf[k * 12 + c];
            J[i * 15 + c] = s;
        }
}

constexpr int POSE_TAB = 42;  // R[9] t[3] Jr[9] R0[9] dR[9] td[3]
__device__ __forceinline__ void pose_table_entry(const double* T0, const double* d6, double* tab) {
    double dR[9], Jr[9], R[9], t[3];
    so3_exp(d6, dR);
    so3_right_jacobian(d6, Jr);
    m3_mul(T0, dR, R);
    m3_vec(T0, d6 + 3, t);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        tab[i] = R[i];
        tab[12 + i] = Jr[i];
        tab[21 + i] = T0[i];
        tab[30 + i] = dR[i];
    }
    tab[9] = t[0] + T0[9]; tab[10] = t[1] + T0[10]; tab[11] = t[2] + T0[11];
    tab[39] = d6[3]; tab[40] = d6[4]; tab[41] = d6[5];
}

// PoseToLandmarkFactor (residuals.hpp:570-595) as a pseudo-observation evaluated from a pose table: two of its three
// rows (half 0: rows 0, 1; half 1: row 2 and a zero row) in the 2-row layout of the visual factors. WANT_J = false
// only needs R | t (the first 12 table entries).
template <bool WANT_J>
__device__ __forceinline__ void p2l_pseudo_obs(const double* tab, const double* q, const double* delta, const double* W, int half,
                                               double* r, double* Jp, double* Jl) {
    double Rq[3], e[3];
    m3_vec(tab, q, Rq);
    for (int a = 0; a < 3; a++) e[a] = Rq[a] + tab[9 + a] - delta[a];
    const int r0 = half ? 2 : 0;
    r[0] = W[3 * r0] * e[0] + W[3 * r0 + 1] * e[1] + W[3 * r0 + 2] * e[2];
    r[1] = half ? 0.0 : W[3] * e[0] + W[4] * e[1] + W[5] * e[2];
    if (WANT_J) {
        double Sq[9], A[9], B[9], C[9], WR0[9], WR[9];
        so3_skew(q, Sq);
        m3_mul(tab + 30, Sq, A);      // dR [q]x
        m3_mul(A, tab + 12, B);       // dR [q]x Jr(w)
        m3_mul(W, tab + 21, WR0);     // W R0
        m3_mul(WR0, B, C);
        m3_mul(W, tab, WR);           // W R
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Jp[j] = -C[3 * r0 + j]; Jp[3 + j] = WR0[3 * r0 + j]; Jl[j] = WR[3 * r0 + j];
            Jp[6 + j] = half ? 0.0 : -C[3 + j]; Jp[9 + j] = half ? 0.0 : WR0[3 + j]; Jl[3 + j] = half ? 0.0 : WR[3 + j];
        }
    }
}

// Pixel reprojection residual (+ Jacobians). cam = K[4] | Tsf[12]. Returns validity
// (Camera.cpp:128-136); invalid => r = 0, Jacobians kept (…Analytic.h:63-65).
template <bool WANT_J>
__device__ __forceinline__ bool pixel_factor(const double* tab, const double* K, const double* Tsf, const double* pw,
                                             double u_meas, double v_meas, double inv_sigma, double* r, double* Jp,
                                             double* Jl) {
    double pf[3], tc[3];
    m3_vec(tab, pw, pf);
    pf[0] += tab[9]; pf[1] += tab[10]; pf[2] += tab[11];
    m3_vec(Tsf, pf, tc);
    tc[0] += Tsf[9]; tc[1] += Tsf[10]; tc[2] += Tsf[11];
    double iz = 1.0 / tc[2];
    double u = (K[0] * tc[0] + K[2] * tc[2]) * iz;
    double v = (K[1] * tc[1] + K[3] * tc[2]) * iz;
    bool valid = !(tc[2] < 0.1) && !(u < 0 || v < 0 || u > 2 * K[2] || v > 2 * K[3]) && isfinite(u) && isfinite(v);
    r[0] = valid ? inv_sigma * (u - u_meas) : 0.0;
    r[1] = valid ? inv_sigma * (v - v_meas) : 0.0;
    if (WANT_J) {
        double a0 = inv_sigma * K[0] * iz, a1 = inv_sigma * K[1] * iz;
        double JhK[6] = {a0, 0.0, -a0 * tc[0] * iz, 0.0, a1, -a1 * tc[1] * iz};
        double A[6];  // JhK * Rsf
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A[j] = JhK[0] * Tsf[j] + JhK[2] * Tsf[6 + j];
            A[3 + j] = JhK[4] * Tsf[3 + j] + JhK[5] * Tsf[6 + j];
        }
        // Jl = A * R
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int j = 0; j < 3; j++) Jl[3 * q + j] = A[3 * q] * tab[j] + A[3 * q + 1] * tab[3 + j] + A[3 * q + 2] * tab[6 + j];
        // Jp[:,0:3] = -Jl [pw]x Jr(w);  Jp[:,3:6] = A * R0
        const double* Jr = tab + 12;
        const double* R0 = tab + 21;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            double l0 = Jl[3 * q], l1 = Jl[3 * q + 1], l2 = Jl[3 * q + 2];
            // row * skew(p) = (l1 p2 - l2 p1, l2 p0 - l0 p2, l0 p1 - l1 p0)
            double s0 = l1 * pw[2] - l2 * pw[1];
            double s1 = l2 * pw[0] - l0 * pw[2];
            double s2 = l0 * pw[1] - l1 * pw[0];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Jp[6 * q + j] = -(s0 * Jr[j] + s1 * Jr[3 + j] + s2 * Jr[6 + j]);
                Jp[6 * q + 3 + j] = A[3 * q] * R0[j] + A[3 * q + 1] * R0[3 + j] + A[3 * q + 2] * R0[6 + j];
            }
        }
    }
    return valid;
}

// Angular (bearing) residual (+ Jacobians), AngularAdjustmentCERESAnalytic.h:55-111.
// The reference evaluates Jr(log(exp(w))); log(exp(w)) = w for |w| < pi, so Jr(w) from the table is used.
template <bool WANT_J>
__device__ __forceinline__ void angular_factor(const double* tab, const double* Tsf, const double* pw,
                                               const double* b, double inv_sigma, double* r, double* Jp, double* Jl) {
    const double* R0 = tab + 21;
    const double* dR = tab + 30;
    // t_s = Tsf * T_f_w0 * dT * pw = Rsf (R pw + t) + tsf with R = R0 dR, t = R0 td + t0 (:62)
    double pf[3], ts[3];
    m3_vec(tab, pw, pf);
    pf[0] += tab[9]; pf[1] += tab[10]; pf[2] += tab[11];
    m3_vec(Tsf, pf, ts);
    ts[0] += Tsf[9]; ts[1] += Tsf[10]; ts[2] += Tsf[11];
    double nrm = v3_norm(ts);
    double inrm = 1.0 / nrm;
    double bs[3] = {ts[0] * inrm, ts[1] * inrm, ts[2] * inrm};
    // tangent basis (:66-77)
    double d[3] = {b[0] - 1, b[1], b[2]};
    double b1[3];
    if (v3_norm(d) > 1e-5) { b1[0] = 0; b1[1] = b[2]; b1[2] = -b[1]; }
    else { b1[0] = b[1]; b1[1] = -b[0]; b1[2] = 0; }
    double n1 = 1.0 / v3_norm(b1);
    b1[0] *= n1; b1[1] *= n1; b1[2] *= n1;
    double b2[3] = {b1[1] * b[2] - b1[2] * b[1], b1[2] * b[0] - b1[0] * b[2], b1[0] * b[1] - b1[1] * b[0]};
    double n2 = 1.0 / v3_norm(b2);
    b2[0] *= n2; b2[1] *= n2; b2[2] *= n2;
    double e[3] = {bs[0] - b[0], bs[1] - b[1], bs[2] - b[2]};
    r[0] = inv_sigma * (b1[0] * e[0] + b1[1] * e[1] + b1[2] * e[2]);
    r[1] = inv_sigma * (b2[0] * e[0] + b2[1] * e[1] + b2[2] * e[2]);
    if (WANT_J) {
        // Je = Pt (I - bs bs^T) Rsf R0 / |t_s|
        double Pt[6] = {b1[0], b1[1], b1[2], b2[0], b2[1], b2[2]};
        double PtM[6];
#pragma unroll
        for (int qq = 0; qq < 2; qq++) {
            double dot = Pt[3 * qq] * bs[0] + Pt[3 * qq + 1] * bs[1] + Pt[3 * qq + 2] * bs[2];
#pragma unroll
            for (int j = 0; j < 3; j++) PtM[3 * qq + j] = Pt[3 * qq + j] - dot * bs[j];
        }
        double Rsw0[9];
        m3_mul(Tsf, R0, Rsw0);
        double Je[6];
#pragma unroll
        for (int qq = 0; qq < 2; qq++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                Je[3 * qq + j] = inv_sigma * inrm *
                                 (PtM[3 * qq] * Rsw0[j] + PtM[3 * qq + 1] * Rsw0[3 + j] + PtM[3 * qq + 2] * Rsw0[6 + j]);
        const double* Jr = tab + 12;
        // Jl = Je dR ; Jp[:,0:3] = -Je dR [pw]x Jr = -Jl [pw]x Jr ; Jp[:,3:6] = Je
#pragma unroll
        for (int qq = 0; qq < 2; qq++) {
#pragma unroll
            for (int j = 0; j < 3; j++)
                Jl[3 * qq + j] = Je[3 * qq] * dR[j] + Je[3 * qq + 1] * dR[3 + j] + Je[3 * qq + 2] * dR[6 + j];
            double l0 = Jl[3 * qq], l1 = Jl[3 * qq + 1], l2 = Jl[3 * qq + 2];
            double s0 = l1 * pw[2] - l2 * pw[1];
            double s1 = l2 * pw[0] - l0 * pw[2];
            double s2 = l0 * pw[1] - l1 * pw[0];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Jp[6 * qq + j] = -(s0 * Jr[j] + s1 * Jr[3 + j] + s2 * Jr[6 + j]);
                Jp[6 * qq + 3 + j] = Je[3 * qq + j];
            }
        }
    }
}

// PosePriordx (residuals.hpp:601-632), sqrt_inf = diag(inf).
__device__ __forceinline__ void pose_prior_factor(const double* T0, const double* Tp, const double* inf,
                                                  const double* d6, double* r, double* J /*6x6 or nullptr*/) {
    double dR[9], R[9], t[3];
    so3_exp(d6, dR);
    m3_mul(T0, dR, R);
    m3_vec(T0, d6 + 3, t);
    t[0] += T0[9]; t[1] += T0[10]; t[2] += T0[11];
    // E = T * Tp^-1 with Eigen::Affine3d::inverse() semantics (general 3x3 inverse of the linear part,
    // residuals.hpp:610): R_E = R Rp^-1, t_E = t - R_E tp
    double Rpi[9], RE[9];
    m3_inverse(Tp, Rpi);
    m3_mul(R, Rpi, RE);
    double w[3], REtp[3];
    so3_log(RE, w);
    m3_vec(RE, Tp + 9, REtp);
    r[0] = inf[0] * w[0]; r[1] = inf[1] * w[1]; r[2] = inf[2] * w[2];
    r[3] = inf[3] * (t[0] - REtp[0]); r[4] = inf[4] * (t[1] - REtp[1]); r[5] = inf[5] * (t[2] - REtp[2]);
    if (J) {
        double Jrw[9], Jrwi[9], Jrd[9], A[9], B00[9], v[3], S[9], RS[9], B10[9], RRpT[9], wj[3];
        m3_mul_t(R, Tp, RRpT);  // the Jacobian uses R Rp^T (residuals.hpp:617)
        so3_log(RRpT, wj);
        so3_right_jacobian(wj, Jrw);
        m3_inverse(Jrw, Jrwi);
        so3_right_jacobian(d6, Jrd);
        m3_mul(Jrwi, Tp, A);
        m3_mul(A, Jrd, B00);
        m3_tvec(Tp, Tp + 9, v);
        so3_skew(v, S);
        m3_mul(R, S, RS);
        m3_mul(RS, Jrd, B10);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                J[6 * i + j] = inf[i] * B00[3 * i + j];
                J[6 * i + 3 + j] = 0.0;
                J[6 * (3 + i) + j] = inf[3 + i] * B10[3 * i + j];
                J[6 * (3 + i) + 3 + j] = inf[3 + i] * T0[3 * i + j];
            }
    }
}

// ---- sparse (NFR) prior factors, addMarginalizationResiduals sparse branch (…Analytic.cpp:363-426) ----
// Output layout for all four: r[rows], J[rows][15] row-major (columns beyond the factor's width untouched).

// IMUPriordx (residuals.hpp:649-695): params pose6 | dv3 | dba3 | dbg3; r = W e. The pose block of the Jacobian is
// W [J6; 0]; the v / ba / bg blocks are plain identities at rows 6 / 9 / 12, NOT whitened (as coded, :679-693).
__device__ __forceinline__ void imu_prior_factor_body(const double* T0, const double* v0, const double* ba0, const double* bg0,
                                                      const double* Tp, const double* vp, const double* bap, const double* bgp,
                                                      const double* W, const double* prm, double* r, double* J) {
    const double ones[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    double e[15], J6[36];
    pose_prior_factor(T0, Tp, ones, prm, e, J ? J6 : nullptr);
    for (int a = 0; a < 3; a++) {
        e[6 + a] = v0[a] + prm[6 + a] - vp[a];
        e[9 + a] = ba0[a] + prm[9 + a] - bap[a];
        e[12 + a] = bg0[a] + prm[12 + a] - bgp[a];
    }
    for (int i = 0; i < 15; i++) {
        double s = 0.0;
        for (int k = 0; k < 15; k++) s += W[i * 15 + k] * e[k];
        r[i] = s;
    }
    if (J) {
        for (int i = 0; i < 15; i++) {
            for (int a = 0; a < 6; a++) {
                double s = 0.0;
                for (int k = 0; k < 6; k++) s += W[i * 15 + k] * J6[k * 6 + a];
                J[i * 15 + a] = s;
            }
            for (int a = 6; a < 15; a++) J[i * 15 + a] = (i == a) ? 1.0 : 0.0;
        }
    }
}

__device__ __noinline__ void imu_prior_factor(const double* T0, const double* v0, const double* ba0, const double* bg0,
                                              const double* Tp, const double* vp, const double* bap, const double* bgp,
                                              const double* W, const double* prm, double* r, double* J) {
    imu_prior_factor_body(T0, v0, ba0, bg0, Tp, vp, bap, bgp, W, prm, r, J);
}

// PoseToLandmarkFactor (residuals.hpp:570-595): r = W (T_f_w (exp w, t) (p + dl) - delta); columns pose6 | lmk3.
__device__ __forceinline__ void pose_to_landmark_factor(const double* T0, const double* q /*p + dl*/, const double* delta,
                                                        const double* W, const double* d6, double* r, double* J) {
    double dR[9], R[9], t[3], Rq[3], e[3];
    so3_exp(d6, dR);
    m3_mul(T0, dR, R);
    m3_vec(T0, d6 + 3, t);
    m3_vec(R, q, Rq);
    for (int a = 0; a < 3; a++) e[a] = Rq[a] + t[a] + T0[9 + a] - delta[a];
    m3_vec(W, e, r);
    if (J) {
        double Sq[9], Jr[9], A[9], B[9], C[9], WR0[9], WR[9];
        so3_skew(q, Sq);
        so3_right_jacobian(d6, Jr);
        m3_mul(dR, Sq, A);
        m3_mul(A, Jr, B);
        m3_mul(W, T0, WR0);
        m3_mul(WR0, B, C);
        m3_mul(W, R, WR);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                J[i * 15 + j] = -C[3 * i + j];
                J[i * 15 + 3 + j] = WR0[3 * i + j];
                J[i * 15 + 6 + j] = WR[3 * i + j];
            }
    }
}

// IMUFactor (residuals.hpp:133-245). Blocks [pose_i 6 | pose_j 6 | dv_i 3 | dv_j 3 | dba_i 3 | dbg_i 3];
// J (optional) is the whitened 9x24 Jacobian, row-major. Quirks kept as coded: the pose_i translation block
// uses the UNperturbed R_i0 (:185), [6:9,0:3] of pose_i uses p_j instead of p_j - p_i (:181-184), pose_j's
// translation block uses exp(w_j)^T (:198).
// WHITEN = false leaves the UN-whitened 9x24 Jacobian in J (the caller multiplies by W with many threads); r is
// whitened in both cases.
// What the Jacobian blocks of the IMU factor read from the residual part: a local (registers) for the one-lane copies below, LDS for
// imu_pair_lin_wg (kernels.h), whose four waves produce and consume these values in phases.
struct ImuMid {
    double Ri[9], Rj[9], dRj[9], RiRjT[9], dR[9], r_dr[3], a[3], cc[3], tj[3], jb[3];
};

template <typename ImuT>
__device__ __forceinline__ void imu_residual_part(const ImuT& f, const double* Ti0, const double* Tj0, const double* vi0,
                                                  const double* vj0, const double* dpi, const double* dpj, const double* dvi,
                                                  const double* dvj, const double* dba, const double* dbg, double* r, ImuMid& m) {
    const double G[3] = {0.0, 0.0, -9.81};  // IMU.h:8
    double dRi[9], dRj[9], Ri[9], Rj[9], ti[3], tj[3];
    so3_exp(dpi, dRi); so3_exp(dpj, dRj);
    m3_mul(Ti0, dRi, Ri); m3_mul(Tj0, dRj, Rj);
    m3_vec(Ti0, dpi + 3, ti); m3_vec(Tj0, dpj + 3, tj);
    for (int k = 0; k < 3; k++) { ti[k] += Ti0[9 + k]; tj[k] += Tj0[9 + k]; }
    double vi[3], vj[3];
    for (int k = 0; k < 3; k++) { vi[k] = vi0[k] + dvi[k]; vj[k] = vj0[k] + dvj[k]; }
    const double dt = f.dt;
    // dR = (DeltaR exp(J_dR_bg dbg))^T R_i R_j^T  (:157-158)
    double jb[3], Eb[9], DRc[9], RiRjT[9], dR[9], r_dr[3];
    m3_vec(f.J_dR_bg, dbg, jb);
    so3_exp(jb, Eb);
    m3_mul(f.dR, Eb, DRc);
    m3_mul_t(Ri, Rj, RiRjT);
    m3_tmul(DRc, RiRjT, dR);
    so3_log(dR, r_dr);
    double a[3], Ra[3], t1[3], t2[3];
    for (int k = 0; k < 3; k++) a[k] = vj[k] - vi[k] - G[k] * dt;
    m3_vec(Ri, a, Ra);
    m3_vec(f.J_dv_bg, dbg, t1); m3_vec(f.J_dv_ba, dba, t2);
    double e[9];
    for (int k = 0; k < 3; k++) { e[k] = r_dr[k]; e[3 + k] = Ra[k] - (f.dv[k] + t1[k] + t2[k]); }
    // positions in world: p = -R^T t
    double pi[3], pj[3], b[3], Rb[3];
    // T.inverse().translation() with Eigen::Affine3d semantics: the linear part is inverted as a general 3x3
    {
        double Rii[9], Rji[9];
        m3_inverse(Ri, Rii); m3_inverse(Rj, Rji);
        m3_vec(Rii, ti, pi); m3_vec(Rji, tj, pj);
    }
    for (int k = 0; k < 3; k++) { pi[k] = -pi[k]; pj[k] = -pj[k]; }
    for (int k = 0; k < 3; k++) b[k] = pj[k] - pi[k] - vi[k] * dt - 0.5 * G[k] * dt * dt;
    m3_vec(Ri, b, Rb);
    m3_vec(f.J_dp_bg, dbg, t1); m3_vec(f.J_dp_ba, dba, t2);
    for (int k = 0; k < 3; k++) e[6 + k] = Rb[k] - (f.dp[k] + t1[k] + t2[k]);
    for (int q = 0; q < 9; q++) {
        double s = 0;
        for (int k = 0; k < 9; k++) s += f.W[9 * q + k] * e[k];
        r[q] = s;
    }
    for (int k = 0; k < 9; k++) { m.Ri[k] = Ri[k]; m.Rj[k] = Rj[k]; m.dRj[k] = dRj[k]; m.RiRjT[k] = RiRjT[k]; m.dR[k] = dR[k]; }
    for (int k = 0; k < 3; k++) {
        m.r_dr[k] = r_dr[k]; m.a[k] = a[k]; m.tj[k] = tj[k]; m.jb[k] = jb[k];
        m.cc[k] = pj[k] - vi[k] * dt - 0.5 * G[k] * dt * dt;
    }
}

// U <- the UN-whitened 9 x 24 Jacobian, block after block
template <typename ImuT>
__device__ __forceinline__ void imu_jacobian_part(const ImuT& f, const double* Ti0, const double* dpi, const double* dpj, const ImuMid& m,
                                                  double* U) {
    const double dt = f.dt;
    for (int i = 0; i < 9 * 24; i++) U[i] = 0.0;
    double Jr_ri[9], A[9], B[9], S[9], RS[9];
    {
        double Jr_r[9];
        so3_right_jacobian(m.r_dr, Jr_r);
        m3_inverse(Jr_r, Jr_ri);
    }
    // pose_i (:174-187)
    {
        double Jrwi[9];
        so3_right_jacobian(dpi, Jrwi);
        m3_mul(Jr_ri, m.Rj, A); m3_mul(A, Jrwi, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[i * 24 + j] = B[3 * i + j];
            so3_skew(m.a, S); m3_mul(m.Ri, S, RS); m3_mul(RS, Jrwi, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[(3 + i) * 24 + j] = -B[3 * i + j];
            so3_skew(m.cc, S); m3_mul(m.Ri, S, RS); m3_mul(RS, Jrwi, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { U[(6 + i) * 24 + j] = -B[3 * i + j]; U[(6 + i) * 24 + 3 + j] = Ti0[3 * i + j]; }
    }
    // pose_j (:190-200)
    {
        double Jrwj[9];
        so3_right_jacobian(dpj, Jrwj);
        m3_mul(Jr_ri, m.Rj, A); m3_mul(A, Jrwj, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[i * 24 + 6 + j] = -B[3 * i + j];
            double C1[9], C2[9];
        so3_skew(m.tj, S); m3_mul(m.RiRjT, S, C1); m3_mul(C1, m.Rj, C2); m3_mul(C2, Jrwj, B);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[(6 + i) * 24 + 6 + j] = -B[3 * i + j];
    }
    m3_mul_t(m.Ri, m.dRj, B);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[(6 + i) * 24 + 9 + j] = -B[3 * i + j];
    // dv_i, dv_j, dba, dbg (:203-237)
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        U[(3 + i) * 24 + 12 + j] = -m.Ri[3 * i + j];
        U[(6 + i) * 24 + 12 + j] = -m.Ri[3 * i + j] * dt;
        U[(3 + i) * 24 + 15 + j] = m.Ri[3 * i + j];
        U[(3 + i) * 24 + 18 + j] = -f.J_dv_ba[3 * i + j];
        U[(6 + i) * 24 + 18 + j] = -f.J_dp_ba[3 * i + j];
        U[(3 + i) * 24 + 21 + j] = -f.J_dv_bg[3 * i + j];
        U[(6 + i) * 24 + 21 + j] = -f.J_dp_bg[3 * i + j];
    }
    double Jrb[9], D1[9], D2[9];
    so3_right_jacobian(m.jb, Jrb);
    m3_mul_t(Jr_ri, m.dR, A);  // Jr^-1 dR^T
    m3_mul(A, Jrb, D1); m3_mul(D1, f.J_dR_bg, D2);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[i * 24 + 21 + j] = -D2[3 * i + j];
}

template <typename ImuT, bool WHITEN = true>
__device__ __forceinline__ void imu_factor_body(const ImuT& f, const double* Ti0, const double* Tj0, const double* vi0,
                                                const double* vj0, const double* dpi, const double* dpj, const double* dvi,
                                                const double* dvj, const double* dba, const double* dbg, double* r, double* J) {
    ImuMid m;
    imu_residual_part(f, Ti0, Tj0, vi0, vj0, dpi, dpj, dvi, dvj, dba, dbg, r, m);
    if (!J) return;
    double Uloc[WHITEN ? 9 * 24 : 1];  // un-whitened Jacobian
    double* U = WHITEN ? Uloc : J;
    imu_jacobian_part<ImuT>(f, Ti0, dpi, dpj, m, U);
    if (WHITEN)
        for (int q = 0; q < 9; q++)
            for (int c = 0; c < 24; c++) {
                double s = 0;
                for (int k = 0; k < 9; k++) s += f.W[9 * q + k] * U[k * 24 + c];
                J[q * 24 + c] = s;
            }
}

// Out-of-line copy for the big kernels (k_solve, k_marg_small: inlining it there costs more registers than the call);
// imu_pair_lin_wg / imu_pair_cost (kernels.h) inline the parts.
template <typename ImuT, bool WHITEN = true>
__device__ __noinline__ void imu_factor(const ImuT& f, const double* Ti0, const double* Tj0, const double* vi0,
                                        const double* vj0, const double* dpi, const double* dpj, const double* dvi,
                                        const double* dvj, const double* dba, const double* dbg, double* r, double* J) {
    imu_factor_body<ImuT, WHITEN>(f, Ti0, Tj0, vi0, vj0, dpi, dpj, dvi, dvj, dba, dbg, r, J);
}

// ---- linexd landmarks ------------------------------------------------------------------------
// ReprojectionErrCeres_linexd_dx (BundleAdjustmentCERESAnalytic.h:104-195), sigma = 1 (…Analytic.cpp:303). tab: pose table
// of the observing key-frame at its current delta. As coded, the line parameter enters the residual as a translation
// of the model points in the line frame by dline[0..2] (:121), while its Jacobian block is J_point [-R_w_l [pt]x | I]
// (:156-160). r4 | Jf 4x6 | Jl 4x6, row-major; an invalid projection has r = 0 and keeps its Jacobian.
__device__ __forceinline__ void line_pixel_factor(const double* tab, const double* K, const double* Tsf, const double* Twl,
                                                  const double* model, const double* uv4, const double* dline, double* r4,
                                                  double* Jf, double* Jl) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const double* pt = model + 3 * i;
        const double q[3] = {pt[0] + dline[0], pt[1] + dline[1], pt[2] + dline[2]};
        double pw[3], Jp[12], J3[6];
        m3_vec(Twl, q, pw);
        pw[0] += Twl[9]; pw[1] += Twl[10]; pw[2] += Twl[11];
        (void)pixel_factor<true>(tab, K, Tsf, pw, uv4[2 * i], uv4[2 * i + 1], 1.0, r4 + 2 * i, Jp, J3);
        if (!Jf) continue;
        double S[9], RS[9];
        so3_skew(pt, S);
        m3_mul(Twl, S, RS);
#pragma unroll
        for (int a = 0; a < 12; a++) Jf[12 * i + a] = Jp[a];
#pragma unroll
        for (int qq = 0; qq < 2; qq++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                Jl[(2 * i + qq) * 6 + a] = -(J3[3 * qq] * RS[a] + J3[3 * qq + 1] * RS[3 + a] + J3[3 * qq + 2] * RS[6 + a]);
                Jl[(2 * i + qq) * 6 + 3 + a] = J3[3 * qq + a];
            }
    }
}

// AngularErrCeres_linexd_dx (AngularAdjustmentCERESAnalytic.h:368-469), weight 1 / sigma^2 with sigma = 1
// (Angular….cpp:326-330). Residual 0: |n_obs x n_line| (parallelism of the two plane normals), residual 1: n_obs . b_line
// (line centre in the observed plane). The Jacobians are the ones the reference writes, with its helper definitions
// J_normalization(X) = (I - X X^T) / |X| on the un-normalised X and J_AcrossX(A) = -[A]x (utilities/geometry.h:327-343).
// r2 | Jf 2x6 | Jl 2x6 (Jf == nullptr: residual only).
__device__ __noinline__ void line_angular_factor(const double* T0, const double* Tsf, const double* Twl, const double* b6,
                                                 const double* dpose, const double* dline, double* r2, double* Jf, double* Jl) {
    double dR[9], dRl[9], Rsw[9], RswdR[9], C[9], Rsl[9];
    so3_exp(dpose, dR); so3_exp(dline, dRl);
    m3_mul(Tsf, T0, Rsw); m3_mul(Rsw, dR, RswdR); m3_mul(RswdR, Twl, C); m3_mul(C, dRl, Rsl);
    // t_s_l = Rsw (dR (R_w_l t_dl + t_w_l) + t_d) + R_s_f t_0 + t_s_f
    double u[3], v[3], t[3], w0[3];
    m3_vec(Twl, dline + 3, u);
    const double p[3] = {u[0] + Twl[9], u[1] + Twl[10], u[2] + Twl[11]};
    m3_vec(dR, p, v);
    v[0] += dpose[3]; v[1] += dpose[4]; v[2] += dpose[5];
    m3_vec(Rsw, v, t);
    m3_vec(Tsf, T0 + 9, w0);
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] += w0[i] + Tsf[9 + i];
    double n_obs[3] = {b6[1] * b6[5] - b6[2] * b6[4], b6[2] * b6[3] - b6[0] * b6[5], b6[0] * b6[4] - b6[1] * b6[3]};
    { const double inv = 1.0 / v3_norm(n_obs); n_obs[0] *= inv; n_obs[1] *= inv; n_obs[2] *= inv; }
    const double tn = v3_norm(t);
    const double bl[3] = {t[0] / tn, t[1] / tn, t[2] / tn};
    double dir[3] = {Rsl[0], Rsl[3], Rsl[6]};
    { const double dn = v3_norm(dir); dir[0] /= dn; dir[1] /= dn; dir[2] /= dn; }
    const double nl[3] = {bl[1] * dir[2] - bl[2] * dir[1], bl[2] * dir[0] - bl[0] * dir[2], bl[0] * dir[1] - bl[1] * dir[0]};
    const double nln = v3_norm(nl);
    const double nlh[3] = {nl[0] / nln, nl[1] / nln, nl[2] / nln};
    const double cx[3] = {n_obs[1] * nlh[2] - n_obs[2] * nlh[1], n_obs[2] * nlh[0] - n_obs[0] * nlh[2], n_obs[0] * nlh[1] - n_obs[1] * nlh[0]};
    const double cxn = v3_norm(cx);
    r2[0] = cxn;
    r2[1] = n_obs[0] * bl[0] + n_obs[1] * bl[1] + n_obs[2] * bl[2];
    if (!Jf) return;
    // e0 = (cx / |cx|)^T (-[n_obs]x) J_normalization(n_l);  e1 = n_obs^T J_normalization(t)
    double Sn[9], e0[3], e1[3], m[3];
    so3_skew(n_obs, Sn);
    // m = -(cx^T [n_obs]x) / |cx|
#pragma unroll
    for (int j = 0; j < 3; j++) m[j] = -(cx[0] * Sn[j] + cx[1] * Sn[3 + j] + cx[2] * Sn[6 + j]) / cxn;
    {
        const double mn = m[0] * nl[0] + m[1] * nl[1] + m[2] * nl[2];
#pragma unroll
        for (int j = 0; j < 3; j++) e0[j] = (m[j] - mn * nl[j]) / nln;
        const double nt = n_obs[0] * t[0] + n_obs[1] * t[1] + n_obs[2] * t[2];
#pragma unroll
        for (int j = 0; j < 3; j++) e1[j] = (n_obs[j] - nt * t[j]) / tn;
    }
    // row-0 weights: g = e0^T ( [R_s_l e_x]x^T J_normalization(t) J_t + [n_l / |n_l|]x J_R ) = a^T J_t + c^T J_R
    double a[3], c[3];
    {
        const double rx[3] = {Rsl[0], Rsl[3], Rsl[6]};
        // e0^T [rx]x^T = ([rx]x e0)^T
        const double f[3] = {rx[1] * e0[2] - rx[2] * e0[1], rx[2] * e0[0] - rx[0] * e0[2], rx[0] * e0[1] - rx[1] * e0[0]};
        const double ft = f[0] * t[0] + f[1] * t[1] + f[2] * t[2];
#pragma unroll
        for (int j = 0; j < 3; j++) a[j] = (f[j] - ft * t[j]) / tn;
        // e0^T [n]x = ([n]x^T e0)^T = (e0 x n)^T
        c[0] = e0[1] * nlh[2] - e0[2] * nlh[1]; c[1] = e0[2] * nlh[0] - e0[0] * nlh[2]; c[2] = e0[0] * nlh[1] - e0[1] * nlh[0];
    }
    // J_t, J_R blocks (3x3 each, the others are zero)
    double Jr_d[9], Jr_l[9], S1[9], S2[9], P1[9], P2[9], Jt_w[9], JR_w[9], JR_lw[9];
    // so3_rightJacobian(log_so3(dR)) as written (Angular….h:416-435): a line's rotation delta does leave |w| < pi during
    // a solve (the coded Jacobians are inexact and the line spins about its own axis), where log(exp(w)) != w
    { double lw[3]; so3_log(dR, lw); so3_right_jacobian(lw, Jr_d); so3_log(dRl, lw); so3_right_jacobian(lw, Jr_l); }
    so3_skew(u, S1); m3_mul(RswdR, S1, P1); m3_mul(P1, Jr_d, P2);
    so3_skew(Twl + 9, S2); m3_mul(RswdR, S2, P1);
#pragma unroll
    for (int i = 0; i < 9; i++) Jt_w[i] = -P2[i] - P1[i];
    {
        double Rwl_dRl[9];
        m3_mul(Twl, dRl, Rwl_dRl);
        const double ex[3] = {Rwl_dRl[0], Rwl_dRl[3], Rwl_dRl[6]};
        so3_skew(ex, S1); m3_mul(RswdR, S1, P1); m3_mul(P1, Jr_d, JR_w);
#pragma unroll
        for (int i = 0; i < 9; i++) JR_w[i] = -JR_w[i];
        const double e_x[3] = {1.0, 0.0, 0.0};
        so3_skew(e_x, S1); m3_mul(Rsl, S1, P2); m3_mul(P2, Jr_l, JR_lw);
#pragma unroll
        for (int i = 0; i < 9; i++) JR_lw[i] = -JR_lw[i];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        // key-frame block: J_t = [Jt_w | Rsw], J_R = [JR_w | 0]
        Jf[j] = a[0] * Jt_w[j] + a[1] * Jt_w[3 + j] + a[2] * Jt_w[6 + j] + c[0] * JR_w[j] + c[1] * JR_w[3 + j] + c[2] * JR_w[6 + j];
        Jf[3 + j] = a[0] * Rsw[j] + a[1] * Rsw[3 + j] + a[2] * Rsw[6 + j];
        Jf[6 + j] = e1[0] * Jt_w[j] + e1[1] * Jt_w[3 + j] + e1[2] * Jt_w[6 + j];
        Jf[9 + j] = e1[0] * Rsw[j] + e1[1] * Rsw[3 + j] + e1[2] * Rsw[6 + j];
        // line block: J_t = [0 | C], J_R = [JR_lw | 0]
        Jl[j] = c[0] * JR_lw[j] + c[1] * JR_lw[3 + j] + c[2] * JR_lw[6 + j];
        Jl[3 + j] = a[0] * C[j] + a[1] * C[3 + j] + a[2] * C[6 + j];
        Jl[6 + j] = 0.0;
        Jl[9 + j] = e1[0] * C[j] + e1[1] * C[3 + j] + e1[2] * C[6 + j];
    }
}

}  // namespace sadvio
