/*
 * oracle/so3.h — TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Plain-C restatement of the SO3/SE3 closed forms of the reference, with the SAME small-angle
 * thresholds (reference: cpp/include/utilities/geometry.h). Matrices are row-major double[9].
 */
#ifndef SADVIO_ORACLE_SO3_H
#define SADVIO_ORACLE_SO3_H
#include <math.h>
#include <string.h>

static inline void m3_set_identity(double *A) {
    memset(A, 0, 9 * sizeof(double));
    A[0] = A[4] = A[8] = 1.0;
}
static inline void m3_mul(const double *A, const double *B, double *C) { /* C = A*B, C may not alias */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static inline void m3_mul_t(const double *A, const double *B, double *C) { /* C = A*B^T */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
static inline void m3_tmul(const double *A, const double *B, double *C) { /* C = A^T*B */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
static inline void m3_vec(const double *A, const double *v, double *o) {
    double x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_tvec(const double *A, const double *v, double *o) { /* o = A^T v */
    double x = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double y = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double z = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_transpose(const double *A, double *T) {
    T[0] = A[0]; T[1] = A[3]; T[2] = A[6];
    T[3] = A[1]; T[4] = A[4]; T[5] = A[7];
    T[6] = A[2]; T[7] = A[5]; T[8] = A[8];
}
/* General 3x3 inverse by cofactors (the reference calls Eigen's .inverse(), a cofactor
 * expansion for fixed 3x3). Returns the determinant. */
static inline double m3_inverse(const double *A, double *I) {
    double c00 = A[4] * A[8] - A[5] * A[7];
    double c01 = A[5] * A[6] - A[3] * A[8];
    double c02 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    double id = 1.0 / det;
    I[0] = c00 * id;
    I[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c01 * id;
    I[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c02 * id;
    I[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return det;
}

/* geometry.h:17-23 */
static inline void so3_skew(const double *w, double *S) {
    S[0] = 0;     S[1] = -w[2]; S[2] = w[1];
    S[3] = w[2];  S[4] = 0;     S[5] = -w[0];
    S[6] = -w[1]; S[7] = w[0];  S[8] = 0;
}
static inline double v3_norm(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* geometry.h:30-37 — exactly I below 1e-5 */
static inline void so3_right_jacobian(const double *w, double *J) {
    double n = v3_norm(w);
    m3_set_identity(J);
    if (n < 1e-5) return;
    double S[9], S2[9];
    so3_skew(w, S);
    m3_mul(S, S, S2);
    double a = (1 - cos(n)) / (n * n);
    double b = (n - sin(n)) / (n * n * n);
    for (int i = 0; i < 9; i++) J[i] = J[i] - a * S[i] + b * S2[i];
}

/* geometry.h:131-147 — first order below 1e-9 */
static inline void so3_exp(const double *v, double *R) {
    double angle = v3_norm(v);
    double S[9];
    m3_set_identity(R);
    if (angle < 1e-9) {
        so3_skew(v, S);
        for (int i = 0; i < 9; i++) R[i] += S[i];
        return;
    }
    double axis[3] = {v[0] / angle, v[1] / angle, v[2] / angle};
    double S2[9];
    so3_skew(axis, S);
    m3_mul(S, S, S2);
    double c = 1. - cos(angle), s = sin(angle);
    for (int i = 0; i < 9; i++) R[i] += c * S2[i] + s * S[i];
}

/* geometry.h:149-166 — first order if |sin| < 1e-9 or angle < 1e-9 (i.e. also near pi) */
static inline void so3_log(const double *M, double *phi) {
    double cos_angle = 0.5 * (M[0] + M[4] + M[8]) - 0.5;
    cos_angle = fmin(fmax(cos_angle, -1.), 1.);
    double angle = acos(cos_angle);
    double v[3] = {M[7] - M[5], M[2] - M[6], M[3] - M[1]}; /* FromskewMatrix(M - M^T) */
    double k;
    if (fabs(sin(angle)) < 1e-9 || angle < 1e-9)
        k = 0.5;
    else
        k = 0.5 * angle / sin(angle);
    phi[0] = k * v[0]; phi[1] = k * v[1]; phi[2] = k * v[2];
}

/* Rigid transform helpers. T = R(9) | t(3). */
static inline void se3_mul(const double *A, const double *B, double *C) { /* C = A*B */
    double R[9], t[3];
    m3_mul(A, B, R);
    m3_vec(A, B + 9, t);
    for (int i = 0; i < 9; i++) C[i] = R[i];
    C[9] = t[0] + A[9]; C[10] = t[1] + A[10]; C[11] = t[2] + A[11];
}
static inline void se3_apply(const double *T, const double *p, double *o) {
    double q[3];
    m3_vec(T, p, q);
    o[0] = q[0] + T[9]; o[1] = q[1] + T[10]; o[2] = q[2] + T[11];
}
/* Eigen::Affine3d::inverse() (Mode = Affine) inverts the linear part as a GENERAL 3x3 matrix, not by
 * transposition; this matters for the reference's tests that feed 8-digit rotation literals
 * (imu_test.cpp:367-369) through 1000 inverse round trips. */
static inline void se3_inverse(const double *T, double *I) {
    double Rt[9], t[3];
    m3_inverse(T, Rt);
    m3_vec(Rt, T + 9, t);
    for (int i = 0; i < 9; i++) I[i] = Rt[i];
    I[9] = -t[0]; I[10] = -t[1]; I[11] = -t[2];
}
/* geometry.h:198-203: (exp(w), t) — NOT the SE3 exponential (parametersBlock.hpp:34-37) */
static inline void se3_from_delta6(const double *d, double *T) {
    so3_exp(d, T);
    T[9] = d[3]; T[10] = d[4]; T[11] = d[5];
}
#endif
