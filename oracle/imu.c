/*
 * oracle/imu.c — TEST INFRASTRUCTURE ONLY (CPU oracle). Not part of the product path.
 *
 * Restatement of the reference's on-manifold IMU pre-integration, the producer of the IMU-factor
 * constants (reference: cpp/src/data/sensors/IMU.cpp:5-91 processIMU, :104-108 biasDeltaCorrection).
 * Used to build synthetic VIO windows and to pin the oracle against imu_test.cpp's known answers.
 */
#include "factors.h"
#include "sadvio_oracle.h"

int oracle_imu_process(oracle_imu_state *cur, const oracle_imu_state *last, const double *kf_ba, const double *kf_bg,
                       double gyr_noise, double acc_noise, double rate_hz) {
    if (cur->ts_ns < last->ts_ns) return 0; /* IMU.cpp:12-14 wrong sync */
    /* bias propagation, :17-18 */
    for (int k = 0; k < 3; k++) { cur->ba[k] = last->ba[k]; cur->bg[k] = last->bg[k]; }
    double dt = (cur->ts_ns - last->ts_ns) * 1e-9; /* :21 */
    if (dt > 1) dt = 1 / rate_hz;                   /* :23-25 */
    double dt22 = 0.5 * dt * dt;
    double a[3], wv[3], wk[3], dv[3], dp[3], dR[9], Jrk[9];
    for (int k = 0; k < 3; k++) {
        a[k] = last->acc[k] - last->ba[k];
        dv[k] = a[k] * dt;
        dp[k] = a[k] * dt22;
        wv[k] = (last->gyr[k] - last->bg[k]) * dt;
        wk[k] = (last->gyr[k] - kf_bg[k]) * dt;
    }
    so3_exp(wv, dR);              /* :29 */
    so3_right_jacobian(wk, Jrk);  /* :30 */

    /* velocity and pose propagation, :33-41 */
    double Rwf[9];
    m3_transpose(last->T_f_w, Rwf);
    double Rdv[3], Rdp[3];
    m3_vec(Rwf, dv, Rdv);
    m3_vec(Rwf, dp, Rdp);
    for (int k = 0; k < 3; k++) cur->v[k] = last->v[k] + Rdv[k] + ORACLE_G[k] * dt;
    double Twf[12], Twf_new[12];
    se3_inverse(last->T_f_w, Twf);
    m3_mul(Rwf, dR, Twf_new);
    for (int k = 0; k < 3; k++) Twf_new[9 + k] = Twf[9 + k] + last->v[k] * dt + Rdp[k] + ORACLE_G[k] * dt22;
    se3_inverse(Twf_new, cur->T_f_w);

    /* B, :44-47 */
    double B[54];
    memset(B, 0, sizeof(B));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            B[i * 6 + j] = Jrk[3 * i + j] * dt;
            B[(3 + i) * 6 + 3 + j] = last->delta_R[3 * i + j] * dt;
            B[(6 + i) * 6 + 3 + j] = last->delta_R[3 * i + j] * dt22;
        }
    double eta[6];
    for (int k = 0; k < 3; k++) {
        eta[k] = gyr_noise * gyr_noise * rate_hz; /* IMU.h:39-41 */
        eta[3 + k] = acc_noise * acc_noise * rate_hz;
    }
    double BEBt[81];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += B[i * 6 + k] * eta[k] * B[j * 6 + k];
            BEBt[i * 9 + j] = s;
        }

    if (last->is_keyframe) { /* :50-61 restart */
        for (int k = 0; k < 3; k++) { cur->delta_p[k] = dp[k]; cur->delta_v[k] = dv[k]; }
        memcpy(cur->delta_R, dR, sizeof(dR));
        memcpy(cur->cov, BEBt, sizeof(BEBt));
        for (int k = 0; k < 3; k++) cur->cov[(6 + k) * 9 + 6 + k] += 0.0001 * dt;
        for (int i = 0; i < 9; i++) {
            double id = (i % 4 == 0) ? 1.0 : 0.0;
            cur->J_dR_bg[i] = -Jrk[i] * dt;
            cur->J_dv_ba[i] = -id * dt;
            cur->J_dv_bg[i] = 0.0;
            cur->J_dp_ba[i] = -dt22 * id;
            cur->J_dp_bg[i] = 0.0;
        }
    } else { /* :63-88 */
        double LR[9];
        memcpy(LR, last->delta_R, sizeof(LR));
        double LRdv[3], LRdp[3];
        m3_vec(LR, dv, LRdv);
        m3_vec(LR, dp, LRdp);
        m3_mul(LR, dR, cur->delta_R);
        for (int k = 0; k < 3; k++) {
            cur->delta_v[k] = last->delta_v[k] + LRdv[k];
            cur->delta_p[k] = last->delta_p[k] + last->delta_v[k] * dt + LRdp[k];
        }
        double ak[3] = {last->acc[0] - kf_ba[0], last->acc[1] - kf_ba[1], last->acc[2] - kf_ba[2]};
        double Sk[9], dR_dA[9];
        so3_skew(ak, Sk);
        m3_mul(LR, Sk, dR_dA);
        double A[81];
        memset(A, 0, sizeof(A));
        for (int i = 0; i < 9; i++) A[i * 9 + i] = 1.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                A[i * 9 + j] = dR[3 * j + i]; /* dR^T */
                A[(3 + i) * 9 + j] = -dR_dA[3 * i + j] * dt;
                A[(6 + i) * 9 + j] = -dR_dA[3 * i + j] * dt22;
                A[(6 + i) * 9 + 3 + j] = (i == j) ? dt : 0.0;
            }
        double AS[81], ASAt[81];
        mat_mul(A, last->cov, AS, 9, 9, 9);
        for (int i = 0; i < 9; i++)
            for (int j = 0; j < 9; j++) {
                double s = 0;
                for (int k = 0; k < 9; k++) s += AS[i * 9 + k] * A[j * 9 + k];
                ASAt[i * 9 + j] = s;
            }
        for (int i = 0; i < 81; i++) cur->cov[i] = ASAt[i] + BEBt[i];
        for (int k = 0; k < 3; k++) cur->cov[(6 + k) * 9 + 6 + k] += 0.0001 * dt;
        /* bias Jacobians, :83-87 */
        double dRt[9], t1[9], t2[9];
        m3_transpose(dR, dRt);
        m3_mul(dRt, last->J_dR_bg, t1);
        m3_mul(dR_dA, last->J_dR_bg, t2);
        for (int i = 0; i < 9; i++) {
            cur->J_dR_bg[i] = t1[i] - Jrk[i] * dt;
            cur->J_dv_ba[i] = last->J_dv_ba[i] - LR[i] * dt;
            cur->J_dv_bg[i] = last->J_dv_bg[i] - t2[i] * dt;
            cur->J_dp_ba[i] = last->J_dp_ba[i] + last->J_dv_ba[i] * dt - dt22 * LR[i];
            cur->J_dp_bg[i] = last->J_dp_bg[i] + last->J_dv_bg[i] * dt - dt22 * t2[i];
        }
    }
    return 1;
}

void oracle_imu_bias_correction(oracle_imu_state *s, const double *d_ba, const double *d_bg) {
    double t1[3], t2[3];
    m3_vec(s->J_dp_ba, d_ba, t1);
    m3_vec(s->J_dp_bg, d_bg, t2);
    for (int k = 0; k < 3; k++) s->delta_p[k] += t1[k] + t2[k];
    m3_vec(s->J_dv_ba, d_ba, t1);
    m3_vec(s->J_dv_bg, d_bg, t2);
    for (int k = 0; k < 3; k++) s->delta_v[k] += t1[k] + t2[k];
    double w[3], E[9], R[9];
    m3_vec(s->J_dR_bg, d_bg, w);
    so3_exp(w, E);
    m3_mul(s->delta_R, E, R);
    memcpy(s->delta_R, R, sizeof(R));
}
