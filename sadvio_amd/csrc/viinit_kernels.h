// viinit_kernels.h — visual-inertial initialisation (AOptimizer::VIInit, AOptimizer.cpp:448-581) on the device.
//
// The problem is tiny (2 + 3 n_frames + 7 unknowns, one 9-row IMUFactorInit per consecutive key-frame pair) and runs
// once per session, so the whole Levenberg-Marquardt solve — factor evaluation, normal equations, damped Cholesky,
// step acceptance (the Ceres-2.2 rules of lm_decide / oracle/viinit.c) — is ONE launch of ONE workgroup: the normal
// matrix lives in LDS (packed lower triangle, the gradient riding along as an extra row so that the forward
// substitution comes with the factorisation), the per-factor Jacobians (9 x 15) in an HBM scratch that stays in L2.
// No host round trip per iteration (the reference: up to 50 ceres iterations).
#pragma once
#include "ba_types.h"
#include "device_math.h"

namespace sadvio {

constexpr int VIINIT_THREADS = 256;
constexpr int VIINIT_MAX_FRAMES = 48;
constexpr int VIINIT_FJ = 9 * 15 + 9;  // whitened Jacobian + residual of one factor

struct ViInitDev {
    int n_frames, n_factors, D, c_ba, c_bg, c_l, optim_bias, pad;
    double isig_ba, isig_bg;
    const double* T;     // [n_frames][12]
    const double* vel;   // [n_frames][3]
    const ImuDev* f;     // kf_i / kf_j index the frames
    const int* vcol;     // [n_frames] column of the frame's velocity delta, -1 = not in the program
    double* scratch;     // [2][n_factors][VIINIT_FJ]
    double* out;         // x[D] | initial_cost final_cost radius iterations termination n_success n_unsuccess
    SolveOpts o;
};

// IMUFactorInit::Evaluate (residuals.hpp:302-410); p15 = r_wi[2] dv_i[3] dv_j[3] dba[3] dbg[3] lambda. J (9 x 15) and r
// are whitened by W = L^T, L L^T = cov^-1 (computed once on the host). The scale column is kept as coded (without
// the exp(lambda) factor of the true derivative, :398-405).
template <bool WANT_J>
__device__ __noinline__ void imu_init_factor(const ImuDev& f, const double* Ti, const double* Tj, const double* vi0, const double* vj0,
                                             const double* p, double* r9, double* J) {
    const double gw[3] = {0.0, 0.0, -9.81};  // IMU.h:8
    const double w[3] = {p[0], p[1], 0.0};
    double Rwi[9], RiRwi[9];
    so3_exp(w, Rwi);
    m3_mul(Ti, Rwi, RiRwi);
    const double* dba = p + 8; const double* dbg = p + 11;
    const double dt = f.dt, es = exp(p[14]);
    double jb[3], E[9], DR[9], RiRjT[9], dR[9], e[9];
    m3_vec(f.J_dR_bg, dbg, jb);
    so3_exp(jb, E);
    m3_mul(f.dR, E, DR);
    m3_mul_t(Ti, Tj, RiRjT);
    m3_tmul(DR, RiRjT, dR);
    so3_log(dR, e);
    double av[3], ap[3], dpv[3];
    for (int a = 0; a < 3; a++) {
        const double pi = -(Ti[a] * Ti[9] + Ti[3 + a] * Ti[10] + Ti[6 + a] * Ti[11]);
        const double pj = -(Tj[a] * Tj[9] + Tj[3 + a] * Tj[10] + Tj[6 + a] * Tj[11]);
        const double vi = vi0[a] + p[2 + a], vj = vj0[a] + p[5 + a];
        dpv[a] = pj - pi;
        av[a] = (vj - vi) - gw[a] * dt;
        ap[a] = es * dpv[a] - vi * dt - 0.5 * gw[a] * dt * dt;
    }
    double t3[3], bv[3], bp[3], t[3];
    m3_vec(f.J_dv_bg, dbg, bv); m3_vec(f.J_dv_ba, dba, t);
    for (int a = 0; a < 3; a++) bv[a] += t[a] + f.dv[a];
    m3_vec(f.J_dp_bg, dbg, bp); m3_vec(f.J_dp_ba, dba, t);
    for (int a = 0; a < 3; a++) bp[a] += t[a] + f.dp[a];
    m3_vec(RiRwi, av, t3);
    for (int a = 0; a < 3; a++) e[3 + a] = t3[a] - bv[a];
    m3_vec(RiRwi, ap, t3);
    for (int a = 0; a < 3; a++) e[6 + a] = t3[a] - bp[a];
    for (int i = 0; i < 9; i++) {
        double s = 0.0;
        for (int k = i; k < 9; k++) s += f.W[i * 9 + k] * e[k];  // W is upper triangular
        r9[i] = s;
    }
    if (!WANT_J) return;
    // un-whitened Jacobian, column by column, whitened straight into J
    double Jrw[9], S[9], M[9], Nv[9], Np[9];
    so3_right_jacobian(w, Jrw);
    so3_skew(av, S); m3_mul(RiRwi, S, M); m3_mul(M, Jrw, Nv);
    so3_skew(ap, S); m3_mul(RiRwi, S, M); m3_mul(M, Jrw, Np);
    double Jre[9], Jrei[9], Jrb[9], A[9], B[9], C9[9];
    so3_right_jacobian(e, Jre);
    m3_inverse(Jre, Jrei);
    so3_right_jacobian(jb, Jrb);
    m3_mul_t(Jrei, dR, A); m3_mul(A, Jrb, B); m3_mul(B, f.J_dR_bg, C9);
    m3_vec(RiRwi, dpv, t3);
    for (int c = 0; c < 15; c++) {
        double u[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (c < 2) {
            for (int i = 0; i < 3; i++) { u[3 + i] = -Nv[3 * i + c]; u[6 + i] = -Np[3 * i + c]; }
        } else if (c < 5) {
            for (int i = 0; i < 3; i++) { u[3 + i] = -RiRwi[3 * i + c - 2]; u[6 + i] = -RiRwi[3 * i + c - 2] * dt; }
        } else if (c < 8) {
            for (int i = 0; i < 3; i++) u[3 + i] = RiRwi[3 * i + c - 5];
        } else if (c < 11) {
            for (int i = 0; i < 3; i++) { u[3 + i] = -f.J_dv_ba[3 * i + c - 8]; u[6 + i] = -f.J_dp_ba[3 * i + c - 8]; }
        } else if (c < 14) {
            for (int i = 0; i < 3; i++) { u[i] = -C9[3 * i + c - 11]; u[3 + i] = -f.J_dv_bg[3 * i + c - 11]; u[6 + i] = -f.J_dp_bg[3 * i + c - 11]; }
        } else {
            for (int i = 0; i < 3; i++) u[6 + i] = t3[i];
        }
        for (int i = 0; i < 9; i++) {
            double s = 0.0;
            for (int k = i; k < 9; k++) s += f.W[i * 9 + k] * u[k];
            J[i * 15 + c] = s;
        }
    }
}

struct ViLm {   // LM state shared by the workgroup (thread 0 writes)
    double radius, decrease_factor, x_cost, x_norm, initial_cost, gmax;
    int iter, n_invalid, n_success, n_unsuccess, termination, done, accepted, fail;
};

__device__ __forceinline__ int vi_tri(int i, int j) { return i * (i + 1) / 2 + j; }

__device__ __forceinline__ double vi_block_sum(double v, double* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int s = VIINIT_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ void vi_unpack(const ViInitDev& P, const ImuDev& f, const double* x, double* p15, int* col) {
    p15[0] = x[0]; p15[1] = x[1]; col[0] = 0; col[1] = 1;
    const int ci = P.vcol[f.kf_i], cj = P.vcol[f.kf_j];
    for (int a = 0; a < 3; a++) {
        col[2 + a] = ci + a; col[5 + a] = cj + a;
        p15[2 + a] = x[ci + a]; p15[5 + a] = x[cj + a];
        col[8 + a] = P.c_ba >= 0 ? P.c_ba + a : -1; col[11 + a] = P.c_bg >= 0 ? P.c_bg + a : -1;
        p15[8 + a] = P.c_ba >= 0 ? x[P.c_ba + a] : 0.0; p15[11 + a] = P.c_bg >= 0 ? x[P.c_bg + a] : 0.0;
    }
    col[14] = P.c_l; p15[14] = P.c_l >= 0 ? x[P.c_l] : 0.0;
}

// 1/2 sum r^2 at x; WANT_J also stores each factor's whitened J | r into `fj`.
template <bool WANT_J>
__device__ double vi_eval(const ViInitDev& P, const double* x, double* fj, double* red) {
    const int t = threadIdx.x;
    double c = 0.0;
    for (int fi = t; fi < P.n_factors; fi += VIINIT_THREADS) {
        const ImuDev& f = P.f[fi];
        double p15[15], r9[9], J[WANT_J ? 135 : 1];
        int col[15];
        vi_unpack(P, f, x, p15, col);
        imu_init_factor<WANT_J>(f, P.T + 12 * f.kf_i, P.T + 12 * f.kf_j, P.vel + 3 * f.kf_i, P.vel + 3 * f.kf_j, p15, r9, J);
        for (int i = 0; i < 9; i++) c += r9[i] * r9[i];
        if (WANT_J) {
            double* o = fj + (long long)fi * VIINIT_FJ;
            for (int i = 0; i < 135; i++) o[i] = J[i];
            for (int i = 0; i < 9; i++) o[135 + i] = r9[i];
        }
    }
    if (P.optim_bias && t == VIINIT_THREADS - 1)  // Landmark3DPrior(0, 0, I / sigma) on dba, dbg (residuals.hpp:512-522)
        for (int a = 0; a < 3; a++) {
            const double ra = P.isig_ba * x[P.c_ba + a], rg = P.isig_bg * x[P.c_bg + a];
            c += ra * ra + rg * rg;
        }
    if (WANT_J) __threadfence_block();
    return 0.5 * vi_block_sum(c, red);
}

__global__ void __launch_bounds__(VIINIT_THREADS) k_viinit(const ViInitDev* Pp) {
    extern __shared__ double lds[];
    __shared__ ViLm lm;
    __shared__ double red[VIINIT_THREADS];
    const ViInitDev P = *Pp;
    const SolveOpts o = P.o;
    const int t = threadIdx.x, D = P.D, nf = P.n_factors;
    double* A = lds;                                  // (D + 1)(D + 2) / 2: packed lower triangle + the rhs row
    double* g = A + (D + 1) * (D + 2) / 2;
    double* hd = g + D; double* sc = hd + D; double* x = sc + D; double* cand = x + D; double* delta = cand + D;
    for (int i = t; i < D; i += VIINIT_THREADS) { x[i] = 0.0; cand[i] = 0.0; delta[i] = 0.0; sc[i] = 1.0; }
    if (t == 0) {
        lm.radius = o.initial_radius; lm.decrease_factor = 2.0; lm.x_norm = 0.0; lm.iter = 0; lm.n_invalid = 0; lm.n_success = 0;
        lm.n_unsuccess = 0; lm.termination = 0; lm.done = 0; lm.accepted = 1; lm.fail = 0;
    }
    __syncthreads();
    int cur = 0;
    double x_cost = vi_eval<true>(P, x, P.scratch, red);
    if (t == 0) { lm.x_cost = x_cost; lm.initial_cost = x_cost; }
    bool first = true;
    while (true) {
        // ---- normal equations of the current linearisation: A = J^T J (lower), g = J^T r ----
        const double* fj = P.scratch + (long long)cur * nf * VIINIT_FJ;
        for (int i = t; i < (D + 1) * (D + 2) / 2; i += VIINIT_THREADS) A[i] = 0.0;
        for (int i = t; i < D; i += VIINIT_THREADS) g[i] = 0.0;
        __syncthreads();
        for (int fi = 0; fi < nf; fi++) {   // one factor at a time: its 15 x 15 block touches distinct entries
            const double* Jf = fj + (long long)fi * VIINIT_FJ;
            const ImuDev& f = P.f[fi];
            if (t < 240) {
                const int a = t < 225 ? t / 15 : t - 225, b = t < 225 ? t % 15 : -1;
                int ca, cb = 0;
                {
                    auto colof = [&](int q) {
                        if (q < 2) return q;
                        if (q < 5) return P.vcol[f.kf_i] + q - 2;
                        if (q < 8) return P.vcol[f.kf_j] + q - 5;
                        if (q < 11) return P.c_ba >= 0 ? P.c_ba + q - 8 : -1;
                        if (q < 14) return P.c_bg >= 0 ? P.c_bg + q - 11 : -1;
                        return P.c_l;
                    };
                    ca = colof(a);
                    if (b >= 0) cb = colof(b);
                }
                if (b >= 0) {
                    if (ca >= 0 && cb >= 0 && ca >= cb) {
                        double s = 0.0;
                        for (int k = 0; k < 9; k++) s += Jf[k * 15 + a] * Jf[k * 15 + b];
                        A[vi_tri(ca, cb)] += s;
                    }
                } else if (ca >= 0) {
                    double s = 0.0;
                    for (int k = 0; k < 9; k++) s += Jf[k * 15 + a] * Jf[135 + k];
                    g[ca] += s;
                }
            }
            __syncthreads();
        }
        if (P.optim_bias && t < 6) {
            const int c = (t < 3 ? P.c_ba : P.c_bg) + t % 3;
            const double is = t < 3 ? P.isig_ba : P.isig_bg;
            A[vi_tri(c, c)] += is * is;
            g[c] += is * is * x[c];
        }
        __syncthreads();
        for (int i = t; i < D; i += VIINIT_THREADS) { hd[i] = A[vi_tri(i, i)]; if (first && o.jacobi_scaling) sc[i] = 1.0 / (1.0 + sqrt(hd[i])); }
        double gm = 0.0;
        for (int i = t; i < D; i += VIINIT_THREADS) gm = fmax(gm, fabs(g[i]));
        red[t] = gm;
        __syncthreads();
        for (int s = VIINIT_THREADS / 2; s > 0; s >>= 1) { if (t < s) red[t] = fmax(red[t], red[t + s]); __syncthreads(); }
        if (t == 0) {   // FinalizeIterationAndCheckIfMinimizerCanContinue (after iteration 0 and after every step attempt)
            lm.gmax = red[0];
            if (lm.iter >= o.max_num_iterations) { lm.done = 1; lm.termination = 0; }
            else if (lm.gmax <= o.gradient_tolerance) { lm.done = 1; lm.termination = 3; }
            else if (lm.radius <= o.min_radius) { lm.done = 1; lm.termination = 4; }
        }
        first = false;
        __syncthreads();
        if (lm.done) break;
        // ---- damped system, Cholesky with the gradient as row D, back substitution ----
        for (int i = t; i < D; i += VIINIT_THREADS) {
            const double s2 = sc[i] * sc[i];
            A[vi_tri(i, i)] += fmin(fmax(s2 * hd[i], o.min_lm_diagonal), o.max_lm_diagonal) / lm.radius / s2;
            A[vi_tri(D, i)] = g[i];
        }
        if (t == 0) lm.fail = 0;
        __syncthreads();
        for (int j = 0; j < D; j++) {
            const double djj = A[vi_tri(j, j)];
            if (!(djj > 0.0) || !isfinite(djj)) { if (t == 0) lm.fail = 1; break; }   // uniform: every thread reads the same LDS word
            const double inv = 1.0 / sqrt(djj);
            __syncthreads();
            for (int i = j + t; i <= D; i += VIINIT_THREADS) A[vi_tri(i, j)] = (i == j) ? sqrt(djj) : A[vi_tri(i, j)] * inv;
            __syncthreads();
            // trailing update of the rows below: entry (i, k), j < k <= i <= D (row D = rhs: k < D only)
            const int m = D - j;   // rows j+1 .. D
            for (int q = t; q < m * (m + 1) / 2; q += VIINIT_THREADS) {
                int ii = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
                while ((ii + 1) * (ii + 2) / 2 <= q) ii++;
                while (ii * (ii + 1) / 2 > q) ii--;
                const int kk = q - ii * (ii + 1) / 2;
                const int i = j + 1 + ii, k = j + 1 + kk;
                if (!(i == D && k == D)) A[vi_tri(i, k)] -= A[vi_tri(i, j)] * A[vi_tri(k, j)];
            }
            __syncthreads();
        }
        __syncthreads();
        if (!lm.fail) {
            // L^T z = y (row D), delta = -z
            for (int j = D - 1; j >= 0; j--) {
                if (t == 0) A[vi_tri(D, j)] /= A[vi_tri(j, j)];
                __syncthreads();
                const double zj = A[vi_tri(D, j)];
                for (int i = t; i < j; i += VIINIT_THREADS) A[vi_tri(D, i)] -= A[vi_tri(j, i)] * zj;
                __syncthreads();
            }
            int bad = 0;
            for (int i = t; i < D; i += VIINIT_THREADS) {
                delta[i] = -A[vi_tri(D, i)];
                cand[i] = x[i] + delta[i];
                if (!isfinite(delta[i])) bad = 1;
            }
            if (bad) lm.fail = 1;
        }
        __syncthreads();
        // ---- model cost change, step / candidate norms, candidate cost ----
        double mcc = 0.0, sn2 = 0.0, cn2 = 0.0, cand_cost = 0.0;
        if (!lm.fail) {
            double m = 0.0;
            for (int fi = t; fi < nf; fi += VIINIT_THREADS) {
                const double* Jf = fj + (long long)fi * VIINIT_FJ;
                const ImuDev& f = P.f[fi];
                double d15[15];
                for (int a = 0; a < 3; a++) {
                    d15[2 + a] = delta[P.vcol[f.kf_i] + a]; d15[5 + a] = delta[P.vcol[f.kf_j] + a];
                    d15[8 + a] = P.c_ba >= 0 ? delta[P.c_ba + a] : 0.0; d15[11 + a] = P.c_bg >= 0 ? delta[P.c_bg + a] : 0.0;
                }
                d15[0] = delta[0]; d15[1] = delta[1]; d15[14] = P.c_l >= 0 ? delta[P.c_l] : 0.0;
                for (int k = 0; k < 9; k++) {
                    double jd = 0.0;
                    for (int a = 0; a < 15; a++) jd += Jf[k * 15 + a] * d15[a];
                    m -= jd * (Jf[135 + k] + 0.5 * jd);
                }
            }
            if (P.optim_bias && t == VIINIT_THREADS - 1)
                for (int a = 0; a < 3; a++) {
                    const double ja = P.isig_ba * delta[P.c_ba + a], jg = P.isig_bg * delta[P.c_bg + a];
                    m -= ja * (P.isig_ba * x[P.c_ba + a] + 0.5 * ja) + jg * (P.isig_bg * x[P.c_bg + a] + 0.5 * jg);
                }
            mcc = vi_block_sum(m, red);
            double a2 = 0.0, b2 = 0.0;
            for (int i = t; i < D; i += VIINIT_THREADS) { a2 += delta[i] * delta[i]; b2 += cand[i] * cand[i]; }
            sn2 = vi_block_sum(a2, red);
            cn2 = vi_block_sum(b2, red);
            if (mcc > 0.0) cand_cost = vi_eval<true>(P, cand, P.scratch + (long long)(1 - cur) * nf * VIINIT_FJ, red);
        }
        // ---- Ceres 2.2 TrustRegionMinimizer bookkeeping (oracle/viinit.c, lm_decide in kernels.h) ----
        if (t == 0) {
            lm.iter += 1;
            lm.accepted = 0;
            if (lm.fail || !(mcc > 0.0)) {
                lm.n_invalid += 1; lm.n_unsuccess += 1;
                if (lm.n_invalid >= o.max_num_consecutive_invalid_steps) { lm.done = 1; lm.termination = 5; }
                else lm.radius *= 0.5;
            } else {
                lm.n_invalid = 0;
                const double cost_change = lm.x_cost - cand_cost;
                if (sqrt(sn2) <= o.parameter_tolerance * (lm.x_norm + o.parameter_tolerance)) { lm.done = 1; lm.termination = 2; }
                else if (fabs(cost_change) <= o.function_tolerance * lm.x_cost) { lm.done = 1; lm.termination = 1; }
                else {
                    const double rel = cost_change / mcc;
                    if (rel > o.min_relative_decrease) {
                        lm.accepted = 1;
                        lm.x_norm = sqrt(cn2); lm.x_cost = cand_cost; lm.n_success += 1;
                        const double tt = 2.0 * rel - 1.0;
                        lm.radius = fmin(o.max_radius, lm.radius / fmax(1.0 / 3.0, 1.0 - tt * tt * tt));
                        lm.decrease_factor = 2.0;
                    } else {
                        lm.radius /= lm.decrease_factor; lm.decrease_factor *= 2.0; lm.n_unsuccess += 1;
                    }
                }
            }
        }
        __syncthreads();
        if (lm.accepted) {
            for (int i = t; i < D; i += VIINIT_THREADS) x[i] = cand[i];
            cur ^= 1;
        }
        __syncthreads();
        if (lm.done) break;
        // a rejected / invalid step re-enters with the same linearisation (fj[cur]) and the new radius; the matrix was
        // factorised in place, so it is rebuilt from the stored Jacobians either way
    }
    __syncthreads();
    for (int i = t; i < D; i += VIINIT_THREADS) P.out[i] = x[i];
    if (t == 0) {
        double* s = P.out + D;
        s[0] = lm.initial_cost; s[1] = lm.x_cost; s[2] = lm.radius; s[3] = lm.iter; s[4] = lm.termination; s[5] = lm.n_success; s[6] = lm.n_unsuccess;
    }
}

}  // namespace sadvio
