// marg_kernels.h — K8: the dense marginalisation prior on the device (BundleAdjustmentCERESAnalytic::marginalize,
// …Analytic.cpp:431-663 -> Marginalization::{computeInformationAndGradient, computeSchurComplement,
// rankReveallingDecomposition, computeJacobiansAndResiduals}, marginalization.cpp:145-265,318-342,516-530).
//
//   A = sum J^T J, b = + sum J^T r over the blocks touching frame0, evaluated at zero deltas   (k_marg_*)
//   Amm^+ by eigen-decomposition (lambda > cut), Ak = Arr - Arm Amm^+ Arm^T, bk likewise        (k_jacobi_*, k_gemm)
//   Ak = U Lambda U^T, J = Lambda^1/2 U^T, r0 = -Lambda^-1/2 U^T bk                             (k_jacobi_*, k_marg_prior)
//
// Eigen-solver: one-sided (Hestenes) Jacobi on the rows of G = A (symmetric, so rows = columns) with the
// round-robin ordering: n/2 independent row pairs per launch, one workgroup per pair (3 dot products + the
// rotation of two rows of G and of V, all unit-stride); n - 1 launches per sweep. For a symmetric PSD matrix the
// rows of V converge to the eigenvectors and lambda_i = v_i . g_i. The prior (J^T J, J^T r0) is invariant to the
// eigenvector sign / order conventions, which is what the parity tests compare.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace sadvio {

constexpr int JAC_THREADS = 256;

__device__ __forceinline__ double block_sum_256(double v, double* sh /*[4]*/) {
    v = wave_sum(v);
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (ln == 0) sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// pair k of round-robin step s over npad (even) players
__device__ __forceinline__ void rr_pair(int npad, int s, int k, int& p, int& q) {
    const int r = npad - 1;
    if (k == 0) { p = r; q = s % r; }
    else { p = (s + k) % r; q = (s - k + r) % r; }
    if (p > q) { const int t = p; p = q; q = t; }
}

// floor2 (device scalar, written by k_jacobi_floor): squared column norm below which a column belongs to the numerical null
// space, (n eps)^2 max_p |g_p|^2. A pair of two such columns is noise against noise: the relative test can never settle
// and only keeps the sweeps going, while both eigenvalues sit at or below the cut of the pseudo-inverse / rank-revealing
// decomposition (marg_cut) and are dropped either way — such pairs are skipped.
__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_floor(const double* __restrict__ G, int n, unsigned long long* amax_bits, double* floor2, int finish) {
    __shared__ double sh[4];
    if (finish) {
        if (threadIdx.x == 0 && blockIdx.x == 0) { const double e = (double)n * 2.220446049250313e-16; *floor2 = e * e * __longlong_as_double((long long)*amax_bits); }
        return;
    }
    const double* g = G + (size_t)blockIdx.x * n;
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += JAC_THREADS) a += g[i] * g[i];
    a = block_sum_256(a, sh);
    if (threadIdx.x == 0) atomicMax(amax_bits, (unsigned long long)__double_as_longlong(a));   // non-negative doubles order like u64
}

__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_step(double* __restrict__ G, double* __restrict__ V, int n, int npad,
                                                             int s, double tol, int* rotated, const double* floor2) {
    int p, q;
    rr_pair(npad, s, blockIdx.x, p, q);
    if (q >= n) return;  // dummy player of an odd n
    __shared__ double sh[4];
    double* gp = G + (size_t)p * n;
    double* gq = G + (size_t)q * n;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += JAC_THREADS) { const double x = gp[i], y = gq[i]; a += x * x; b += y * y; c += x * y; }
    a = block_sum_256(a, sh); b = block_sum_256(b, sh); c = block_sum_256(c, sh);
    if (a == 0.0 || b == 0.0 || c * c <= tol * tol * a * b) return;
    if (a < *floor2 && b < *floor2) return;
    if (threadIdx.x == 0) atomicAdd(rotated, 1);   // rotations of this sweep (0 = converged)
    const double zeta = (b - a) / (2.0 * c);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
    double* vp = V + (size_t)p * n;
    double* vq = V + (size_t)q * n;
    for (int i = threadIdx.x; i < n; i += JAC_THREADS) {
        const double x = gp[i], y = gq[i];
        gp[i] = cs * x - sn * y; gq[i] = sn * x + cs * y;
        const double u = vp[i], w = vq[i];
        vp[i] = cs * u - sn * w; vq[i] = sn * u + cs * w;
    }
}

// ---- Cholesky-preconditioned one-sided Jacobi (Veselic / Hari / Drmac): A = P L L^T P^T by diagonal pivoting, then the
// rows of G = L^T are orthogonalised by Jacobi rotations: G -> Sigma U^T with A = U Sigma^2 U^T, so the rows ARE sqrt(lambda) u^T
// (no accumulation of rotations). On the rank-deficient, 10-decades-graded Schur complements of a marginalisation this takes
// 9-10 sweeps where Jacobi on A itself takes 30+ (profiled on the config-3 problem).

// Pivoted Cholesky in panels of PCH_NB columns (the LAPACK dpstrf scheme): the panel kernel (ONE workgroup - every step
// needs the arg max of the remaining diagonal) picks the pivot, applies the symmetric swap to the trailing matrix S and to
// the rows of G = L^T written so far, and forms row k of G from row k of S minus the contributions of the CURRENT panel's
// rows only (<= 31 terms); the rank-PCH_NB update of the whole trailing matrix is k_pchol_syrk on all CUs. (One workgroup
// doing the full left-looking dot products is bound by the L2 bandwidth of a single CU: 20 ms at n = 915 against 3.)
// S: n x n symmetric working copy (both triangles kept current), G: n x n row-major output, dg: remaining diagonal,
// piv: permutation, ctl[0] = rank once the factorisation has stopped (else -1), ctl[1] = tau (as double bits in dctl).
constexpr int PCH_THREADS = 1024;
constexpr int PCH_MAXN = 2048;
constexpr int PCH_NB = 32;
constexpr double PCH_THETA = 0.1;    // relaxed pivoting: a pivot is at least this fraction of the largest remaining diagonal
constexpr int PCH_STRICT_TAIL = 32;   // relaxed pivoting (k_pchol_panel_rx): the last indices are pivoted one arg max at a time
__global__ __launch_bounds__(PCH_THREADS) void k_pchol_panel(double* __restrict__ S, int n, double* __restrict__ G, int* __restrict__ piv,
                                                             double* __restrict__ dg, int* __restrict__ ctl, double* __restrict__ dctl, int k0, double tau_rel) {
    __shared__ double d[PCH_MAXN];
    __shared__ double lk[PCH_NB];         // G[c][k] of the current panel's rows
    __shared__ int pv[PCH_MAXN];
    __shared__ double wmax[PCH_THREADS / 64];
    __shared__ int widx[PCH_THREADS / 64];
    __shared__ int s_j;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    if (ctl[0] >= 0) return;              // the factorisation stopped in an earlier panel
    for (int i = tid; i < n; i += PCH_THREADS) {
        if (k0 == 0) { d[i] = S[(size_t)i * n + i]; pv[i] = i; } else { d[i] = dg[i]; pv[i] = piv[i]; }
    }
    __syncthreads();
    if (k0 == 0 && tid == 0) { double m = 0.0; for (int i = 0; i < n; i++) m = fmax(m, d[i]); dctl[0] = tau_rel >= 0.0 ? tau_rel * m : -tau_rel; }
    __syncthreads();
    const double tau = dctl[0];
    int rank = -1;
    const int k1 = min(n, k0 + PCH_NB);
    for (int k = k0; k < k1; k++) {
        double best = -1.0; int bi = k;
        for (int i = k + tid; i < n; i += PCH_THREADS) if (d[i] > best) { best = d[i]; bi = i; }
        const double wm = wave_max(best);
        const unsigned long long who = __ballot(best == wm);
        const int src = __ffsll((long long)who) - 1;
        const int wi = __builtin_amdgcn_readlane(bi, src);
        if (ln == 0) { wmax[wv] = wm; widx[wv] = wi; }
        __syncthreads();
        if (tid == 0) {
            double m = wmax[0]; int j = widx[0];
            for (int q = 1; q < PCH_THREADS / 64; q++) if (wmax[q] > m) { m = wmax[q]; j = widx[q]; }
            s_j = (m > tau && m > 0.0) ? j : -1;
        }
        __syncthreads();
        const int j = s_j;
        if (j < 0) { rank = k; break; }
        if (j != k) {
            // symmetric swap k <-> j of S (rows, then columns), the same column swap in the rows of G written so far
            for (int i = tid; i < n; i += PCH_THREADS) { const double a = S[(size_t)k * n + i], b = S[(size_t)j * n + i]; S[(size_t)k * n + i] = b; S[(size_t)j * n + i] = a; }
            __syncthreads();
            for (int i = tid; i < n; i += PCH_THREADS) { const double a = S[(size_t)i * n + k], b = S[(size_t)i * n + j]; S[(size_t)i * n + k] = b; S[(size_t)i * n + j] = a; }
            for (int c = tid; c < k; c += PCH_THREADS) { double* row = G + (size_t)c * n; const double a = row[k], b = row[j]; row[k] = b; row[j] = a; }
            if (tid == 0) { const double t = d[k]; d[k] = d[j]; d[j] = t; const int q = pv[k]; pv[k] = pv[j]; pv[j] = q; }
            __syncthreads();
        }
        if (tid < k - k0) lk[tid] = G[(size_t)(k0 + tid) * n + k];
        __syncthreads();
        const double lkk = sqrt(d[k]);
        const double inv = 1.0 / lkk;
        const int np = k - k0;
        double* rowk = G + (size_t)k * n;
        for (int i = tid; i < n; i += PCH_THREADS) {
            double v = 0.0;
            if (i > k) {
                double s0 = S[(size_t)k * n + i];
                for (int c = 0; c < np; c++) s0 -= G[(size_t)(k0 + c) * n + i] * lk[c];
                v = s0 * inv;
                d[i] -= v * v;
            } else if (i == k) v = lkk;
            rowk[i] = v;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += PCH_THREADS) { dg[i] = d[i]; piv[i] = pv[i]; }
    if (rank < 0 && k1 == n) rank = n;
    if (rank >= 0) {
        for (int c = rank; c < n; c++) for (int i = tid; i < n; i += PCH_THREADS) G[(size_t)c * n + i] = 0.0;
        if (tid == 0) ctl[0] = rank;
    }
}

// trailing update after a panel: S[i][j] -= sum_{c in panel} G[c][i] G[c][j] for i, j >= k1 (both triangles), 64 x 64 tiles
__global__ __launch_bounds__(256) void k_pchol_syrk(double* __restrict__ S, int n, const double* __restrict__ G, const int* __restrict__ ctl, int k0) {
    if (ctl[0] >= 0) return;
    const int k1 = k0 + PCH_NB;
    __shared__ double Ai[PCH_NB][64], Aj[PCH_NB][64];
    const int ti = k1 + blockIdx.y * 64, tj = k1 + blockIdx.x * 64;
    for (int e = threadIdx.x; e < PCH_NB * 64; e += 256) {
        const int c = e / 64, x = e % 64;
        Ai[c][x] = ti + x < n ? G[(size_t)(k0 + c) * n + ti + x] : 0.0;
        Aj[c][x] = tj + x < n ? G[(size_t)(k0 + c) * n + tj + x] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 4 x 4 micro-tile per thread
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
#pragma unroll 4
    for (int c = 0; c < PCH_NB; c++) {
        double vi[4], vj[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { vi[a] = Ai[c][ty * 4 + a]; vj[a] = Aj[c][tx * 4 + a]; }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] += vi[a] * vj[b];
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = ti + ty * 4 + a, j = tj + tx * 4 + b;
            if (i < n && j < n) S[(size_t)i * n + j] -= acc[a][b];
        }
}

// The same factorisation WITHOUT data movement (the version used): the permutation stays implicit - G is stored by ORIGINAL
// column index (A = G^T G needs no permutation at all), a pivoted index just drops out of the arg max and gets zeros in the
// later rows. Thread i owns index i (i + 1024 for n > 1024): its remaining diagonal, its "pivoted" flag and its entries of
// the current panel's rows live in registers, so a column costs two barriers, the broadcast of the pivot's panel entries
// through LDS and ONE global round trip (row j of S, known only after the arg max). The trailing update is k_pchol_syrk_full
// over the whole matrix (rows / columns of pivoted indices are updated too and never read again).
// done[i] = step at which index i was chosen (or -1); dg, ctl, dctl as above.
template <int EPT, int NB>
__global__ __launch_bounds__(PCH_THREADS) void k_pchol_panel_np(const double* __restrict__ S, int n, double* __restrict__ G, int* __restrict__ done_g,
                                                                double* __restrict__ dg, int* __restrict__ ctl, double* __restrict__ dctl, int k0, double tau_rel) {
    __shared__ double lk[NB];
    __shared__ double wmax[PCH_THREADS / 64];
    __shared__ int widx[PCH_THREADS / 64];
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    if (ctl[0] >= 0) return;              // the factorisation stopped in an earlier panel
    double d[EPT], g[NB][EPT];
    bool done[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = tid + e * PCH_THREADS;
        if (i < n) { d[e] = k0 == 0 ? S[(size_t)i * n + i] : dg[i]; done[e] = k0 == 0 ? false : done_g[i] >= 0; }
        else { d[e] = -1.0; done[e] = true; }
    }
    if (k0 == 0) {
        double m = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; e++) m = fmax(m, d[e]);
        m = wave_max(m);
        if (ln == 0) wmax[wv] = m;
        __syncthreads();
        if (tid == 0) { double mm = 0.0; for (int q = 0; q < PCH_THREADS / 64; q++) mm = fmax(mm, wmax[q]); const double t0 = tau_rel >= 0.0 ? tau_rel * mm : -tau_rel; dctl[0] = t0; lk[0] = t0; }
        __syncthreads();
    }
    const double tau = k0 == 0 ? lk[0] : dctl[0];
    __syncthreads();
    int rank = -1;
#pragma unroll
    for (int kk = 0; kk < NB; kk++) {
        const int k = k0 + kk;
        if (k >= n) break;
        double best = -1.0; int bi = 0;
#pragma unroll
        for (int e = 0; e < EPT; e++) if (!done[e] && d[e] > best) { best = d[e]; bi = tid + e * PCH_THREADS; }
        const double wm = wave_max(best);
        const unsigned long long who = __ballot(best == wm);
        const int src = __ffsll((long long)who) - 1;
        const int wi = __builtin_amdgcn_readlane(bi, src);
        if (ln == 0) { wmax[wv] = wm; widx[wv] = wi; }
        __syncthreads();
        double m = wmax[0]; int j = widx[0];
#pragma unroll
        for (int q = 1; q < PCH_THREADS / 64; q++) if (wmax[q] > m) { m = wmax[q]; j = widx[q]; }
        if (!(m > tau && m > 0.0)) { rank = k; break; }
        double srow[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) { const int i = tid + e * PCH_THREADS; srow[e] = i < n ? S[(size_t)j * n + i] : 0.0; }
        if (tid == (j & (PCH_THREADS - 1))) {
#pragma unroll
            for (int e = 0; e < EPT; e++) if (j == tid + e * PCH_THREADS) {
#pragma unroll
                for (int c = 0; c < kk; c++) lk[c] = g[c][e];
            }
        }
        __syncthreads();
        const double lkk = sqrt(m), inv = 1.0 / lkk;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = tid + e * PCH_THREADS;
            double v = 0.0;
            if (i == j) { v = lkk; done[e] = true; done_g[i] = k; }
            else if (!done[e]) {
                double s0 = srow[e];
#pragma unroll
                for (int c = 0; c < kk; c++) s0 -= g[c][e] * lk[c];
                v = s0 * inv;
                d[e] -= v * v;
            }
            g[kk][e] = v;
            if (i < n) G[(size_t)k * n + i] = v;
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; e++) { const int i = tid + e * PCH_THREADS; if (i < n) dg[i] = d[e]; }
    if (rank < 0 && k0 + NB >= n) rank = n;
    if (rank >= 0) {
        for (int c = rank; c < n; c++) for (int i = tid; i < n; i += PCH_THREADS) G[(size_t)c * n + i] = 0.0;
        if (tid == 0) ctl[0] = rank;
    }
}

// ---- relaxed pivoting (round 4): a whole panel's pivots are chosen at once ----------------------------------------------------------
// k_pchol_panel_np spends 1.6 us per column, most of it waiting for ONE global round trip: row j of S can only be requested once the
// arg max of the remaining diagonal is known. Any index whose remaining diagonal is safely above its own rounding noise is a valid
// pivot of a positive semi-definite matrix (the factorisation is backward stable in every order; the order only decides WHICH
// dependent indices are left over at the end), so here a panel picks its NB pivots up front - the two (one for n > 1024) largest
// "safe" remaining diagonals of every wave, sorted - requests their NB rows of S together, and then runs the NB columns with one
// barrier each: the owner of pivot c publishes its remaining diagonal and its entries of the panel's earlier rows, every thread
// finishes its entry of row c from registers. A candidate whose diagonal has meanwhile dropped below the safe bound (it depends on
// the pivots taken before it in this panel) is skipped and competes again in the next panel. "Safe": d_i > tau,
// d_i >= PCH_THETA max_j d_j (threshold pivoting) and d_i >= safe_rel * a_ii (a_ii = the original diagonal, safe_rel ~ 1e3 n eps). When no safe index is left but some diagonal is still
// above tau - and always for the last PCH_STRICT_TAIL indices - the panel falls back to strict pivoting (arg max per column, as
// k_pchol_panel_np): the rank decision is taken by the same rule, in the same greedy order, as before.
// The panel's first row is no longer known to the host: ctl[1] = rows written so far, ctl[2] = first row of the last panel (what
// the trailing update reads); rows [ctl[1], ctl[2] + NB) are zeroed so that the update can always contract NB rows.
// dg: [0, n) remaining diagonal, [n, 2n) original diagonal.
template <int EPT, int NB>
__global__ __launch_bounds__(PCH_THREADS) void k_pchol_panel_rx(const double* __restrict__ S, int n, double* __restrict__ G, int* __restrict__ done_g,
                                                                double* __restrict__ dg, int* __restrict__ ctl, double* __restrict__ dctl, int first, double tau_rel,
                                                                double safe_rel) {
    constexpr int NW = PCH_THREADS / 64;
    __shared__ double lk[2][NB];
    __shared__ double pd[2][2];
    __shared__ double cval[NB];
    __shared__ int cidx[NB], ord[NB];
    __shared__ double wmax[NW];
    __shared__ int widx[NW], wcnt[NW];
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    if (ctl[0] >= 0) return;              // the factorisation stopped in an earlier panel
    const int k0 = first ? 0 : ctl[1];
    double d[EPT], a0[EPT], g[NB][EPT];
    bool done[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = tid + e * PCH_THREADS;
        if (i < n) {
            d[e] = first ? S[(size_t)i * n + i] : dg[i];
            a0[e] = first ? d[e] : dg[n + i];
            done[e] = first ? false : done_g[i] >= 0;
        } else { d[e] = -1.0; a0[e] = 0.0; done[e] = true; }
    }
    if (first) {
        double m = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; e++) m = fmax(m, d[e]);
        m = wave_max(m);
        if (ln == 0) wmax[wv] = m;
        __syncthreads();
        if (tid == 0) { double mm = 0.0; for (int q = 0; q < NW; q++) mm = fmax(mm, wmax[q]); const double t0 = tau_rel >= 0.0 ? tau_rel * mm : -tau_rel; dctl[0] = t0; pd[0][0] = t0; }
        __syncthreads();
    }
    const double tau = first ? pd[0][0] : dctl[0];
    __syncthreads();
    // ---- the arg max over ALL remaining indices (the stop test, the threshold, the strict panel's first pivot) ----
    {
        double ball = -1.0; int iall = 0, lw = 0;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            if (!done[e] && d[e] > ball) { ball = d[e]; iall = tid + e * PCH_THREADS; }
            lw += __popcll(__ballot(!done[e]));
        }
        const double wm = wave_max(ball);
        const int src = __ffsll((long long)__ballot(ball == wm)) - 1;
        const int wi = __builtin_amdgcn_readlane(iall, src);
        if (ln == 0) { wmax[wv] = wm; widx[wv] = wi; wcnt[wv] = lw; }
        if (tid < NB) cval[tid] = -1.0;
    }
    __syncthreads();
    double m_all = wmax[0]; int j_all = widx[0], left = 0;
#pragma unroll
    for (int q = 1; q < NW; q++) if (wmax[q] > m_all) { m_all = wmax[q]; j_all = widx[q]; }
#pragma unroll
    for (int q = 0; q < NW; q++) left += wcnt[q];
    // ---- candidates: the largest remaining diagonals of every wave that pass the threshold test d_i >= theta max_j d_j (threshold
    //      pivoting: the growth of the factor, hence the accuracy of the pivots that follow, stays within 1 / theta of the greedy order's)
    //      and the safety test against their own rounding noise ----
    const double thr = PCH_THETA * m_all;
    {
        bool taken[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) taken[e] = false;
        // slot r * nwa + wv: round r of wave wv, nwa = waves that own indices at all (a small system gives its few waves more rounds)
        const int nwa = EPT > 1 ? NW : min(NW, (n + 63) >> 6);
        const int pw = (NB + nwa - 1) / nwa;
        for (int r = 0; r < pw; r++) {
            const int slot = r * nwa + wv;
            double best = -1.0; int bi = 0, be = 0;
#pragma unroll
            for (int e = 0; e < EPT; e++)
                if (!done[e] && !taken[e] && d[e] > tau && d[e] >= thr && d[e] >= safe_rel * a0[e] && d[e] > best) { best = d[e]; bi = tid + e * PCH_THREADS; be = e; }
            const double cm = wave_max(best);
            const int cs = __ffsll((long long)__ballot(best == cm)) - 1;
            const int ci = __builtin_amdgcn_readlane(bi, cs);
            if (ln == cs && cm > 0.0) {
#pragma unroll
                for (int e = 0; e < EPT; e++) if (e == be) taken[e] = true;
            }
            if (ln == 0 && wv < nwa && slot < NB) { cval[slot] = cm > 0.0 ? cm : -1.0; cidx[slot] = ci; }
        }
    }
    __syncthreads();
    int nc = 0;
#pragma unroll
    for (int q = 0; q < NB; q++) nc += cval[q] > 0.0 ? 1 : 0;
    // The END of the factorisation stays strictly pivoted: which index is left over when a null direction finally shows decides how
    // well its (zero) pivot is computed - the greedy order keeps the eliminated block well conditioned, an arbitrary one need not.
    nc = min(nc, max(left - PCH_STRICT_TAIL, 0));
    if (tid < NB) {
        const double v = cval[tid];
        int rk = 0;
#pragma unroll
        for (int q = 0; q < NB; q++) { const double u = cval[q]; rk += (u > v || (u == v && q < tid)) ? 1 : 0; }
        ord[rk] = tid;        // descending by value; the invalid candidates (-1) come last
    }
    __syncthreads();
    int rank = -1, cnt = 0;
    if (!(m_all > tau && m_all > 0.0)) rank = k0;
    else if (nc > 0) {
        // ---- fast panel ----
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int j = c < nc ? cidx[ord[c]] : -1;
#pragma unroll
            for (int e = 0; e < EPT; e++) { const int i = tid + e * PCH_THREADS; g[c][e] = (j >= 0 && i < n) ? S[(size_t)j * n + i] : 0.0; }
        }
#pragma unroll
        for (int c = 0; c < NB; c++) {
            if (c < nc) {      // (uniform)
            const int j = cidx[ord[c]], p = c & 1;
            if (tid == (j & (PCH_THREADS - 1))) {
#pragma unroll
                for (int e = 0; e < EPT; e++) if (j == tid + e * PCH_THREADS) {
                    pd[p][0] = d[e]; pd[p][1] = a0[e];
#pragma unroll
                    for (int q = 0; q < c; q++) lk[p][q] = g[q][e];
                }
            }
            __syncthreads();
            const double dj = pd[p][0];
            if (dj > tau && dj >= thr && dj >= safe_rel * pd[p][1]) {
                const double lkk = sqrt(dj), inv = 1.0 / lkk;
                const int k = k0 + cnt;
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const int i = tid + e * PCH_THREADS;
                    double v = 0.0;
                    if (i == j) { v = lkk; done[e] = true; done_g[i] = k; }
                    else if (!done[e]) {
                        double s0 = g[c][e];
#pragma unroll
                        for (int q = 0; q < c; q++) s0 -= g[q][e] * lk[p][q];
                        v = s0 * inv;
                        d[e] -= v * v;
                    }
                    g[c][e] = v;
                    if (i < n) G[(size_t)k * n + i] = v;
                }
                cnt++;
            } else {
#pragma unroll
                for (int e = 0; e < EPT; e++) g[c][e] = 0.0;
            }
            }
        }
    } else {
        // ---- strict panel: arg max of the remaining diagonal per column ----
        bool active = true;     // (uniform)
#pragma unroll
        for (int kk = 0; kk < NB; kk++) {
            const int k = k0 + kk;
            if (k >= n) active = false;
            double m = m_all; int j = j_all;
            if (active && kk > 0) {
                double best = -1.0; int bi = 0;
#pragma unroll
                for (int e = 0; e < EPT; e++) if (!done[e] && d[e] > best) { best = d[e]; bi = tid + e * PCH_THREADS; }
                const double wm = wave_max(best);
                const int src = __ffsll((long long)__ballot(best == wm)) - 1;
                const int wi = __builtin_amdgcn_readlane(bi, src);
                __syncthreads();      // (the reads of wmax / lk of the column before)
                if (ln == 0) { wmax[wv] = wm; widx[wv] = wi; }
                __syncthreads();
                m = wmax[0]; j = widx[0];
#pragma unroll
                for (int q = 1; q < NW; q++) if (wmax[q] > m) { m = wmax[q]; j = widx[q]; }
                if (!(m > tau && m > 0.0)) { rank = k; active = false; }
            }
            if (active) {
            double srow[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) { const int i = tid + e * PCH_THREADS; srow[e] = i < n ? S[(size_t)j * n + i] : 0.0; }
            if (tid == (j & (PCH_THREADS - 1))) {
#pragma unroll
                for (int e = 0; e < EPT; e++) if (j == tid + e * PCH_THREADS) {
#pragma unroll
                    for (int q = 0; q < kk; q++) lk[0][q] = g[q][e];
                }
            }
            __syncthreads();
            const double lkk = sqrt(m), inv = 1.0 / lkk;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = tid + e * PCH_THREADS;
                double v = 0.0;
                if (i == j) { v = lkk; done[e] = true; done_g[i] = k; }
                else if (!done[e]) {
                    double s0 = srow[e];
#pragma unroll
                    for (int q = 0; q < kk; q++) s0 -= g[q][e] * lk[0][q];
                    v = s0 * inv;
                    d[e] -= v * v;
                }
                g[kk][e] = v;
                if (i < n) G[(size_t)k * n + i] = v;
            }
            cnt++;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; e++) { const int i = tid + e * PCH_THREADS; if (i < n) { dg[i] = d[e]; if (first) dg[n + i] = a0[e]; } }
    const int k1 = k0 + cnt;
    if (rank < 0 && k1 >= n) rank = n;
    if (rank >= 0) {
        for (int c = rank; c < n; c++) for (int i = tid; i < n; i += PCH_THREADS) G[(size_t)c * n + i] = 0.0;
        if (tid == 0) ctl[0] = rank;
    } else {
        for (int c = k1; c < k0 + NB && c < n; c++) for (int i = tid; i < n; i += PCH_THREADS) G[(size_t)c * n + i] = 0.0;
        if (tid == 0) { ctl[1] = k1; ctl[2] = k0; ctl[3] += 1; if (nc == 0) ctl[4] += 1; ctl[5] += nc - cnt > 0 ? nc - cnt : 0; }   // [3..5]: panels, strict panels, skipped candidates (SADVIO_DEBUG & 16384)
    }
}

// S[i][j] -= sum_{c in panel} G[c][i] G[c][j] over the WHOLE matrix (implicit pivoting: nothing is ordered), 64 x 64 tiles
template <int NB>
__global__ __launch_bounds__(256) void k_pchol_syrk_full(double* __restrict__ S, int n, const double* __restrict__ G, const int* __restrict__ ctl, int k0) {
    if (ctl[0] >= 0) return;
    __shared__ double Ai[NB][64], Aj[NB][64];
    const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
    for (int e = threadIdx.x; e < NB * 64; e += 256) {
        const int c = e / 64, x = e % 64;
        Ai[c][x] = ti + x < n ? G[(size_t)(k0 + c) * n + ti + x] : 0.0;
        Aj[c][x] = tj + x < n ? G[(size_t)(k0 + c) * n + tj + x] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 4 x 4 micro-tile per thread
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
#pragma unroll 4
    for (int c = 0; c < NB; c++) {
        double vi[4], vj[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { vi[a] = Ai[c][ty * 4 + a]; vj[a] = Aj[c][tx * 4 + a]; }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] += vi[a] * vj[b];
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = ti + ty * 4 + a, j = tj + tx * 4 + b;
            if (i < n && j < n) S[(size_t)i * n + j] -= acc[a][b];
        }
}

// Block one-sided Jacobi on the r rows (length n) of G: one launch = one round-robin step over BLOCKS of JB rows, one
// workgroup per block pair. The workgroup holds its 2 JB rows in registers, forms their Gram matrix (2 JB x 2 JB) with one
// block reduction, diagonalises it in LDS (cyclic two-sided Jacobi by one wave, rotations accumulated in U), and replaces the
// rows by U^T rows: afterwards the 2 JB rows are mutually orthogonal. (r / JB - 1) launches per sweep instead of r - 1; a
// resident-grid version with one launch per solve was measured and dropped: a barrier or a flag across the 8 XCDs costs
// 60 - 90 us, an order of magnitude more than a launch boundary.)
constexpr int JB = 4;            // rows per block
constexpr int JB2 = 2 * JB;
constexpr int JB_INNER = 1;     // inner sweeps of a block's 2 JB x 2 JB diagonalisation
template <int EPT>               // elements per thread and row: n <= 256 * EPT
__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_block(double* __restrict__ G, int r, int n, int nbpad, int s, double tol, int* rotated) {
    __shared__ double part[JAC_THREADS / 64][JB2 * (JB2 + 1) / 2];
    __shared__ double M[JB2][JB2 + 1], U[JB2][JB2 + 1];
    __shared__ double rc[JB], rs[JB];
    __shared__ int s_work;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    int bp, bq;
    rr_pair(nbpad, s, blockIdx.x, bp, bq);
    int rows[JB2];
#pragma unroll
    for (int k = 0; k < JB; k++) { rows[k] = bp * JB + k; rows[JB + k] = bq * JB + k; }
    double x[JB2][EPT];
#pragma unroll
    for (int k = 0; k < JB2; k++)
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = tid + e * JAC_THREADS;
            x[k][e] = (rows[k] < r && i < n) ? G[(size_t)rows[k] * n + i] : 0.0;
        }
    // Gram matrix of the 2 JB rows
    {
        int idx = 0;
#pragma unroll
        for (int a = 0; a < JB2; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) {
                double v = 0.0;
#pragma unroll
                for (int e = 0; e < EPT; e++) v += x[a][e] * x[b][e];
                v = wave_sum(v);
                if (ln == 0) part[wv][idx] = v;
                idx++;
            }
    }
    __syncthreads();
    if (tid < JB2 * (JB2 + 1) / 2) {
        int a = 0, b = tid;
        while (b >= a + 1) { b -= a + 1; a++; }
        const double v = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        M[a][b] = v; M[b][a] = v;
    }
    if (tid < JB2 * JB2) U[tid / JB2][tid % JB2] = (tid / JB2 == tid % JB2) ? 1.0 : 0.0;
    __syncthreads();
    if (tid == 0) {
        int work = 0;
        for (int a = 0; a < JB2; a++) for (int b = 0; b < a; b++) if (M[a][a] != 0.0 && M[b][b] != 0.0 && M[a][b] * M[a][b] > tol * tol * M[a][a] * M[b][b]) work = 1;
        s_work = work;
        if (work) atomicAdd(rotated, 1);
    }
    __syncthreads();
    if (!s_work) return;
    // cyclic two-sided Jacobi on M by the first wave: JB disjoint rotations per step (round-robin over 2 JB indices)
    if (wv == 0) {
        for (int sw = 0; sw < JB_INNER; sw++) {   // an inexact inner diagonalisation could only cost outer sweeps (measured: none, even with one inner sweep)
            bool any = false;
            for (int st = 0; st < JB2 - 1; st++) {
                if (ln < JB) {
                    int p, q;
                    rr_pair(JB2, st, ln, p, q);
                    const double app = M[p][p], aqq = M[q][q], apq = M[p][q];
                    double c = 1.0, sn = 0.0;
                    if (apq != 0.0 && apq * apq > 1e-32 * fabs(app * aqq)) {
                        const double zeta = (aqq - app) / (2.0 * apq);
                        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        c = 1.0 / sqrt(1.0 + t * t); sn = c * t;
                    }
                    rc[ln] = c; rs[ln] = sn;
                }
                wave_lds_fence();
                bool rot = false;
                for (int k = 0; k < JB; k++) rot |= rs[k] != 0.0;
                any |= rot;
                if (rot) {
                    // rows: M <- J^T M ; then columns: M <- M J ; U <- U J   (lane = (pair k, column j))
                    const bool act = ln < JB * JB2;
                    const int k = act ? ln / JB2 : 0, j = ln % JB2;
                    int p, q;
                    rr_pair(JB2, st, k, p, q);
                    const double c = rc[k], sn = rs[k];
                    const double mp = M[p][j], mq = M[q][j];     // a lane reads and writes its own two entries: no fence between
                    if (act) { M[p][j] = c * mp - sn * mq; M[q][j] = sn * mp + c * mq; }
                    wave_lds_fence();
                    const double cp = M[j][p], cq = M[j][q], up = U[j][p], uq = U[j][q];
                    if (act) {
                        M[j][p] = c * cp - sn * cq; M[j][q] = sn * cp + c * cq;
                        U[j][p] = c * up - sn * uq; U[j][q] = sn * up + c * uq;
                    }
                    wave_lds_fence();
                }
            }
            if (!any) break;
        }
    }
    __syncthreads();
    // rows <- U^T rows  (new row a = sum_b U[b][a] old row b)
    double u[JB2][JB2];
#pragma unroll
    for (int a = 0; a < JB2; a++)
#pragma unroll
        for (int b = 0; b < JB2; b++) u[a][b] = U[b][a];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = tid + e * JAC_THREADS;
        if (i >= n) continue;
#pragma unroll
        for (int a = 0; a < JB2; a++) {
            if (rows[a] >= r) continue;
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < JB2; b++) v += u[a][b] * x[b][e];
            G[(size_t)rows[a] * n + i] = v;
        }
    }
}

// The same step on BLOCKS of JM = 8 rows (16 rows per workgroup, n <= JM_MAXN) with the Gram matrix and the row update on the
// FP64 matrix cores: half the launches per sweep. The 16 rows are read ONCE, coalesced, into an LDS image X[16][ldx]
// (ldx = 2 mod 32: the operand reads below are conflict-free per half wave); every wave owns a quarter of the 16-column
// groups for the load, the Gram matrix and the update, so only the Gram reduction and U cross waves. Gram: lane (row = l % 16,
// k = l / 16) feeds the same register as A and B operand of v_mfma_f64_16x16x4_f64 (X X^T contracts over columns); update
// X <- U^T X: one 16 x 16 x 16 product per 16-column tile (A = U^T, B = X[4 sl + l / 16][16 t + l % 16]), stored from the
// accumulators as full 128-byte row segments.
constexpr int JM = 8, JM2 = 16, JM_MAXN = 1024;
__host__ __device__ constexpr int jm_ldx(int n) { return ((n + 15) / 16 * 16 + 29) / 32 * 32 + 2; }
// Jacobi rotation (c, s) annihilating apq of [[app, apq], [apq, aqq]]: t = sign(d) b / (|d| + sqrt(d^2 + b^2)) with d = aqq - app,
// b = 2 apq (the smaller root of t^2 + 2 zeta t - 1); c = h w, s = sign(d) b w with h = |d| + sqrt(d^2 + b^2), w = rsqrt(h^2 + b^2):
// no division, two v_rsq_f64 + Newton (c^2 + s^2 = 1 to rounding); d and b are pre-scaled by a power of two.
__device__ __forceinline__ void jm_rotation(double app, double aqq, double apq, double& c, double& s) {
    c = 1.0; s = 0.0;
    if (apq == 0.0 || !(apq * apq > 1e-32 * fabs(app * aqq))) return;
    double d = aqq - app, b = 2.0 * apq;
    const int e = -ilogb(fmax(fabs(d), fabs(b)));
    d = ldexp(d, e); b = ldexp(b, e);
    const double n2 = d * d + b * b;
    const double h = fabs(d) + n2 * rsqrt_nr(n2);
    const double w = rsqrt_nr(h * h + b * b);
    c = h * w;
    s = (d >= 0.0 ? b : -b) * w;
}
__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_mma(double* __restrict__ G, int r, int n, int ldx, int nbpad, int s, double tol, int* rotated, long long* ts) {
#ifdef SADVIO_KERNEL_TS
#define JM_TS(i_) do { if (ts && blockIdx.x == 1 && threadIdx.x == 0) ts[i_] = wall_clock64(); } while (0)
#else
#define JM_TS(i_) do { } while (0)
#endif
    typedef double d4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char jm_smem[];
    double* X = (double*)jm_smem;
    __shared__ double part[JAC_THREADS / 64][JM2][JM2 + 1];
    __shared__ double M[JM2][JM2 + 1], U[JM2][JM2 + 1];
    __shared__ int s_work;
    JM_TS(0);
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    const int ri = ln & 15, kq = ln >> 4;
    int bp, bq;
    rr_pair(nbpad, s, blockIdx.x, bp, bq);
    const bool full = s == 0;
    const int ngrp = (n + 15) >> 4;
    const int g0 = wv * ngrp / (JAC_THREADS / 64), g1 = (wv + 1) * ngrp / (JAC_THREADS / 64);   // <= 16 groups per wave
    const int c0 = 16 * g0, c1 = min(16 * g1, n), c1p = 16 * g1;
    // rows -> LDS (this wave's columns; zeros for absent rows and up to the end of the last group) and their Gram matrix, in
    // chunks of 64 columns: all global loads are issued first, chunk j is written to the image and contracted while the
    // later chunks are still in flight
    {
        double v[4][JM2];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int row = 0; row < JM2; row++) {
                const int grow = row < JM ? bp * JM + row : bq * JM + row - JM;
                const int c = c0 + 64 * j + ln;
                v[j][row] = (grow < r && c < c1) ? G[(size_t)grow * n + c] : 0.0;
            }
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double* xr = X + ri * ldx + kq;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gb = g0 + 4 * j;
            if (gb >= g1) break;
            const int c = c0 + 64 * j + ln;
            if (c < c1p) {
#pragma unroll
                for (int row = 0; row < JM2; row++) X[row * ldx + c] = v[j][row];
            }
            wave_lds_fence();
            double a[4][4];
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                const int g = min(gb + tt, g1 - 1);
#pragma unroll
                for (int e = 0; e < 4; e++) { const double x = xr[16 * g + 4 * e]; a[tt][e] = gb + tt < g1 ? x : 0.0; }
            }
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt][0], a[tt][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt][1], a[tt][1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt][2], a[tt][2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt][3], a[tt][3], acc1, 0, 0, 0);
            }
        }
        JM_TS(1);
#pragma unroll
        for (int vv = 0; vv < 4; vv++) part[wv][kq + 4 * vv][ri] = acc0[vv] + acc1[vv];
    }
    JM_TS(2);
    __syncthreads();
    JM_TS(3);
    {
        const int i = tid >> 4, j = tid & 15;
        M[i][j] = part[0][i][j] + part[1][i][j] + part[2][i][j] + part[3][i][j];
        U[i][j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    JM_TS(4);
    if (wv == 0) {
        bool w = false;
        {
            // pairs checked / rotated: all 120 in the first round of a sweep (full), afterwards only the 64 pairs ACROSS the two
            // blocks - every block is in exactly one pair of round 0, so each within-block pair is still visited once per sweep
            const int a = ln >> 2, b0 = (ln & 3) * 4;
            const double maa = M[a][a];
            double mbb[4], mab[4];
#pragma unroll
            for (int e = 0; e < 4; e++) { mbb[e] = M[b0 + e][b0 + e]; mab[e] = M[a][b0 + e]; }
#pragma unroll
            for (int e = 0; e < 4; e++) w |= b0 + e < a && (full || (a >= JM && b0 + e < JM)) && maa != 0.0 && mbb[e] != 0.0 && mab[e] * mab[e] > tol * tol * maa * mbb[e];
        }
        const bool work = __ballot(w) != 0ull;
        if (ln == 0) { s_work = work; if (work) atomicAdd(rotated, 1); }
        JM_TS(5);
        if (work) {
            // one sweep of the cyclic two-sided Jacobi on M: JM disjoint rotations per step (an inexact inner diagonalisation
            // costs no outer sweeps, measured on the 4-row version). Lane (k1, k2) owns the 2 x 2 block (pair k1 rows x pair k2
            // columns) of M and a 2 x 2 block of U; it derives both rotations itself from the pairs' diagonal blocks
            // (broadcast reads), so a step is ONE LDS round trip: all reads, the math, all writes, one fence.
            const int k1 = ln >> 3, k2 = ln & 7;
            const int nsteps = full ? JM2 - 1 : JM;
            for (int st = 0; st < nsteps; st++) {
                int p1, q1, p2, q2;
                if (full) { rr_pair(JM2, st, k1, p1, q1); rr_pair(JM2, st, k2, p2, q2); }
                else { p1 = k1; q1 = JM + ((k1 + st) & (JM - 1)); p2 = k2; q2 = JM + ((k2 + st) & (JM - 1)); }
                const double a1 = M[p1][p1], b1 = M[q1][q1], x1 = M[p1][q1];
                const double a2 = M[p2][p2], b2 = M[q2][q2], x2 = M[p2][q2];
                const double B00 = M[p1][p2], B01 = M[p1][q2], B10 = M[q1][p2], B11 = M[q1][q2];
                const double u00 = U[2 * k1][p2], u01 = U[2 * k1][q2], u10 = U[2 * k1 + 1][p2], u11 = U[2 * k1 + 1][q2];
                double c1, s1, c2, s2;
                jm_rotation(a1, b1, x1, c1, s1);
                jm_rotation(a2, b2, x2, c2, s2);
                if (__ballot(s1 != 0.0) == 0ull) continue;     // every pair of this step is already orthogonal
                const double t00 = c1 * B00 - s1 * B10, t01 = c1 * B01 - s1 * B11, t10 = s1 * B00 + c1 * B10, t11 = s1 * B01 + c1 * B11;
                M[p1][p2] = c2 * t00 - s2 * t01; M[p1][q2] = s2 * t00 + c2 * t01;
                M[q1][p2] = c2 * t10 - s2 * t11; M[q1][q2] = s2 * t10 + c2 * t11;
                U[2 * k1][p2] = c2 * u00 - s2 * u01; U[2 * k1][q2] = s2 * u00 + c2 * u01;
                U[2 * k1 + 1][p2] = c2 * u10 - s2 * u11; U[2 * k1 + 1][q2] = s2 * u10 + c2 * u11;
                wave_lds_fence();
            }
        }
    }
    JM_TS(6);
    __syncthreads();
    if (!s_work) return;
    // rows <- U^T rows: D[a][c] = sum_b U[b][a] X[b][c]
    double ua[4];
#pragma unroll
    for (int sl = 0; sl < 4; sl++) ua[sl] = U[4 * sl + kq][ri];
    int orow[4];
#pragma unroll
    for (int v = 0; v < 4; v++) { const int a = kq + 4 * v; orow[v] = a < JM ? bp * JM + a : bq * JM + a - JM; }
    double x[4][4], xn[4][4];
    auto load_x = [&](int gb, double (&dst)[4][4]) {
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int g = min(gb + tt, g1 - 1);
#pragma unroll
            for (int sl = 0; sl < 4; sl++) dst[tt][sl] = X[(4 * sl + kq) * ldx + 16 * g + ri];
        }
    };
    load_x(g0, x);
    for (int gb = g0; gb < g1; gb += 4) {              // 4 tiles at a time: independent accumulators; the next 4 tiles' operands are read ahead
        if (gb + 4 < g1) load_x(gb + 4, xn);
        d4 acc[4];
#pragma unroll
        for (int tt = 0; tt < 4; tt++) acc[tt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sl = 0; sl < 4; sl++)
#pragma unroll
            for (int tt = 0; tt < 4; tt++) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[sl], x[tt][sl], acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int c = 16 * (gb + tt) + ri;
            if (gb + tt < g1 && c < n) {
#pragma unroll
                for (int v = 0; v < 4; v++) if (orow[v] < r) G[(size_t)orow[v] * n + c] = acc[tt][v];
            }
        }
#pragma unroll
        for (int tt = 0; tt < 4; tt++)
#pragma unroll
            for (int sl = 0; sl < 4; sl++) x[tt][sl] = xn[tt][sl];
    }
    JM_TS(7);
}

// eigen-pairs from the orthogonalised rows: lambda_i = |g_i|^2, v_i = g_i / |g_i| scattered back through the pivoting
// (rows >= rank: lambda = 0, v = 0 - they are below every cut and only ever multiplied by zero)
__global__ __launch_bounds__(JAC_THREADS) void k_eig_from_rows(const double* __restrict__ G, const int* __restrict__ piv, const int* __restrict__ rank, int n,
                                                               double* __restrict__ V, double* __restrict__ ev) {
    __shared__ double sh[4];
    const int i = blockIdx.x;
    const double* g = G + (size_t)i * n;
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += JAC_THREADS) s += g[k] * g[k];
    s = block_sum_256(s, sh);
    const bool live = i < *rank && s > 0.0;
    const double inv = live ? 1.0 / sqrt(s) : 0.0;
    for (int k = threadIdx.x; k < n; k += JAC_THREADS) V[(size_t)i * n + (piv ? piv[k] : k)] = live ? g[k] * inv : 0.0;
    if (threadIdx.x == 0) ev[i] = live ? s : 0.0;
}

// G = sym(A block), V = I
__global__ void k_jacobi_init(const double* __restrict__ A, long long lda, int n, double* G, double* V, int lower_only) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx - (long long)i * n);
    const double aij = A[(size_t)i * lda + j], aji = A[(size_t)j * lda + i];
    G[idx] = lower_only ? (i >= j ? aij : aji) : 0.5 * (aij + aji);
    V[idx] = i == j ? 1.0 : 0.0;
}

// lambda_i = v_i . g_i (one workgroup per row)
__global__ __launch_bounds__(JAC_THREADS) void k_jacobi_eigenvalues(const double* G, const double* V, int n, double* ev) {
    __shared__ double sh[4];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += JAC_THREADS) s += G[(size_t)i * n + k] * V[(size_t)i * n + k];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) ev[i] = s;
}

// C[i][j] = beta C[i][j] + alpha sum_k A(i,k) B(k,j) with arbitrary strides (covers every transpose); 16x16 tiles.
__global__ __launch_bounds__(256) void k_gemm(double* C, long long ldc, const double* A, long long sai, long long sak,
                                              const double* B, long long sbk, long long sbj, int M, int N, int K, double alpha,
                                              double beta) {
    __shared__ double As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int ka = k0 + tx, kb = k0 + ty;
        As[ty][tx] = (i < M && ka < K) ? A[(size_t)i * sai + (size_t)ka * sak] : 0.0;
        Bs[ty][tx] = (kb < K && j < N) ? B[(size_t)kb * sbk + (size_t)j * sbj] : 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) acc += As[ty][k] * Bs[k][tx];
        __syncthreads();
    }
    if (i < M && j < N) C[(size_t)i * ldc + j] = (beta == 0.0 ? 0.0 : beta * C[(size_t)i * ldc + j]) + alpha * acc;
}

// ---- FP64 matrix-core GEMM (round 4) -----------------------------------------------------------------------------------
// One 64 x 64 tile of C = alpha op(A) op(B) + beta C per 256-thread workgroup, K streamed through LDS in chunks of 16:
// wave w owns rows [16 w, 16 w + 16) of the tile, four v_mfma_f64_16x16x4_f64 accumulators across its 64 columns. Arbitrary
// element strides cover every transpose (the staging loop picks the lane order that is unit-stride in memory); [kbeg, kend)
// restricts the contraction (triangular operands). Used by the Schur complement of the marginalisation and by the
// triangular inverse below — the contractions of K8 that ARE dense ("GEMM Arm Amm+ Arm^T (MFMA-eligible)").
__device__ __forceinline__ void mgemm_tile(double* __restrict__ C, long long ldc, const double* __restrict__ A, long long sai, long long sak,
                                           const double* __restrict__ B, long long sbk, long long sbj, int M, int N, int kbeg, int kend,
                                           double alpha, double beta, int ti, int tj, double (*As)[17], double (*Bs)[17]) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63, lr = ln & 15, lk = ln >> 4;
    d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int idx = tid + 256 * e;
            int r, c;
            if (sak == 1) { r = idx >> 4; c = idx & 15; } else { c = idx >> 6; r = idx & 63; }
            As[r][c] = (ti + r < M && k0 + c < kend) ? A[(long long)(ti + r) * sai + (long long)(k0 + c) * sak] : 0.0;
            int j, k;
            if (sbj == 1) { k = idx >> 6; j = idx & 63; } else { j = idx >> 4; k = idx & 15; }
            Bs[j][k] = (tj + j < N && k0 + k < kend) ? B[(long long)(k0 + k) * sbk + (long long)(tj + j) * sbj] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double a = As[16 * wv + lr][4 * u + lk];
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[16 * q + lr][4 * u + lk], acc[q], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = ti + 16 * wv + lk + 4 * rg, col = tj + 16 * q + lr;
            if (row < M && col < N) {
                double* c = C + (long long)row * ldc + col;
                *c = (beta == 0.0 ? 0.0 : beta * *c) + alpha * acc[q][rg];
            }
        }
}

// C[i][j] = beta C[i][j] + alpha sum_k A(i,k) B(k,j), same calling convention as k_gemm; grid (ceil(N/64), ceil(M/64))
__global__ __launch_bounds__(256) void k_mgemm(double* C, long long ldc, const double* A, long long sai, long long sak, const double* B, long long sbk,
                                               long long sbj, int M, int N, int K, double alpha, double beta) {
    __shared__ double As[64][17], Bs[64][17];
    mgemm_tile(C, ldc, A, sai, sak, B, sbk, sbj, M, N, 0, K, alpha, beta, blockIdx.y * 64, blockIdx.x * 64, As, Bs);
}

// the panel's update of the WHOLE working matrix on the FP64 matrix cores: S -= Gp^T Gp, Gp = rows [ctl[2], ctl[2] + NB) of G
// (rows at or beyond n: none)
template <int NB>
__global__ __launch_bounds__(256) void k_pchol_syrk_mma(double* __restrict__ S, int n, const double* __restrict__ G, const int* __restrict__ ctl) {
    if (ctl[0] >= 0) return;
    __shared__ double As[64][17], Bs[64][17];
    const int base = ctl[2];
    const int K = min(NB, n - base);
    const double* Gp = G + (size_t)base * n;
    mgemm_tile(S, n, Gp, 1, n, Gp, n, 1, n, n, 0, K, -1.0, 1.0, blockIdx.y * 64, blockIdx.x * 64, As, Bs);
}

// ---- Cholesky form of the prior (SADVIO_PRIOR_FORM_CHOLESKY) ---------------------------------------------------------------------
// S' = [[sym_lower(Ak), bk], [bk^T, -1]] ((n + 1) x (n + 1)): the pivoted Cholesky kernels above run on it unchanged — the extra
// index has a negative diagonal, so it is never chosen as a pivot, and as a not-yet-pivoted column it receives
// G[k][n] = (bk[p_k] - sum_{c<k} G[c][p_k] G[c][n]) / G[k][p_k], i.e. G^-T bk restricted to the pivots: r0 = -G[:, n].
__global__ void k_marg_aug_init(const double* __restrict__ Ak, const double* __restrict__ bk, int n, double* __restrict__ S) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n1 = n + 1;
    if (idx >= (long long)n1 * n1) return;
    const int i = (int)(idx / n1), j = (int)(idx - (long long)i * n1);
    double v;
    if (i < n && j < n) v = i >= j ? Ak[(size_t)i * n + j] : Ak[(size_t)j * n + i];   // Eigen reads the lower triangle (marginalization.cpp:321)
    else if (i == n && j == n) v = -1.0;
    else v = bk[i < n ? i : j];
    S[idx] = v;
}

// J[k][i] = G'[k][i], r0[k] = -G'[k][n] for the rank rows (the factor's rows beyond the rank are zero)
__global__ void k_marg_pack_chol(const double* __restrict__ G, int n, const int* __restrict__ rank, double* __restrict__ J, double* __restrict__ r0) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n1 = n + 1, r = *rank;
    if (idx >= (long long)r * n1) return;
    const int k = (int)(idx / n1), i = (int)(idx - (long long)k * n1);
    const double v = G[idx];
    if (i < n) J[(size_t)k * n + i] = v; else r0[k] = -v;
}

// x = G^-1 e_s for the rank refinement of the rank-revealing route (ba_capi.hip: refine_rank_by_eigenvalue): G (rows by ORIGINAL column,
// leading dimension n1) is upper triangular in pivot order, col_of[t] = the column pivoted at step t. One workgroup; the rows are taken in
// blocks of 32 from step s upwards: (A) every row of the block gets its dot product with the part of x that is already solved - x is
// kept by original column in LDS, zero where unsolved, so the sum is a contiguous, coalesced pass over the row (32 threads per row) - and
// the block's 32 x 32 triangle is gathered into LDS; (B) one wave back-substitutes the block (lane t owns x_t, column-oriented updates).
// The host's version of this loop was 0.8 ms at n = 915 (DRAM-bound on the 6.7 MB the DMA had just written); this is ~ 0.1 ms.
constexpr int RB_THREADS = 1024, RB_BLK = 32;
__global__ __launch_bounds__(RB_THREADS) void k_rank_backsub(const double* __restrict__ G, int n1, const int* __restrict__ col_of, int s_last, int ldx, double* __restrict__ xall) {
    __shared__ double xo[PCH_MAXN + 1];
    __shared__ double T[RB_BLK][RB_BLK + 1];
    __shared__ double rhs[RB_BLK];
    const int tid = threadIdx.x;
    const int s = s_last - (int)blockIdx.x;                  // workgroup q solves for step s_last - q
    double* __restrict__ xout = xall + (size_t)blockIdx.x * ldx;
    for (int c = tid; c < n1; c += RB_THREADS) xo[c] = 0.0;
    __syncthreads();
    for (int i1 = s + 1; i1 > 0; i1 -= RB_BLK) {          // block rows [i0, i1), i1 <= s + 1
        const int i0 = i1 > RB_BLK ? i1 - RB_BLK : 0, nb = i1 - i0;
        const int r = tid >> 5, l = tid & 31;             // 32 threads per row
        if (r < nb) {
            const double* row = G + (size_t)(i0 + r) * n1;
            double a0 = 0.0, a1 = 0.0;
            int c = l;
            for (; c + 32 < n1; c += 64) { a0 += row[c] * xo[c]; a1 += row[c + 32] * xo[c + 32]; }
            if (c < n1) a0 += row[c] * xo[c];
            double a = a0 + a1;
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
            if (l == 0) rhs[r] = (i0 + r == s ? 1.0 : 0.0) - a;
            if (l < nb) T[r][l] = row[col_of[i0 + l]];
        }
        __syncthreads();
        if (tid < 64) {                                   // one wave; lanes >= nb idle
            const int t = tid;
            double x = 0.0, b = t < nb ? rhs[t] : 0.0;
            const double rinv = t < nb ? 1.0 / T[t][t] : 0.0;
            for (int q = nb - 1; q >= 0; q--) {
                const double xq = __shfl(b, q) * __shfl(rinv, q);   // x_q = (its remaining right-hand side) / g_qq
                if (t == q) x = xq;
                if (t < q) b -= T[t][q] * xq;
            }
            if (t < nb) { xo[col_of[i0 + t]] = x; xout[i0 + t] = x; }
        }
        __syncthreads();
    }
}

// ---- triangular inverse: Sigma_k = Ak^-1 = Z^T Z for sadvio_ba_sparsify from the Cholesky form --------------------------------
// With p(q) = the index chosen at pivot step q, L[q][k] = G[k][p(q)] is lower triangular and Ak[p(a)][p(b)] = (L L^T)[a][b], so
// Z[k][p(a)] = (L^-1)[k][a]. L^-1 by recursive halving on 32-row leaves: inv([[A, 0], [C, B]]) = [[A^-1, 0], [-B^-1 C A^-1, B^-1]]
// — every level is two batched matrix-core GEMMs over its nodes (k_tri_level), ceil(log2(n / 32)) levels, no pivot chain.
__global__ void k_tri_gather(const double* __restrict__ G, int n, const int* __restrict__ step_of, int* __restrict__ piv_of, int npad, double* __restrict__ L, int phase) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (phase == 0) { if (idx < n && step_of[idx] >= 0 && step_of[idx] < n) piv_of[step_of[idx]] = (int)idx; return; }
    if (idx >= (long long)npad * npad) return;
    const int q = (int)(idx / npad), k = (int)(idx - (long long)q * npad);
    double v = q == k ? 1.0 : 0.0;
    if (q < n && k < n) v = k <= q ? G[(size_t)k * n + piv_of[q]] : 0.0;
    L[idx] = v;
}

struct TriNode { int lo, mid, hi, pad; };

// leaves: in-place inverse of the lower-triangular diagonal block [lo, hi) (hi - lo <= 32), one wave per leaf
__global__ __launch_bounds__(64) void k_tri_leaf(double* __restrict__ L, int npad, const TriNode* __restrict__ leaves) {
    __shared__ double B[32][33];
    const TriNode nd = leaves[blockIdx.x];
    const int s = nd.hi - nd.lo, t = threadIdx.x;
    for (int e = t; e < s * s; e += 64) { const int r = e / s, c = e - r * s; B[r][c] = L[(size_t)(nd.lo + r) * npad + nd.lo + c]; }
    __syncthreads();
    if (t < s) {
        double x[32];
#pragma unroll
        for (int r = 0; r < 32; r++) x[r] = 0.0;
#pragma unroll
        for (int r = 0; r < 32; r++) {
            if (r < s && r >= t) {
                double acc = r == t ? 1.0 : 0.0;
#pragma unroll
                for (int j = 0; j < 32; j++) if (j < r && j >= t) acc -= B[r][j] * x[j];
                x[r] = acc / B[r][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 32; r++) if (r < s) L[(size_t)(nd.lo + r) * npad + nd.lo + t] = r >= t ? x[r] : 0.0;
    }
}

// one level: phase 0: T = C A^-1 (C = L[mid:hi, lo:mid], A^-1 = L[lo:mid, lo:mid]) into Tb at the same coordinates;
// phase 1: L[mid:hi, lo:mid] = -B^-1 T (B^-1 = L[mid:hi, mid:hi]). grid (tiles j, tiles i, node); the contraction range follows
// the triangles (A^-1[k][j] = 0 for k < j, B^-1[i][k] = 0 for k > i).
__global__ __launch_bounds__(256) void k_tri_level(double* __restrict__ L, double* __restrict__ Tb, int npad, const TriNode* __restrict__ nodes, int phase) {
    __shared__ double As[64][17], Bs[64][17];
    const TriNode nd = nodes[blockIdx.z];
    const int M = nd.hi - nd.mid, N = nd.mid - nd.lo;
    const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
    if (ti >= M || tj >= N) return;
    const long long ld = npad;
    if (phase == 0) {
        double* C = Tb + (size_t)nd.mid * ld + nd.lo;
        const double* A = L + (size_t)nd.mid * ld + nd.lo;        // C block, M x N (K = N)
        const double* Bm = L + (size_t)nd.lo * ld + nd.lo;        // A^-1, N x N lower
        mgemm_tile(C, ld, A, ld, 1, Bm, ld, 1, M, N, tj & ~15, N, 1.0, 0.0, ti, tj, As, Bs);
    } else {
        double* C = L + (size_t)nd.mid * ld + nd.lo;
        const double* A = L + (size_t)nd.mid * ld + nd.mid;       // B^-1, M x M lower (K = M)
        const double* Bm = Tb + (size_t)nd.mid * ld + nd.lo;      // T, M x N
        const int kend = ti + 64 < M ? ti + 64 : M;
        mgemm_tile(C, ld, A, ld, 1, Bm, ld, 1, M, N, 0, kend, -1.0, 0.0, ti, tj, As, Bs);
    }
}

// Z[k][i] = (L^-1)[k][step_of[i]] (n x n, original column order)
__global__ void k_tri_scatter(const double* __restrict__ Li, int npad, int n, const int* __restrict__ step_of, double* __restrict__ Z) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const int k = (int)(idx / n), i = (int)(idx - (long long)k * n);
    const int a = step_of[i];
    Z[idx] = (a >= 0 && a <= k) ? Li[(size_t)k * npad + a] : 0.0;
}

// eigen form: Z[c][:] = J[c][:] / |J_c|^2 (Sigma = Lambda^-1, U[:, c] = J_c / sqrt(lambda_c): Sigma_k = sum_c z_c z_c^T). Rows whose
// lambda_c = |J_c|^2 is not above the eigenvalue cut are dropped, as rankReveallingDecomposition drops them (marginalization.cpp:318-342):
// the rows of an orthogonalised rank-deficient Cholesky factor reach below it (its pivots are cut lower than the eigenvalues).
// cut[0] = the cut (k_z_cut); lam = |J_c|^2 from k_row_norm2.
__global__ void k_z_cut(const double* __restrict__ lam, int nf, int noise_floor, double* __restrict__ cut) {
    __shared__ double sh[256];
    double m = 0.0;
    for (int c = threadIdx.x; c < nf; c += blockDim.x) m = fmax(m, lam[c]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = blockDim.x >> 1; s2 > 0; s2 >>= 1) { if ((int)threadIdx.x < s2) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s2]); __syncthreads(); }
    if (threadIdx.x == 0) cut[0] = noise_floor ? fmax(1e-12, (double)nf * 2.220446049250313e-16 * sh[0]) : 1e-12;
}
__global__ __launch_bounds__(JAC_THREADS) void k_z_from_eig(const double* __restrict__ J, int n, const double* __restrict__ lam, const double* __restrict__ cut,
                                                            double* __restrict__ Z) {
    const int c = blockIdx.x;
    const double s = lam[c];
    const double il = s > cut[0] ? 1.0 / s : 0.0;
    for (int i = threadIdx.x; i < n; i += JAC_THREADS) Z[(size_t)c * n + i] = J[(size_t)c * n + i] * il;
}

// rows of V (eigenvectors) scaled by sel[i] (0 = dropped): Vs[i][:] = sel[i] * V[i][:]
__global__ void k_scale_rows(const double* V, const double* sel, int n, double* Vs) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    Vs[idx] = sel[idx / n] * V[idx];
}

// ---- assembly of A, b ---------------------------------------------------------------------------------------
// Generic block: J (rows x ncols, row-major), r (rows), col map -> A (N x N, both triangles) and b.
__device__ __forceinline__ void marg_accumulate(double* A, double* b, int N, const double* J, const double* r, int rows, int ncols,
                                                const int* col) {
    for (int a = 0; a < ncols; a++) {
        if (col[a] < 0) continue;
        double g = 0.0;
        for (int q = 0; q < rows; q++) g += J[q * ncols + a] * r[q];
        atomic_add_f64(&b[col[a]], g);
        for (int c2 = 0; c2 < ncols; c2++) {
            if (col[c2] < 0) continue;
            double h = 0.0;
            for (int q = 0; q < rows; q++) h += J[q * ncols + a] * J[q * ncols + c2];
            atomic_add_f64(&A[(size_t)col[a] * N + col[c2]], h);
        }
    }
}

// reprojection factors of the kept / marginalised landmarks seen from frame0 at zero deltas (…Analytic.cpp:510-572). One thread per
// observation. Every observation adds to the SAME 6 x 6 block and 6 gradient entries of frame0: as global atomics that were ~700 threads
// on 42 addresses (98 us, 2.0 M busy cycles for 93 k VALU instructions); those 27 + 6 sums are reduced over the wave first and lane 0 adds
// them once (both triangles). The landmark blocks go to distinct addresses (two cameras of a landmark share theirs).
template <int FACTOR>
__global__ __launch_bounds__(128) void k_marg_obs(DevPtrs P, const int* items /*[n][2]: device obs index, first column of the landmark*/, int n_items,
                                                  double* A, double* b, int N) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = e < n_items;
    double r[2] = {0.0, 0.0}, Jp[12], Jl[6];
#pragma unroll
    for (int i = 0; i < 12; i++) Jp[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) Jl[i] = 0.0;
    int lc = 0;
    if (have) {
        const int o = items[2 * e];
        lc = items[2 * e + 1];
        const int kf = P.obs_kf[o], cam = P.obs_cam[o];
        double d6[6] = {0, 0, 0, 0, 0, 0}, tab[POSE_TAB];
        pose_table_entry(P.kf_T0 + 12 * (long long)kf, d6, tab);
        const int gl = items[2 * n_items + e];    // the landmark of the item (third slot of the list)
        const double pw[3] = {P.lmk_p[3 * (long long)gl], P.lmk_p[3 * (long long)gl + 1], P.lmk_p[3 * (long long)gl + 2]};
        if (FACTOR == 0) {
            const double* m = P.obs_meas + 2 * (long long)o;
            pixel_factor<true>(tab, P.cam_K + 4 * (long long)cam, P.cam_T + 12 * (long long)cam, pw, m[0], m[1], P.cam_isig[cam], r, Jp, Jl);
        } else {
            const double* m = P.obs_meas + 3 * (long long)o;
            double bb[3] = {m[0], m[1], m[2]};
            angular_factor<true>(tab, P.cam_T + 12 * (long long)cam, pw, bb, P.cam_isig[cam], r, Jp, Jl);
        }
    }
    // frame0's block: wave sums, one add per entry and wave
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = 0; c <= a; c++) {
            const double h = wave_sum(Jp[a] * Jp[c] + Jp[6 + a] * Jp[6 + c]);
            if (ln == 0 && h != 0.0) { atomic_add_f64(&A[(size_t)a * N + c], h); if (c != a) atomic_add_f64(&A[(size_t)c * N + a], h); }
        }
        const double g = wave_sum(Jp[a] * r[0] + Jp[6 + a] * r[1]);
        if (ln == 0 && g != 0.0) atomic_add_f64(&b[a], g);
    }
    if (!have) return;
    // the landmark's rows / columns
#pragma unroll
    for (int a = 0; a < 3; a++) {
        atomic_add_f64(&b[lc + a], Jl[a] * r[0] + Jl[3 + a] * r[1]);
#pragma unroll
        for (int c = 0; c < 3; c++) atomic_add_f64(&A[(size_t)(lc + a) * N + lc + c], Jl[a] * Jl[c] + Jl[3 + a] * Jl[3 + c]);
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double h = Jl[a] * Jp[c] + Jl[3 + a] * Jp[6 + c];
            atomic_add_f64(&A[(size_t)(lc + a) * N + c], h);
            atomic_add_f64(&A[(size_t)c * N + lc + a], h);
        }
    }
}

// IMUFactor + IMUBiasFactor (frame0, frame1) and the PosePriordx blocks: a handful of blocks, one thread each.
struct MargSmall {
    int has_imu, kf_i, kf_j, kf_keep_col;  // global key-frame indices
    ImuDev imu;
    int n_prior;
    int prior_kf[4], prior_base[4];
    double prior_T[4][12], prior_inf[4][6];
};

// One wave: lanes 0, 1, 2.. evaluate the IMU factor, the bias factor and the pose priors (one lane each: the evaluators are scalar
// code) into LDS, then ALL lanes accumulate J^T J / J^T r (as one lane's loop the 24 x 24 x 9 products + atomics were most of the kernel).
__device__ __forceinline__ void marg_accumulate_wave(double* A, double* b, int N, const double* J, const double* r, int rows, int ncols, const int* col, int t) {
    for (int idx = t; idx < ncols * (ncols + 1); idx += 64) {
        const int a = idx / (ncols + 1), c2 = idx - a * (ncols + 1);
        if (col[a] < 0) continue;
        double h = 0.0;
        if (c2 == ncols) {
            for (int q = 0; q < rows; q++) h += J[q * ncols + a] * r[q];
            atomic_add_f64(&b[col[a]], h);
        } else if (col[c2] >= 0) {
            for (int q = 0; q < rows; q++) h += J[q * ncols + a] * J[q * ncols + c2];
            atomic_add_f64(&A[(size_t)col[a] * N + col[c2]], h);
        }
    }
}
__global__ __launch_bounds__(64) void k_marg_small(DevPtrs P, const MargSmall* Sp, double* A, double* b, int N) {
    // the request (1.6 KB: the IMU constants, the priors) comes in with one coalesced copy; the IMU Jacobian is written straight into LDS
    // (un-whitened) and whitened by 24 lanes, as imu_pair_eval did — as one lane's private 9 x 24 arrays it lived in scratch (3.6 KB per
    // lane) and the kernel took 73 us
    __shared__ MargSmall S;
    const int t = threadIdx.x;
    {
        const unsigned long long* src = (const unsigned long long*)Sp;
        unsigned long long* dst = (unsigned long long*)&S;
        for (int i = t; i < (int)(sizeof(MargSmall) / 8); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const double z[6] = {0, 0, 0, 0, 0, 0};
    __shared__ double sJ[9 * 24], sr[9], sJb[72], srb[6], sJp[4][36], srp[4][6];
    __shared__ int scol[24], scolb[12], scolp[4][6];
    if (t == 0 && S.has_imu) {
        double r[9];
        imu_factor_body<ImuDev, false>(S.imu, P.kf_T0 + 12 * (long long)S.kf_i, P.kf_T0 + 12 * (long long)S.kf_j, P.kf_vel + 3 * (long long)S.kf_i,
                                       P.kf_vel + 3 * (long long)S.kf_j, z, z, z, z, z, z, r, sJ);
        for (int i = 0; i < 9; i++) sr[i] = r[i];
        for (int a = 0; a < 6; a++) { scol[a] = a; scol[6 + a] = S.kf_keep_col + a; }
        for (int a = 0; a < 3; a++) { scol[12 + a] = 6 + a; scol[15 + a] = S.kf_keep_col + 6 + a; scol[18 + a] = 9 + a; scol[21 + a] = 12 + a; }
    }
    if (t == 1 && S.has_imu) {
        for (int i = 0; i < 72; i++) sJb[i] = 0.0;
        for (int a = 0; a < 3; a++) {
            srb[a] = S.imu.sa * (P.kf_ba[3 * (long long)S.kf_j + a] - P.kf_ba[3 * (long long)S.kf_i + a]);
            srb[3 + a] = S.imu.sg * (P.kf_bg[3 * (long long)S.kf_j + a] - P.kf_bg[3 * (long long)S.kf_i + a]);
            sJb[a * 12 + a] = -S.imu.sa; sJb[(3 + a) * 12 + 3 + a] = -S.imu.sg; sJb[a * 12 + 6 + a] = S.imu.sa; sJb[(3 + a) * 12 + 9 + a] = S.imu.sg;
            scolb[a] = 9 + a; scolb[3 + a] = 12 + a; scolb[6 + a] = S.kf_keep_col + 9 + a; scolb[9 + a] = S.kf_keep_col + 12 + a;
        }
    }
    if (t >= 2 && t - 2 < S.n_prior) {
        const int k = t - 2;
        double r[6], J[36];
        pose_prior_factor(P.kf_T0 + 12 * (long long)S.prior_kf[k], S.prior_T[k], S.prior_inf[k], z, r, J);
        for (int i = 0; i < 36; i++) sJp[k][i] = J[i];
        for (int a = 0; a < 6; a++) { srp[k][a] = r[a]; scolp[k][a] = S.prior_base[k] + a; }
    }
    __syncthreads();
    if (S.has_imu && t < 24) {   // J <- W J, one column per lane (W = L^T upper triangular, residuals.hpp:151-154)
        double u[9];
#pragma unroll
        for (int q = 0; q < 9; q++) u[q] = sJ[q * 24 + t];
#pragma unroll
        for (int q = 0; q < 9; q++) {
            double v = 0.0;
#pragma unroll
            for (int kk = q; kk < 9; kk++) v += S.imu.W[9 * q + kk] * u[kk];
            sJ[q * 24 + t] = v;
        }
    }
    __syncthreads();
    if (S.has_imu) {
        marg_accumulate_wave(A, b, N, sJ, sr, 9, 24, scol, t);
        marg_accumulate_wave(A, b, N, sJb, srb, 6, 12, scolb, t);
    }
    for (int k = 0; k < S.n_prior; k++) marg_accumulate_wave(A, b, N, sJp[k], srp[k], 6, 6, scolp[k], t);
}

// previous dense prior at zero deltas: A[col a][col c] += sum_q J[q][a] J[q][c], b[col a] += sum_q J[q][a] r0[q]
__global__ void k_marg_last_prior(const double* J, const double* r0, const int* col, int nf, int nl, double* A, double* b, int N) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nl * nl) return;
    const int a = (int)(idx / nl), c = (int)(idx - (long long)a * nl);
    if (col[a] < 0 || col[c] < 0) return;
    double h = 0.0;
    for (int q = 0; q < nf; q++) h += J[(size_t)q * nl + a] * J[(size_t)q * nl + c];
    atomic_add_f64(&A[(size_t)col[a] * N + col[c]], h);
    if (c == a) {
        double g = 0.0;
        for (int q = 0; q < nf; q++) g += J[(size_t)q * nl + a] * r0[q];
        atomic_add_f64(&b[col[a]], g);
    }
}

// the same for a LARGE previous prior (the steady state: n_last ~ n): H = J^T J comes from k_mgemm (FP64 matrix cores), this kernel
// scatters it through the column map and forms J^T r0 (one thread per (a, c); the column loop of the small kernel above streams all
// of J through every workgroup: 3 270 workgroups x 6.7 MB at n_last = 915)
__global__ void k_marg_last_scatter(const double* __restrict__ H, const double* __restrict__ J, const double* __restrict__ r0, const int* __restrict__ col, int nf, int nl,
                                    double* __restrict__ A, double* __restrict__ b, int N) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nl * nl) return;
    const int a = (int)(idx / nl), c = (int)(idx - (long long)a * nl);
    if (a == 0 && col[c] >= 0) {      // thread (0, c): entry c of J^T r0, unit stride across the threads
        double g = 0.0;
        for (int q = 0; q < nf; q++) g += J[(size_t)q * nl + c] * r0[q];
        atomic_add_f64(&b[col[c]], g);
    }
    if (col[a] < 0 || col[c] < 0) return;
    atomic_add_f64(&A[(size_t)col[a] * N + col[c]], H[idx]);
}

// ... and when the resident prior still holds the (H, g) = (Ak, bk) it was factorised from: A += H through the column map (lower
// triangle of H, as the factorisation read it), b += J^T r0 = -g
__global__ void k_marg_last_scatter_h(const double* __restrict__ H, const double* __restrict__ g, const int* __restrict__ col, int nl, double* __restrict__ A,
                                      double* __restrict__ b, int N) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nl * nl) return;
    const int a = (int)(idx / nl), c = (int)(idx - (long long)a * nl);
    if (a == 0 && col[c] >= 0) atomic_add_f64(&b[col[c]], -g[c]);
    if (col[a] < 0 || col[c] < 0) return;
    atomic_add_f64(&A[(size_t)col[a] * N + col[c]], a >= c ? H[idx] : H[(size_t)c * nl + a]);
}

// bk = brr - T bmm (one thread per row of T: n x m)
__global__ void k_marg_bk(const double* T, const double* b, int n, int m, double* bk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = b[m + i];
    for (int k = 0; k < m; k++) s -= T[(size_t)i * m + k] * b[k];
    bk[i] = s;
}

// prior rows from the selected eigenpairs of Ak: J[c][:] = sqrt(lam_c) v_c, r0[c] = -(v_c . bk) / sqrt(lam_c)
__global__ __launch_bounds__(JAC_THREADS) void k_marg_prior(const double* V, const double* ev, const int* sel_rows, int n_full, int n,
                                                            const double* bk, double* J, double* r0) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    const int k = sel_rows[c];
    const double lam = ev[k], sq = sqrt(lam);
    double dot = 0.0;
    for (int i = threadIdx.x; i < n; i += JAC_THREADS) {
        const double v = V[(size_t)k * n + i];
        J[(size_t)c * n + i] = sq * v;
        dot += v * bk[i];
    }
    dot = block_sum_256(dot, sh);
    if (threadIdx.x == 0) r0[c] = -sqrt(1.0 / lam) * dot;
}


// ---- NFR sparsification (Marginalization::sparsifyVIO / sparsifyVO, marginalization.cpp:362-514) ------------------
// Everything is computed from the dense prior J = Lambda^1/2 U^T: lambda_c = |J_c|^2, U[:,c] = J_c / sqrt(lambda_c),
// Sigma = Lambda^-1, so the covariance of an NFR factor with (sparse) Jacobian J_f is
//   cov = (J_f U) Sigma (J_f U)^T = sum_c w_c w_c^T,  w_c = Jsel J_c[cidx] / lambda_c.
struct NfrSpec {
    int rows, cols;
    int cidx[15];
    double Jsel[225];  // rows x cols row-major
};

__global__ __launch_bounds__(JAC_THREADS) void k_row_norm2(const double* J, int nf, int n, double* lam) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += JAC_THREADS) { const double v = J[(size_t)c * n + i]; s += v * v; }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) lam[c] = s;
}

// one workgroup per factor: S[f] (15 x 15 slot, rows x rows used) = sum_c w_c w_c^T
__global__ __launch_bounds__(JAC_THREADS) void k_nfr_cov(const double* J, int nf, int n, const double* lam, const NfrSpec* specs, double* S) {
    __shared__ double acc[225];
    const NfrSpec& sp = specs[blockIdx.x];
    const int rows = sp.rows, cols = sp.cols;
    for (int i = threadIdx.x; i < 225; i += JAC_THREADS) acc[i] = 0.0;
    __syncthreads();
    for (int c = threadIdx.x; c < nf; c += JAC_THREADS) {
        const double il = 1.0 / lam[c];
        double u[15], w[15];
        for (int k = 0; k < cols; k++) u[k] = J[(size_t)c * n + sp.cidx[k]];
        for (int a = 0; a < rows; a++) {
            double s = 0.0;
            for (int k = 0; k < cols; k++) s += sp.Jsel[a * cols + k] * u[k];
            w[a] = s * il;
        }
        for (int a = 0; a < rows; a++)
            for (int b = 0; b <= a; b++) atomic_add_f64(&acc[a * 15 + b], w[a] * w[b]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * rows; i += JAC_THREADS) {
        const int a = i / rows, b = i - a * rows;
        S[(size_t)blockIdx.x * 225 + i] = a >= b ? acc[a * 15 + b] : acc[b * 15 + a];
    }
}

// Round 4: the same covariances from Z with Z^T Z = Sigma_k (either form of the prior), compact factor descriptions — the
// selector matrices of a sparsification come from a table of at most four shapes (IMUPriordx 15 x 15, PoseToLandmark 3 x 9,
// identity 3 x 3, [I -I] 3 x 6) instead of 225 doubles per factor: cov_f = (Jsel Z[:, cidx]^T)(..)^T, packed rows x rows at out_off.
struct NfrSpecC {
    int rows, cols, jsel, out_off;
    int cidx[16];
    int fin, pad;   // fin = 1 (3-row factors): the kernel also takes the factor's information square root, W = (cov^-1)^1/2, in place of cov
};

// Symmetric information square root of a 3 x 3 factor covariance (marginalization.cpp:385-392: Lambda = cov^-1, its symmetric square
// root through the eigen-decomposition, eigenvalues <= 1e-12 dropped) on one lane: the arithmetic of the host's nfr_sqrt_info
// (Gauss-Jordan inverse with partial pivoting, cyclic Jacobi). 300 of these per key-frame were ~0.3 ms of host time in sadvio_ba_sparsify.
// A singular covariance leaves NaNs (the host reports SADVIO_E_NOT_USABLE).
__device__ __forceinline__ void nfr_sqrt_info3(const double* S, double* W) {
    double M[3][6];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { M[i][j] = S[3 * i + j]; M[i][3 + j] = i == j ? 1.0 : 0.0; }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int p = c;
#pragma unroll
        for (int r = 0; r < 3; r++) if (r > c && fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (p != c) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                // (p is c + 1 or c + 2: select without dynamic indexing)
                const double a = M[c][j];
                const double bq = p == 1 ? M[1][j] : M[2][j];
                M[c][j] = bq;
                if (p == 1) M[1][j] = a; else M[2][j] = a;
            }
        }
        if (M[c][c] == 0.0) ok = false;
        const double d = 1.0 / M[c][c];
#pragma unroll
        for (int j = 0; j < 6; j++) M[c][j] *= d;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            if (r == c) continue;
            const double f = M[r][c];
#pragma unroll
            for (int j = 0; j < 6; j++) M[r][j] -= f * M[c][j];
        }
    }
    double A[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { A[i][j] = 0.5 * (M[i][3 + j] + M[j][3 + i]); V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 100; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;     // (0,1) (0,2) (1,2)
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
            for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - sn * b; A[k][q] = sn * a + c * b; }
#pragma unroll
            for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - sn * b; A[q][k] = sn * a + c * b; }
#pragma unroll
            for (int k = 0; k < 3; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - sn * b; V[k][q] = sn * a + c * b; }
        }
    }
    double se[3];
#pragma unroll
    for (int k = 0; k < 3; k++) se[k] = A[k][k] > 1e-12 ? sqrt(A[k][k]) : 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double w = V[i][0] * se[0] * V[j][0] + V[i][1] * se[1] * V[j][1] + V[i][2] * se[2] * V[j][2];
            W[3 * i + j] = ok ? w : __longlong_as_double(0x7ff8000000000000LL);
        }
}

__global__ __launch_bounds__(JAC_THREADS) void k_nfr_cov_z(const double* __restrict__ Z, int nf, int n, const NfrSpecC* __restrict__ specs,
                                                           const double* __restrict__ jsel_tab /*[4][225]*/, double* __restrict__ S) {
    __shared__ double acc[225];
    __shared__ double Js[225];
    __shared__ int ci[16];
    const NfrSpecC& sp = specs[blockIdx.x];
    const int rows = sp.rows, cols = sp.cols;
    const int tid = threadIdx.x;
    for (int i = tid; i < rows * cols; i += JAC_THREADS) Js[i] = jsel_tab[(size_t)sp.jsel * 225 + i];
    for (int i = tid; i < 225; i += JAC_THREADS) acc[i] = 0.0;
    if (tid < 16) ci[tid] = sp.cidx[tid];
    __syncthreads();
    if (rows == 3) {   // the n_keep landmark factors: six partial sums in registers, one wave reduction each
        double p00 = 0, p10 = 0, p11 = 0, p20 = 0, p21 = 0, p22 = 0;
        for (int c = tid; c < nf; c += JAC_THREADS) {
            double w0 = 0, w1 = 0, w2 = 0;
            for (int k = 0; k < cols; k++) {
                const double u = Z[(size_t)c * n + ci[k]];
                w0 += Js[k] * u; w1 += Js[cols + k] * u; w2 += Js[2 * cols + k] * u;
            }
            p00 += w0 * w0; p10 += w1 * w0; p11 += w1 * w1; p20 += w2 * w0; p21 += w2 * w1; p22 += w2 * w2;
        }
        p00 = wave_sum(p00); p10 = wave_sum(p10); p11 = wave_sum(p11); p20 = wave_sum(p20); p21 = wave_sum(p21); p22 = wave_sum(p22);
        if ((tid & 63) == 0) {
            atomic_add_f64(&acc[0], p00); atomic_add_f64(&acc[15], p10); atomic_add_f64(&acc[16], p11);
            atomic_add_f64(&acc[30], p20); atomic_add_f64(&acc[31], p21); atomic_add_f64(&acc[32], p22);
        }
    } else {           // the one 15-row factor of a VIO prior
        for (int c = tid; c < nf; c += JAC_THREADS) {
            double u[15], w[15];
            for (int k = 0; k < cols; k++) u[k] = Z[(size_t)c * n + ci[k]];
            for (int a = 0; a < rows; a++) {
                double s = 0.0;
                for (int k = 0; k < cols; k++) s += Js[a * cols + k] * u[k];
                w[a] = s;
            }
            for (int a = 0; a < rows; a++)
                for (int b = 0; b <= a; b++) atomic_add_f64(&acc[a * 15 + b], w[a] * w[b]);
        }
    }
    __syncthreads();
    if (rows == 3 && sp.fin) {
        if (tid == 0) {
            const double C[9] = {acc[0], acc[15], acc[30], acc[15], acc[16], acc[31], acc[30], acc[31], acc[32]};
            double W[9];
            nfr_sqrt_info3(C, W);
#pragma unroll
            for (int i = 0; i < 9; i++) S[(size_t)sp.out_off + i] = W[i];
        }
        return;
    }
    for (int i = tid; i < rows * rows; i += JAC_THREADS) {
        const int a = i / rows, b = i - a * rows;
        S[(size_t)sp.out_off + i] = a >= b ? acc[a * 15 + b] : acc[b * 15 + a];
    }
}

// |trace of the 3x3 block (a, b) of J^T J| for every pair of kept landmarks (computeOffDiag, :304-316)
__global__ void k_nfr_trace(const double* J, int nf, int n, const int* lcols, int K, double* mi) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * K) return;
    const int a = idx / K, b = idx - a * K;
    if (a >= b) return;
    double tr = 0.0;
    for (int c = 0; c < nf; c++)
        for (int q = 0; q < 3; q++) tr += J[(size_t)c * n + lcols[a] + q] * J[(size_t)c * n + lcols[b] + q];
    mi[a * K + b] = fabs(tr);
    mi[b * K + a] = fabs(tr);
}

// ---- relative-pose information between two key-frames (marginalizeRelative, …Analytic.cpp:665-809) ---------------------------
// Every landmark shared by the two frames is marginalised; Amm is block diagonal (3 x 3 per landmark), so the reference's
// eigen pseudo-inverse of Amm is taken block by block: one thread per landmark evaluates its reprojection factors in the two
// frames at zero deltas, forms H_ll, E = [J_a^T J_l ; J_b^T J_l] (12 x 3) and the pose blocks, diagonalises H_ll (cyclic
// Jacobi) and leaves (lambda, V, E) in a scratch row; the second kernel applies the cut (which needs the largest eigenvalue
// of ALL blocks, marginalization.cpp:237 with the noise floor of oracle/marg.c) and subtracts E H_ll^+ E^T from Ak.
// `mult`: how many times the reference enters the landmark (once per feature in frame b, marginalization.cpp:548-559).
constexpr int RELM_ROW = 3 + 9 + 36;
template <int FACTOR>
__global__ __launch_bounds__(64) void k_relmarg_lmk(DevPtrs P, const int* items /*[n][2]: global landmark, multiplicity*/, int n_items, int ga, int gb,
                              double* scratch, double* Ak /*144, pose blocks accumulated here*/, unsigned long long* evmax_bits) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_items) return;
    const int gl = items[2 * e];
    const double mult = (double)items[2 * e + 1];
    const double pw[3] = {P.lmk_p[3 * (long long)gl], P.lmk_p[3 * (long long)gl + 1], P.lmk_p[3 * (long long)gl + 2]};
    double Hll[9], E[36], Hp[2][36];
    for (int i = 0; i < 9; i++) Hll[i] = 0.0;
    for (int i = 0; i < 36; i++) { E[i] = 0.0; Hp[0][i] = 0.0; Hp[1][i] = 0.0; }
    const double d6[6] = {0, 0, 0, 0, 0, 0};
    for (int o = P.lmk_ob[gl]; o < P.lmk_oe[gl]; o++) {
        const int kf = P.obs_kf[o], cam = P.obs_cam[o];
        if (cam < 0 || (kf != ga && kf != gb)) continue;
        const int side = kf == ga ? 0 : 1;
        double tab[POSE_TAB], r[2], Jp[12], Jl[6];
        pose_table_entry(P.kf_T0 + 12 * (long long)kf, d6, tab);
        if (FACTOR == 0) {
            const double* m = P.obs_meas + 2 * (long long)o;
            pixel_factor<true>(tab, P.cam_K + 4 * (long long)cam, P.cam_T + 12 * (long long)cam, pw, m[0], m[1], P.cam_isig[cam], r, Jp, Jl);
        } else {
            const double* m = P.obs_meas + 3 * (long long)o;
            double bb[3] = {m[0], m[1], m[2]};
            angular_factor<true>(tab, P.cam_T + 12 * (long long)cam, pw, bb, P.cam_isig[cam], r, Jp, Jl);
        }
        for (int q = 0; q < 2; q++) {
            for (int a = 0; a < 3; a++) {
                for (int b = 0; b < 3; b++) Hll[3 * a + b] += Jl[3 * q + a] * Jl[3 * q + b];
                for (int p = 0; p < 6; p++) E[(6 * side + p) * 3 + a] += Jp[6 * q + p] * Jl[3 * q + a];
            }
            for (int p = 0; p < 6; p++) for (int p2 = 0; p2 < 6; p2++) Hp[side][6 * p + p2] += Jp[6 * q + p] * Jp[6 * q + p2];
        }
    }
    for (int i = 0; i < 9; i++) Hll[i] *= mult;
    for (int i = 0; i < 36; i++) { E[i] *= mult; Hp[0][i] *= mult; Hp[1][i] *= mult; }
    for (int side = 0; side < 2; side++)
        for (int p = 0; p < 6; p++)
            for (int p2 = 0; p2 < 6; p2++)
                if (Hp[side][6 * p + p2] != 0.0) atomic_add_f64(&Ak[(6 * side + p) * 12 + 6 * side + p2], Hp[side][6 * p + p2]);
    // cyclic Jacobi on the symmetrised 3 x 3 block
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[3 * i + j] = 0.5 * (Hll[3 * i + j] + Hll[3 * j + i]);
    for (int sweep = 0; sweep < 30; sweep++) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5], diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                const double apq = A[3 * p + q];
                if (apq == 0.0) continue;
                const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) { const double a = A[3 * k + p], b = A[3 * k + q]; A[3 * k + p] = c * a - s * b; A[3 * k + q] = s * a + c * b; }
                for (int k = 0; k < 3; k++) { const double a = A[3 * p + k], b = A[3 * q + k]; A[3 * p + k] = c * a - s * b; A[3 * q + k] = s * a + c * b; }
                for (int k = 0; k < 3; k++) { const double a = V[3 * k + p], b = V[3 * k + q]; V[3 * k + p] = c * a - s * b; V[3 * k + q] = s * a + c * b; }
            }
    }
    double* row = scratch + (long long)e * RELM_ROW;
    double mx = 0.0;
    for (int k = 0; k < 3; k++) { row[k] = A[4 * k]; mx = fmax(mx, fabs(A[4 * k])); }
    for (int i = 0; i < 9; i++) row[3 + i] = V[i];
    for (int i = 0; i < 36; i++) row[12 + i] = E[i];
    atomic_max_u64(evmax_bits, (unsigned long long)__double_as_longlong(mx));
}

__global__ void k_relmarg_apply(const double* scratch, int n_items, int m, const unsigned long long* evmax_bits, double* Ak, int noise_floor) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_items) return;
    const double cut = noise_floor ? fmax(1e-12, (double)m * 2.220446049250313e-16 * __longlong_as_double((long long)*evmax_bits)) : 1e-12;   // SADVIO_EIG_CUT_*
    const double* row = scratch + (long long)e * RELM_ROW;
    double Pi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 3; k++) {
        if (!(row[k] > cut)) continue;
        const double iv = 1.0 / row[k];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Pi[3 * a + b] += row[3 + 3 * a + k] * iv * row[3 + 3 * b + k];
    }
    const double* E = row + 12;
    for (int i = 0; i < 12; i++) {
        double t[3];
        for (int b = 0; b < 3; b++) t[b] = E[3 * i] * Pi[b] + E[3 * i + 1] * Pi[3 + b] + E[3 * i + 2] * Pi[6 + b];
        for (int j = 0; j < 12; j++) {
            const double v = t[0] * E[3 * j] + t[1] * E[3 * j + 1] + t[2] * E[3 * j + 2];
            if (v != 0.0) atomic_add_f64(&Ak[12 * i + j], -v);
        }
    }
}

// J (6 x 12) of Relative6DPose(T_w_a, T_w_b, T_a_w T_w_b, I) at zero deltas (…Analytic.cpp:784-800)
__global__ __launch_bounds__(64) void k_relmarg_jac(DevPtrs P, int ga, int gb, double* J72) {
    if (threadIdx.x != 0) return;
    const double* Ta = P.kf_T0 + 12 * (long long)ga;   // T_f_w
    const double* Tb = P.kf_T0 + 12 * (long long)gb;
    double Twa[12], Twb[12], Tab[12], v[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Twa[3 * i + j] = Ta[3 * j + i]; Twb[3 * i + j] = Tb[3 * j + i]; }
    m3_tvec(Ta, Ta + 9, v); for (int i = 0; i < 3; i++) Twa[9 + i] = -v[i];
    m3_tvec(Tb, Tb + 9, v); for (int i = 0; i < 3; i++) Twb[9 + i] = -v[i];
    m3_mul(Ta, Twb, Tab);                                   // T_a_w T_w_b
    m3_vec(Ta, Twb + 9, v); for (int i = 0; i < 3; i++) Tab[9 + i] = v[i] + Ta[9 + i];
    const double W[36] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1};
    const double z[6] = {0, 0, 0, 0, 0, 0};
    double r[6], J[90];
    relative_pose_factor(Twa, Twb, Tab, W, z, z, r, J);
    for (int i = 0; i < 6; i++) for (int c = 0; c < 12; c++) J72[12 * i + c] = J[15 * i + c];
}

}  // namespace sadvio
