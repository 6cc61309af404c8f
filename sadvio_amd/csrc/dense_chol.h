// dense_chol.h — blocked right-looking Cholesky + solve of a reduced system that lives in HBM (N_p > 174).
//
// Storage: full row-major N x N, lower triangle used (element (i >= j) at A[i * ld + j]); the right-hand side y
// is a separate vector that is carried through the factorisation as an extra matrix row (forward substitution
// for free), so after the last block column y = L^-1 y; k_chol_backsolve then overwrites it with L^-T y.
//
// Per block column k0 (NB = 32 columns), two launches:
//   k_chol_panel   every workgroup loads the 32x32 diagonal block into LDS and factors it redundantly (one wave,
//                  rows in registers, v_readlane broadcasts: ~1.5 us), then solves its 256 panel rows
//                  L_rk = A_rk L_kk^-T by forward substitution against the LDS copy; workgroup 0 also files
//                  L_kk and substitutes the right-hand-side row.
//   k_chol_update  trailing update C -= L_ik L_jk^T on 64x64 tiles (lower-triangular tile pairs), K = 32, on the
//                  FP64 matrix cores (v_mfma_f64_16x16x4_f64, 8 per 16x16 sub-tile); a second set of workgroups
//                  updates the right-hand-side row.
// `rows_end` bounds the rows below the block column that can be non-zero (half bandwidth known on the host from
// the co-visibility structure; = N for a dense system): the work per step is O(bw^2), not O(N^2), for the
// block-banded systems of long trajectories (configs 4 / 5).
#pragma once
#include "chol16.h"
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace sadvio {

constexpr int CH_NB = 32;
constexpr int CH_TS = 64;
constexpr int CH_THREADS = 256;

__device__ __forceinline__ double ch_readlane(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// In-register Cholesky of an nb x nb (nb <= 32) block held one ROW per lane (lane r: a[c] = D[r][c], c <= r).
// Returns false in all lanes if a pivot is not positive. On return a[] holds row r of L.
__device__ __forceinline__ bool ch_factor_rows(double (&a)[CH_NB], int nb, int ln) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < CH_NB; c++) {
        if (c < nb) {
            const double d = ch_readlane(a[c], c);  // pivot (row c, column c), wave-uniform
            if (!(d > 0.0)) ok = false;
            const double inv = rsqrt_nr(d);  // v_rsq_f64 + 2 Newton steps (kernels.h): ~4x shorter chain than sqrt + divide
            a[c] = (ln == c) ? d * inv : a[c] * inv;  // column c of L (rows >= c)
#pragma unroll
            for (int j = c + 1; j < CH_NB; j++) {
                if (j < nb) {
                    const double ljc = ch_readlane(a[c], j);  // L[j][c], wave-uniform
                    a[j] -= a[c] * ljc;                       // rows r >= j matter; others hold garbage never read
                }
            }
        }
    }
    return ok;
}

// One block column: diagonal factor + panel solve (+ right-hand-side row).
__global__ __launch_bounds__(CH_THREADS) void k_chol_panel(double* __restrict__ A, long long ld, double* __restrict__ y,
                                                           int N, int k0, int rows_end, int* info, const int* skip, long long* ts) {
    if (skip && *skip) return;
    if (*info != 0) return;
#define CH_TS_MARK(i_) do { if (ts && blockIdx.x == 0 && threadIdx.x == 0) ts[i_] = wall_clock64(); } while (0)
    CH_TS_MARK(0);
    __shared__ double Ld[CH_NB][CH_NB + 1];
    __shared__ double dinv[CH_NB];
    __shared__ int s_ok;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    const int nb = min(CH_NB, N - k0);
    if (wv == 0) {
        double a[CH_NB];
        const int r = ln < nb ? ln : nb - 1;
#pragma unroll
        for (int c = 0; c < CH_NB; c++) a[c] = (c < nb && c <= r) ? A[(long long)(k0 + r) * ld + k0 + c] : 0.0;
        CH_TS_MARK(1);
        const bool ok = ch_factor_rows(a, nb, ln);
        CH_TS_MARK(2);
        if (ln < nb) {
#pragma unroll
            for (int c = 0; c < CH_NB; c++)
                if (c < nb) Ld[ln][c] = (c <= ln) ? a[c] : 0.0;
        }
        if (ln == 0) s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) {
        if (blockIdx.x == 0 && tid == 0) *info = k0 + 1;
        return;
    }
    if (tid < nb) dinv[tid] = 1.0 / Ld[tid][tid];
    __syncthreads();
    if (blockIdx.x == 0) {
        // file L_kk (lower triangle) into A
        for (int e = tid; e < nb * nb; e += CH_THREADS) {
            const int r = e / nb, c = e - r * nb;
            if (c <= r) A[(long long)(k0 + r) * ld + k0 + c] = Ld[r][c];
        }
    }
    CH_TS_MARK(3);
    // panel rows (and, as the last "row", the right-hand side)
    const int s = k0 + nb;
    const int m = rows_end - s;
    const int idx = blockIdx.x * CH_THREADS + tid;  // 0 .. m-1 panel rows, m = rhs row (workgroup that owns it)
    const bool is_rhs = idx == m;
    if (idx > m) return;
    double x[CH_NB];
    double* row = is_rhs ? y + k0 : A + (long long)(s + idx) * ld + k0;
#pragma unroll
    for (int c = 0; c < CH_NB; c++) x[c] = c < nb ? row[c] : 0.0;
    CH_TS_MARK(4);
#pragma unroll
    for (int c = 0; c < CH_NB; c++) {
        if (c < nb) {
            const double xc = x[c] * dinv[c];
            x[c] = xc;
#pragma unroll
            for (int j = c + 1; j < CH_NB; j++)
                if (j < nb) x[j] -= xc * Ld[j][c];
        }
    }
    CH_TS_MARK(5);
#pragma unroll
    for (int c = 0; c < CH_NB; c++)
        if (c < nb) row[c] = x[c];
    CH_TS_MARK(6);
}

// Trailing update of the rows / columns [s, rows_end) by the panel of block column k0, and of the rhs row.
__global__ __launch_bounds__(CH_THREADS) void k_chol_update(double* __restrict__ A, long long ld, double* __restrict__ y,
                                                            int N, int k0, int rows_end, const int* info, const int* skip) {
    if (skip && *skip) return;
    if (*info != 0) return;
    const int nb = min(CH_NB, N - k0);
    const int s = k0 + nb;
    const int m = rows_end - s;
    if (m <= 0) return;
    const int nt = (m + CH_TS - 1) / CH_TS;
    const int npair = nt * (nt + 1) / 2;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    __shared__ double Pi[CH_TS][CH_NB + 1];
    __shared__ double Pj[CH_TS][CH_NB + 1];
    if ((int)blockIdx.x >= npair) {
        // rhs row: y[s + j] -= sum_c z_c L[s + j][k0 + c], one thread per j
        const int t = blockIdx.x - npair;
        const int j = t * CH_TS * 4 + tid;  // 256 entries per workgroup
        if (j < m) {
            const double* lrow = A + (long long)(s + j) * ld + k0;
            double acc = 0.0;
#pragma unroll 8
            for (int c = 0; c < nb; c++) acc += y[k0 + c] * lrow[c];
            y[s + j] -= acc;
        }
        return;
    }
    int ti = 0, rem = blockIdx.x;
    while (rem >= ti + 1) { rem -= ti + 1; ti++; }
    const int tj = rem;
    const int i0 = ti * CH_TS, j0 = tj * CH_TS;
    // stage the two panel slabs (64 rows x nb columns each; zero beyond the edge)
    for (int e = tid; e < CH_TS * CH_NB; e += CH_THREADS) {
        const int r = e / CH_NB, c = e - r * CH_NB;
        const int gi = i0 + r, gj = j0 + r;
        Pi[r][c] = (gi < m && c < nb) ? A[(long long)(s + gi) * ld + k0 + c] : 0.0;
        Pj[r][c] = (gj < m && c < nb) ? A[(long long)(s + gj) * ld + k0 + c] : 0.0;
    }
    __syncthreads();
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lr = ln & 15, lk = ln >> 4;
    // wave wv owns the 16-row band wv of the tile; 4 column sub-tiles (skipping those above the diagonal)
    const int ib = wv;
    for (int jb = 0; jb < 4; jb++) {
        if (ti == tj && jb > ib) continue;
        const int rbase = i0 + 16 * ib, cbase = j0 + 16 * jb;
        if (rbase >= m || cbase >= m) continue;
        d4 c;
        long long addr[4];
        bool ok[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = rbase + lk + 4 * rg, col = cbase + lr;
            ok[rg] = row < m && col < m && col <= row;
            addr[rg] = (long long)(s + (row < m ? row : m - 1)) * ld + s + (col < m ? col : m - 1);
            c[rg] = ok[rg] ? A[addr[rg]] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < CH_NB; kk += 4) {
            const double a = -Pi[16 * ib + lr][kk + lk];
            const double b = Pj[16 * jb + lr][kk + lk];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
            if (ok[rg]) A[addr[rg]] = c[rg];
    }
}

// x = L^-T y (in place), block by block from the bottom; one workgroup. `bw` = number of rows below a block
// column that can be non-zero (N for dense).
__global__ __launch_bounds__(CH_THREADS) void k_chol_backsolve(const double* __restrict__ A, long long ld,
                                                               double* __restrict__ y, int N, int bw, const int* info,
                                                               const int* skip) {
    if (skip && *skip) return;
    if (*info != 0) return;
    __shared__ double part[CH_THREADS / CH_NB][CH_NB];
    __shared__ double xk[CH_NB];
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    const int c = tid & (CH_NB - 1), g = tid / CH_NB;  // column within the block, row group
    constexpr int NG = CH_THREADS / CH_NB;
    const int nblk = (N + CH_NB - 1) / CH_NB;
    for (int k = nblk - 1; k >= 0; k--) {
        const int k0 = k * CH_NB;
        const int nb = min(CH_NB, N - k0);
        const int s = k0 + nb;
        const int rows_end = min(N, s + bw);
        // t_c = sum_{i >= s} L[i][k0 + c] x_i
        double acc = 0.0;
        if (c < nb)
            for (int i = s + g; i < rows_end; i += NG) acc += A[(long long)i * ld + k0 + c] * y[i];
        part[g][c] = acc;
        // the diagonal block, one column per lane of wave 0: l[j] = L[k0 + j][k0 + lane]
        double l[CH_NB];
        if (wv == 0) {
#pragma unroll
            for (int j = 0; j < CH_NB; j++) l[j] = (j < nb && ln < nb && ln <= j) ? A[(long long)(k0 + j) * ld + k0 + ln] : 0.0;
        }
        __syncthreads();
        if (wv == 0) {
            double t = 0.0;
            if (ln < nb) {
                t = y[k0 + ln];
#pragma unroll
                for (int q = 0; q < NG; q++) t -= part[q][ln];
            }
            // solve L_kk^T x = t: x_j for j = nb-1 .. 0; lane c keeps t_c
#pragma unroll
            for (int j = CH_NB - 1; j >= 0; j--) {
                if (j < nb) {
                    const double tj = ch_readlane(t, j);
                    const double ljj = ch_readlane(l[j], j);
                    const double xj = tj / ljj;
                    if (ln == j) t = xj;
                    else if (ln < j) t -= l[j] * xj;  // L[k0 + j][k0 + ln]
                }
            }
            if (ln < nb) y[k0 + ln] = t;
        }
        __syncthreads();
    }
}


// ---- block-banded systems: sliding LDS window over the tuned in-LDS factorisation ----------------------------------
// A long trajectory gives a block-banded S (half bandwidth bw scalars, known on the host). A window of R = bw + C
// rows is held packed in LDS; chol_solve_packed<NB, PARTIAL> eliminates its first C columns (their L entries only
// reach bw rows further down, so they are final) and leaves the updated bw x bw Schur complement, which is carried
// to the top-left corner of the next window while the new rows are loaded from HBM. L (off-diagonal blocks) is
// written in place, the inverse pivot blocks to `linv_g`, the right-hand side rides along as the extra row. The
// backward pass walks the windows in reverse. ONE launch per solve instead of 2 N / 32: the whole factorisation is a
// single workgroup -- the dependency chain of a banded Cholesky is sequential anyway.
// packed lower-triangle index g -> (row i, column j <= i)
__device__ __forceinline__ void tri_decode(int g, int& i, int& j) {
    i = (int)((sqrt(8.0 * (double)g + 1.0) - 1.0) * 0.5);
    while (tri(i + 1, 0) <= g) i++;
    while (tri(i, 0) > g) i--;
    j = g - tri(i, 0);
}

// Re-zero the band of a banded S after a step (everything outside the band is never written): N x bw doubles instead of
// the N x N memset (72 MB per LM step at config 5).
__global__ void k_band_zero(double* __restrict__ S, long long ld, int N, int bw) {
    const long long total = (long long)N * bw;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / bw), j = i - (bw - 1) + (int)(e - (long long)i * bw);
        if (j >= 0) S[(long long)i * ld + j] = 0.0;
    }
}

// A window sharded over several GPUs all-reduces its reduced system every step; for a banded S only the band
// (N x bw entries instead of N x N: 45x fewer bytes at config 5) travels: pack -> all-reduce -> unpack.
// buf = [band: row i holds S[i][i - bw + 1 .. i] | tail: the `tail` doubles that follow S (gradients, diagonal, partials)].
__global__ void k_band_pack(const double* __restrict__ S, long long ld, int N, int bw, long long tail, double* __restrict__ buf, int unpack) {
    const long long nb = (long long)N * bw, total = nb + tail;
    double* Sm = const_cast<double*>(S);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        if (e < nb) {
            const int i = (int)(e / bw), t = (int)(e - (long long)i * bw);
            const int j = i - (bw - 1) + t;
            if (j < 0) { if (!unpack) buf[e] = 0.0; continue; }
            if (unpack) Sm[(long long)i * ld + j] = buf[e]; else buf[e] = S[(long long)i * ld + j];
        } else {
            const long long q = (long long)N * ld + (e - nb);   // the tail follows the full N x ld matrix
            if (unpack) Sm[q] = buf[e]; else buf[e] = S[q];
        }
    }
}

// Twisted factorisation (twist_M >= 0): the band is eliminated from BOTH ends at once by two workgroups — half 0 runs
// columns [0, M) top-down, half 1 runs columns [M + bw, N) bottom-up (the same code on the index-reversed matrix; the
// symmetric element is read from the stored lower triangle) — each leaving its bw x bw Schur complement on the middle
// block [M, M + bw) in `mid`. k_band_mid adds the two, solves the middle block, and the backward passes of both halves
// (phase 1) start from it. The sequential pivot chain, which is all this solver's time (3 us per 6 columns), is halved.
// twist_M < 0: one workgroup does the whole band, forward and backward (short systems).
struct BandMap {
    int rev, N;
    __device__ __forceinline__ long long a(long long ld, int li, int lj) const {  // local li >= lj -> element of the stored lower triangle
        return rev ? (long long)(N - 1 - lj) * ld + (N - 1 - li) : (long long)li * ld + lj;
    }
    __device__ __forceinline__ int v(int li) const { return rev ? N - 1 - li : li; }
    __device__ __forceinline__ long long blk(int lb, int nblk_tot) const { return rev ? nblk_tot - 1 - lb : lb; }
};

template <int NB>
__global__ __launch_bounds__(SOLVE_THREADS) void k_band_solve(double* __restrict__ A, long long ld, double* __restrict__ y,
                                                              double* __restrict__ linv_g, int N, int bw, int C, int* info,
                                                              const int* skip, long long* dbg, int twist_M, int phase, double* __restrict__ mid) {
#define BAND_TS(win_, idx_) do { if (dbg && tid == 0 && blockIdx.x == 0 && (win_)) dbg[idx_] = wall_clock64(); } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool twisted = twist_M >= 0;
    BandMap mp;
    mp.rev = twisted && blockIdx.x == 1; mp.N = N;
    // columns this workgroup eliminates / rows of its (local) system
    const int Ne = !twisted ? N : (mp.rev ? N - twist_M - bw : twist_M);
    const int Nh = !twisted ? N : Ne + bw;
    const int nblk_tot = N / NB;
    const int Rmax = bw + C;
    double* LpT = (double*)smem;                              // [NBP][Rmax + 2]
    double* Pk = LpT + (size_t)(Rmax + 2) * NBP;              // packed window incl. rhs row
    double* xv = Pk + (size_t)(Rmax + 1) * (Rmax + 2) / 2;   // [Rmax] work vector of the backward pass
    double* xs = xv + Rmax;                                   // [Rmax]
    double* linvTab = xs + Rmax;                              // [Rmax / NB][NB * NB]
    if (!twisted || phase == 0) {
    int carried = 0, Rprev = 0, Cprev = 0;
    for (int c0 = 0; c0 < Ne; c0 += C) {
        const int Cw = min(C, Ne - c0);
        const int R = min(bw + Cw, Nh - c0);
        const bool tsw = c0 == 2 * C;
        BAND_TS(tsw, 0);
        // (1) carry: old (i + Cprev, j + Cprev) -> new (i, j) for i, j < carried, rhs row likewise; through registers
        {
            constexpr int MAXC = (MAX_LDS_NP * (MAX_LDS_NP + 1) / 2 + MAX_LDS_NP + SOLVE_THREADS - 1) / SOLVE_THREADS;
            double v[MAXC];
            const int ntri = carried * (carried + 1) / 2;
            const int nc = ntri + carried;  // + the rhs entries
#pragma unroll
            for (int q = 0; q < MAXC; q++) {
                const int e = tid + q * SOLVE_THREADS;
                if (e < nc) {
                    int i, j;
                    if (e < ntri) { tri_decode(e, i, j); v[q] = Pk[tri(i + Cprev, j + Cprev)]; }
                    else v[q] = Pk[tri(Rprev, e - ntri + Cprev)];
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MAXC; q++) {
                const int e = tid + q * SOLVE_THREADS;
                if (e < nc) Pk[e < ntri ? e : tri(R, e - ntri)] = v[q];
            }
        }
        BAND_TS(tsw, 1);
        // (2) fresh rows [carried, R) from HBM (zero outside the band by construction of A); the packed index runs
        //     linearly, four independent loads in flight per thread
        {
            const int g0 = tri(carried, 0), g1 = tri(R, 0);
            for (int gb = g0 + tid; gb < g1; gb += 4 * nt) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int g = gb + u * nt;
                    if (g < g1) { int i, j; tri_decode(g, i, j); v[u] = A[mp.a(ld, c0 + i, c0 + j)]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int g = gb + u * nt;
                    if (g < g1) Pk[g] = v[u];
                }
            }
        }
        for (int j = carried + tid; j < R; j += nt) Pk[tri(R, j)] = y[mp.v(c0 + j)];
        if (tid == 0) Pk[tri(R, R)] = 0.0;
        __syncthreads();
        BAND_TS(tsw, 2);
        // (3) eliminate the first Cw columns
        const int nsteps = Cw / NB;
        const bool ok = chol_solve_packed<NB, true, true>(Pk, R, xv, xs, LpT, linvTab, nullptr, nsteps, bw);
        __syncthreads();
        BAND_TS(tsw, 3);
        if (!ok) { if (tid == 0) *info = c0 + 1; return; }
        // (4) file L (rows below each pivot block), the inverse pivot blocks and the substituted rhs
        for (int g = tid; g < tri(R, 0); g += nt) {
            int i, j;
            tri_decode(g, i, j);
            if (j < min(Cw, (i / NB) * NB)) A[mp.a(ld, c0 + i, c0 + j)] = Pk[g];  // left of row i's own pivot block
        }
        for (int e = tid; e < nsteps * NB * NB; e += nt) {
            const int lb = e / (NB * NB);
            linv_g[mp.blk(c0 / NB + lb, nblk_tot) * NB * NB + (e - lb * NB * NB)] = linvTab[e];
        }
        for (int j = tid; j < Cw; j += nt) y[mp.v(c0 + j)] = Pk[tri(R, j)];
        carried = R - Cw; Rprev = R; Cprev = Cw;
        __syncthreads();
        BAND_TS(tsw, 4);
        BAND_TS(c0 + C >= Ne, 5);
    }
    if (twisted) {
        // the Schur complement this half leaves on the middle block (bw x bw + rhs), in the middle block's own order
        double* mo = mid + (size_t)blockIdx.x * (tri(bw, 0) + bw);
        for (int g = tid; g < tri(bw, 0); g += nt) {
            int i, j;
            tri_decode(g, i, j);
            const double v = Pk[tri(i + Cprev, j + Cprev)];
            if (mp.rev) mo[tri(bw - 1 - j, bw - 1 - i)] = v; else mo[g] = v;
        }
        for (int i = tid; i < bw; i += nt) mo[tri(bw, 0) + (mp.rev ? bw - 1 - i : i)] = Pk[tri(Rprev, i + Cprev)];
        return;
    }
    // backward pass: x = L^-T z, windows in reverse. The L entries were stored over addresses this CU has read before:
    // drop possibly stale L1 lines first.
    __threadfence();
    __syncthreads();
    }
    const int nwin = (Ne + C - 1) / C;
    for (int wi = nwin - 1; wi >= 0; wi--) {
        const int c0 = wi * C;
        const int Cw = min(C, Ne - c0);
        const int R = min(bw + Cw, Nh - c0);
        const int nsteps = Cw / NB;
        const bool tsw = wi == 2;
        BAND_TS(tsw, 6);
        {
            const int g1 = tri(R, 0);
            for (int gb = tid; gb < g1; gb += 4 * nt) {
                double v[4];
                bool use[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int g = gb + u * nt;
                    use[u] = false;
                    if (g < g1) {
                        int i, j;
                        tri_decode(g, i, j);
                        use[u] = j < min(Cw, (i / NB) * NB);
                        if (use[u]) v[u] = A[mp.a(ld, c0 + i, c0 + j)];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (use[u]) Pk[gb + u * nt] = v[u];
            }
        }
        for (int e = tid; e < nsteps * NB * NB; e += nt) {
            const int lb = e / (NB * NB);
            linvTab[e] = linv_g[mp.blk(c0 / NB + lb, nblk_tot) * NB * NB + (e - lb * NB * NB)];
        }
        for (int i = tid; i < R; i += nt) xv[i] = y[mp.v(c0 + i)];  // i < Cw: z, i >= Cw: x already solved
        __syncthreads();
        BAND_TS(tsw, 7);
        // rows below the window's columns: x[j] -= sum_{i >= Cw} L[i][j] x_i
        for (int j = tid; j < Cw; j += nt) {
            double t = xv[j];
            for (int i = Cw; i < R; i++) t -= Pk[tri(i, j)] * xv[i];
            xs[j] = t;
        }
        __syncthreads();
        for (int j = tid; j < Cw; j += nt) xv[j] = xs[j];
        __syncthreads();
        BAND_TS(tsw, 8);
        for (int k = nsteps - 1; k >= 0; k--) {
            const int b0 = k * NB;
            double xk[NB];
#pragma unroll
            for (int cc = 0; cc < NB; cc++) {
                double t = 0.0;
#pragma unroll
                for (int q = cc; q < NB; q++) t += linvTab[k * NB * NB + q * NB + cc] * xv[b0 + q];
                xk[cc] = t;
            }
            if (tid < NB) {
                double v = xk[0];
#pragma unroll
                for (int cc = 1; cc < NB; cc++) if (tid == cc) v = xk[cc];
                xs[b0 + tid] = v;
            }
            for (int j = tid; j < b0; j += nt) {
                double t = xv[j];
#pragma unroll
                for (int cc = 0; cc < NB; cc++) t -= Pk[tri(b0 + cc, j)] * xk[cc];
                xv[j] = t;
            }
            __syncthreads();
        }
        for (int j = tid; j < Cw; j += nt) y[mp.v(c0 + j)] = xs[j];
        __syncthreads();
        BAND_TS(tsw, 9);
        BAND_TS(wi == 0, 10);
    }
#undef BAND_TS
}

// Middle block of the twisted factorisation: S_mid = (A_mid + fwd updates) + (A_mid + bwd updates) - A_mid, likewise
// the right-hand side; solved in LDS; x_mid goes to y[M .. M + bw), where both backward passes pick it up.
template <int NB>
__global__ __launch_bounds__(SOLVE_THREADS) void k_band_mid(const double* __restrict__ A, long long ld, double* __restrict__ y, int N, int bw, int M,
                                                            const double* __restrict__ mid, int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* LpT = (double*)smem;
    double* Pk = LpT + (size_t)(bw + 2) * NBP;
    double* xv = Pk + (size_t)(bw + 1) * (bw + 2) / 2;
    double* xs = xv + bw;
    double* linvTab = xs + bw;
    const int ntri = tri(bw, 0), half = ntri + bw;
    for (int g = tid; g < ntri; g += nt) {
        int i, j;
        tri_decode(g, i, j);
        Pk[g] = mid[g] + mid[half + g] - A[(long long)(M + i) * ld + M + j];
    }
    for (int i = tid; i < bw; i += nt) Pk[tri(bw, i)] = mid[ntri + i] + mid[half + ntri + i] - y[M + i];
    if (tid == 0) Pk[tri(bw, bw)] = 0.0;
    __syncthreads();
    const bool ok = chol_solve_packed<NB, false>(Pk, bw, xv, xs, LpT, linvTab, nullptr);
    __syncthreads();
    if (!ok) { if (tid == 0) *info = M + 1; return; }
    for (int i = tid; i < bw; i += nt) y[M + i] = xs[i];
}

// ---- long bands: block cyclic reduction ------------------------------------------------------------------------------
// With blocks of b = bw columns the banded S is block tridiagonal (K = N / b diagonal blocks D_k, couplings E_k = rows of
// block k+1 x columns of block k). Cyclic reduction eliminates every other block of the active set at once: log2(K)
// levels instead of N / 6 sequential pivot steps. Eliminating block i with active neighbours p < i < n:
//     D_i = L L^T,  W_X = A_Xi L^-T (X = p, n),  y_i = L^-1 g_i
//     D_X -= W_X W_X^T,  g_X -= W_X y_i,  new coupling A'_np = -W_n W_p^T
// One workgroup per (i, X) runs the tuned LDS solver on the window [D_i ; A_Xi | 0 ; g_i | 0] (partial factorisation of
// the first b columns: the panel rows ARE W_X, the trailing block IS -W_X W_X^T, the rhs row holds y_i and -W_X y_i);
// k_bcr_combine adds the two updates an even block receives and forms the fill blocks; the last block is solved
// directly; the backward levels compute x_i = L^-T (y_i - W_p^T x_p - W_n^T x_n). Storage per block: b x b row-major.
struct BcrPtrs {
    double *D, *E, *g;          // active block-tridiagonal system (E_k: rows of the NEXT ACTIVE block x columns of k)
    double *Lp, *linv, *yv;     // per eliminated block: packed factor rows, inverse pivot blocks, y = L^-1 g
    double *Wp, *Wn;            // per eliminated block: W_p (rows p x cols i), W_n
    double *Ul, *Ur, *gl, *gr;  // per surviving block: update from the elimination of its right / left neighbour
    double *X;                  // solution blocks
    int K, b, N;
};

__global__ void k_bcr_extract(const double* __restrict__ S, long long ld, const double* __restrict__ y, BcrPtrs B, const int* skip, const int* info) {
    if ((skip && *skip) || *info != 0) return;
    const int k = blockIdx.x, b = B.b;
    const int r0 = k * b;
    double* D = B.D + (size_t)k * b * b;
    double* E = B.E + (size_t)k * b * b;
    for (int e = threadIdx.x; e < b * b; e += blockDim.x) {
        const int r = e / b, c = e - r * b;
        const int gr = r0 + r, gc = r0 + c;
        double v;
        if (gr < B.N && gc < B.N) v = (c <= r) ? S[(long long)gr * ld + gc] : S[(long long)gc * ld + gr];
        else v = (r == c) ? 1.0 : 0.0;   // identity padding of the last block
        D[e] = v;
        const int er = r0 + b + r;       // row of block k+1; columns of block k; outside the band the matrix is zero
        E[e] = (k + 1 < B.K && er < B.N && er - gc < b) ? S[(long long)er * ld + gc] : 0.0;
    }
    for (int r = threadIdx.x; r < b; r += blockDim.x) B.g[(size_t)k * b + r] = (r0 + r < B.N) ? y[r0 + r] : 0.0;
}

template <int NB>
__global__ __launch_bounds__(SOLVE_THREADS) void k_bcr_elim(BcrPtrs B, int s, int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((skip && *skip) || *info != 0) return;
    const int tid = threadIdx.x, nt = blockDim.x, b = B.b;
    const int i = (2 * blockIdx.x + 1) * s, X = blockIdx.y;   // X = 0: previous active block p, 1: next active block n
    const int p = i - s, n = i + s;
    if (i >= B.K) return;
    if (X == 1 && n >= B.K) return;
    const int R = 2 * b;
    double* LpT = (double*)smem;
    double* Pk = LpT + (size_t)(R + 2) * NBP;
    double* xv = Pk + (size_t)(R + 1) * (R + 2) / 2;
    double* xs = xv + R;
    double* linvTab = xs + R;
    const size_t bb = (size_t)b * b;
    const double* D = B.D + (size_t)i * bb;
    // rows 0 .. b-1: D_i (lower); rows b .. 2b-1: [A_Xi | 0]; row 2b: [g_i | 0]
    for (int g = tid; g < tri(R + 1, 0); g += nt) {
        int r, c;
        tri_decode(g, r, c);
        double v = 0.0;
        if (r < b) v = D[(size_t)r * b + c];
        else if (r < R) {
            if (c < b) v = X ? B.E[(size_t)i * bb + (size_t)(r - b) * b + c]      // A_ni = E_i (rows n, cols i)
                             : B.E[(size_t)p * bb + (size_t)c * b + (r - b)];     // A_pi = E_p^T (E_p: rows i, cols p)
        } else if (c < b) v = B.g[(size_t)i * b + c];
        Pk[g] = v;
    }
    __syncthreads();
    const bool ok = chol_solve_packed<NB, true>(Pk, R, xv, xs, LpT, linvTab, nullptr, b / NB);
    __syncthreads();
    if (!ok) { if (tid == 0) *info = i * b + 1; return; }
    double* W = (X ? B.Wn : B.Wp) + (size_t)i * bb;
    const int e_blk = X ? n : p;
    double* U = (X ? B.Ur : B.Ul) + (size_t)e_blk * bb;    // X = n: the update comes from n's LEFT neighbour -> Ur[n]; X = p: Ul[p]
    double* gu = (X ? B.gr : B.gl) + (size_t)e_blk * b;
    for (int e = tid; e < b * b; e += nt) {
        const int r = e / b, c = e - r * b;
        W[e] = Pk[tri(b + r, c)];
        U[e] = (c <= r) ? Pk[tri(b + r, b + c)] : Pk[tri(b + c, b + r)];
    }
    for (int r = tid; r < b; r += nt) gu[r] = Pk[tri(R, b + r)];
    if (X == 0) {   // p always exists for an eliminated block: this workgroup files the factor
        double* Lp = B.Lp + (size_t)i * (b * (b + 1) / 2);
        for (int g = tid; g < b * (b + 1) / 2; g += nt) Lp[g] = Pk[g];
        for (int e = tid; e < (b / NB) * NB * NB; e += nt) B.linv[(size_t)i * (b / NB) * NB * NB + e] = linvTab[e];
        for (int r = tid; r < b; r += nt) B.yv[(size_t)i * b + r] = Pk[tri(R, r)];
    }
}

// Survivors of a level absorb their updates; eliminated blocks with two neighbours leave the fill block A'_np = -W_n W_p^T.
__global__ __launch_bounds__(512) void k_bcr_combine(BcrPtrs B, int s, const int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((skip && *skip) || *info != 0) return;
    const int j = blockIdx.x, blk = j * s, b = B.b, tid = threadIdx.x, nt = blockDim.x;
    if (blk >= B.K) return;
    const size_t bb = (size_t)b * b;
    if ((j & 1) == 0) {
        double* D = B.D + (size_t)blk * bb;
        double* g = B.g + (size_t)blk * b;
        const bool hl = blk + s < B.K, hr = blk - s >= 0;
        for (int e = tid; e < b * b; e += nt) {
            double v = D[e];
            if (hl) v += B.Ul[(size_t)blk * bb + e];
            if (hr) v += B.Ur[(size_t)blk * bb + e];
            D[e] = v;
        }
        for (int r = tid; r < b; r += nt) {
            double v = g[r];
            if (hl) v += B.gl[(size_t)blk * b + r];
            if (hr) v += B.gr[(size_t)blk * b + r];
            g[r] = v;
        }
        return;
    }
    const int p = blk - s, n = blk + s;
    if (n >= B.K) return;
    double* Wn = (double*)smem;          // [b][b + 1] padded rows: conflict-free column walks
    double* Wp = Wn + (size_t)b * (b + 1);
    for (int e = tid; e < b * b; e += nt) {
        const int r = e / b, c = e - r * b;
        Wn[r * (b + 1) + c] = B.Wn[(size_t)blk * bb + e];
        Wp[r * (b + 1) + c] = B.Wp[(size_t)blk * bb + e];
    }
    __syncthreads();
    double* E = B.E + (size_t)p * bb;    // new coupling of the pair (p, n): rows n, columns p
    // E = -W_n W_p^T on the FP64 matrix cores: one 16 x 16 output tile per wave and round, K = b in steps of 4
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int ls = b + 1, nt16 = (b + 15) >> 4;
    const int wv = tid >> 6, ln = tid & 63, lr = ln & 15, lk = ln >> 4, nw = nt >> 6;
    for (int tile = wv; tile < nt16 * nt16; tile += nw) {
        const int ti = tile / nt16, tj = tile - ti * nt16;
        const int ar = 16 * ti + lr, br = 16 * tj + lr;
        const double* pa = Wn + min(ar, b - 1) * ls;
        const double* pb = Wp + min(br, b - 1) * ls;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < b; k0 += 4) {
            const int k = k0 + lk;
            const double av = pa[min(k, b - 1)], bv = pb[min(k, b - 1)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ar < b && k < b) ? av : 0.0, (br < b && k < b) ? bv : 0.0, acc, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = 16 * ti + lk + 4 * rg, col = 16 * tj + lr;
            if (row < b && col < b) E[(size_t)row * b + col] = -acc[rg];
        }
    }
}

// The last one or two active blocks (0 and s2 > 0) are solved together in LDS: [D_0 ; E_0 D_s2] is a dense system of
// at most 2 b <= 173 columns, cheaper than one more reduction level.
template <int NB>
__global__ __launch_bounds__(SOLVE_THREADS) void k_bcr_root(BcrPtrs B, int s2, int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((skip && *skip) || *info != 0) return;
    const int tid = threadIdx.x, nt = blockDim.x, b = B.b;
    const int n = s2 > 0 ? 2 * b : b;
    double* LpT = (double*)smem;
    double* Pk = LpT + (size_t)(n + 2) * NBP;
    double* xv = Pk + (size_t)(n + 1) * (n + 2) / 2;
    double* xs = xv + n;
    double* linvTab = xs + n;
    const size_t bb = (size_t)b * b;
    for (int g = tid; g < tri(n + 1, 0); g += nt) {
        int r, c;
        tri_decode(g, r, c);
        double v = 0.0;
        if (r < b) v = B.D[(size_t)r * b + c];
        else if (r < n) v = c < b ? B.E[(size_t)(r - b) * b + c] : B.D[(size_t)s2 * bb + (size_t)(r - b) * b + (c - b)];
        else if (c < n) v = c < b ? B.g[c] : B.g[(size_t)s2 * b + (c - b)];
        Pk[g] = v;
    }
    __syncthreads();
    const bool ok = chol_solve_packed<NB, false>(Pk, n, xv, xs, LpT, linvTab, nullptr);
    __syncthreads();
    if (!ok) { if (tid == 0) *info = 1; return; }
    for (int r = tid; r < n; r += nt) B.X[r < b ? (size_t)r : (size_t)s2 * b + (r - b)] = xs[r];
}

template <int NB>
__global__ __launch_bounds__(256) void k_bcr_back(BcrPtrs B, int s, const int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((skip && *skip) || *info != 0) return;
    const int tid = threadIdx.x, nt = blockDim.x, b = B.b;
    const int i = (2 * blockIdx.x + 1) * s, p = i - s, n = i + s;
    if (i >= B.K) return;
    const size_t bb = (size_t)b * b;
    double* Pk = (double*)smem;                       // packed factor rows of block i
    double* x = Pk + b * (b + 1) / 2;
    double* xs = x + b;
    double* linvTab = xs + b;
    double* xp = linvTab + (b / NB) * NB * NB;
    double* xn = xp + b;
    for (int g = tid; g < b * (b + 1) / 2; g += nt) Pk[g] = B.Lp[(size_t)i * (b * (b + 1) / 2) + g];
    for (int e = tid; e < (b / NB) * NB * NB; e += nt) linvTab[e] = B.linv[(size_t)i * (b / NB) * NB * NB + e];
    for (int r = tid; r < b; r += nt) { xp[r] = B.X[(size_t)p * b + r]; xn[r] = n < B.K ? B.X[(size_t)n * b + r] : 0.0; }
    __syncthreads();
    for (int c = tid; c < b; c += nt) {   // t = y_i - W_p^T x_p - W_n^T x_n
        double t = B.yv[(size_t)i * b + c];
        const double* Wp = B.Wp + (size_t)i * bb;
        for (int r = 0; r < b; r++) t -= Wp[(size_t)r * b + c] * xp[r];
        if (n < B.K) {
            const double* Wn = B.Wn + (size_t)i * bb;
            for (int r = 0; r < b; r++) t -= Wn[(size_t)r * b + c] * xn[r];
        }
        x[c] = t;
    }
    __syncthreads();
    block_backsub<NB>(Pk, b / NB, x, xs, linvTab);
    for (int r = tid; r < b; r += nt) B.X[(size_t)i * b + r] = xs[r];
}

__global__ void k_bcr_writeback(BcrPtrs B, double* __restrict__ y, const int* info, const int* skip) {
    if ((skip && *skip) || *info != 0) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < B.N) y[e] = B.X[e];   // X blocks are contiguous b-vectors: block k, row r sits at k * b + r
}

// ---- dense systems: wide panels built on the in-LDS factorisation -------------------------------------------------
// A dense reduced system (a dense prior couples the kept landmarks; N_p ~ 1 000) is factorised 96 columns at a time:
//   k_wchol_diag   one workgroup: the 96 x 96 diagonal block (+ its rhs entries) goes through the tuned LDS solver
//                  (chol_solve_packed<6, PARTIAL>, identity padding beyond N), then M = L_dd^-1 is formed from the
//                  inverse pivot blocks by block anti-diagonals and stored; z = L_dd^-1 y_d
//   k_wchol_trsm   X = A_panel M^T: one (rows x 96) x (96 x 96) product per 64-row tile on the FP64 matrix cores
//   k_wchol_syrk   A_trailing -= X X^T on 64 x 64 tiles with K = 96 (FP64 MFMA), rhs row y -= X z
//   k_wchol_backsolve   x_d = M^T (z_d - X_p^T x_p), super-steps in reverse, one workgroup
// 3 launches per 96 columns instead of 2 per 32, and the panel / update work runs on MFMA instead of per-thread
// substitution chains.
constexpr int WD = 96;
constexpr int WD_NBK = WD / 6;
constexpr int WDS = WD + 4;  // LDS row stride: (lane & 15) * 100 + (lane >> 4) hits every bank pair exactly twice

// stage `rows` x WD doubles from global (row stride ld) into an LDS slab [..][WDS]; rows beyond `valid` are zero.
// Eight independent loads are in flight per thread before the first LDS store.
__device__ __forceinline__ void stage_slab(double (*dst)[WDS], const double* __restrict__ src, long long ld, int rows, int valid) {
    const int total = rows * WD;
    for (int eb = threadIdx.x; eb < total; eb += 8 * blockDim.x) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = eb + u * blockDim.x;
            const int r = e / WD, cc = e - r * WD;
            v[u] = (e < total && r < valid) ? src[(long long)r * ld + cc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = eb + u * blockDim.x;
            if (e < total) dst[e / WD][e - (e / WD) * WD] = v[u];
        }
    }
}

__global__ __launch_bounds__(SOLVE_THREADS) void k_wchol_diag(double* __restrict__ A, long long ld, double* __restrict__ y,
                                                              double* __restrict__ Mg, int N, int c0, int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    constexpr int NB = 6;
    const int tid = threadIdx.x, nt = blockDim.x;
    double* LpT = (double*)smem;                         // [NBP][WD + 2]
    double* Pk = LpT + (size_t)(WD + 2) * NBP;           // packed window incl. rhs row
    double* xv = Pk + (size_t)(WD + 1) * (WD + 2) / 2;
    double* xs = xv + WD;
    double* linvTab = xs + WD;                           // [WD_NBK][36]
    double* Ml = linvTab + WD_NBK * 36;                  // [WD][WD + 1] inverse of the block factor
    double* Tt = Ml + (size_t)WD * (WD + 1);             // [WD_NBK][36] temporaries of one anti-diagonal
    // load (identity padding beyond N)
    for (int gb = tid; gb < tri(WD, 0); gb += 5 * nt) {   // five independent loads in flight per thread
        double v[5];
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int g = gb + u * nt;
            if (g < tri(WD, 0)) {
                int i, j;
                tri_decode(g, i, j);
                v[u] = (c0 + i < N) ? A[(long long)(c0 + i) * ld + c0 + j] : (i == j ? 1.0 : 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int g = gb + u * nt;
            if (g < tri(WD, 0)) Pk[g] = v[u];
        }
    }
    for (int j = tid; j < WD; j += nt) Pk[tri(WD, j)] = (c0 + j < N) ? y[c0 + j] : 0.0;
    if (tid == 0) Pk[tri(WD, WD)] = 0.0;
    __syncthreads();
    const bool ok = chol_solve_packed<NB, true>(Pk, WD, xv, xs, LpT, linvTab, nullptr, WD_NBK);
    __syncthreads();
    if (!ok) { if (tid == 0) *info = c0 + 1; return; }
    for (int j = tid; j < WD && c0 + j < N; j += nt) y[c0 + j] = Pk[tri(WD, j)];
    // M = L^-1 by block anti-diagonals: M_kk = Linv_k; M_kj = -Linv_k sum_{i=j}^{k-1} L_ki M_ij
    for (int e = tid; e < WD * (WD + 1); e += nt) Ml[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < WD_NBK * 36; e += nt) {
        const int k = e / 36, a = (e % 36) / 6, b = e % 6;
        Ml[(6 * k + a) * (WD + 1) + 6 * k + b] = linvTab[k * 36 + a * 6 + b];
    }
    __syncthreads();
    for (int d = 1; d < WD_NBK; d++) {
        const int nblk = WD_NBK - d;
        for (int e = tid; e < nblk * 36; e += nt) {
            const int j = e / 36, a = (e % 36) / 6, b = e % 6, k = j + d;
            // row 6k+a of L left of its pivot block is contiguous in the packed layout: columns 6j .. 6k-1
            const double* lrow = Pk + tri(6 * k + a, 6 * j);
            const double* mcol = Ml + (size_t)(6 * j) * (WD + 1) + 6 * j + b;
            double t0 = 0.0, t1 = 0.0;
            const int len = 6 * d;
#pragma unroll 6
            for (int q = 0; q < len; q += 2) {
                t0 += lrow[q] * mcol[(size_t)q * (WD + 1)];
                t1 += lrow[q + 1] * mcol[(size_t)(q + 1) * (WD + 1)];
            }
            Tt[e] = t0 + t1;
        }
        __syncthreads();
        for (int e = tid; e < nblk * 36; e += nt) {
            const int j = e / 36, a = (e % 36) / 6, b = e % 6, k = j + d;
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 6; q++) t += linvTab[k * 36 + a * 6 + q] * Tt[j * 36 + q * 6 + b];
            Ml[(6 * k + a) * (WD + 1) + 6 * j + b] = -t;
        }
        __syncthreads();
    }
    for (int e = tid; e < WD * WD; e += nt) Mg[e] = Ml[(e / WD) * (WD + 1) + (e % WD)];
}

// ---- the 96 x 96 diagonal block on 16 x 16 MFMA tiles (chol16.h) -------------------------------------------------------
// C(16 x 16) = sum over the listed (A, B) tile pairs of A B, tiles column-major in LDS (element (r, c) at c * 16 + r); the
// result comes back in the MFMA accumulator layout: register s of lane (lr = lane % 16, lk = lane / 16) = C[lk + 4 s][lr].
// TA: A is given transposed (the tile holds A^T).
template <bool TA>
__device__ __forceinline__ c16_d4 wd_tile_mma(const double* At, const double* Bt, c16_d4 acc, int lr, int lk) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int k = 4 * kk + lk;
        const double a = TA ? At[lr * 16 + k] : At[k * 16 + lr];   // A[row lr][col k]
        const double b = Bt[lr * 16 + k];                           // B[row k][col lr]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ void wd_tile_store(double* Ct, c16_d4 acc, int lr, int lk, double scale) {
#pragma unroll
    for (int s = 0; s < 4; s++) Ct[lr * 16 + lk + 4 * s] = scale * acc[s];
}

constexpr int WD_T = WD / 16;   // 6 tile rows
// LDS: image of the 96-column system + rhs row | chol16 exchange areas | M image (lower block triangle, 21 tiles) | one scratch tile per wave
__host__ __device__ constexpr size_t wd16_lds_doubles() {
    return (size_t)c16_size(WD) + WD + C16_WORK + 16 * (WD_T + 1) + (size_t)(WD_T * (WD_T + 1) / 2) * 256 + (size_t)(SOLVE_THREADS / 64) * 256;
}
__device__ __forceinline__ void wd16_factor_and_invert(double* Im, double* __restrict__ y, double* __restrict__ Mg, int N, int c0, int* info, double* __restrict__ Lt = nullptr);
__global__ __launch_bounds__(SOLVE_THREADS) void k_wchol_diag16(double* __restrict__ A, long long ld, double* __restrict__ y,
                                                                double* __restrict__ Mg, int N, int c0, int* info, const int* skip, double* __restrict__ Lt = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int tid = threadIdx.x;
    double* Im = (double*)smem;                    // tile-packed image, WD columns + the right-hand-side row
    // load: lower triangle of the block (identity beyond N), the rhs entries as row WD, zeros elsewhere
    for (int e = tid; e < c16_size(WD); e += SOLVE_THREADS) Im[e] = 0.0;
    __syncthreads();
    for (int gb = tid; gb < WD * (WD + 1) / 2; gb += 4 * SOLVE_THREADS) {   // four independent loads in flight per thread
        double v[4]; int ii[4], jj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = gb + u * SOLVE_THREADS;
            ii[u] = -1;
            if (g < WD * (WD + 1) / 2) {
                int i, j;
                tri_decode(g, i, j);
                ii[u] = i; jj[u] = j;
                v[u] = (c0 + i < N) ? A[(long long)(c0 + i) * ld + c0 + j] : (i == j ? 1.0 : 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (ii[u] >= 0) Im[c16_index(ii[u], jj[u])] = v[u];
    }
    for (int j = tid; j < WD; j += SOLVE_THREADS) Im[c16_index(WD, j)] = (c0 + j < N) ? y[c0 + j] : 0.0;
    __syncthreads();
    wd16_factor_and_invert(Im, y, Mg, N, c0, info, Lt);
}

// The factor of a diagonal block as the later launches take it (k_wchol_step): the 21 tiles of the image after c16_solve<., false> —
// panel tiles L_IJ, diagonal tiles L_JJ^-T with exact zeros below the diagonal — copied to Lt (WD_LT doubles). Returns "not finite".
constexpr int WD_LT = (WD_T * (WD_T + 1) / 2) * 256;
__device__ __forceinline__ bool wd16_store_tiles(const double* Im, double* __restrict__ Lt) {
    bool bad = false;
    for (int e = threadIdx.x; e < WD_LT; e += SOLVE_THREADS) {
        const int t = e >> 8, q = e & 255, r = q & 15, c = q >> 4;
        int I = 0, rem = t;
        while (rem >= I + 1) { rem -= I + 1; I++; }
        double v = Im[e];
        if (rem == I && r > c) v = 0.0;        // diagonal tile: L_II^-T is upper triangular (element (r, c) at c * 16 + r)
        if (!(fabs(v) < 1e300)) bad = true;    // a non-positive pivot turned into NaN / inf
        Lt[e] = v;
    }
    return bad;
}

// M = L^-1 (row-major WD x WD -> Mg) from the tiles of a factored block (Im: panel tiles L_IJ, diagonal tiles L_JJ^-T) by block
// anti-diagonals: M_II = L_II^-1 (the diagonal tile holds its transpose), M_IJ = -L_II^-1 sum_{K=J}^{I-1} L_IK M_KJ. Mi: 21 tiles of
// LDS, scr: one tile per wave. Every thread of the workgroup calls it.
__device__ __forceinline__ void wd16_invert(const double* Im, double* Mi, double* scr_base, double* __restrict__ Mg, int c0, int* info) {
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6, nwv = SOLVE_THREADS / 64;
    double* scr = scr_base + (size_t)wv * 256;
    const int lr = ln & 15, lk = ln >> 4;
    for (int e = tid; e < WD_T * 256; e += SOLVE_THREADS) {
        const int I = e >> 8, q = e & 255, r = q & 15, c = q >> 4;
        Mi[(c16_tile(I, I) << 8) + c * 16 + r] = r >= c ? Im[(c16_tile(I, I) << 8) + r * 16 + c] : 0.0;   // transpose, exact zeros above the diagonal
    }
    __syncthreads();
    for (int d = 1; d < WD_T; d++) {
        for (int J = wv; J + d < WD_T; J += nwv) {
            const int I = J + d;
            c16_d4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int K = J; K < I; K++) acc = wd_tile_mma<false>(Im + (c16_tile(I, K) << 8), Mi + (c16_tile(K, J) << 8), acc, lr, lk);
            wd_tile_store(scr, acc, lr, lk, 1.0);
            wave_lds_fence();
            c16_d4 m = {0.0, 0.0, 0.0, 0.0};
            m = wd_tile_mma<true>(Im + (c16_tile(I, I) << 8), scr, m, lr, lk);       // L_II^-1 T: the tile holds L_II^-T
            wd_tile_store(Mi + (c16_tile(I, J) << 8), m, lr, lk, -1.0);
            wave_lds_fence();
        }
        __syncthreads();
    }
    bool bad = false;
    for (int e = tid; e < WD * WD; e += SOLVE_THREADS) {
        const int i = e / WD, j = e - i * WD;
        const double v = j <= i ? Mi[(c16_tile(i >> 4, j >> 4) << 8) + (j & 15) * 16 + (i & 15)] : 0.0;
        if (!(fabs(v) < 1e300)) bad = true;    // a non-positive pivot turned into NaN / inf
        Mg[e] = v;
    }
    if (bad) *info = c0 + 1;
}

// The image of a diagonal block (lower halves of the tiles + the right-hand-side row) -> its factor in place, the forward-substituted
// right-hand side in y, M = L^-1 in Mg (if given), the factor's tiles in Lt (if given). The LDS carve behind the image is the one of
// k_wchol_diag16 (wd16_lds_doubles).
__device__ __forceinline__ void wd16_factor_and_invert(double* Im, double* __restrict__ y, double* __restrict__ Mg, int N, int c0, int* info, double* __restrict__ Lt) {
    const int tid = threadIdx.x;
    double* xs = Im + c16_size(WD);
    double* pub = xs + WD;
    double* yv = pub + C16_WORK;
    double* Mi = yv + 16 * (WD_T + 1);             // M = L^-1, tile (I, J), I >= J, at c16_tile(I, J) * 256
    double* scr = Mi + (size_t)(WD_T * (WD_T + 1) / 2) * 256;
    c16_symmetrize(Im, WD_T + 1);
    __syncthreads();
    (void)c16_solve<1, false>(Im, WD, xs, pub, yv, nullptr);
    // forward-substituted rhs: first rows of the tiles of row WD
    for (int j = tid; j < WD && c0 + j < N; j += SOLVE_THREADS) y[c0 + j] = Im[(c16_tile(WD_T, j >> 4) << 8) + (j & 15) * 16];
    if (Lt && wd16_store_tiles(Im, Lt)) *info = c0 + 1;
    if (Mg) wd16_invert(Im, Mi, scr, Mg, c0, info);
}

// X = A[s .. N, c0 .. c0 + WD) * M^T, 64 rows per workgroup
__global__ __launch_bounds__(CH_THREADS) void k_wchol_trsm(double* __restrict__ A, long long ld, const double* __restrict__ Mg,
                                                           int N, int c0, const int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int s = c0 + WD;
    const int r0 = s + blockIdx.x * CH_TS;
    if (r0 >= N) return;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    double (*Xa)[WDS] = (double (*)[WDS])smem;
    double (*Ms)[WDS] = (double (*)[WDS])(smem + sizeof(double) * CH_TS * WDS);
    stage_slab(Xa, A + (long long)r0 * ld + c0, ld, CH_TS, N - r0);
    stage_slab(Ms, Mg, WD, WD, WD);
    __syncthreads();
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lr = ln & 15, lk = ln >> 4;
    d4 acc[WD / 16];
#pragma unroll
    for (int cb = 0; cb < WD / 16; cb++) {
        d4 c4 = {0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < 16 * (cb + 1); kk += 16) {   // M is lower triangular: k <= column; 4 k-steps per batch of loads
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { av[u] = Xa[16 * wv + lr][kk + 4 * u + lk]; bv[u] = Ms[16 * cb + lr][kk + 4 * u + lk]; }
#pragma unroll
            for (int u = 0; u < 4; u++) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], c4, 0, 0, 0);
        }
        acc[cb] = c4;
    }
#pragma unroll
    for (int cb = 0; cb < WD / 16; cb++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = r0 + 16 * wv + lk + 4 * rg;
            if (row < N) A[(long long)row * ld + c0 + 16 * cb + lr] = acc[cb][rg];
        }
}

// The same with 512-thread workgroups for the look-ahead loop, where this kernel sits between two diagonal blocks: every global load of
// the two slabs (30 per thread) is in flight before the first wait, and a wave takes three of the six column blocks of its 16 rows.
__global__ __launch_bounds__(SOLVE_THREADS) void k_wchol_trsm8(double* __restrict__ A, long long ld, const double* __restrict__ Mg,
                                                               int N, int c0, const int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int s = c0 + WD;
    const int r0 = s + blockIdx.x * CH_TS;
    if (r0 >= N) return;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    double (*Xa)[WDS] = (double (*)[WDS])smem;
    double (*Ms)[WDS] = (double (*)[WDS])(smem + sizeof(double) * CH_TS * WDS);
    {
        constexpr int NXA = CH_TS * WD / SOLVE_THREADS, NM = WD * WD / SOLVE_THREADS;   // 12, 18
        double vx[NXA], vm[NM];
        const int valid = N - r0;
#pragma unroll
        for (int u = 0; u < NXA; u++) {
            const int e = tid + u * SOLVE_THREADS;
            const int r = e / WD, cc = e - r * WD;
            vx[u] = r < valid ? A[(long long)(r0 + r) * ld + c0 + cc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NM; u++) vm[u] = Mg[tid + u * SOLVE_THREADS];
#pragma unroll
        for (int u = 0; u < NXA; u++) { const int e = tid + u * SOLVE_THREADS; Xa[e / WD][e - (e / WD) * WD] = vx[u]; }
#pragma unroll
        for (int u = 0; u < NM; u++) { const int e = tid + u * SOLVE_THREADS; Ms[e / WD][e - (e / WD) * WD] = vm[u]; }
    }
    __syncthreads();
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lr = ln & 15, lk = ln >> 4;
    const int rb = wv & 3, cb0 = 3 * (wv >> 2);
    d4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int cb = cb0 + q;
        d4 c4 = {0.0, 0.0, 0.0, 0.0}, c5 = {0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < 16 * (cb + 1); kk += 16) {   // M is lower triangular: k <= column; 4 k-steps per batch of loads
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { av[u] = Xa[16 * rb + lr][kk + 4 * u + lk]; bv[u] = Ms[16 * cb + lr][kk + 4 * u + lk]; }
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], c5, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], c5, 0, 0, 0);
        }
        acc[q] = c4 + c5;
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = r0 + 16 * rb + lk + 4 * rg;
            if (row < N) A[(long long)row * ld + c0 + 16 * (cb0 + q) + lr] = acc[q][rg];
        }
}

// trailing update by the WD-wide panel X and rhs row; K = WD
__global__ __launch_bounds__(CH_THREADS) void k_wchol_syrk(double* __restrict__ A, long long ld, double* __restrict__ y, int N, int c0,
                                                           const int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int s = c0 + WD;
    const int m = N - s;
    if (m <= 0) return;
    const int nt = (m + CH_TS - 1) / CH_TS;
    const int npair = nt * (nt + 1) / 2;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    if ((int)blockIdx.x >= npair) {
        const int j = (blockIdx.x - npair) * CH_THREADS + tid;
        if (j < m) {
            const double* xrow = A + (long long)(s + j) * ld + c0;
            double acc = 0.0;
#pragma unroll 8
            for (int cc = 0; cc < WD; cc++) acc += y[c0 + cc] * xrow[cc];
            y[s + j] -= acc;
        }
        return;
    }
    double (*Pi)[WDS] = (double (*)[WDS])smem;
    double (*Pj)[WDS] = (double (*)[WDS])(smem + sizeof(double) * CH_TS * WDS);
    int ti = 0, rem = blockIdx.x;
    while (rem >= ti + 1) { rem -= ti + 1; ti++; }
    const int tj = rem;
    const int i0 = ti * CH_TS, j0 = tj * CH_TS;
    stage_slab(Pi, A + (long long)(s + i0) * ld + c0, ld, CH_TS, m - i0);
    stage_slab(Pj, A + (long long)(s + j0) * ld + c0, ld, CH_TS, m - j0);
    __syncthreads();
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lr = ln & 15, lk = ln >> 4;
    const int ib = wv;
    for (int jb = 0; jb < 4; jb++) {
        if (ti == tj && jb > ib) continue;
        const int rbase = i0 + 16 * ib, cbase = j0 + 16 * jb;
        if (rbase >= m || cbase >= m) continue;
        d4 c;
        long long addr[4];
        bool ok[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = rbase + lk + 4 * rg, col = cbase + lr;
            ok[rg] = row < m && col < m && col <= row;
            addr[rg] = (long long)(s + (row < m ? row : m - 1)) * ld + s + (col < m ? col : m - 1);
            c[rg] = ok[rg] ? A[addr[rg]] : 0.0;
        }
        for (int kk = 0; kk < WD; kk += 24) {   // 6 k-steps per batch of LDS loads
            double a[6], b[6];
#pragma unroll
            for (int u = 0; u < 6; u++) { a[u] = -Pi[16 * ib + lr][kk + 4 * u + lk]; b[u] = Pj[16 * jb + lr][kk + 4 * u + lk]; }
#pragma unroll
            for (int u = 0; u < 6; u++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], c, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
            if (ok[rg]) A[addr[rg]] = c[rg];
    }
}

// ---- the same back-substitution, one LAUNCH per super-step ------------------------------------------------------------------
// k_wchol_backsolve streams every panel of L through one CU (4.9 MB at N = 1 065: 8.5 us per step). Here step st is a launch of its
// own: workgroup 0 finishes block st - it applies the step before it, x_{st+1}, to its own 96 right-hand-side entries (one
// 96 x 96 block of L) and multiplies by M_st^T - while the other workgroups apply x_{st+1} to all EARLIER columns, 64 columns each.
// The updates of x_{st+2}, x_{st+3}, ... were applied by the launches before. Twelve launches of ~3.5 us instead of 102 us.
__global__ __launch_bounds__(SOLVE_THREADS) void k_wchol_backstep(const double* __restrict__ A, long long ld, double* __restrict__ y,
                                                                  const double* __restrict__ Mg_all, int N, int st, const int* info, const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int tid = threadIdx.x;
    const int nsteps = (N + WD - 1) / WD;
    const int c0 = st * WD, c1 = c0 + WD;              // this block | the rows of the step before it
    const bool have_prev = st + 1 < nsteps;
    const int nr1 = have_prev ? min(WD, N - c1) : 0;
    __shared__ double xp[WD], part[8][64];
    if (blockIdx.x > 0) {
        // y[cc] -= sum_r L[c1 + r][cc] x_{st+1}[r] for 64 earlier columns; eight row groups of 12
        if (tid < WD) xp[tid] = tid < nr1 ? y[c1 + tid] : 0.0;
        const int cc = ((int)blockIdx.x - 1) * 64 + (tid & 63), rg = tid >> 6;
        double v[12];
        const double* col = A + (long long)c1 * ld + min(cc, c0 - 1);
#pragma unroll
        for (int u = 0; u < 12; u++) { const int r = 12 * rg + u; v[u] = col[(long long)min(r, max(nr1 - 1, 0)) * ld]; }
        __syncthreads();
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < 12; u += 2) { a0 += v[u] * xp[12 * rg + u]; a1 += v[u + 1] * xp[12 * rg + u + 1]; }   // (xp = 0 beyond nr1)
        part[rg][tid & 63] = a0 + a1;
        __syncthreads();
        if (tid < 64 && cc < c0) y[cc] -= ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + ((part[4][tid] + part[5][tid]) + (part[6][tid] + part[7][tid]));
        return;
    }
    double (*Ms)[WDS] = (double (*)[WDS])smem;
    __shared__ double tv[WD], p4[4][WD];
    const int nr = min(WD, N - c0);
    // every load first: M_st (18 per thread), the block L[c1 .., c0 ..] for the near update (24 per thread of the first 384), the two
    // right-hand-side slices
    constexpr int MV = WD * WD / SOLVE_THREADS;
    double mreg[MV];
    {
        const double* Mg = Mg_all + (size_t)st * WD * WD;
#pragma unroll
        for (int q = 0; q < MV; q++) mreg[q] = Mg[tid + q * SOLVE_THREADS];
    }
    const int c = tid % WD, q4 = tid / WD;                 // q4 < 4 for the first 384 threads
    double lv[24];
#pragma unroll
    for (int u = 0; u < 24; u++) {
        const int r = q4 + 4 * u;
        lv[u] = A[(long long)(have_prev ? c1 + min(r, nr1 - 1) : c0) * ld + min(c0 + c, N - 1)];      // (clamped into the matrix: weighted by zero below; the last block has no rows behind it)
    }
    if (tid < WD) { xp[tid] = tid < nr1 ? y[c1 + tid] : 0.0; tv[tid] = tid < nr ? y[c0 + tid] : 0.0; }
#pragma unroll
    for (int q = 0; q < MV; q++) { const int e = tid + q * SOLVE_THREADS; Ms[e / WD][e % WD] = mreg[q]; }
    __syncthreads();
    if (tid < 4 * WD) {
        double a0 = 0.0, a1 = 0.0;
        if (have_prev) {
#pragma unroll
            for (int u = 0; u < 24; u += 2) { a0 += lv[u] * xp[q4 + 4 * u]; a1 += lv[u + 1] * xp[q4 + 4 * (u + 1)]; }
        }
        p4[q4][c] = a0 + a1;
    }
    __syncthreads();
    if (tid < WD) tv[tid] -= (p4[0][tid] + p4[1][tid]) + (p4[2][tid] + p4[3][tid]);
    __syncthreads();
    if (tid < 4 * WD) {
        double x = 0.0;                                    // (M^T t)_c = sum_{k >= c} M[k][c] t_k, k = q4, q4 + 4, ...
        for (int k = c + ((q4 - c) & 3); k < WD; k += 4) x += Ms[k][c] * tv[k];
        p4[q4][c] = x;
    }
    __syncthreads();
    if (tid < nr) y[c0 + tid] = (p4[0][tid] + p4[1][tid]) + (p4[2][tid] + p4[3][tid]);
}

// ---- trailing update + LOOK-AHEAD: the next diagonal block is factored inside the same launch ---------------------------------
// The panel loop diag -> trsm -> syrk is a chain of single-workgroup diagonal kernels (33 us each, half of the solve) with two wide
// kernels between them. Here the last workgroup of the trailing update takes the NEXT diagonal block: it applies this panel's update to
// it itself (X_next X_next^T from the panel rows the preceding k_wchol_trsm left, 96 x 96 x 96 on the matrix cores, and the right-hand
// side), then factors and inverts it (wd16_factor_and_invert) while the other workgroups update the rest of the trailing matrix;
// they leave the elements of that block and its right-hand-side rows alone. 512-thread workgroups: a 64 x 64 tile is 8 waves x 2
// sub-tiles. LDS: max(two 64 x 96 slabs, the image + the 96 x 96 slab of X_next, over which M is built later).
constexpr int WXS = 113;   // row stride of the transposed X_next slab of the look-ahead workgroup (see there)
__host__ __device__ constexpr size_t wdla_lds_doubles() { return wd16_lds_doubles() + (size_t)WD * WXS - (size_t)(WD_T * (WD_T + 1) / 2 + SOLVE_THREADS / 64) * 256; }
__global__ __launch_bounds__(SOLVE_THREADS) void k_wchol_syrk_la(double* __restrict__ A, long long ld, double* __restrict__ y, double* __restrict__ Mg_next,
                                                                 int N, int c0, int* info, const int* skip, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int s = c0 + WD;
    const int m = N - s;
    if (m <= 0) return;
    const int nt = (m + CH_TS - 1) / CH_TS;
    const int npair = nt * (nt + 1) / 2;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lr = ln & 15, lk = ln >> 4;
    const int bid = (int)blockIdx.x - 1;   // workgroup 0 is dispatched first: the look-ahead block, the longest chain of the launch
    if (bid < 0) {
        // ---- the next diagonal block [s, s + WD) ----
        double* Im = (double*)smem;
        // X_next TRANSPOSED, XT[panel column][block row], with the panel's right-hand side as block row WD (rows WD + 1 .. WD + 15 zero):
        // the MFMA fragments (16 consecutive block rows of one column per 16-lane group) are contiguous 128-byte reads, and the
        // right-hand-side update is the seventh tile row of the same product - as in chol16. Row stride WXS = 113 doubles: 16 consecutive
        // columns of one block row (what a wave stores from its unit-stride HBM reads) fall into 16 different bank pairs. (Row-major with
        // the stride of the trsm / syrk slabs the compiler's ds_read2_b64 pairs hit four bank pairs per 16 lanes: 9.4 us for the tile
        // updates.) M and the scratch tiles reuse the area later.
        double (*XT)[WXS] = (double (*)[WXS])(Im + c16_size(WD) + WD + C16_WORK + 16 * (WD_T + 1));
        if (dbg && tid == 0) dbg[0] = wall_clock64();
        // every global load of the prologue is issued before the first wait: X_next = the panel rows s .. s + WD (18 per thread, unit
        // stride), the panel's right-hand side, and this wave's tiles of the block itself straight in the accumulator layout (register q of
        // lane (lr, lk) = element (lk + 4 q, lr): 128-byte row segments), which spares a staging pass through LDS. Addresses are clamped
        // into the stored triangle and the values selected afterwards: no branch around a load.
        constexpr int NX = WD * WD / SOLVE_THREADS, NTL = WD_T * (WD_T + 1) / 2 + WD_T, NW = SOLVE_THREADS / 64, TPW = (NTL + NW - 1) / NW;
        double vx[NX];
#pragma unroll
        for (int u = 0; u < NX; u++) {
            const int e = tid + u * SOLVE_THREADS;
            const int r = e / WD, c = e - r * WD;
            const double v = A[(long long)min(s + r, N - 1) * ld + c0 + c];
            vx[u] = (s + r < N) ? v : 0.0;
        }
        const double vy = y[c0 + min(tid, WD - 1)];
        d4 acc0[TPW];
#pragma unroll
        for (int w = 0; w < TPW; w++) {
            const int t = min(wv + w * NW, NTL - 1);
            int I = 0, r = t;
            while (r >= I + 1) { r -= I + 1; I++; }          // t < 21: lower tile (I, J) of the block; t = 21 + J: the right-hand-side tile (WD_T, J)
            const int J = r;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * I + lk + 4 * q, j = 16 * J + lr;
                const double v = A[(long long)min(s + i, N - 1) * ld + min(s + min(j, i), N - 1)];
                const double vr = y[min(s + j, N - 1)];
                acc0[w][q] = I == WD_T ? ((lk + 4 * q == 0 && s + j < N) ? vr : 0.0) : (j <= i ? ((s + i < N) ? v : (i == j ? 1.0 : 0.0)) : 0.0);
            }
        }
        for (int e = tid; e < 256; e += SOLVE_THREADS) Im[(c16_tile(WD_T, WD_T) << 8) + e] = 0.0;   // the tile behind the right-hand side's last column block
        for (int e = tid; e < 15 * WD; e += SOLVE_THREADS) XT[e / 15][WD + 1 + e % 15] = 0.0;
#pragma unroll
        for (int u = 0; u < NX; u++) { const int e = tid + u * SOLVE_THREADS; XT[e - (e / WD) * WD][e / WD] = vx[u]; }
        if (tid < WD) XT[tid][WD] = vy;
        __syncthreads();
        if (dbg && tid == 0) dbg[1] = wall_clock64();
        // image tile (I, J) = block tile (right-hand-side tile) - sum_K X(I, K) X(J, K)^T: the 48 operand fragments of a tile are fetched
        // before its 24 MFMAs (two chains)
#pragma unroll
        for (int w = 0; w < TPW; w++) {
            const int t = wv + w * NW;
            if (t >= NTL) break;
            int I = 0, r = t;
            while (r >= I + 1) { r -= I + 1; I++; }
            const int J = r;
            double a[4 * WD_T], b[4 * WD_T];
#pragma unroll
            for (int k4 = 0; k4 < 4 * WD_T; k4++) { a[k4] = -XT[4 * k4 + lk][16 * I + lr]; b[k4] = XT[4 * k4 + lk][16 * J + lr]; }
            d4 acc = acc0[w], acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k4 = 0; k4 < 4 * WD_T; k4 += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4], b[k4], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4 + 1], b[k4 + 1], acc2, 0, 0, 0);
            }
            double* ct = Im + (c16_tile(I, J) << 8);
#pragma unroll
            for (int q = 0; q < 4; q++) ct[lr * 16 + lk + 4 * q] = acc[q] + acc2[q];   // register q of lane (lr, lk) = C[lk + 4 q][lr]
        }
        if (dbg && ln == 0) dbg[8 + wv] = wall_clock64();
        __syncthreads();
        if (dbg && tid == 0) dbg[4] = wall_clock64();
        wd16_factor_and_invert(Im, y, Mg_next, N, s, info);
        if (dbg && tid == 0) dbg[5] = wall_clock64();
        return;
    }
    if (bid >= npair) {
        const int j = (bid - npair) * SOLVE_THREADS + tid;
        if (j < m && j >= WD) {          // the first WD rows belong to the look-ahead workgroup
            const double* xrow = A + (long long)(s + j) * ld + c0;
            double acc = 0.0;
#pragma unroll 8
            for (int cc = 0; cc < WD; cc++) acc += y[c0 + cc] * xrow[cc];
            y[s + j] -= acc;
        }
        return;
    }
    double (*Pi)[WDS] = (double (*)[WDS])smem;
    double (*Pj)[WDS] = (double (*)[WDS])(smem + sizeof(double) * CH_TS * WDS);
    int ti = 0, rem = bid;
    while (rem >= ti + 1) { rem -= ti + 1; ti++; }
    const int tj = rem;
    const int i0 = ti * CH_TS, j0 = tj * CH_TS;
    stage_slab(Pi, A + (long long)(s + i0) * ld + c0, ld, CH_TS, m - i0);
    stage_slab(Pj, A + (long long)(s + j0) * ld + c0, ld, CH_TS, m - j0);
    __syncthreads();
    const int ib = wv & 3;
    for (int jb = 2 * (wv >> 2); jb < 2 * (wv >> 2) + 2; jb++) {
        if (ti == tj && jb > ib) continue;
        const int rbase = i0 + 16 * ib, cbase = j0 + 16 * jb;
        if (rbase >= m || cbase >= m) continue;
        d4 c;
        long long addr[4];
        bool ok[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = rbase + lk + 4 * rg, col = cbase + lr;
            ok[rg] = row < m && col < m && col <= row && !(row < WD && col < WD);   // (the look-ahead workgroup's block)
            addr[rg] = (long long)(s + (row < m ? row : m - 1)) * ld + s + (col < m ? col : m - 1);
            c[rg] = ok[rg] ? A[addr[rg]] : 0.0;
        }
        for (int kk = 0; kk < WD; kk += 24) {   // 6 k-steps per batch of LDS loads
            double a[6], b[6];
#pragma unroll
            for (int u = 0; u < 6; u++) { a[u] = -Pi[16 * ib + lr][kk + 4 * u + lk]; b[u] = Pj[16 * jb + lr][kk + 4 * u + lk]; }
#pragma unroll
            for (int u = 0; u < 6; u++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], c, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
            if (ok[rg]) A[addr[rg]] = c[rg];
    }
}

// ---- round 4: ONE launch per 96 columns ---------------------------------------------------------------------------------------------
// The round-3 loop was trsm8 -> syrk_la per panel: two dependent launches, and the look-ahead workgroup's chain was
// factor (13.7 us) + M = L^-1 (12 us) because the NEXT launch's panel product X = A M^T needed the explicit inverse. Here the panel
// solve is a block SUBSTITUTION against the factor's own tiles (wd_subst_strip: X_j = (A_j - sum_{q<j} X_q L_jq^T) L_jj^-T on 16 x 16
// MFMA tiles, L_jj^-T being what chol16 leaves in the diagonal tile), so
//   * the inverse leaves the chain: M (which only the back-substitution reads) is built by a spare workgroup of the FOLLOWING launch,
//   * the panel solve needs no launch of its own: every tile workgroup substitutes the two 64-row strips of the panel it is about to
//     contract (redundantly - 136 workgroups on 256 CUs - instead of waiting for a launch that does it once),
//   * the substituted panel goes OUT of place (Lx), because other workgroups still read the unsubstituted rows.
// Launch st (columns c0 = 96 st): workgroup 0 = look-ahead (substitute the next block's 96 panel rows, update the block, factor it,
// store its tiles for launch st + 1; the last block is inverted on the spot), workgroups 1 .. npair = 64 x 64 tiles of the trailing
// update (the diagonal ones also write their strip of X and apply it to the right-hand side), the last workgroup = M_st.
// strip element (r < 16, c < WD) at `at(r, c)`; Lt: the factor's tiles in LDS
template <class At>
__device__ __forceinline__ void wd_subst_strip(const double* Lt, At at, int lr, int lk) {
#pragma unroll
    for (int j = 0; j < WD_T; j++) {
        c16_d4 t, t2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int rg = 0; rg < 4; rg++) t[rg] = *at(lk + 4 * rg, 16 * j + lr);
#pragma unroll
        for (int q = 0; q < j; q++) {
            const double* Lq = Lt + (c16_tile(j, q) << 8);
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = -*at(lr, 16 * q + 4 * u + lk); b[u] = Lq[(4 * u + lk) * 16 + lr]; }   // X_q[row][k], L_jq[col][k]
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], t, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], t2, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], t, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], t2, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++) *at(lk + 4 * rg, 16 * j + lr) = t[rg] + t2[rg];
        wave_lds_fence();
        const double* Ld = Lt + (c16_tile(j, j) << 8);
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = *at(lr, 16 * j + 4 * u + lk); b[u] = Ld[lr * 16 + 4 * u + lk]; }          // T[row][k], (L_jj^-T)[k][col]
        c16_d4 x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], x2, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], x, 0, 0, 0);
        x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], x2, 0, 0, 0);
        wave_lds_fence();
#pragma unroll
        for (int rg = 0; rg < 4; rg++) *at(lk + 4 * rg, 16 * j + lr) = x[rg] + x2[rg];
        wave_lds_fence();
    }
}

__host__ __device__ constexpr size_t wdstep_lds_doubles() {
    return wdla_lds_doubles() > (size_t)2 * CH_TS * WDS + WD_LT ? wdla_lds_doubles() : (size_t)2 * CH_TS * WDS + WD_LT;
}
__global__ __launch_bounds__(SOLVE_THREADS) void k_wchol_step(double* __restrict__ A, long long ld, double* __restrict__ Lx, double* __restrict__ y,
                                                              const double* __restrict__ Lt_cur, double* __restrict__ Lt_next, double* __restrict__ Mg_cur,
                                                              double* __restrict__ Mg_next, int N, int c0, int* info, const int* skip, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    const int s = c0 + WD;
    const int m = N - s;
    if (m <= 0) return;
    const int nt = (m + CH_TS - 1) / CH_TS;
    const int npair = nt * (nt + 1) / 2;
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lr = ln & 15, lk = ln >> 4;
    const int bid = (int)blockIdx.x - 1;   // workgroup 0 is dispatched first: the look-ahead block, the longest chain of the launch
    long long* dts = dbg && tid == 0 ? (bid < 0 ? dbg : bid == 7 ? dbg + 8 : bid == npair ? dbg + 14 : nullptr) : nullptr;   // SADVIO_DEBUG & 4096: phase stamps
    if (dts) dts[0] = wall_clock64();
    constexpr int NLT = (WD_LT + SOLVE_THREADS - 1) / SOLVE_THREADS;   // 11 loads per thread bring the factor's tiles
    if (bid < 0) {
        // ---- the next diagonal block [s, s + WD) ----
        double* Im = (double*)smem;
        double (*XT)[WXS] = (double (*)[WXS])(Im + c16_size(WD) + WD + C16_WORK + 16 * (WD_T + 1));   // transposed slab, see k_wchol_syrk_la
        // NTL: the block's 21 lower tiles, updated on the matrix cores. The right-hand side's tile row (one useful row of 16 per tile: 144 of
        // 648 MFMAs when it rode the same product) is 96 dot products on the VALU, by the last 96 threads - waves that hold two tiles
        // where the others hold three.
        constexpr int NX = WD * WD / SOLVE_THREADS, NTL = WD_T * (WD_T + 1) / 2, NW = SOLVE_THREADS / 64, TPW = (NTL + NW - 1) / NW;
        double vx[NX], vl[NLT];
#pragma unroll
        for (int u = 0; u < NLT; u++) { const int e = tid + u * SOLVE_THREADS; vl[u] = Lt_cur[min(e, WD_LT - 1)]; }
#pragma unroll
        for (int u = 0; u < NX; u++) {
            const int e = tid + u * SOLVE_THREADS;
            const int r = e / WD, c = e - r * WD;
            const double v = A[(long long)min(s + r, N - 1) * ld + c0 + c];
            vx[u] = (s + r < N) ? v : 0.0;
        }
        const double vy = y[c0 + min(tid, WD - 1)];
        const int jr = tid - (SOLVE_THREADS - WD);                       // >= 0: this thread's column of the right-hand side
        const double vrhs = y[min(s + max(jr, 0), N - 1)];
        d4 acc0[TPW];
#pragma unroll
        for (int w = 0; w < TPW; w++) {
            const int t = min(wv + w * NW, NTL - 1);
            int I = 0, r = t;
            while (r >= I + 1) { r -= I + 1; I++; }          // lower tile (I, J) of the block
            const int J = r;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * I + lk + 4 * q, j = 16 * J + lr;
                const double v = A[(long long)min(s + i, N - 1) * ld + min(s + min(j, i), N - 1)];
                acc0[w][q] = j <= i ? ((s + i < N) ? v : (i == j ? 1.0 : 0.0)) : 0.0;
            }
        }
        for (int e = tid; e < 256; e += SOLVE_THREADS) Im[(c16_tile(WD_T, WD_T) << 8) + e] = 0.0;   // the tile behind the right-hand side's last column block
        for (int e = tid; e < 15 * WD; e += SOLVE_THREADS) XT[e / 15][WD + 1 + e % 15] = 0.0;
#pragma unroll
        for (int u = 0; u < NLT; u++) { const int e = tid + u * SOLVE_THREADS; if (e < WD_LT) Im[e] = vl[u]; }   // the factor's tiles sit in the image area until the update
#pragma unroll
        for (int u = 0; u < NX; u++) { const int e = tid + u * SOLVE_THREADS; XT[e - (e / WD) * WD][e / WD] = vx[u]; }
        if (tid < WD) XT[tid][WD] = vy;
        __syncthreads();
        if (dts) dts[1] = wall_clock64();
        if (wv < WD_T) wd_subst_strip(Im, [&](int r, int c) { return &XT[c][16 * wv + r]; }, lr, lk);   // the block's 96 panel rows -> X_next
        __syncthreads();
        if (dts) dts[2] = wall_clock64();
#pragma unroll
        for (int w = 0; w < TPW; w++) {
            const int t = wv + w * NW;
            if (t >= NTL) break;
            int I = 0, r = t;
            while (r >= I + 1) { r -= I + 1; I++; }
            const int J = r;
            double a[4 * WD_T], b[4 * WD_T];
#pragma unroll
            for (int k4 = 0; k4 < 4 * WD_T; k4++) { a[k4] = -XT[4 * k4 + lk][16 * I + lr]; b[k4] = XT[4 * k4 + lk][16 * J + lr]; }
            d4 acc = acc0[w], acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k4 = 0; k4 < 4 * WD_T; k4 += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4], b[k4], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k4 + 1], b[k4 + 1], acc2, 0, 0, 0);
            }
            double* ct = Im + (c16_tile(I, J) << 8);     // (the factor's tiles that sat here were last read before the barrier above)
#pragma unroll
            for (int q = 0; q < 4; q++) ct[lr * 16 + lk + 4 * q] = acc[q] + acc2[q];   // register q of lane (lr, lk) = C[lk + 4 q][lr]
        }
        if (jr >= 0) {
            // rhs_next[j] = rhs[s + j] - sum_k z_k X[j][k]: row 0 of the tile (WD_T, j / 16) of the image, the tile's other rows zero
            double v0 = s + jr < N ? vrhs : 0.0, v1 = 0.0;
#pragma unroll 8
            for (int k = 0; k < WD; k += 2) {
                v0 = __builtin_fma(-XT[k][WD], XT[k][jr], v0);
                v1 = __builtin_fma(-XT[k + 1][WD], XT[k + 1][jr], v1);
            }
            double* ct = Im + (c16_tile(WD_T, jr >> 4) << 8) + (jr & 15) * 16;
            ct[0] = v0 + v1;
#pragma unroll
            for (int r = 1; r < 16; r++) ct[r] = 0.0;
        }
        __syncthreads();
        if (dts) dts[3] = wall_clock64();
        const bool last = s + WD >= N;
        wd16_factor_and_invert(Im, y, last ? Mg_next : (double*)nullptr, N, s, info, Lt_next);
        if (dts) dts[4] = wall_clock64();
        return;
    }
    if (bid == npair) {
        // ---- M of THIS panel's block, for the back-substitution ----
        if (!Mg_cur) return;       // (factor only: sadvio_ba_marginalize)
        double* Ls = (double*)smem;
        double* Mi = Ls + WD_LT;
        double* scr = Mi + WD_LT;
        for (int e = tid; e < WD_LT; e += SOLVE_THREADS) Ls[e] = Lt_cur[e];
        __syncthreads();
        wd16_invert(Ls, Mi, scr, Mg_cur, c0, info);
        if (dts) dts[1] = wall_clock64();
        return;
    }
    // ---- tile (ti, tj) of the trailing matrix ----
    double (*Pi)[WDS] = (double (*)[WDS])smem;
    double (*Pj)[WDS] = (double (*)[WDS])(smem + sizeof(double) * CH_TS * WDS);
    double* Ls = (double*)(smem + sizeof(double) * 2 * CH_TS * WDS);
    int ti = 0, rem = bid;
    while (rem >= ti + 1) { rem -= ti + 1; ti++; }
    const int tj = rem;
    const int i0 = ti * CH_TS, j0 = tj * CH_TS;
    {
        constexpr int NP = CH_TS * WD / SOLVE_THREADS;   // 12
        double vi[NP], vj[NP], vl[NLT];
#pragma unroll
        for (int u = 0; u < NLT; u++) { const int e = tid + u * SOLVE_THREADS; vl[u] = Lt_cur[min(e, WD_LT - 1)]; }
#pragma unroll
        for (int u = 0; u < NP; u++) {
            const int e = tid + u * SOLVE_THREADS;
            const int r = e / WD, c = e - r * WD;
            const double a = A[(long long)(s + min(i0 + r, m - 1)) * ld + c0 + c];
            const double b = A[(long long)(s + min(j0 + r, m - 1)) * ld + c0 + c];
            vi[u] = i0 + r < m ? a : 0.0;
            vj[u] = j0 + r < m ? b : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NLT; u++) { const int e = tid + u * SOLVE_THREADS; if (e < WD_LT) Ls[e] = vl[u]; }
#pragma unroll
        for (int u = 0; u < NP; u++) { const int e = tid + u * SOLVE_THREADS; const int r = e / WD, c = e - r * WD; Pi[r][c] = vi[u]; Pj[r][c] = vj[u]; }
    }
    __syncthreads();
    if (dts) dts[1] = wall_clock64();
    {
        const int rb = wv & 3;
        if (wv < 4) wd_subst_strip(Ls, [&](int r, int c) { return &Pi[16 * rb + r][c]; }, lr, lk);
        else if (ti != tj) wd_subst_strip(Ls, [&](int r, int c) { return &Pj[16 * rb + r][c]; }, lr, lk);
    }
    __syncthreads();
    if (dts) dts[2] = wall_clock64();
    if (ti == tj) Pj = Pi;
    const int ib = wv & 3;
    for (int jb = 2 * (wv >> 2); jb < 2 * (wv >> 2) + 2; jb++) {
        if (ti == tj && jb > ib) continue;
        const int rbase = i0 + 16 * ib, cbase = j0 + 16 * jb;
        if (rbase >= m || cbase >= m) continue;
        d4 c;
        long long addr[4];
        bool ok[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const int row = rbase + lk + 4 * rg, col = cbase + lr;
            ok[rg] = row < m && col < m && col <= row && !(row < WD && col < WD);   // (the look-ahead workgroup's block)
            addr[rg] = (long long)(s + (row < m ? row : m - 1)) * ld + s + (col < m ? col : m - 1);
            c[rg] = ok[rg] ? A[addr[rg]] : 0.0;
        }
        for (int kk = 0; kk < WD; kk += 24) {   // 6 k-steps per batch of LDS loads
            double a[6], b[6];
#pragma unroll
            for (int u = 0; u < 6; u++) { a[u] = -Pi[16 * ib + lr][kk + 4 * u + lk]; b[u] = Pj[16 * jb + lr][kk + 4 * u + lk]; }
#pragma unroll
            for (int u = 0; u < 6; u++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], c, 0, 0, 0);
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
            if (ok[rg]) A[addr[rg]] = c[rg];
    }
    if (ti == tj) {
        // this strip of the substituted panel -> Lx, and its share of the right-hand side (rows of the look-ahead block: that workgroup's)
        for (int e = tid; e < CH_TS * WD; e += SOLVE_THREADS) {
            const int r = e / WD, c = e - r * WD;
            if (i0 + r < m) Lx[(long long)(s + i0 + r) * ld + c0 + c] = Pi[r][c];
        }
        const int r = tid >> 3, part = tid & 7;      // 64 rows x 8 column groups of 12
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < WD / 8; u++) { const int c = part + 8 * u; acc += Pi[r][c] * y[c0 + c]; }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (part == 0 && i0 + r < m && i0 + r >= WD) y[s + i0 + r] -= acc;
    }
    if (dts) dts[3] = wall_clock64();
}

// x = L^-T z in place: super-steps in reverse, one workgroup. Right-looking: once the 96 unknowns x_d of a step are
// known (x_d = M^T t_d), every earlier right-hand-side entry is updated, y[c] -= sum_r X[d rows r][c] x_d[r] -- row
// reads of X, unit stride across the threads.
__global__ __launch_bounds__(SOLVE_THREADS) void k_wchol_backsolve(const double* __restrict__ A, long long ld, double* __restrict__ y,
                                                                   const double* __restrict__ Mg_all, int N, const int* info,
                                                                   const int* skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (skip && *skip) return;
    if (*info != 0) return;
    double (*Ms)[WDS] = (double (*)[WDS])smem;
    __shared__ double tv[WD], xd[WD], part[4][WD];
    const int tid = threadIdx.x;
    const int nsteps = (N + WD - 1) / WD;
    // M of a step is fetched into registers while the previous step's update streams its panel (18 loads per thread, issued
    // before the update loop, stored to LDS behind it); the 96-long dot products of x_d = M^T t_d are split over four threads;
    // the update keeps 24 panel rows in flight per thread instead of 8 (the kernel is one workgroup: its speed is the number
    // of loads it keeps outstanding).
    constexpr int MV = (WD * WD + SOLVE_THREADS - 1) / SOLVE_THREADS;
    double mreg[MV];
    auto fetch_m = [&](int st) {
        const double* Mg = Mg_all + (size_t)st * WD * WD;
#pragma unroll
        for (int q = 0; q < MV; q++) { const int e = tid + q * SOLVE_THREADS; mreg[q] = e < WD * WD ? Mg[e] : 0.0; }
    };
    auto store_m = [&]() {
#pragma unroll
        for (int q = 0; q < MV; q++) { const int e = tid + q * SOLVE_THREADS; if (e < WD * WD) Ms[e / WD][e % WD] = mreg[q]; }
    };
    fetch_m(nsteps - 1);
    for (int st = nsteps - 1; st >= 0; st--) {
        const int c0 = st * WD;
        const int nr = min(WD, N - c0);
        store_m();
        if (tid < WD) tv[tid] = tid < nr ? y[c0 + tid] : 0.0;
        __syncthreads();
        if (st > 0) fetch_m(st - 1);       // in flight under the dot products and the update below
        if (tid < 4 * WD) {
            const int c = tid % WD, q4 = tid / WD;      // (M^T t)_c = sum_{k >= c} M[k][c] t_k, k = q4, q4 + 4, ...
            double x = 0.0;
            for (int k = c + ((q4 - c) & 3); k < WD; k += 4) x += Ms[k][c] * tv[k];
            part[q4][c] = x;
        }
        __syncthreads();
        if (tid < WD) {
            const double x = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
            xd[tid] = x;
            if (tid < nr) y[c0 + tid] = x;
        }
        __syncthreads();
        for (int cc = tid; cc < c0; cc += SOLVE_THREADS) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            const double* col = A + (long long)c0 * ld + cc;
            int r = 0;
            for (; r + 24 <= nr; r += 24) {
                double v[24];
#pragma unroll
                for (int u = 0; u < 24; u++) v[u] = col[(long long)(r + u) * ld];
#pragma unroll
                for (int u = 0; u < 24; u += 3) { a0 += v[u] * xd[r + u]; a1 += v[u + 1] * xd[r + u + 1]; a2 += v[u + 2] * xd[r + u + 2]; }
            }
            for (; r < nr; r++) a0 += col[(long long)r * ld] * xd[r];
            y[cc] -= (a0 + a1) + a2;
        }
        __syncthreads();
    }
}

// ---- the wide-panel factor as a prior (sadvio_ba_marginalize, Cholesky form, full-rank Ak) ------------------------------------------
// k_wchol_diag16 + k_wchol_step leave L in three places: the panels below the 96 x 96 diagonal blocks (Lx, row-major), the tiles of
// the diagonal blocks (Lt: L_IK for I > K; the diagonal tiles hold U = L_II^-T), and z = L^-1 rhs in y. k_wfac_diag inverts the
// diagonal tiles back (L_II = (U^-1)^T, 16 threads per tile) and tests the pivots; k_wfac_pack writes J = L^T, r0 = -z.
// flag |= 1: a pivot is not finite, not above tau, or below safe_rel of its original diagonal entry (the caller falls back to the
// pivoted factorisation).
__global__ void k_diag_max(const double* __restrict__ A, long long ld, int n, double* __restrict__ out) {      // out[0] = max_i A[i][i], one workgroup
    __shared__ double sh[256];
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, A[(size_t)i * ld + i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = blockDim.x >> 1; s2 > 0; s2 >>= 1) { if ((int)threadIdx.x < s2) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s2]); __syncthreads(); }
    if (threadIdx.x == 0) out[0] = sh[0];
}
__global__ void k_iota(int* __restrict__ v, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = i; }
// tau_rel >= 0: tau = tau_rel * dmax[0], else tau = -tau_rel (the pivoted factorisation's convention)
__global__ __launch_bounds__(64) void k_wfac_diag(const double* __restrict__ Ltw, int N, const double* __restrict__ A0, long long ld0, double tau_rel,
                                                  const double* __restrict__ dmax, double safe_rel, double* __restrict__ Ld, int* __restrict__ flag) {
    const double tau = tau_rel >= 0.0 ? tau_rel * dmax[0] : -tau_rel;
    const int b = blockIdx.x / WD_T, I = blockIdx.x - b * WD_T, c = threadIdx.x;
    const double* U = Ltw + (size_t)b * WD_LT + ((size_t)c16_tile(I, I) << 8);   // element (r, c) at c * 16 + r, upper triangular
    double* out = Ld + (size_t)blockIdx.x * 256;
    if (c >= 16) return;
    const int g = b * WD + 16 * I + c;          // global index of this thread's column
    // column c of U^-1: x[c] = 1 / U[c][c], x[r] = -(sum_{q = r + 1 .. c} U[r][q] x[q]) / U[r][r]
    double x[16];
#pragma unroll
    for (int r = 15; r >= 0; r--) {
        double acc = r == c ? 1.0 : 0.0;
#pragma unroll
        for (int q = 15; q > r; q--) acc -= (q <= c) ? U[q * 16 + r] * x[q] : 0.0;
        x[r] = r <= c ? acc / U[r * 16 + r] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) out[c * 16 + k] = x[k];       // L_II[i = c][k] = (U^-1)[k][c]
    if (g < N) {
        const double lkk = 1.0 / U[c * 16 + c], piv = lkk * lkk, a0 = A0[(long long)g * ld0 + g];
        if (!(piv > tau && piv >= safe_rel * a0 && piv < 1e300 && lkk > 0.0)) atomicOr(flag, 1);
    }
}

__global__ void k_wfac_pack(const double* __restrict__ Lx, long long ld, const double* __restrict__ Ltw, const double* __restrict__ Ld, const double* z, int N,
                            double* __restrict__ J, double* r0) {      // (r0 may be z: entry k is read and written by one thread)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * N) return;
    const int k = (int)(idx / N), i = (int)(idx - (long long)k * N);     // J[k][i] = L[i][k]
    double v = 0.0;
    if (i >= k) {
        const int bi = i / WD, bk = k / WD;
        if (bi > bk) v = Lx[(long long)i * ld + k];
        else {
            const int I = (i - bi * WD) >> 4, K = (k - bk * WD) >> 4;
            v = I > K ? Ltw[(size_t)bi * WD_LT + ((size_t)c16_tile(I, K) << 8) + (k & 15) * 16 + (i & 15)] : Ld[((size_t)bi * WD_T + I) * 256 + (i & 15) * 16 + (k & 15)];
        }
    }
    J[idx] = v;
    if (i == 0) r0[k] = -z[k];
}

}  // namespace sadvio
