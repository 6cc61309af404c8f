// chol16.h — in-LDS Cholesky solve of the reduced pose system (N <= 174) on 16x16 tiles, one 512-thread workgroup.
//
// Layout ("tile-packed"): the lower block triangle of the (N + 1) x (N + 1) matrix [S ; rhs^T] in 16 x 16 tiles,
// tile (I, J), I >= J, at ((I (I + 1) / 2 + J) * 256 doubles, COLUMN-major inside the tile (element (r, c) at c * 16 + r).
// A wave holds a tile as 4 registers per lane: lane l = 16 k + i, register s <-> element (i, 4 s + k) ("X layout") — the
// register s of a tile is 64 consecutive doubles (conflict-free ds_read_b64 / ds_write_b64), and it is at the same time
//   * the A / B operand of v_mfma_f64_16x16x4_f64 for the 4 columns 4 s .. 4 s + 3 of the tile, and
//   * the accumulator layout of the TRANSPOSED tile (D[row = k + 4 s][col = i], measured: scripts/probe/uarch_probe.hip),
// so  X(C) += mfma(A = X(P)[s], B = X(Q)[s])  accumulates  C += Q P^T  with no data movement between products, and a
// SYMMETRIC tile's register s is directly the operand of its own rank-4 update.
//
// Factorisation (right-looking, block columns of 16, 4-column steps inside a block):
//   pivot wave (wave 0)  one diagonal tile D in registers. Per 4-column step: the 4 x 4 pivot block goes to SGPRs
//                        (v_readlane), its Cholesky factor is computed uniformly (v_rsq_f64 + 2 Newton steps per column:
//                        65 cycles, the dependency floor), every lane gets the 4 entries of its row with three
//                        v_permlane{16,32}_swap pairs and forward-substitutes them (y = a L_ss^-T), D -= y y^T is ONE MFMA.
//                        The step's y (= 4 columns of L_kk) and the 10 numbers of the 4 x 4 factor are published in LDS.
//   all waves            "replay" the same 4 steps on the panel tiles below (A_Ik -> L_Ik = A_Ik L_kk^-T, one MFMA per step)
//                        and on an identity tile (-> L_kk^-T, kept in the dead diagonal tile for the back-substitution);
//   bulk waves           trailing update C_IJ -= L_Ik L_Jk^T, 4 MFMAs per tile, while the pivot wave updates and factors
//                        the next diagonal tile (look-ahead).
// Two workgroup barriers per block column (8 for N = 114 instead of 38 with 6-column blocks). The right-hand side is row N
// of the matrix (forward substitution for free); columns >= N of the last tile are dummy pivots (inverse 0: no effect).
// Back-substitution: thread c owns y_c; per block, x_J = L_JJ^-T v_J by 16 lanes, y_c -= L_Jc^T x_J by everyone.
#pragma once
#include <hip/hip_runtime.h>

namespace sadvio {

typedef double c16_d4 __attribute__((ext_vector_type(4)));

constexpr int C16_STEP = 64;             // published per 4-column step: y = the step's 4 columns of L_kk (the back-substitution reads the rhs row's)
constexpr int C16_PUB = 4 * C16_STEP;    // per block column; double-buffered by block parity
constexpr int C16_WT = 16 * 17;          // L_kk^-T of the current block, element (r, c) at c * 17 + r (read transposed without bank conflicts)
#ifndef C16_ALL_TILES
#define C16_ALL_TILES 4
#endif
constexpr int C16_PAIR_TILES = 7;        // GATHER == 3: the pivot / helper pair runs while the trailing update is at most this many tiles per bulk wave
constexpr int C16_GSLOTS = 4 * 64;       // the pivot wave's gather buffer, one 64-double slot per 4-column step (the helper wave reads them too)
constexpr int C16_WORK = 2 * C16_PUB + C16_GSLOTS + C16_WT + 8;   // doubles of the exchange area (`pub`); the last 8: the two step counters of the pivot / helper pair

__host__ __device__ constexpr int c16_tile(int I, int J) { return ((I * (I + 1)) >> 1) + J; }
// element (i, j), i >= j, of the tile-packed lower triangle (diagonal tiles: the lower half; see c16_symmetrize)
__host__ __device__ constexpr int c16_index(int i, int j) { return (c16_tile(i >> 4, j >> 4) << 8) + ((j & 15) << 4) + (i & 15); }
__host__ __device__ constexpr int c16_blocks(int n_rows) { return (n_rows + 15) >> 4; }
// doubles of the image of an N-column system (+ the right-hand-side row)
__host__ __device__ constexpr int c16_size(int N) { return (c16_blocks(N + 1) * (c16_blocks(N + 1) + 1) / 2) << 8; }

__device__ __forceinline__ double c16_readlane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double c16_rsqrt(double d) {   // v_rsq_f64 (2^-24) + ONE third-order (Halley) step: e = 1 - d y^2,
    double y = __builtin_amdgcn_rsq(d);                    // y <- y (1 + e / 2 + 3 e^2 / 8): error O(e^3) ~ 1e-22 before rounding;
    const double t = d * y;                                // five dependent operations instead of the six of two Newton steps
    const double e = __builtin_fma(-t, y, 1.0);
    const double p = __builtin_fma(0.375, e, 0.5);
    const double q = e * p;
    return __builtin_fma(y, q, y);
}
// (even-row member, odd-row member) of the lane pair {l, l ^ 16} in both lanes; likewise (lower, upper) of {l, l ^ 32}
__device__ __forceinline__ void c16_pair16(double v, double& e, double& o) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    e = __hiloint2double(b[0], a[0]); o = __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ void c16_pair32(double v, double& l, double& u) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    l = __hiloint2double(b[0], a[0]); u = __hiloint2double(b[1], a[1]);
}

// value of the lane (i - 1) % 16 of the same 16-lane row: DPP row_ror:1 (a lane receives from the lane 1 below, cyclically),
// i.e. after d applications lane i holds the value lane (i - d) % 16 started with. The back-substitution wants (i + d) % 16:
// it applies row_ror:15 = one step the other way.
__device__ __forceinline__ double c16_row_ror1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x12F, 0xF, 0xF, true);   // row_ror:15
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x12F, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ c16_d4 c16_load(const double* t, int ln) {
    c16_d4 v;
    v[0] = t[ln]; v[1] = t[64 + ln]; v[2] = t[128 + ln]; v[3] = t[192 + ln];
    return v;
}
__device__ __forceinline__ void c16_store(double* t, int ln, c16_d4 v) {
    t[ln] = v[0]; t[64 + ln] = v[1]; t[128 + ln] = v[2]; t[192 + ln] = v[3];
}

// Mirror the lower halves of the diagonal tiles into their upper halves (the assembly only writes i >= j).
__device__ __forceinline__ void c16_symmetrize(double* A, int nb) {
    for (int e = threadIdx.x; e < nb * 120; e += blockDim.x) {
        const int I = e / 120;
        int q = e - I * 120;
        const int c = (int)((1.0f + __builtin_sqrtf((float)(1 + 8 * q))) * 0.5f);   // q = c (c - 1) / 2 + r, r < c <= 15 (exact: 1 + 8 q <= 953)
        q -= (c * (c - 1)) >> 1;
        double* t = A + (c16_tile(I, I) << 8);
        t[c * 16 + q] = t[q * 16 + c];
    }
}

// ---- one 4-column step ------------------------------------------------------------------------------------------------
// The 4 x 4 pivot block P = D[4S .. 4S+3][4S .. 4S+3] is made uniform (v_readlane or an LDS broadcast), its Cholesky factor
// L_ss and M = L_ss^-1 are computed in every lane; M goes into an MFMA A operand "Mpad" (lane (r < 4, k): M[r][k]) so that
//     y = (tile register of the step) L_ss^-T  =  first accumulator register of  mfma(A = Mpad, B = register)
// for the diagonal tile AND for every panel tile: the replay is two MFMAs per step, no cross-lane VALU work.
// No sign test on the pivots: a non-positive pivot turns into NaN / inf (v_rsq_f64) and reaches the solution, which the
// caller tests; columns >= nreal are dummies (inverse 0: they neither change nor produce anything).
struct C16Lane {          // per-lane constants
    int e;                // Mpad select: index into the 10 entries of M (row-major lower: 00 10 11 20 21 22 30 31 32 33) or -1
    double k0, k1, k2, k3;   // 1.0 where lane / 16 == q: the lane's own column of a step, selected by multiplication
    double w[10];            // 1.0 for the lane's entry of M (Mpad), else 0
    long long* dbg;          // probe builds: timestamps inside the first pivot steps (null in the library)
};
__device__ __forceinline__ void c16_stamp(const C16Lane& lc, int slot, double& tie) {
    if (lc.dbg) {
        asm volatile("s_nop 0" : "+v"(tie) :: "memory");
        long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        asm volatile("s_nop 0" : "+v"(tie) :: "memory");
        if ((threadIdx.x & 63) == 0) lc.dbg[slot] = t;
    }
}
__device__ __forceinline__ C16Lane c16_lane(int ln) {
    const int r = ln & 15, k = ln >> 4;
    C16Lane c;
    c.e = (r < 4 && k <= r) ? (r * (r + 1) / 2 + k) : -1;
    c.k0 = k == 0 ? 1.0 : 0.0; c.k1 = k == 1 ? 1.0 : 0.0; c.k2 = k == 2 ? 1.0 : 0.0; c.k3 = k == 3 ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 10; q++) c.w[q] = c.e == q ? 1.0 : 0.0;
    c.dbg = nullptr;
    return c;
}

template <int S, int GATHER>
__device__ __forceinline__ void c16_gather(double u, double* gbuf, int ln, double (&a)[10]) {
    if (GATHER == 0) {
        // D[4S + r][4S + c] (r >= c) = D[4S + c][4S + r] sits in lane 16 r + 4 S + c
        a[0] = c16_readlane(u, 4 * S);
        a[1] = c16_readlane(u, 16 + 4 * S); a[2] = c16_readlane(u, 16 + 4 * S + 1);
        a[3] = c16_readlane(u, 32 + 4 * S); a[4] = c16_readlane(u, 32 + 4 * S + 1); a[5] = c16_readlane(u, 32 + 4 * S + 2);
        a[6] = c16_readlane(u, 48 + 4 * S); a[7] = c16_readlane(u, 48 + 4 * S + 1); a[8] = c16_readlane(u, 48 + 4 * S + 2);
        a[9] = c16_readlane(u, 48 + 4 * S + 3);
    } else {
        gbuf[ln] = u;      // wave-private 64 doubles; same-wave LDS accesses complete in order
        const double2 r1 = *(const double2*)(gbuf + 16 + 4 * S), r2 = *(const double2*)(gbuf + 32 + 4 * S);
        const double2 r3 = *(const double2*)(gbuf + 48 + 4 * S), r3b = *(const double2*)(gbuf + 48 + 4 * S + 2);
        a[0] = gbuf[4 * S]; a[1] = r1.x; a[2] = r1.y; a[3] = r2.x; a[4] = r2.y; a[5] = gbuf[32 + 4 * S + 2];
        a[6] = r3.x; a[7] = r3.y; a[8] = r3b.x; a[9] = r3b.y;
    }
}

// Cholesky factor of the 4 x 4 pivot block a (row-major lower: 00 10 11 20 21 22 30 31 32 33), uniform in every lane: the inverse
// diagonal i_r = 1 / l_rr and the off-diagonal entries; columns >= nreal are dummies (inverse 0)
struct C16Fac { double i0, i1, i2, i3, l10, l20, l30, l21, l31, l32; };
__device__ __forceinline__ C16Fac c16_factor4(const double (&a)[10], int nreal) {
    C16Fac f;
    f.i0 = nreal > 0 ? c16_rsqrt(a[0]) : 0.0;
    f.l10 = a[1] * f.i0; f.l20 = a[3] * f.i0; f.l30 = a[6] * f.i0;
    f.i1 = nreal > 1 ? c16_rsqrt(__builtin_fma(-f.l10, f.l10, a[2])) : 0.0;
    f.l21 = __builtin_fma(-f.l20, f.l10, a[4]) * f.i1; f.l31 = __builtin_fma(-f.l30, f.l10, a[7]) * f.i1;
    f.i2 = nreal > 2 ? c16_rsqrt(__builtin_fma(-f.l21, f.l21, __builtin_fma(-f.l20, f.l20, a[5]))) : 0.0;
    f.l32 = __builtin_fma(-f.l31, f.l21, __builtin_fma(-f.l30, f.l20, a[8])) * f.i2;
    f.i3 = nreal > 3 ? c16_rsqrt(__builtin_fma(-f.l32, f.l32, __builtin_fma(-f.l31, f.l31, __builtin_fma(-f.l30, f.l30, a[9])))) : 0.0;
    return f;
}
// M = L_ss^-1 as the lane's entry of the MFMA A operand "Mpad" (selected by multiplication with 0 / 1 weights: half the instructions of a select chain)
__device__ __forceinline__ double c16_mpad(const C16Fac& f, const C16Lane& lc) {
    const double m10 = -(f.l10 * f.i0) * f.i1;
    const double m21 = -(f.l21 * f.i1) * f.i2;
    const double m32 = -(f.l32 * f.i2) * f.i3;
    const double m20 = -__builtin_fma(f.l21, m10, f.l20 * f.i0) * f.i2;
    const double m31 = -__builtin_fma(f.l32, m21, f.l31 * f.i1) * f.i3;
    const double m30 = -__builtin_fma(f.l32, m20, __builtin_fma(f.l31, m10, f.l30 * f.i0)) * f.i3;
    double mp = lc.w[0] * f.i0;
    mp = __builtin_fma(lc.w[1], m10, mp); mp = __builtin_fma(lc.w[2], f.i1, mp); mp = __builtin_fma(lc.w[3], m20, mp);
    mp = __builtin_fma(lc.w[4], m21, mp); mp = __builtin_fma(lc.w[5], f.i2, mp); mp = __builtin_fma(lc.w[6], m30, mp);
    mp = __builtin_fma(lc.w[7], m31, mp); mp = __builtin_fma(lc.w[8], m32, mp); mp = __builtin_fma(lc.w[9], f.i3, mp);
    return mp;
}
// y = (the lane's row: 4 entries) L_ss^-T by forward substitution; the lane keeps its own column
__device__ __forceinline__ double c16_rowsolve(const C16Fac& f, double a0, double a1, double a2, double a3, const C16Lane& lc) {
    const double y0 = a0 * f.i0;
    const double y1 = __builtin_fma(-f.l10, y0, a1) * f.i1;
    const double y2 = __builtin_fma(-f.l21, y1, __builtin_fma(-f.l20, y0, a2)) * f.i2;
    const double y3 = __builtin_fma(-f.l32, y2, __builtin_fma(-f.l31, y1, __builtin_fma(-f.l30, y0, a3))) * f.i3;
    return __builtin_fma(lc.k3, y3, __builtin_fma(lc.k2, y2, __builtin_fma(lc.k1, y1, lc.k0 * y0)));
}
// the 10 entries of the pivot block of step S out of a gather slot (the 64 doubles of register S, lane order)
template <int S>
__device__ __forceinline__ void c16_slot_pivots(const double* slot, double (&a)[10]) {
    const double2 r1 = *(const double2*)(slot + 16 + 4 * S), r2 = *(const double2*)(slot + 32 + 4 * S);
    const double2 r3 = *(const double2*)(slot + 48 + 4 * S), r3b = *(const double2*)(slot + 48 + 4 * S + 2);
    a[0] = slot[4 * S]; a[1] = r1.x; a[2] = r1.y; a[3] = r2.x; a[4] = r2.y; a[5] = slot[32 + 4 * S + 2];
    a[6] = r3.x; a[7] = r3.y; a[8] = r3b.x; a[9] = r3b.y;
}

template <int S, int GATHER>
__device__ __forceinline__ void c16_pivot_step(c16_d4& D, c16_d4& E, c16_d4& W, double& mp_io, double& y_io, int nreal, double* pub, double* gbuf, int ln, const C16Lane& lc) {
    double u = D[S];
    c16_stamp(lc, 8 * S + 0, u);
    double a0, a1, a2, a3;   // the 4 entries of the lane's row (independent of the factor)
    double a[10];
    c16_d4 zw = {0.0, 0.0, 0.0, 0.0};
    if (GATHER == 0) {
        double ev, od;
        c16_pair16(u, ev, od);
        c16_pair32(ev, a0, a2);
        c16_pair32(od, a1, a3);
        c16_stamp(lc, 8 * S + 1, a3);
        c16_gather<S, 0>(u, gbuf, ln, a);
    } else {
        gbuf[ln] = u;          // wave-private; same-wave LDS accesses complete in order
        const double* row = gbuf + (ln & 15);
        if (S > 0) zw = __builtin_amdgcn_mfma_f64_16x16x4f64(mp_io, E[S - 1], zw, 0, 0, 0);   // the previous step on the identity tile, see below
        a0 = row[0]; a1 = row[16]; a2 = row[32]; a3 = row[48];
        const double2 r1 = *(const double2*)(gbuf + 16 + 4 * S), r2 = *(const double2*)(gbuf + 32 + 4 * S);
        const double2 r3 = *(const double2*)(gbuf + 48 + 4 * S), r3b = *(const double2*)(gbuf + 48 + 4 * S + 2);
        a[0] = gbuf[4 * S]; a[1] = r1.x; a[2] = r1.y; a[3] = r2.x; a[4] = r2.y; a[5] = gbuf[32 + 4 * S + 2];
        a[6] = r3.x; a[7] = r3.y; a[8] = r3b.x; a[9] = r3b.y;
        c16_stamp(lc, 8 * S + 1, a3);
    }
    c16_stamp(lc, 8 * S + 2, a[9]);
    C16Fac f = c16_factor4(a, nreal);
    c16_stamp(lc, 8 * S + 3, f.i3);
    if (GATHER == 1 && S > 0) {   // the matrix pipe finished zw during the factor chain
        W[S - 1] = zw[0];
        E = __builtin_amdgcn_mfma_f64_16x16x4f64(-y_io, zw[0], E, 0, 0, 0);
    }
    double y = c16_rowsolve(f, a0, a1, a2, a3, lc);
    c16_stamp(lc, 8 * S + 4, y);
    if (S < 3) D = __builtin_amdgcn_mfma_f64_16x16x4f64(-y, y, D, 0, 0, 0);
    if (S < 3) { double t = D[S + 1]; c16_stamp(lc, 8 * S + 5, t); D[S + 1] = t; }
    // for the other waves (in the shadow of the MFMA): M = L_ss^-1 as an MFMA A operand
    double mp = c16_mpad(f, lc);
    c16_stamp(lc, 8 * S + 6, mp);
    pub[S * C16_STEP + ln] = y;
    // The same step on the identity tile E (after the four steps W = I L_kk^-T) is two MFMAs, zw = M E[S] and E -= y zw^T.
    // They are issued one step LATE, inside the next step (zw before its factor chain, the update after it), so that the
    // matrix pipe works under VALU instructions the wave has to issue anyway instead of stalling it; the last step's is
    // finished by c16_pivot_block.
    if (GATHER == 0) {
        c16_d4 zw = {0.0, 0.0, 0.0, 0.0};
        zw = __builtin_amdgcn_mfma_f64_16x16x4f64(mp, E[S], zw, 0, 0, 0);
        W[S] = zw[0];
        if (S < 3) E = __builtin_amdgcn_mfma_f64_16x16x4f64(-y, zw[0], E, 0, 0, 0);
    }
    mp_io = mp; y_io = y;
}

// Factor the diagonal tile D (nreal real pivot columns): publishes the steps' y, returns X(L_kk^-T).
// A 4-column step whose columns are ALL dummies (4 S >= nreal: the last block column of a system whose size is not a multiple
// of 16) would compute y = 0, M = 0, W[S] = 0 and leave D alone: it is skipped (a step is ~1 000 cycles of pivot chain;
// N = 114 has three of them in its last block column).
template <int GATHER>
__device__ __forceinline__ c16_d4 c16_pivot_block(c16_d4 D, int nreal, double* pub, double* gbuf, int ln, const C16Lane& lc) {
    c16_d4 E, W = {0.0, 0.0, 0.0, 0.0};
    const int i = ln & 15, k = ln >> 4;
#pragma unroll
    for (int s = 0; s < 4; s++) E[s] = (i == 4 * s + k) ? 1.0 : 0.0;
    double mp = 0.0, y = 0.0;
    c16_d4 zw = {0.0, 0.0, 0.0, 0.0};
    c16_pivot_step<0, GATHER>(D, E, W, mp, y, nreal, pub, gbuf, ln, lc);
    if (nreal > 4) {
        c16_pivot_step<1, GATHER>(D, E, W, mp, y, nreal - 4, pub, gbuf, ln, lc);
        if (nreal > 8) {
            c16_pivot_step<2, GATHER>(D, E, W, mp, y, nreal - 8, pub, gbuf, ln, lc);
            if (nreal > 12) {
                c16_pivot_step<3, GATHER>(D, E, W, mp, y, nreal - 12, pub, gbuf, ln, lc);
                if (GATHER == 1) { zw = __builtin_amdgcn_mfma_f64_16x16x4f64(mp, E[3], zw, 0, 0, 0); W[3] = zw[0]; }
            } else {
                if (GATHER == 1) { zw = __builtin_amdgcn_mfma_f64_16x16x4f64(mp, E[2], zw, 0, 0, 0); W[2] = zw[0]; }
                W[3] = 0.0; pub[3 * C16_STEP + ln] = 0.0;
            }
        } else {
            if (GATHER == 1) { zw = __builtin_amdgcn_mfma_f64_16x16x4f64(mp, E[1], zw, 0, 0, 0); W[1] = zw[0]; }
            W[2] = 0.0; W[3] = 0.0; pub[2 * C16_STEP + ln] = 0.0; pub[3 * C16_STEP + ln] = 0.0;
        }
    } else {
        if (GATHER == 1) { zw = __builtin_amdgcn_mfma_f64_16x16x4f64(mp, E[0], zw, 0, 0, 0); W[0] = zw[0]; }
        W[1] = 0.0; W[2] = 0.0; W[3] = 0.0;
        pub[1 * C16_STEP + ln] = 0.0; pub[2 * C16_STEP + ln] = 0.0; pub[3 * C16_STEP + ln] = 0.0;
    }
    return W;
}

// ---- the pivot / helper pair (GATHER == 3) ------------------------------------------------------------------------------------
// A wave issues one VALU instruction per ~4.8 cycles whatever it is (scripts/probe/lat_probe.hip), and a pivot step above is ~125
// of them: the chain is bound by the ISSUE SLOTS of the pivot wave. A third of a step (M = L_ss^-1, its Mpad select, the two MFMAs
// of the identity tile) is not on the chain — it only has to be finished when the block is. With GATHER == 3 a HELPER wave on another
// SIMD takes it over: the pivot wave keeps gather -> 4 x 4 factor -> row solve -> rank-4 update -> publish y, and posts two step
// counters in LDS (slot written | y published; LDS operations of one wave complete in order, so a counter follows its data); the
// helper reads the same gather slot, factors the same 4 x 4 block in step with the pivot wave (same arithmetic, same bits), forms Mpad,
// and replays the step on the identity tile with the published y. It publishes L_kk^-T.
typedef __attribute__((address_space(3))) int c16_lds_int;   // the counters are polled with ds_read, not through the flat aperture
__device__ __forceinline__ void c16_post(volatile c16_lds_int* flag, int v, int ln) {
    asm volatile("" ::: "memory");
    if (ln == 0) *flag = v;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void c16_await(volatile c16_lds_int* flag, int v) {
    for (int it = 0; *flag < v && it < (1 << 20); it++) { }   // (bounded: a lost counter ends as a wrong solution the caller's residual test sees, not as a hung queue)
    asm volatile("" ::: "memory");
}

template <int S>
__device__ __forceinline__ void c16_lean_step(c16_d4& D, int nreal, double* pub, double* gslots, volatile c16_lds_int* flags, int seq, int ln, const C16Lane& lc) {
    double* slot = gslots + 64 * S;
    { double t = D[S]; c16_stamp(lc, 8 * S + 0, t); D[S] = t; }
    slot[ln] = D[S];                 // wave-private write, read back below; the helper reads the pivot entries
    c16_post(flags, seq + S + 1, ln);
    const double* row = slot + (ln & 15);
    const double a0 = row[0], a1 = row[16], a2 = row[32], a3 = row[48];
    double a[10];
    c16_slot_pivots<S>(slot, a);
    const C16Fac f = c16_factor4(a, nreal);
    const double y = c16_rowsolve(f, a0, a1, a2, a3, lc);
    if (S < 3) D = __builtin_amdgcn_mfma_f64_16x16x4f64(-y, y, D, 0, 0, 0);
    pub[S * C16_STEP + ln] = y;
    c16_post(flags + 1, seq + S + 1, ln);
    { double t = y; c16_stamp(lc, 8 * S + 1, t); }
}
// pivot wave: factor the diagonal tile D (nreal real pivot columns); publishes the steps' y (zeros for all-dummy steps)
__device__ __forceinline__ void c16_lean_block(c16_d4 D, int nreal, double* pub, double* gslots, volatile c16_lds_int* flags, int seq, int ln, const C16Lane& lc) {
    c16_lean_step<0>(D, nreal, pub, gslots, flags, seq, ln, lc);
    if (nreal > 4) c16_lean_step<1>(D, nreal - 4, pub, gslots, flags, seq, ln, lc); else pub[1 * C16_STEP + ln] = 0.0;
    if (nreal > 8) c16_lean_step<2>(D, nreal - 8, pub, gslots, flags, seq, ln, lc); else pub[2 * C16_STEP + ln] = 0.0;
    if (nreal > 12) c16_lean_step<3>(D, nreal - 12, pub, gslots, flags, seq, ln, lc); else pub[3 * C16_STEP + ln] = 0.0;
}
template <int S>
__device__ __forceinline__ void c16_helper_step(c16_d4& E, c16_d4& W, double& zprev, int nreal, const double* pub, const double* gslots, volatile c16_lds_int* flags, int seq, int ln, const C16Lane& lc) {
    c16_await(flags, seq + S + 1);
    double a[10];
    c16_slot_pivots<S>(gslots + 64 * S, a);
    c16_stamp(lc, 8 * S + 2, a[9]);
    const C16Fac f = c16_factor4(a, nreal);
    double mp = c16_mpad(f, lc);
    c16_stamp(lc, 8 * S + 3, mp);
    if (S > 0) {   // the previous step's update of the identity tile: its y was published while this step's factor chain ran
        c16_await(flags + 1, seq + S);
        double y = pub[(S - 1) * C16_STEP + ln];
        c16_stamp(lc, 8 * S + 4, y);
        E = __builtin_amdgcn_mfma_f64_16x16x4f64(-y, zprev, E, 0, 0, 0);
    }
    c16_d4 zw = {0.0, 0.0, 0.0, 0.0};
    zw = __builtin_amdgcn_mfma_f64_16x16x4f64(mp, E[S], zw, 0, 0, 0);
    W[S] = zw[0];
    zprev = zw[0];
    { double t = W[S]; c16_stamp(lc, 8 * S + 5, t); W[S] = t; }
}
// helper wave: X(L_kk^-T) of the block the pivot wave is factoring
__device__ __forceinline__ c16_d4 c16_helper_block(int nreal, const double* pub, const double* gslots, volatile c16_lds_int* flags, int seq, int ln, const C16Lane& lc) {
    c16_d4 E, W = {0.0, 0.0, 0.0, 0.0};
    const int i = ln & 15, k = ln >> 4;
#pragma unroll
    for (int s = 0; s < 4; s++) E[s] = (i == 4 * s + k) ? 1.0 : 0.0;
    double zprev = 0.0;
    c16_helper_step<0>(E, W, zprev, nreal, pub, gslots, flags, seq, ln, lc);
    if (nreal > 4) c16_helper_step<1>(E, W, zprev, nreal - 4, pub, gslots, flags, seq, ln, lc);
    if (nreal > 8) c16_helper_step<2>(E, W, zprev, nreal - 8, pub, gslots, flags, seq, ln, lc);
    if (nreal > 12) c16_helper_step<3>(E, W, zprev, nreal - 12, pub, gslots, flags, seq, ln, lc);
    return W;
}

// X(L_kk^-T) -> its tile (for the back-substitution, X layout) and the padded transposed buffer (for the panel products)
__device__ __forceinline__ void c16_publish_w(double* tile, double* wt, int ln, c16_d4 W) {
    c16_store(tile, ln, W);
    const int i = ln & 15, k = ln >> 4;
#pragma unroll
    for (int s = 0; s < 4; s++) wt[(4 * s + k) * 17 + i] = W[s];    // element (r = i, c = 4 s + k)
}

// L_Ik = A_Ik L_kk^-T: X(Q P^T) = sum_s mfma(A = X(P)[s], B = X(Q)[s]) with Q = A_Ik and P = L_kk^-1, whose X layout
// (lane (i, k), register s: L_kk^-1[i][4 s + k] = L_kk^-T[4 s + k][i]) is the transposed read of the buffer
__device__ __forceinline__ c16_d4 c16_panel(c16_d4 X, const double* wt, int ln) {
    const int i = ln & 15, k = ln >> 4;
    const double* w = wt + i * 17 + k;
    c16_d4 Y = {0.0, 0.0, 0.0, 0.0}, Y2 = {0.0, 0.0, 0.0, 0.0};
    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(w[0], X[0], Y, 0, 0, 0);
    Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[4], X[1], Y2, 0, 0, 0);
    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(w[8], X[2], Y, 0, 0, 0);
    Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[12], X[3], Y2, 0, 0, 0);
    return Y + Y2;
}

// H = L_IJ L_JJ^-1 (stored transposed) in place of a panel tile (the diagonal tile holds X(L_JJ^-T)): with it the back-substitution is
// x_J = z_J - sum_{I > J} H_IJ^T x_I with z = the right-hand-side row of H — no triangular solve per block.
__device__ __forceinline__ void c16_to_h(double* A, int I, int J, int ln) {
    double* t = A + (c16_tile(I, J) << 8);
    const c16_d4 L = c16_load(t, ln);
    const c16_d4 W = c16_load(A + (c16_tile(J, J) << 8), ln);
    // operands swapped: the accumulator then holds H^T, i.e. the tile is stored TRANSPOSED (H[i][c] at i * 16 + c): the
    // back-substitution reads, for a fixed row i, the 16 columns of a tile with unit stride across lanes (no bank conflicts;
    // the column-major tile put the 64 lanes' 16-byte reads 128 bytes apart)
    c16_d4 H = {0.0, 0.0, 0.0, 0.0}, H2 = {0.0, 0.0, 0.0, 0.0};
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(L[0], W[0], H, 0, 0, 0);
    H2 = __builtin_amdgcn_mfma_f64_16x16x4f64(L[1], W[1], H2, 0, 0, 0);
    H = __builtin_amdgcn_mfma_f64_16x16x4f64(L[2], W[2], H, 0, 0, 0);
    H2 = __builtin_amdgcn_mfma_f64_16x16x4f64(L[3], W[3], H2, 0, 0, 0);
    c16_store(t, ln, H + H2);
}

// Solve [S] x = rhs for the tile-packed image A (N columns, rhs = row N; diagonal tiles already symmetric). On return
// xs[0 .. N) = S^-1 rhs. Every thread of the 512-thread workgroup must call it. pub: C16_WORK doubles; yv: 16 * nb
// doubles. Returns false if the solution is not finite (a non-positive pivot). ts (may be null): phase timestamps.
// SOLVE = false stops after the factorisation: the panel tiles hold L_IJ, the diagonal tiles L_JJ^-T, the tiles of row N the
// forward-substituted right-hand side (L^-1 rhs)^T in their first rows — what the wide-panel dense solver takes (dense_chol.h).
template <int GATHER = 1, bool SOLVE = true>
__device__ __forceinline__ bool c16_solve(double* A, int N, double* xs, double* pub, double* yv, long long* ts, long long* dbg = nullptr) {
    const int tid = threadIdx.x, ln = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
    const int nb = c16_blocks(N + 1);     // tile rows incl. the right-hand-side row
    const int nbc = c16_blocks(N);        // block columns with real pivots
    C16Lane lc = c16_lane(ln);
    lc.dbg = dbg;
    double* gbuf = pub + 2 * C16_PUB;     // the pivot wave's gather buffer (GATHER == 3: one slot per step)
    double* wt = gbuf + C16_GSLOTS;       // L_kk^-T of the block just factored, transposed read
    volatile c16_lds_int* flags = (volatile c16_lds_int*)(wt + C16_WT);   // GATHER == 3: step counters of the pivot / helper pair (slot written | y published)
    constexpr bool PAIR = GATHER == 3;
    constexpr int G1 = PAIR ? 1 : GATHER;
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    if (PAIR) {
        if (tid == 0) { flags[0] = 0; flags[1] = 0; }
        __syncthreads();
    }
    c16_d4 D = {0.0, 0.0, 0.0, 0.0};      // pivot wave: the next diagonal tile
    if (PAIR && nwv >= 4) {
        if (wv == 0) c16_lean_block(c16_load(A, ln), N < 16 ? N : 16, pub, gbuf, flags, 0, ln, lc);
        else if (wv == 1) c16_publish_w(A, wt, ln, c16_helper_block(N < 16 ? N : 16, pub, gbuf, flags, 0, ln, lc));
    } else if (wv == 0) c16_publish_w(A, wt, ln, c16_pivot_block<G1>(c16_load(A, ln), N < 16 ? N : 16, pub, gbuf, ln, lc));
    __syncthreads();
    if (ts && tid == 0) ts[0] = clock64();
    for (int kb = 0; kb < nbc; kb++) {
        const int m = nb - kb - 1;        // tile rows below the diagonal tile
        // ---- phase A: L_Ik = A_Ik L_kk^-T, one product per panel tile. The pivot wave takes the tile it needs next,
        //      (kb + 1, kb), and updates the next diagonal tile with it ----
        if (wv == 0) {
            if (m >= 1) {
                double* tp = A + (c16_tile(kb + 1, kb) << 8);
                const c16_d4 Y = c16_panel(c16_load(tp, ln), wt, ln);
                c16_store(tp, ln, Y);
                if (kb + 1 < nbc) {
                    D = c16_load(A + (c16_tile(kb + 1, kb + 1) << 8), ln);
                    c16_d4 D2 = {0.0, 0.0, 0.0, 0.0};     // two accumulators: two dependent MFMAs instead of four
                    D = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[0], Y[0], D, 0, 0, 0);
                    D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[1], Y[1], D2, 0, 0, 0);
                    D = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[2], Y[2], D, 0, 0, 0);
                    D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[3], Y[3], D2, 0, 0, 0);
                    D += D2;
                }
            }
        } else {
            for (int I = kb + 1 + wv; I < nb; I += nwv - 1) {
                double* tp = A + (c16_tile(I, kb) << 8);
                c16_store(tp, ln, c16_panel(c16_load(tp, ln), wt, ln));
            }
        }
        __syncthreads();
        if (ts && tid == 0 && kb < 8) ts[1 + 2 * kb] = clock64();
        // ---- phase B: factorisation of the next diagonal tile (look-ahead) | trailing update of everything else, and the
        //      panel of the PREVIOUS block column (no longer needed as L) turned into H = L L_JJ^-1 ----
        // GATHER == 3: while the pivot block is the longer side of the phase, wave 1 is its helper (see above) and the trailing tiles
        // go to the waves of the other SIMD slots; with many tile rows left the trailing update is the longer side and takes every wave
        const int ntl = (nb - kb) * (nb - kb - 1) / 2;
        const bool pair = PAIR && nwv >= 4 && kb + 1 < nbc && ntl <= C16_PAIR_TILES * (nwv - 3) + 1;
        const int left = N - 16 * (kb + 1);
        if (wv == 0) {
            if (kb + 1 < nbc) {
                if (pair) c16_lean_block(D, left < 16 ? left : 16, pub + ((kb + 1) & 1) * C16_PUB, gbuf, flags, 4 * (kb + 1), ln, lc);
                else {
                    const c16_d4 W = c16_pivot_block<G1>(D, left < 16 ? left : 16, pub + ((kb + 1) & 1) * C16_PUB, gbuf, ln, lc);
                    c16_publish_w(A + (c16_tile(kb + 1, kb + 1) << 8), wt, ln, W);   // wt: every wave is past its phase-A reads (barrier)
                }
            }
        } else if (pair && wv == 1) {
            const c16_d4 W = c16_helper_block(left < 16 ? left : 16, pub + ((kb + 1) & 1) * C16_PUB, gbuf, flags, 4 * (kb + 1), ln, lc);
            c16_publish_w(A + (c16_tile(kb + 1, kb + 1) << 8), wt, ln, W);
        } else if (pair) {
            // bulk waves of a pair phase: every wave but 0, 1 and the one that shares the pivot wave's SIMD (nwv / 2)
            const int nbw = nwv - 3;
            const int bw = wv < nwv / 2 ? wv - 2 : wv - 3;
            if (wv != nwv / 2) {
                int I = kb + 1, J = kb + 2 + bw;
                while (true) {
                    int jmax = I < nbc ? I : nbc - 1;
                    while (I < nb && J > jmax) { J -= jmax - kb; I++; jmax = I < nbc ? I : nbc - 1; }
                    if (I >= nb) break;
                    double* ct = A + (c16_tile(I, J) << 8);
                    c16_d4 C = c16_load(ct, ln);
                    const c16_d4 LI = c16_load(A + (c16_tile(I, kb) << 8), ln);
                    const c16_d4 LJ = c16_load(A + (c16_tile(J, kb) << 8), ln);
                    c16_d4 C2 = {0.0, 0.0, 0.0, 0.0};
                    C = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[0], LI[0], C, 0, 0, 0);
                    C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[1], LI[1], C2, 0, 0, 0);
                    C = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[2], LI[2], C, 0, 0, 0);
                    C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[3], LI[3], C2, 0, 0, 0);
                    c16_store(ct, ln, C + C2);
                    J += nbw;
                }
                if (SOLVE && kb >= 1)
                    for (int Ic = kb + (nbw - 1 - bw); Ic < nb; Ic += nbw) c16_to_h(A, Ic, kb - 1, ln);
            }
        } else {
            // tiles (I, J), kb < J <= I, J < nbc, minus the pivot wave's (kb + 1, kb + 1), dealt round-robin in row order.
            // The wave that shares the pivot wave's SIMD (wave nwv / 2 with waves placed round-robin on the 4 SIMDs) stays
            // out of it: whatever it issues is taken from the pivot chain (+40 % on a pivot step, measured)
            // ... unless the trailing update is the longer side (many tile rows left: N > ~130)
            const bool all = (nb - kb) * (nb - kb - 1) / 2 > C16_ALL_TILES * (nwv - 2) + 2;
            const int nbw = all ? nwv - 1 : nwv - 2;                   // bulk waves of this phase
            const int bw = (all || wv < nwv / 2) ? wv - 1 : wv - 2;    // index among them
            if (all || wv != nwv / 2) {
                int I = kb + 1, J = kb + 2 + bw;
                while (true) {
                    int jmax = I < nbc ? I : nbc - 1;
                    while (I < nb && J > jmax) { J -= jmax - kb; I++; jmax = I < nbc ? I : nbc - 1; }   // row I holds jmax - kb tiles
                    if (I >= nb) break;
                    double* ct = A + (c16_tile(I, J) << 8);
                    c16_d4 C = c16_load(ct, ln);
                    const c16_d4 LI = c16_load(A + (c16_tile(I, kb) << 8), ln);
                    const c16_d4 LJ = c16_load(A + (c16_tile(J, kb) << 8), ln);
                    c16_d4 C2 = {0.0, 0.0, 0.0, 0.0};
                    C = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[0], LI[0], C, 0, 0, 0);
                    C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[1], LI[1], C2, 0, 0, 0);
                    C = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[2], LI[2], C, 0, 0, 0);
                    C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-LJ[3], LI[3], C2, 0, 0, 0);
                    c16_store(ct, ln, C + C2);
                    J += nbw;
                }
                if (SOLVE && kb >= 1)
                    for (int Ic = kb + (nbw - 1 - bw); Ic < nb; Ic += nbw) c16_to_h(A, Ic, kb - 1, ln);   // last waves first: they got fewer tiles above
            }
        }
        __syncthreads();
        if (ts && tid == 0 && kb < 8) ts[2 + 2 * kb] = clock64();
    }
    if (!SOLVE) { __syncthreads(); return true; }
    // the last block column's panel (at most the tile row of the right-hand side when N is a multiple of 16)
    for (int Ic = nbc + wv; Ic < nb; Ic += nwv) c16_to_h(A, Ic, nbc - 1, ln);
    // ---- back-substitution. z_c = H[N][c] = (L_JJ^-T y_J)[c]; x_I = z_I once the blocks above are in; then every thread
    //      c < 16 I subtracts H_I,J(c)^T x_I (16 consecutive doubles of tile (I, c / 16)). One barrier per block. ----
    const int IB = N >> 4, r = N & 15;
    __syncthreads();
    double z = 0.0;
    if (tid < N) {
        const int J = tid >> 4, cl = tid & 15;
        if (J < IB) z = A[(c16_tile(IB, J) << 8) + r * 16 + cl];   // H tiles are stored transposed (c16_to_h)
    }
    if (r > 0 && (tid >> 4) == IB) {       // the block that shares its diagonal tile with the right-hand side: z = L_JJ^-T y_J
        const int cl = tid & 15;
        yv[tid] = tid < N ? pub[(IB & 1) * C16_PUB + (cl >> 2) * C16_STEP + 16 * (cl & 3) + r] : 0.0;   // lane (row r, k = cl & 3) of step cl / 4
        const double* lt = A + (c16_tile(IB, IB) << 8) + cl;
        const double* v = yv + 16 * IB;
        double xi = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c++) xi = __builtin_fma(lt[c * 16], v[c], xi);
        z = tid < N ? xi : 0.0;
    }
    bool bad = false;
    for (int I = nbc - 1; I >= 0; I--) {
        double hcol[16];
        if (tid < 16 * I) {                // issued before the barrier: independent of x_I
            const double* ht = A + (c16_tile(I, tid >> 4) << 8) + (tid & 15);
#pragma unroll
            for (int i = 0; i < 16; i++) hcol[i] = ht[i * 16];
        }
        if ((tid >> 4) == I) {
            yv[tid] = z;
            if (tid < N) { xs[tid] = z; if (!(fabs(z) < 1e300)) bad = true; }
        }
        if (I == 0) break;
        __syncthreads();
        if (tid < 16 * I) {
            const double* xi = yv + 16 * I;
#pragma unroll
            for (int i = 0; i < 16; i++) z = __builtin_fma(-hcol[i], xi[i], z);
        }
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (ts && tid == 0) ts[20] = clock64();
    return s_bad == 0;
}

}  // namespace sadvio
