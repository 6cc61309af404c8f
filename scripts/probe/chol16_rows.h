// chol16_rows.h — MEASUREMENT ARTEFACT of round 6, not part of the library: sadvio_amd/csrc/chol16.h plus a third pivot-block variant
// (GATHER == 2: row layout on FP64 DPP broadcasts, see "the pivot block in ROW layout" below). scripts/probe/chol16_probe.hip times it
// against the product's 4-column MFMA steps (profiles/r06_chol16_rows_probe.txt): 5 % faster in the probe, 1.1 us slower inside the
// product build of k_solve - not kept. Everything else in this file is a copy of chol16.h as of that measurement.
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
constexpr int C16_IDENT = 256;           // a 16 x 16 identity tile (the odd lane rows of the row-layout pivot block start from it)
constexpr int C16_WORK = 2 * C16_PUB + 64 + C16_WT + C16_IDENT;   // doubles of the exchange area (`pub`)

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
    const double i0 = nreal > 0 ? c16_rsqrt(a[0]) : 0.0;
    const double l10 = a[1] * i0, l20 = a[3] * i0, l30 = a[6] * i0;
    const double i1 = nreal > 1 ? c16_rsqrt(__builtin_fma(-l10, l10, a[2])) : 0.0;
    const double l21 = __builtin_fma(-l20, l10, a[4]) * i1, l31 = __builtin_fma(-l30, l10, a[7]) * i1;
    const double i2 = nreal > 2 ? c16_rsqrt(__builtin_fma(-l21, l21, __builtin_fma(-l20, l20, a[5]))) : 0.0;
    const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, a[8])) * i2;
    double i3 = nreal > 3 ? c16_rsqrt(__builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, a[9])))) : 0.0;
    c16_stamp(lc, 8 * S + 3, i3);
    if (GATHER == 1 && S > 0) {   // the matrix pipe finished zw during the factor chain
        W[S - 1] = zw[0];
        E = __builtin_amdgcn_mfma_f64_16x16x4f64(-y_io, zw[0], E, 0, 0, 0);
    }
    // y = (row's 4 entries) L_ss^-T by forward substitution; the lane keeps its own column
    const double y0 = a0 * i0;
    const double y1 = __builtin_fma(-l10, y0, a1) * i1;
    const double y2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, y0, a2)) * i2;
    const double y3 = __builtin_fma(-l32, y2, __builtin_fma(-l31, y1, __builtin_fma(-l30, y0, a3))) * i3;
    double y = __builtin_fma(lc.k3, y3, __builtin_fma(lc.k2, y2, __builtin_fma(lc.k1, y1, lc.k0 * y0)));
    c16_stamp(lc, 8 * S + 4, y);
    if (S < 3) D = __builtin_amdgcn_mfma_f64_16x16x4f64(-y, y, D, 0, 0, 0);
    if (S < 3) { double t = D[S + 1]; c16_stamp(lc, 8 * S + 5, t); D[S + 1] = t; }
    // for the other waves (in the shadow of the MFMA): M = L_ss^-1 as an MFMA A operand
    const double m10 = -(l10 * i0) * i1;
    const double m21 = -(l21 * i1) * i2;
    const double m32 = -(l32 * i2) * i3;
    const double m20 = -__builtin_fma(l21, m10, l20 * i0) * i2;
    const double m31 = -__builtin_fma(l32, m21, l31 * i1) * i3;
    const double m30 = -__builtin_fma(l32, m20, __builtin_fma(l31, m10, l30 * i0)) * i3;
    double mp;   // the lane's entry of M by multiplication with 0 / 1 weights (half the instructions of a select chain)
    mp = lc.w[0] * i0;
    mp = __builtin_fma(lc.w[1], m10, mp); mp = __builtin_fma(lc.w[2], i1, mp); mp = __builtin_fma(lc.w[3], m20, mp);
    mp = __builtin_fma(lc.w[4], m21, mp); mp = __builtin_fma(lc.w[5], i2, mp); mp = __builtin_fma(lc.w[6], m30, mp);
    mp = __builtin_fma(lc.w[7], m31, mp); mp = __builtin_fma(lc.w[8], m32, mp); mp = __builtin_fma(lc.w[9], i3, mp);
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

// ---- the pivot block in ROW layout (round 6; GATHER == 2) -----------------------------------------------------------------
// The four 16-lane rows of the pivot wave: lane i (= ln & 15) of the EVEN rows keeps ROW i of the symmetric diagonal tile in 16
// registers, lane i of the ODD rows keeps row i of an identity tile in the SAME 16 registers. A right-looking column step k is
// then, for every lane at once,
//     inv = rsqrt(D[k][k])               v_readlane of lane k's register k -> uniform, v_rsq_f64 + one third-order step
//     m   = R[k] * inv                   the lane's entry of column k: of L in the even rows (c), of the replayed identity in the odd ones
//     cb  = c in all four rows           the even rows' m copied into the odd rows (v_permlane16_swap)
//     R[j] -= cb[lane j] * m   (j > k)   ONE instruction each: v_fmac_f64_dpp ... row_newbcast:j — gfx90a+ lets a DP ALU VOP2
//                                        take lane j of its own 16-lane row as the first operand (the only DPP mode FP64 has)
// which leaves L_kk in the even rows and L_kk^-T (the identity under the same column operations) in the odd ones, row by row. No
// 4 x 4 uniform factor, no forward substitution, no matrix-core round trip, no gather through LDS per step: the chain per COLUMN
// is readlane -> rsqrt -> multiply -> swap -> one fmac (~ 110 cycles against ~ 250 for a quarter of a 4-column step), and the
// 14 - k other fmacs of a step issue in the shadow of the next column's rsqrt chain (a DPP FP64 fmac occupies the pipe for 8
// cycles, measured: the first version carried the identity rows in 16 more registers of every lane, 240 fmacs per block, and was
// bound by their issue: 3 300 cycles per block against 4 250 for the 4-column steps; this one issues 120).
// Rows above the pivot keep receiving (meaningless) updates: nobody reads them — a broadcast takes lane j > k, the pivot lane k.
// Every instruction of a column step is a VOLATILE asm statement: volatile asms keep their order, and the order IS the design (the
// machine scheduler otherwise clusters the five dependent operations of the rsqrt chain and sinks the updates behind them).
template <int J>
__device__ __forceinline__ void c16_fmac_bcast(double& acc, double cb, double m) {   // acc -= cb[lane J of the row] * m
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(cb), "v"(m), "n"(J));
}
template <int J0, int J1>
struct C16Cols {   // columns [J0, J1) of a step
    static __device__ __forceinline__ void run(double (&R)[16], double cb, double m) {
        c16_fmac_bcast<J0>(R[J0], cb, m);
        C16Cols<J0 + 1, J1>::run(R, cb, m);
    }
};
template <int J1>
struct C16Cols<J1, J1> { static __device__ __forceinline__ void run(double (&)[16], double, double) {} };
// m = R[K] * inv and cb = the even rows' m in every row (see c16_row_iter), ending with the two wait states of the DPP read
__device__ __forceinline__ void c16_col_scale(double rk, double inv, double& m, double& cb) {
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(m) : "v"(rk), "v"(inv));
    asm volatile("v_mov_b64 v[60:61], %1\n\tv_mov_b64 v[62:63], %1\n\ts_nop 1\n\tv_permlane16_swap_b32 v60, v62\n\tv_permlane16_swap_b32 v61, v63\n\t"
                 "v_mov_b64 %0, v[60:61]\n\ts_nop 1" : "=&v"(cb) : "v"(m) : "v60", "v61", "v62", "v63");
}
// One column step, software-pipelined by hand (the wave issues in order and is alone on its SIMD; a dependent FP64 operation
// costs ~ 20 cycles, measured: 11 levels per column were 215 cycles). On entry d = D[K][K] (uniform) and y = v_rsq_f64(d) are
// under way; the chain of step K is
//     t = d y, ry = R[K] y | e = 1 - t y | p = 1/2 + 3/8 e, rye = ry e | m = ry + rye p   ( = R[K] rsqrt(d): the third-order step
//                                                                                          of c16_rsqrt applied to the product)
//     R[K + 1] -= m[lane K + 1] m   in the EVEN rows only (row_mask: they need no copy of m)  | readlane -> d | v_rsq_f64 -> y
// seven levels. The copy of m into the odd rows (cb), their update of column K + 1 and the 14 - K other updates of the step are
// off the chain: the latter are DEFERRED into the gaps of the next step's chain (cbp, mp = the previous step's cb, m).
// k375 = 0.375 in a register (VOP3 takes no literal on gfx9).
template <int K>
__device__ __forceinline__ void c16_row_iter(double (&R)[16], double& d, double& y, double& cbp, double& mp, double k375) {
    constexpr int n = K >= 1 ? 15 - K : 0;            // deferred updates of step K - 1: columns K + 1 .. 15
    constexpr int b0 = K + 1, b1 = b0 + (n + 3) / 4, b2 = b0 + (2 * n + 3) / 4, b3 = b0 + (3 * n + 3) / 4, b4 = b0 + n;
    double t, ry, e, p, rye, m;
    asm volatile("v_mul_f64 %0, %2, %3\n\tv_mul_f64 %1, %4, %3" : "=&v"(t), "=&v"(ry) : "s"(d), "v"(y), "v"(R[K]));
    C16Cols<b0, b1>::run(R, cbp, mp);                 // (column K + 1 first: the pivot read below wants it complete)
    asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(t), "v"(y));
    C16Cols<b1, b2>::run(R, cbp, mp);
    asm volatile("v_fma_f64 %0, %2, %3, 0.5\n\tv_mul_f64 %1, %4, %3" : "=&v"(p), "=&v"(rye) : "v"(k375), "v"(e), "v"(ry));
    C16Cols<b2, b3>::run(R, cbp, mp);
    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(m) : "v"(rye), "v"(p), "v"(ry));
    R[K] = m;
    if (K < 15) {
        constexpr int K1 = K < 15 ? K + 1 : 15;
        double cb;
        // (two wait states between the VALU write of a VGPR and a DPP read of it; the hazard recogniser does not look into inline asm)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %1 row_newbcast:%2 row_mask:0x5 bank_mask:0xf" : "+v"(R[K1]) : "v"(m), "n"(K1));
        d = c16_readlane(R[K1], K1);
        asm volatile("v_rsq_f64 %0, %1" : "=v"(y) : "s"(d));
        // cb = the even rows' m in every row: (r0, r0, r2, r2) is what v_permlane16_swap leaves in its first operand when both
        // start as copies of m (it swaps the odd rows of the first with the even rows of the second). In ONE statement on fixed
        // scratch registers: the halves of a 64-bit asm operand cannot be named, and a register copy the compiler places between the
        // swaps and the first DPP read of cb would sit inside the two wait states (seen: wrong factors). The s_nop between the copies and
        // the swaps is needed as well (VALU write -> v_permlane*_swap read: without it the low words came out wrong, errors of 1e-6).
        asm volatile("v_mov_b64 v[60:61], %1\n\tv_mov_b64 v[62:63], %1\n\ts_nop 1\n\tv_permlane16_swap_b32 v60, v62\n\tv_permlane16_swap_b32 v61, v63\n\t"
                     "v_mov_b64 %0, v[60:61]\n\ts_nop 1" : "=&v"(cb) : "v"(m) : "v60", "v61", "v62", "v63");
        C16Cols<b3, b4>::run(R, cbp, mp);
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xa bank_mask:0xf" : "+v"(R[K1]) : "v"(cb), "v"(m), "n"(K1));
        cbp = cb; mp = m;
    }
}
// the steps of a block with dummy pivots (the last block column of a system whose size is not a multiple of 16): columns
// 0 .. nreal - 1, then out (one exit, no work for the dummy columns; c16_rows_out zeroes what they would have produced)
template <int K>
struct C16Partial {
    static __device__ __forceinline__ void run(double (&R)[16], int nreal) {
        const double inv = c16_rsqrt(c16_readlane(R[K], K));
        double m, cb;
        c16_col_scale(R[K], inv, m, cb);
        R[K] = m;
        C16Cols<(K < 15 ? K + 1 : 16), 16>::run(R, cb, m);
        if (K + 1 < nreal) C16Partial<K + 1>::run(R, nreal);
    }
};
template <>
struct C16Partial<15> { static __device__ __forceinline__ void run(double (&)[16], int) {} };   // (nreal = 16 takes the other path)
// The lane's row: row i of the symmetric column-major tile t (even rows of the wave; 16 consecutive lanes read 16 consecutive
// doubles) or of the identity tile `ident` (odd rows).
__device__ __forceinline__ void c16_load_rows(const double* t, const double* ident, int ln, double (&R)[16]) {
    const double* p = ((ln & 16) ? ident : t) + (ln & 15);
#pragma unroll
    for (int c = 0; c < 16; c++) R[c] = p[c * 16];
}
// Results of the block: rows of L_kk^-T (odd rows of the wave) -> the dead diagonal tile (column-major: X layout in memory, for
// c16_to_h and the back-substitution) and the padded transposed buffer wt (element (r, c) at c * 17 + r, for the panel products);
// with want_y the columns of L_kk (even rows) in the X layout of the 4-column steps (pub[S * C16_STEP + 16 q + i] = L[i][4 S + q]:
// what the back-substitution reads for the block that holds the right-hand-side row). Dummy pivot columns: zero.
__device__ __forceinline__ void c16_rows_out(const double (&R)[16], int nreal, double* tile, double* wt, double* pub, bool want_y, int ln) {
    const int i = ln & 15, g = ln >> 4;
    if (g & 1) {      // row 1 of the wave writes the tile, row 3 the padded buffer, with the same instructions (per-lane address and stride)
        double* p = (g == 1 ? tile : wt) + i;
        const int st = g == 1 ? 16 : 17;
#pragma unroll
        for (int c = 0; c < 16; c++) p[c * st] = c < nreal ? R[c] : 0.0;
    } else if (want_y && g == 0) {
#pragma unroll
        for (int c = 0; c < 16; c++) pub[(c >> 2) * C16_STEP + 16 * (c & 3) + i] = c < nreal ? R[c] : 0.0;
    }
}
// Factor the diagonal tile whose rows are in R (see c16_load_rows; nreal real pivot columns) and publish the results.
__device__ __forceinline__ void c16_row_stamp(long long* dbg, int slot) {
    if (dbg) {
        long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        if ((threadIdx.x & 63) == 0) dbg[slot] = t;
    }
}
__device__ __forceinline__ void c16_pivot_block_rows(double (&R)[16], int nreal, double* tile, double* wt, double* pub, bool want_y, int ln, long long* dbg = nullptr) {
    c16_row_stamp(dbg, 0);
    if (nreal == 16) {      // straight-line code (a test per step would put register copies at every join)
        double d = c16_readlane(R[0], 0), y, cbp = 0.0, mp = 0.0;
        asm volatile("v_rsq_f64 %0, %1\n\ts_nop 0" : "=v"(y) : "s"(d));
        double k375 = 0.375;
        asm volatile("" : "+v"(k375));
        c16_row_iter<0>(R, d, y, cbp, mp, k375); c16_row_iter<1>(R, d, y, cbp, mp, k375); c16_row_iter<2>(R, d, y, cbp, mp, k375); c16_row_iter<3>(R, d, y, cbp, mp, k375);
        c16_row_stamp(dbg, 3);
        c16_row_iter<4>(R, d, y, cbp, mp, k375); c16_row_iter<5>(R, d, y, cbp, mp, k375); c16_row_iter<6>(R, d, y, cbp, mp, k375); c16_row_iter<7>(R, d, y, cbp, mp, k375);
        c16_row_iter<8>(R, d, y, cbp, mp, k375); c16_row_iter<9>(R, d, y, cbp, mp, k375); c16_row_iter<10>(R, d, y, cbp, mp, k375); c16_row_iter<11>(R, d, y, cbp, mp, k375);
        c16_row_iter<12>(R, d, y, cbp, mp, k375); c16_row_iter<13>(R, d, y, cbp, mp, k375); c16_row_iter<14>(R, d, y, cbp, mp, k375); c16_row_iter<15>(R, d, y, cbp, mp, k375);
        c16_row_stamp(dbg, 1);
        c16_rows_out(R, 16, tile, wt, pub, want_y, ln);
        c16_row_stamp(dbg, 2);
        return;
    }
    C16Partial<0>::run(R, nreal);
    c16_rows_out(R, nreal, tile, wt, pub, want_y, ln);
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
    double* gbuf = pub + 2 * C16_PUB;     // the pivot wave's gather buffer
    double* wt = gbuf + 64;               // L_kk^-T of the block just factored, transposed read
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    c16_d4 D = {0.0, 0.0, 0.0, 0.0};      // pivot wave: the next diagonal tile
    double R[16];                         // ... its rows (GATHER == 2)
    const int IBr = (N & 15) ? (N >> 4) : -1;   // the block whose diagonal tile holds the right-hand-side row: the back-substitution reads its y
    double* ident = wt + C16_WT;
    if (GATHER == 2) {
        for (int e = tid; e < 256; e += blockDim.x) ident[e] = (e >> 4) == (e & 15) ? 1.0 : 0.0;
        __syncthreads();
    }
    if (wv == 0) {
        if (GATHER == 2) {
            c16_load_rows(A, ident, ln, R);
            c16_pivot_block_rows(R, N < 16 ? N : 16, A, wt, pub, IBr == 0, ln);
        } else {
            c16_publish_w(A, wt, ln, c16_pivot_block<GATHER == 2 ? 1 : GATHER>(c16_load(A, ln), N < 16 ? N : 16, pub, gbuf, ln, lc));
        }
    }
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
                    if (GATHER == 2) {   // X layout -> rows through the tile's own home (same-wave LDS accesses complete in order; the
                        double* dt = A + (c16_tile(kb + 1, kb + 1) << 8);   // reads are in flight across the barrier)
                        c16_store(dt, ln, D);
                        c16_load_rows(dt, ident, ln, R);
                    }
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
        if (wv == 0) {
            if (kb + 1 < nbc) {
                const int left = N - 16 * (kb + 1);
                double* dt = A + (c16_tile(kb + 1, kb + 1) << 8);
                // wt: every wave is past its phase-A reads (barrier)
                if (GATHER == 2) c16_pivot_block_rows(R, left < 16 ? left : 16, dt, wt, pub + ((kb + 1) & 1) * C16_PUB, kb + 1 == IBr, ln, kb == 3 ? dbg : nullptr);
                else c16_publish_w(dt, wt, ln, c16_pivot_block<GATHER == 2 ? 1 : GATHER>(D, left < 16 ? left : 16, pub + ((kb + 1) & 1) * C16_PUB, gbuf, ln, lc));
            }
        } else {
            // tiles (I, J), kb < J <= I, J < nbc, minus the pivot wave's (kb + 1, kb + 1), dealt round-robin in row order.
            // The wave that shares the pivot wave's SIMD (wave nwv / 2 with waves placed round-robin on the 4 SIMDs) stays
            // out of it: whatever it issues is taken from the pivot chain (+40 % on a pivot step, measured)
            // ... unless the trailing update is the longer side (many tile rows left: N > ~130)
            const bool all = (nb - kb) * (nb - kb - 1) / 2 > 4 * (nwv - 2) + 2;
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
