// lm_kernels.h — throughput variants of K5 / K7 for large batches of plain visual windows (no robust loss, no prior-kept
// landmarks, no pseudo-observations, one GPU per window).
//
// kernels.h maps one LANE to one OBSERVATION with a landmark's observations in a group of G lanes: every reduction over a
// landmark is a lane shuffle and the dependent chain of a tile is as short as it gets, which is what one window (250
// workgroups on 256 CUs) needs. A batch of 64 windows is instruction-bound instead (rocprofv3, round-2 notes in DESIGN.md:
// VALU busy 46 %, 2 waves / SIMD): 5 of 8 lanes carry an observation, the 3 x 3 elimination is repeated by every lane of
// a group and the block sums travel through DPP. Here the same arithmetic is laid out for throughput:
//   k_elim        one lane per LANDMARK, a serial loop over its observations: H_ll, g_l, damped inverse, its Cholesky
//                 factor L (M^-1 = L L^T) and M^-1 g_l  -> 9 doubles per landmark in HBM; cost / gradient-max partials
//   k_build_obs   one lane per observation, flat (no padding): with W = Jl L the landmark's Schur term is
//                 E M^-1 E^T = Z Z^T, Z = sum_a Jp_a^T W_a, so ONE strip Z[row][4 * landmark + c] per wave feeds both MFMA
//                 operands (v_mfma_f64_16x16x4_f64, K = the landmarks of a chunk); a spare strip row with w = L^T g_l makes
//                 the same pass return Z w = E M^-1 g_l, the landmark part of the reduced gradient; tile flush as in k_build
//   k_diag        key-frame major: sum Jp^T Jp and sum Jp^T r over the observations of one key-frame, in registers
//   k_backsub_lm  one lane per landmark: delta_l = -M^-1 sum Jl^T (r + Jp dp); the model cost change of the landmark's
//                 residual blocks from sums accumulated in the same pass; candidate cost in a second residual-only loop
// The tiles are those of k_build; a tile's landmarks are cut into chunks of <= 12 landmarks and <= 64 observations
// (chunk tables built with the tiles).
#pragma once
#include "kernels.h"

namespace sadvio {

constexpr int LM_CHUNK = 12;            // landmarks per MFMA chunk (12 x 5 observations fill 60 of 64 lanes)
constexpr int LM_KS = 3 * LM_CHUNK + 2; // strip row stride (doubles): 3 columns per landmark + 2 (bank spread); 32 x 38 doubles = 9.5 KB per wave
constexpr int LM_ELIM = 9;              // per landmark: L (6, lower, row-major) | w = L^T g_l (3)

// the per-observation fields, loaded ahead of their use (plain path: no pseudo-observations, no loss function)
struct ObsIn { int sl, cam; double m[3]; };
template <int FACTOR>
__device__ __forceinline__ ObsIn lm_load_obs(const DevPtrs& P, int o, int oe) {
    ObsIn a;
    a.sl = 0; a.cam = 0; a.m[0] = a.m[1] = a.m[2] = 0.0;
    if (o < oe) {
        a.sl = P.obs_slot[o]; a.cam = P.obs_cam[o];
        if (FACTOR == 0) { const double2 mm = *(const double2*)(P.obs_meas + 2 * (long long)o); a.m[0] = mm.x; a.m[1] = mm.y; }
        else { const double* mm = P.obs_meas + 3 * (long long)o; a.m[0] = mm[0]; a.m[1] = mm[1]; a.m[2] = mm[2]; }
    }
    return a;
}
template <int FACTOR, bool WANT_J>
__device__ __forceinline__ void lm_linearize_in(const double* tab, const double* ct, const ObsIn& a, const double* pw, double* r, double* Jp, double* Jl) {
    if (FACTOR == 0) pixel_factor<WANT_J>(tab, ct, ct + 4, pw, a.m[0], a.m[1], ct[16], r, Jp, Jl);
    else angular_factor<WANT_J>(tab, ct + 4, pw, a.m, ct[16], r, Jp, Jl);
}

// LM-damped inverse of a landmark's 3 x 3 block (the arithmetic of group_eliminate, one lane)
__device__ __forceinline__ void lm_damped_inverse(const DevPtrs& P, const double* H, const double* s, double radius, double* Mi) {
    const double ir = 1.0 / radius;
    const double s0 = s[0] * s[0], s1 = s[1] * s[1], s2 = s[2] * s[2];
    double M[6] = {H[0], H[1], H[2], H[3], H[4], H[5]};
    M[0] += fmin(fmax(s0 * H[0], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s0;
    M[3] += fmin(fmax(s1 * H[3], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s1;
    M[5] += fmin(fmax(s2 * H[5], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s2;
    sym3_inverse(M, Mi);
}

// ---- K5a: per-landmark elimination ------------------------------------------------------------------------------------
template <int FACTOR>
__global__ __launch_bounds__(BUILD_THREADS) void k_elim(DevPtrs P, int slot, int max_tile_kf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ti = P.tile_perm[blockIdx.x];   // largest tiles first (host: LPT order), so that the launch does not end on a long tile
    const Tile T = P.tiles[ti];
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const LmState st = P.states[(long long)T.w * P.state_stride + slot];   // decided by k_decide (large batches)
    if (st.done) return;
    double* poseTab = (double*)smem;
    double* camTab = poseTab + (size_t)max_tile_kf * POSE_TAB;
    int* rowTab = (int*)(camTab + MAX_WIN_CAM * 17);
    stage_tables(P, T, st.cur, poseTab, camTab, rowTab);
    __syncthreads();
    const int nl = T.lmk1 - T.lmk0;
    const double* xl = P.xl + (long long)st.cur * P.xl_stride;
    double cost_part = 0.0, fixed_part = 0.0, gmax_part = 0.0;
    for (int lm = tid; lm < nl; lm += (int)blockDim.x) {
        const int gl = T.lmk0 + lm;
        const int ob = P.lmk_ob[gl], oe = P.lmk_oe[gl];
        const int lcode = P.lmk_const ? P.lmk_const[gl] : 0;
        const double pw[3] = {P.lmk_p[3 * (long long)gl] + xl[3 * (long long)gl], P.lmk_p[3 * (long long)gl + 1] + xl[3 * (long long)gl + 1],
                              P.lmk_p[3 * (long long)gl + 2] + xl[3 * (long long)gl + 2]};
        double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        ObsIn nx = lm_load_obs<FACTOR>(P, ob, oe);
        for (int o = ob; o < oe; o++) {
            const ObsIn cu = nx;
            nx = lm_load_obs<FACTOR>(P, o + 1, oe);   // the next observation's loads are in flight during this one's arithmetic
            const int sl = cu.sl;
            const double* ct = camTab + (cu.cam - T.cam_base) * 17;
            double r[2], Jp[12], Jl[6];
            lm_linearize_in<FACTOR, true>(poseTab + sl * POSE_TAB, ct, cu, pw, r, Jp, Jl);
            const double c = r[0] * r[0] + r[1] * r[1];
            if (rowTab[sl] >= 0 || lcode != 1) cost_part += c; else fixed_part += c;   // a block of constant parameters: fixed cost
            if (lcode == 0) {
                H[0] += Jl[0] * Jl[0] + Jl[3] * Jl[3]; H[1] += Jl[0] * Jl[1] + Jl[3] * Jl[4]; H[2] += Jl[0] * Jl[2] + Jl[3] * Jl[5];
                H[3] += Jl[1] * Jl[1] + Jl[4] * Jl[4]; H[4] += Jl[1] * Jl[2] + Jl[4] * Jl[5]; H[5] += Jl[2] * Jl[2] + Jl[5] * Jl[5];
                g[0] += Jl[0] * r[0] + Jl[3] * r[1]; g[1] += Jl[1] * r[0] + Jl[4] * r[1]; g[2] += Jl[2] * r[0] + Jl[5] * r[1];
            }
        }
        double out[LM_ELIM] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (lcode == 0 && oe > ob) {
            double s[3];
            if (slot == 0) {
                s[0] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[0])) : 1.0;
                s[1] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[3])) : 1.0;
                s[2] = P.o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[5])) : 1.0;
                P.s_lmk[3 * (long long)gl] = s[0]; P.s_lmk[3 * (long long)gl + 1] = s[1]; P.s_lmk[3 * (long long)gl + 2] = s[2];
            } else { s[0] = P.s_lmk[3 * (long long)gl]; s[1] = P.s_lmk[3 * (long long)gl + 1]; s[2] = P.s_lmk[3 * (long long)gl + 2]; }
            double Mi[6];
            lm_damped_inverse(P, H, s, st.radius, Mi);
            gmax_part = fmax(gmax_part, fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))));
            // M^-1 = L L^T (lower Cholesky; M^-1 is positive definite with M)
            const double l00 = sqrt(fmax(Mi[0], 0.0)), i00 = l00 > 0.0 ? 1.0 / l00 : 0.0;
            const double l10 = Mi[1] * i00, l20 = Mi[2] * i00;
            const double l11 = sqrt(fmax(Mi[3] - l10 * l10, 0.0)), i11 = l11 > 0.0 ? 1.0 / l11 : 0.0;
            const double l21 = (Mi[4] - l20 * l10) * i11;
            const double l22 = sqrt(fmax(Mi[5] - l20 * l20 - l21 * l21, 0.0));
            out[0] = l00; out[1] = l10; out[2] = l11; out[3] = l20; out[4] = l21; out[5] = l22;
            // w = L^T g_l: the landmark's part of the reduced gradient is E M^-1 g = (E L)(L^T g) = Z w
            out[6] = l00 * g[0] + l10 * g[1] + l20 * g[2];
            out[7] = l11 * g[1] + l21 * g[2];
            out[8] = l22 * g[2];
        }
        double* dst = P.lm_elim + (long long)gl * LM_ELIM;
#pragma unroll
        for (int i = 0; i < LM_ELIM; i++) dst[i] = out[i];
    }
    __shared__ double s_part[BUILD_WAVES * 4];
    const double c = wave_sum(cost_part), f = wave_sum(fixed_part), gm = wave_max(gmax_part);
    if (ln == 0) { s_part[wv * 4] = c; s_part[wv * 4 + 1] = f; s_part[wv * 4 + 2] = gm; }
    __syncthreads();
    if (tid == 0) {
        double cs = 0.0, fs = 0.0, gs = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) { cs += s_part[k * 4]; fs += s_part[k * 4 + 1]; gs = fmax(gs, s_part[k * 4 + 2]); }
        TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + ti;
        ta->lin_cost = cs; ta->fixed_cost = fs; ta->gmax = gs;
    }
}

// acc += strip strip^T over the first `ksteps` K = 4 column groups of a wave's strip (<= 9 here, <= 16 supported): the operand loads of the next
// group of four steps are issued before the MFMAs of the current one (the trip count is dynamic, the compiler does not
// pipeline the loop by itself and every step would expose the LDS latency).
typedef double lm_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lm_syrk_pass(const double* Zb, int lr, int lk, int nt16, int ksteps, lm_d4* acc) {
    double a0[4], a1[4], b0[4], b1[4];
    const double* p0 = Zb + lr * LM_KS + lk;
    const double* p1 = Zb + (16 + lr) * LM_KS + lk;
    const bool two = nt16 > 1;
#pragma unroll
    for (int j = 0; j < 4; j++) { a0[j] = j < ksteps ? p0[4 * j] : 0.0; a1[j] = (two && j < ksteps) ? p1[4 * j] : 0.0; }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (4 * g >= ksteps) break;
        if (g < 3 && 4 * (g + 1) < ksteps) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool in = 4 * (g + 1) + j < ksteps;   // the strip row ends after the chunk's columns
                b0[j] = in ? p0[16 * (g + 1) + 4 * j] : 0.0; b1[j] = (two && in) ? p1[16 * (g + 1) + 4 * j] : 0.0;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {   // steps past the last one multiply zeros: whole groups of four need no guard
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], a0[j], acc[0], 0, 0, 0);
            if (two) {
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], a0[j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], a1[j], acc[2], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { a0[j] = b0[j]; a1[j] = b1[j]; }
    }
}

// ---- K5b: the landmark (Schur) part of the reduced system from flat observations -------------------------------------------
// Per chunk (<= 12 landmarks, <= 64 observations, lane = observation) the wave's strip [row][38] holds, in column
// 3 * landmark + c, Z = sum_a Jp_a^T (Jl_a L) in the rows of the observing key-frames and w = L^T g_l in a spare row. One
// v_mfma_f64_16x16x4_f64 pass with A = B = the strip gives strip strip^T: the tile's rows are Z Z^T = E M^-1 E^T (to be
// subtracted from S) and the spare row is Z w = E M^-1 g_l (to be subtracted from the reduced gradient). The two cameras of a
// key-frame (adjacent lanes, the same rows) are summed with DPP before the store; no LDS atomics except one add of the
// accumulators into the tile at the end. The observation-diagonal terms (J_p^T J_p, J_p^T r) are k_diag's.
template <int FACTOR>
__global__ __launch_bounds__(BUILD_THREADS, 3) void k_build_obs(DevPtrs P, int slot, int max_tile_kf, int Rp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ti = P.tile_perm[blockIdx.x];   // largest tiles first (host: LPT order), so that the launch does not end on a long tile
    const Tile T = P.tiles[ti];
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const LmState st = P.states[(long long)T.w * P.state_stride + slot];
    if (st.done) return;
    SADVIO_TS(3, 32);
    double* poseTab = (double*)smem;
    double* camTab = poseTab + (size_t)max_tile_kf * POSE_TAB;
    int* rowTab = (int*)(camTab + MAX_WIN_CAM * 17);
    double* strips = (double*)(smem + tile_tables_bytes(max_tile_kf));   // [BUILD_WAVES][Rp * LM_KS]
    const int strip_doubles = Rp * LM_KS;
    double* Stile = strips + BUILD_WAVES * strip_doubles;
    const int Nt = 6 * T.n_free;
    const int tri_n = Nt * (Nt + 1) / 2;
    double* gT = Stile + tri_n;
    stage_tables(P, T, st.cur, poseTab, camTab, rowTab);
    for (int i = tid; i < tri_n + Nt; i += blockDim.x) Stile[i] = 0.0;
    __syncthreads();
    SADVIO_TS(3, 34);
    const double* xl = P.xl + (long long)st.cur * P.xl_stride;
    double* Zb = strips + wv * strip_doubles;
    typedef lm_d4 d4;
    d4 accZ[3] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    const int lr = ln & 15, lk = ln >> 4;
    const int nt16 = (Nt + 15) >> 4;
    const int rows_used = 16 * nt16;          // the strip rows this tile touches; Nt <= rows_used - 2
    const int w_row = rows_used - 1;          // spare row: w = L^T g_l
    // level 1 = the observation's fields, level 2 = its landmark's position and elimination record (prefetching the next
    // chunk's was measured: no gain at 3 waves / SIMD, and it costs registers)
    struct In1 { int ob0, ob1, lm0, nlm, sl, cam, ls; double m[3]; };
    struct In2 { double pw[3], E[LM_ELIM]; int lcode; };
    auto load1 = [&](int ch, In1& a) {
        a.ob0 = P.chunk_ob[ch]; a.ob1 = P.chunk_ob[ch + 1];
        a.lm0 = P.chunk_lm[ch]; a.nlm = P.chunk_lm[ch + 1] - a.lm0;
        const int o = a.ob0 + ln;
        a.sl = 0; a.cam = T.cam_base; a.ls = 0; a.m[0] = a.m[1] = a.m[2] = 0.0;
        if (o < a.ob1) {
            a.ls = P.obs_lslot[o]; a.sl = P.obs_slot[o]; a.cam = P.obs_cam[o];
            if (FACTOR == 0) { const double2 mm = *(const double2*)(P.obs_meas + 2 * (long long)o); a.m[0] = mm.x; a.m[1] = mm.y; }
            else { const double* mm = P.obs_meas + 3 * (long long)o; a.m[0] = mm[0]; a.m[1] = mm[1]; a.m[2] = mm[2]; }
        }
    };
    auto load2 = [&](const In1& a, In2& b) {
        b.lcode = 1;
#pragma unroll
        for (int i = 0; i < 3; i++) b.pw[i] = 0.0;
#pragma unroll
        for (int i = 0; i < LM_ELIM; i++) b.E[i] = 0.0;
        if (a.ob0 + ln < a.ob1) {
            const long long gl = a.lm0 + a.ls;
            b.lcode = P.lmk_const ? P.lmk_const[gl] : 0;
#pragma unroll
            for (int i = 0; i < 3; i++) b.pw[i] = P.lmk_p[3 * gl + i] + xl[3 * gl + i];
            const double* E = P.lm_elim + gl * LM_ELIM;
#pragma unroll
            for (int i = 0; i < LM_ELIM; i++) b.E[i] = E[i];
        }
    };
    In1 A1;
    In2 A2;
    for (int ch = T.chunk0 + wv; ch < T.chunk1; ch += BUILD_WAVES) {
        load1(ch, A1); load2(A1, A2);
        {
            double2* z = (double2*)Zb;
            const double2 zero2 = make_double2(0.0, 0.0);
            for (int i = ln; i < rows_used * LM_KS / 2; i += 64) z[i] = zero2;
        }
        const bool have = A1.ob0 + ln < A1.ob1;
        const int ls = A1.ls;
        int row = -1;
        double z[18];
#pragma unroll
        for (int i = 0; i < 18; i++) z[i] = 0.0;
        const bool lfree = have && A2.lcode == 0;
        if (lfree) {
            row = rowTab[A1.sl];
            if (row >= 0) {   // a constant key-frame has no rows in the reduced system
                const double* tab = poseTab + A1.sl * POSE_TAB;
                const double* ct = camTab + (A1.cam - T.cam_base) * 17;
                double r[2], Jp[12], Jl[6];
                if (FACTOR == 0) pixel_factor<true>(tab, ct, ct + 4, A2.pw, A1.m[0], A1.m[1], ct[16], r, Jp, Jl);
                else angular_factor<true>(tab, ct + 4, A2.pw, A1.m, ct[16], r, Jp, Jl);
                const double* L = A2.E;
                double W[6];   // W = Jl L (2 x 3), L lower: rows (l00) (l10 l11) (l20 l21 l22)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    W[3 * q] = Jl[3 * q] * L[0] + Jl[3 * q + 1] * L[1] + Jl[3 * q + 2] * L[3];
                    W[3 * q + 1] = Jl[3 * q + 1] * L[2] + Jl[3 * q + 2] * L[4];
                    W[3 * q + 2] = Jl[3 * q + 2] * L[5];
                }
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int c = 0; c < 3; c++) z[3 * i + c] = Jp[i] * W[c] + Jp[6 + i] * W[3 + c];
            }
        }
        SADVIO_TS(3, 35);
        // the (at most two, adjacent) observations of a landmark in one key-frame share their strip rows: the first one stores the sum
        const int key = row >= 0 ? (row << 8) | ls : -1 - ln;            // unique negative where there is nothing to store
        const int key_prev = dpp_i32<0x138>(key), key_next = dpp_i32<0x130>(key);   // wave_shr:1 / wave_shl:1
        const int ls_prev = dpp_i32<0x138>(have ? ls : -1);
        const bool follower = row >= 0 && ln > 0 && key_prev == key;
        const bool has_follower = row >= 0 && ln < 63 && key_next == key;
#pragma unroll
        for (int i = 0; i < 18; i++) {
            const double zn = dpp_f64<0x130>(z[i]);
            if (has_follower) z[i] += zn;
        }
        wave_lds_fence();   // the zeros are in place
        if (row >= 0 && !follower) {
            double* zrow = Zb + row * LM_KS + 3 * ls;
#pragma unroll
            for (int i = 0; i < 6; i++) { zrow[i * LM_KS] = z[3 * i]; zrow[i * LM_KS + 1] = z[3 * i + 1]; zrow[i * LM_KS + 2] = z[3 * i + 2]; }
        }
        if (lfree && (ln == 0 || ls_prev != ls)) {   // the landmark's first observation writes w
            double* d = Zb + w_row * LM_KS + 3 * ls;
            d[0] = A2.E[6]; d[1] = A2.E[7]; d[2] = A2.E[8];
        }
        wave_lds_fence();
        SADVIO_TS(3, 36);
        lm_syrk_pass(Zb, lr, lk, nt16, (3 * A1.nlm + 3) >> 2, accZ);   // accZ += strip strip^T over the chunk's 3 * nlm columns
        SADVIO_TS(3, 37);
        wave_lds_fence();   // the strip is zeroed again by the next chunk
        SADVIO_TS(3, 38);
    }
    SADVIO_TS(3, 39);
    if (T.chunk0 + wv < T.chunk1) {
#pragma unroll
        for (int pp = 0; pp < 3; pp++) {
            const int tr = pp < 1 ? 0 : 1, tc = pp - tr;
            if (tr < nt16) {
                const int col = 16 * tc + lr;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int row = 16 * tr + lk + 4 * rg;
                    if (row < Nt && col <= row) atomic_add_f64(&Stile[tri(row, col)], -accZ[pp][rg]);
                    else if (row == w_row && col < Nt) atomic_add_f64(&gT[col], -accZ[pp][rg]);
                }
            }
        }
    }
    __syncthreads();
    SADVIO_TS(3, 40);
    // flush the tile's non-zeros (same mapping as k_build: local rows 6 * rank, list sorted by global index)
    double* Sg = P.S + T.S_off;
    int* growTab = (int*)strips;
    for (int k = tid; k < T.n_kf; k += blockDim.x) {
        const int r = rowTab[k];
        if (r >= 0) {
            const int fi = P.kf_fidx[P.tile_kf[T.kf_off + k]];
#pragma unroll
            for (int i = 0; i < 6; i++) growTab[r + i] = fi * T.dpf + i;
        }
    }
    __syncthreads();
    if (T.ld) {
        for (int row = wv; row < Nt; row += BUILD_WAVES) {
            const long long grow = (long long)growTab[row] * T.ld;
            const double* srow = Stile + tri(row, 0);
            for (int col = ln; col <= row; col += 64) {
                const double v = srow[col];
                if (v != 0.0) atomic_add_f64(&Sg[grow + growTab[col]], v);
            }
        }
    } else {
        for (int col = wv; col < Nt; col += BUILD_WAVES) {
            const int gc = growTab[col];
            for (int row = col + ln; row < Nt; row += 64) {
                const double v = Stile[tri(row, col)];
                if (v != 0.0) atomic_add_f64(&Sg[c16_index(growTab[row], gc)], v);
            }
        }
    }
    for (int i = tid; i < Nt; i += blockDim.x)
        if (gT[i] != 0.0) atomic_add_f64(&P.gred[T.red_off + growTab[i]], gT[i]);
    SADVIO_TS(3, 42);
}

// ---- K5c: the observation-diagonal part, key-frame major -------------------------------------------------------------------
// D_f = sum over the observations of key-frame f of Jp^T Jp (6 x 6), g_f = sum Jp^T r: one workgroup per segment of a
// key-frame's observations (the window's observations sorted by key-frame, built with the tiles), every lane keeps its 27
// sums in registers over its share of the segment; one reduction per workgroup, then 39 global atomics (the diagonal block of
// S, gred, gfull, hdiag). Independent of the landmark elimination.
struct DiagSeg { int w, kf, begin, end; };   // window, global key-frame (free), slice of the key-frame-sorted observation list
constexpr int DIAG_SEG = 4096;               // observations per workgroup (16 per lane: the 27-value reduction is paid once per workgroup, one workgroup per key-frame at config 2)

template <int FACTOR>
__global__ __launch_bounds__(BUILD_THREADS, 2) void k_diag(   // the bearing factor spills 188 B per lane at 168 VGPRs
    DevPtrs P, const DiagSeg* segs, const int* kf_lmk, const int* kf_cam, const double* kf_meas, int slot) {
    const DiagSeg sg = segs[blockIdx.x];
    const WinDev& W = P.win[sg.w];
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const LmState st = P.states[(long long)sg.w * P.state_stride + slot];
    if (st.done) return;
    __shared__ double tab[POSE_TAB];
    __shared__ double camTab[MAX_WIN_CAM * 17];
    __shared__ double red[BUILD_WAVES][27];
    {
        const double* src = P.ptab + (long long)st.cur * P.ptab_stride + (long long)sg.kf * POSE_TAB;
        for (int i = tid; i < POSE_TAB; i += blockDim.x) tab[i] = src[i];
        for (int i = tid; i < W.n_cam * 17; i += blockDim.x) {
            const int c = i / 17, e = i - 17 * c;
            const int gc = W.cam_base + c;
            camTab[i] = e < 4 ? P.cam_K[4 * (long long)gc + e] : (e < 16 ? P.cam_T[12 * (long long)gc + e - 4] : P.cam_isig[gc]);
        }
    }
    __syncthreads();
    const double* xl = P.xl + (long long)st.cur * P.xl_stride;
    double D[21], gf[6];
#pragma unroll
    for (int i = 0; i < 21; i++) D[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) gf[i] = 0.0;
    // Software pipeline over the lane's observations, two levels deep: the landmark index of observation j + 2 and the landmark
    // position / measurement of observation j + 1 are in flight while observation j is linearised. The landmark is a gather (the
    // list is sorted by key-frame), and at 2 waves / SIMD nothing else hides its latency: without the pipeline the kernel ran at
    // 12 % VALU utilisation (175 us per 64 windows beside k_elim / k_build_obs, round-3 counters).
    struct DObs { double p[3], x[3], m[3]; int cam; };
    auto ld_idx = [&](long long i) -> long long { return i < sg.end ? (long long)kf_lmk[i] : -1LL; };
    auto ld_obs = [&](long long i, long long gl, DObs& o) {
        if (gl < 0) return;
#pragma unroll
        for (int k = 0; k < 3; k++) { o.p[k] = P.lmk_p[3 * gl + k]; o.x[k] = xl[3 * gl + k]; }
        o.cam = kf_cam[i];
        if (FACTOR == 0) { const double2 mm = *(const double2*)(kf_meas + 2 * i); o.m[0] = mm.x; o.m[1] = mm.y; o.m[2] = 0.0; }
        else { o.m[0] = kf_meas[3 * i]; o.m[1] = kf_meas[3 * i + 1]; o.m[2] = kf_meas[3 * i + 2]; }
    };
    DObs A{}, B{};
    long long i = sg.begin + tid;
    long long gl0 = ld_idx(i), gl1 = ld_idx(i + BUILD_THREADS);
    ld_obs(i, gl0, A);
    for (; i < sg.end; i += BUILD_THREADS) {
        const long long gl2 = ld_idx(i + 2 * BUILD_THREADS);
        ld_obs(i + BUILD_THREADS, gl1, B);
        const double pw[3] = {A.p[0] + A.x[0], A.p[1] + A.x[1], A.p[2] + A.x[2]};
        const double* ct = camTab + (A.cam - W.cam_base) * 17;
        double r[2], Jp[12], Jl[6];
        if (FACTOR == 0) pixel_factor<true>(tab, ct, ct + 4, pw, A.m[0], A.m[1], ct[16], r, Jp, Jl);
        else angular_factor<true>(tab, ct + 4, pw, A.m, ct[16], r, Jp, Jl);
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b <= a; b++) D[e++] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
            gf[a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
        }
        A = B; gl1 = gl2;
    }
#pragma unroll
    for (int i = 0; i < 21; i++) { const double v = group_sum(D[i], 64); if (ln == 0) red[wv][i] = v; }   // DPP / permlane swaps: no LDS shuffles
#pragma unroll
    for (int i = 0; i < 6; i++) { const double v = group_sum(gf[i], 64); if (ln == 0) red[wv][21 + i] = v; }
    __syncthreads();
    if (tid < 27) {
        double v = 0.0;
        for (int k = 0; k < BUILD_WAVES; k++) v += red[k][tid];
        const int base = P.kf_fidx[sg.kf] * W.dpf;
        if (tid < 21) {
            int a = 0, b = tid;
            while (b >= a + 1) { b -= a + 1; a++; }
            atomic_add_f64(&P.S[W.S_off + s_index(W.ld, base + a, base + b)], v);
            if (a == b) atomic_add_f64(&P.hdiag[W.red_off + base + a], v);
        } else {
            atomic_add_f64(&P.gred[W.red_off + base + tid - 21], v);
            atomic_add_f64(&P.gfull[W.red_off + base + tid - 21], v);
        }
    }
}

// ---- K7: back-substitution, one lane per landmark ------------------------------------------------------------------------
template <int FACTOR>
__global__ __launch_bounds__(BUILD_THREADS, 3) void k_backsub_lm(DevPtrs P, int slot, int max_tile_kf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ti = P.tile_perm[blockIdx.x];   // largest tiles first (host: LPT order), so that the launch does not end on a long tile
    const Tile T = P.tiles[ti];
    const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
    const LmState st = P.states[(long long)T.w * P.state_stride + slot];
    IterAcc* acc = P.acc + (long long)T.w * P.state_stride + slot;
    zero_s_slice(P, T, ti);
    if (st.done || acc->chol_fail) {
        if (tid == 0) {
            TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + ti;
            ta->cand_cost = 0.0; ta->mcc = 0.0; ta->step_norm2 = 0.0; ta->cand_norm2 = 0.0;
        }
        return;
    }
    double* poseTab = (double*)smem;
    double* camTab = poseTab + (size_t)max_tile_kf * POSE_TAB;
    int* rowTab = (int*)(camTab + MAX_WIN_CAM * 17);
    double* candTab = (double*)(smem + tile_tables_bytes(max_tile_kf));  // [n_kf][12] R|t at the candidate poses
    double* dpTab = candTab + (size_t)max_tile_kf * 12;                  // [n_kf][6] pose step of each listed key-frame
    const int cur = st.cur;
    stage_tables(P, T, cur, poseTab, camTab, rowTab);
    {
        const double* src = P.ptab + (long long)(1 - cur) * P.ptab_stride;
        for (int i = tid; i < T.n_kf * 12; i += blockDim.x) {
            const int k = i / 12, e = i - 12 * k;
            candTab[i] = src[(long long)P.tile_kf[T.kf_off + k] * POSE_TAB + e];
        }
        const double* dp = P.delta + T.red_off;
        for (int i = tid; i < T.n_kf * 6; i += blockDim.x) {
            const int k = i / 6, e = i - 6 * k;
            const int fi = P.kf_fidx[P.tile_kf[T.kf_off + k]];
            dpTab[i] = fi < 0 ? 0.0 : dp[fi * T.dpf + e];
        }
    }
    __syncthreads();
    const int nl = T.lmk1 - T.lmk0;
    const double* xl = P.xl + (long long)cur * P.xl_stride;
    double* xlc = P.xl + (long long)(1 - cur) * P.xl_stride;
    double sn = 0.0, cn = 0.0, mcc = 0.0, cc = 0.0;
    for (int lm = tid; lm < nl; lm += (int)blockDim.x) {
        const int gl = T.lmk0 + lm;
        const int ob = P.lmk_ob[gl], oe = P.lmk_oe[gl];
        const int lcode = P.lmk_const ? P.lmk_const[gl] : 0;
        const double p0[3] = {P.lmk_p[3 * (long long)gl], P.lmk_p[3 * (long long)gl + 1], P.lmk_p[3 * (long long)gl + 2]};
        const double x0[3] = {xl[3 * (long long)gl], xl[3 * (long long)gl + 1], xl[3 * (long long)gl + 2]};
        const double pw[3] = {p0[0] + x0[0], p0[1] + x0[1], p0[2] + x0[2]};
        // sums over the landmark's residual blocks that are in the program (u = Jp dp):
        //   H = sum Jl^T Jl, g = sum Jl^T r, t = sum Jl^T (r + u), a1 = sum u . r, a2 = sum u . u
        double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, t[3] = {0, 0, 0}, a1 = 0.0, a2 = 0.0;
        ObsIn nx = lm_load_obs<FACTOR>(P, ob, oe);
        for (int o = ob; o < oe; o++) {
            const ObsIn cu = nx;
            nx = lm_load_obs<FACTOR>(P, o + 1, oe);
            const int sl = cu.sl;
            const int row = rowTab[sl];
            if (row < 0 && lcode == 1) continue;   // a block of constant parameters
            const double* ct = camTab + (cu.cam - T.cam_base) * 17;
            double r[2], Jp[12], Jl[6];
            lm_linearize_in<FACTOR, true>(poseTab + sl * POSE_TAB, ct, cu, pw, r, Jp, Jl);
            double u0 = 0.0, u1 = 0.0;
            if (row >= 0) {
                const double* d = dpTab + sl * 6;
#pragma unroll
                for (int i = 0; i < 6; i++) { u0 += Jp[i] * d[i]; u1 += Jp[6 + i] * d[i]; }
            }
            a1 += u0 * r[0] + u1 * r[1];
            a2 += u0 * u0 + u1 * u1;
            if (lcode == 0) {
                const double e0 = r[0] + u0, e1 = r[1] + u1;
                H[0] += Jl[0] * Jl[0] + Jl[3] * Jl[3]; H[1] += Jl[0] * Jl[1] + Jl[3] * Jl[4]; H[2] += Jl[0] * Jl[2] + Jl[3] * Jl[5];
                H[3] += Jl[1] * Jl[1] + Jl[4] * Jl[4]; H[4] += Jl[1] * Jl[2] + Jl[4] * Jl[5]; H[5] += Jl[2] * Jl[2] + Jl[5] * Jl[5];
                g[0] += Jl[0] * r[0] + Jl[3] * r[1]; g[1] += Jl[1] * r[0] + Jl[4] * r[1]; g[2] += Jl[2] * r[0] + Jl[5] * r[1];
                t[0] += Jl[0] * e0 + Jl[3] * e1; t[1] += Jl[1] * e0 + Jl[4] * e1; t[2] += Jl[2] * e0 + Jl[5] * e1;
            }
        }
        double d0 = 0.0, d1 = 0.0, d2 = 0.0;
        const bool active = lcode == 0 && oe > ob;
        if (active) {
            const double s[3] = {P.s_lmk[3 * (long long)gl], P.s_lmk[3 * (long long)gl + 1], P.s_lmk[3 * (long long)gl + 2]};
            double Mi[6];
            lm_damped_inverse(P, H, s, st.radius, Mi);
            d0 = -(Mi[0] * t[0] + Mi[1] * t[1] + Mi[2] * t[2]);
            d1 = -(Mi[1] * t[0] + Mi[3] * t[1] + Mi[4] * t[2]);
            d2 = -(Mi[2] * t[0] + Mi[4] * t[1] + Mi[5] * t[2]);
            sn += d0 * d0 + d1 * d1 + d2 * d2;
        }
        const double c0 = x0[0] + d0, c1 = x0[1] + d1, c2 = x0[2] + d2;
        xlc[3 * (long long)gl] = c0; xlc[3 * (long long)gl + 1] = c1; xlc[3 * (long long)gl + 2] = c2;
        if (active) cn += c0 * c0 + c1 * c1 + c2 * c2;
        // model cost change -sum m . (r + m / 2), m = u + Jl delta:
        //   sum m . r = a1 + delta . g;  sum m . m = a2 + 2 delta . (t - g) + delta^T H delta
        {
            const double dg = d0 * g[0] + d1 * g[1] + d2 * g[2];
            const double dc = d0 * (t[0] - g[0]) + d1 * (t[1] - g[1]) + d2 * (t[2] - g[2]);
            const double dHd = d0 * (H[0] * d0 + H[1] * d1 + H[2] * d2) + d1 * (H[1] * d0 + H[3] * d1 + H[4] * d2) + d2 * (H[2] * d0 + H[4] * d1 + H[5] * d2);
            mcc += -(a1 + dg) - 0.5 * (a2 + 2.0 * dc + dHd);
        }
        // residuals at the candidate point
        const double pc[3] = {p0[0] + c0, p0[1] + c1, p0[2] + c2};
        nx = lm_load_obs<FACTOR>(P, ob, oe);
        for (int o = ob; o < oe; o++) {
            const ObsIn cu = nx;
            nx = lm_load_obs<FACTOR>(P, o + 1, oe);
            const int sl = cu.sl;
            if (rowTab[sl] < 0 && lcode == 1) continue;
            const double* ct = camTab + (cu.cam - T.cam_base) * 17;
            double r[2];
            lm_linearize_in<FACTOR, false>(candTab + sl * 12, ct, cu, pc, r, nullptr, nullptr);
            cc += r[0] * r[0] + r[1] * r[1];
        }
    }
    sn = wave_sum(sn); cn = wave_sum(cn); mcc = wave_sum(mcc); cc = wave_sum(cc);
    __shared__ double s_part[BUILD_WAVES * 4];
    if (ln == 0) { s_part[wv * 4] = cc; s_part[wv * 4 + 1] = mcc; s_part[wv * 4 + 2] = sn; s_part[wv * 4 + 3] = cn; }
    __syncthreads();
    if (tid == 0) {
        double a0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) { a0 += s_part[k * 4]; b1 += s_part[k * 4 + 1]; b2 += s_part[k * 4 + 2]; b3 += s_part[k * 4 + 3]; }
        TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + ti;
        ta->cand_cost = a0; ta->mcc = b1; ta->step_norm2 = b2; ta->cand_norm2 = b3;
    }
}

}  // namespace sadvio
