// lm_kernels.h — throughput variants of K5 / K7 for large batches of plain visual windows (no robust loss, no prior-kept
// landmarks, no pseudo-observations, one GPU per window).
//
// kernels.h maps one LANE to one OBSERVATION with a landmark's observations in a group of G lanes: every reduction over a
// landmark is a lane shuffle and the dependent chain of a tile is as short as it gets, which is what one window (250
// workgroups on 256 CUs) needs. A batch of 64 windows is instruction-bound instead (rocprofv3, round-2 notes in DESIGN.md:
// VALU busy 46 %, 2 waves / SIMD): 5 of 8 lanes carry an observation, the 3 x 3 elimination is repeated by every lane of
// a group and the block sums travel through DPP. Here the same arithmetic is laid out for throughput, and (round 5) in THREE
// passes over the observation list per LM step — the algorithmic count of SURVEY.md 8d — instead of five:
//   k_build_obs   pass 1, one lane per OBSERVATION, flat (no padding). Every lane forms its landmark's damping from the H_ll, g_l record
//                 k_lm_pass left: M = H_ll + D = L L^T, L^-1, w = L^-1 g_l (Cholesky form: the block step of a landmark-first
//                 Cholesky of the un-reduced system); with W = Jl L^-T the landmark's Schur term is E M^-1 E^T = Z Z^T,
//                 Z = sum_a Jp_a^T W_a, so ONE strip
//                 Z[row][3 * landmark + c] per wave feeds both MFMA operands (v_mfma_f64_16x16x4_f64, K = the landmarks of a
//                 chunk); a spare strip row with w makes the same pass return Z w = E M^-1 g_l, the landmark part of the reduced
//                 gradient; the key-frame blocks sum Jp^T Jp, sum Jp^T r come from the tile's record; tile flush as in k_build
//   k_lm_pass     passes 2 and 3, one lane per landmark over LDS-staged observation constants: delta_l = -M^-1 sum Jl^T (r + Jp dp)
//                 and the model cost change from sums accumulated in the same loop; then the candidate pass, which LINEARISES at
//                 x + delta: candidate cost, and the H_ll / g_l record of every landmark and the key-frame sums of the tile in the
//                 buffer of the candidate — what pass 1 of the next step needs if the step is accepted (a rejected step leaves x
//                 and its records untouched). INIT = true is the opening launch of a solve: the same pass at x itself.
// The tiles are those of k_build; a tile's landmarks are cut into chunks of <= 12 landmarks and <= 64 observations
// (chunk tables built with the tiles).
#pragma once
#include "kernels.h"

namespace sadvio {

constexpr int LM_CHUNK = 12;            // landmarks per MFMA chunk (12 x 5 observations fill 60 of 64 lanes)
constexpr int LM_KS = 3 * LM_CHUNK + 2; // strip row stride (doubles): 3 columns per landmark + 2 (bank spread); 32 x 38 doubles = 9.5 KB per wave
constexpr int LM_HG = 9;                // per landmark and delta buffer: H_ll (00 01 02 11 12 22) | g_l (3) at that buffer's point
constexpr int LM_DT_RANK = 28;          // per tile, delta buffer and free key-frame of the tile: sum Jp^T Jp (21, lower, row-major) | sum Jp^T r (6) | -
constexpr int LM_DT_COST = MAX_GEMM_FREE_KF * LM_DT_RANK;   // then: sum r^2 (blocks in the program) | of the constant blocks | max |g_l| | -
constexpr int LM_DT = LM_DT_COST + 4;
constexpr int LM_PASS_THREADS = 64;     // k_lm_pass: one wave per tile and sub-block of 64 landmarks (measured: 64 > 128 > 256 > 192 threads, 211 / 202 / 188 / 163 k it/s)

// LM damping of a landmark's 3 x 3 block and the inverse of its Cholesky factor (the arithmetic of group_eliminate, one lane):
// Li = L^-1 (packed lower, sym3_chol_inverse), M = H_ll + D = L L^T
__device__ __forceinline__ void lm_damped_inverse(const DevPtrs& P, const double* H, const double* s, double radius, double* Mi) {
    const double ir = 1.0 / radius;
    const double s0 = s[0] * s[0], s1 = s[1] * s[1], s2 = s[2] * s[2];
    double M[6] = {H[0], H[1], H[2], H[3], H[4], H[5]};
    M[0] += fmin(fmax(s0 * H[0], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s0;
    M[3] += fmin(fmax(s1 * H[3], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s1;
    M[5] += fmin(fmax(s2 * H[5], P.o.min_lm_diagonal), P.o.max_lm_diagonal) * ir / s2;
    sym3_chol_inverse(M, Mi);
}

// View tables of the pixel factor (throughput kernels): per (key-frame slot, camera) of a tile the products every observation of
// that view shares — M1 = Rsf R, t1 = Rsf t + tsf (p_c = M1 p_w + t1), M0 = Rsf R0 (translation block of Jp), the intrinsics and
// 1 / sigma: 26 doubles instead of 47 table entries per observation, 9 + 12 + 12 fused multiply-adds instead of 18 + 8 + 18 + 18.
// Built once per workgroup from the staged pose / camera tables (LM_VT doubles per view, view = slot * n_cam + camera).
constexpr int LM_VT = 26;   // M1 9 | t1 3 | M0 9 | fx fy cx cy | 1 / sigma
__device__ __forceinline__ void lm_stage_views(const double* poseTab, const double* camTab, int n_kf, int n_cam, double* vt) {
    for (int v = threadIdx.x; v < n_kf * n_cam; v += blockDim.x) {
        const int k = v / n_cam, c = v - k * n_cam;
        const double* tab = poseTab + k * POSE_TAB;
        const double* ct = camTab + c * 17;     // K[4] Tsf[12] isig
        const double* Rsf = ct + 4;
        double* o = vt + v * LM_VT;
        m3_mul(Rsf, tab, o);
        double t1[3];
        m3_vec(Rsf, tab + 9, t1);
        o[9] = t1[0] + Rsf[9]; o[10] = t1[1] + Rsf[10]; o[11] = t1[2] + Rsf[11];
        m3_mul(Rsf, tab + 21, o + 12);
        o[21] = ct[0]; o[22] = ct[1]; o[23] = ct[2]; o[24] = ct[3]; o[25] = ct[16];
    }
}
// pixel_factor (device_math.h) from a view table: the same residual, validity rule (Camera.cpp:128-136) and Jacobians
template <bool WANT_J>
__device__ __forceinline__ void pixel_factor_view(const double* vt, const double* Jr, const double* pw, double u_meas, double v_meas,
                                                  double* r, double* Jp, double* Jl) {
    double tc[3];
    m3_vec(vt, pw, tc);
    tc[0] += vt[9]; tc[1] += vt[10]; tc[2] += vt[11];
    const double fx = vt[21], fy = vt[22], cx = vt[23], cy = vt[24], isig = vt[25];
    const double iz = 1.0 / tc[2];
    const double u = (fx * tc[0] + cx * tc[2]) * iz;
    const double v = (fy * tc[1] + cy * tc[2]) * iz;
    const bool valid = !(tc[2] < 0.1) && !(u < 0 || v < 0 || u > 2 * cx || v > 2 * cy) && isfinite(u) && isfinite(v);
    r[0] = valid ? isig * (u - u_meas) : 0.0;
    r[1] = valid ? isig * (v - v_meas) : 0.0;
    if (WANT_J) {
        const double a0 = isig * fx * iz, a1 = isig * fy * iz;
        const double b0 = -a0 * tc[0] * iz, b1 = -a1 * tc[1] * iz;
        const double* M0 = vt + 12;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Jl[j] = a0 * vt[j] + b0 * vt[6 + j];
            Jl[3 + j] = a1 * vt[3 + j] + b1 * vt[6 + j];
            Jp[3 + j] = a0 * M0[j] + b0 * M0[6 + j];
            Jp[9 + j] = a1 * M0[3 + j] + b1 * M0[6 + j];
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double l0 = Jl[3 * q], l1 = Jl[3 * q + 1], l2 = Jl[3 * q + 2];
            const double s0 = l1 * pw[2] - l2 * pw[1];
            const double s1 = l2 * pw[0] - l0 * pw[2];
            const double s2 = l0 * pw[1] - l1 * pw[0];
#pragma unroll
            for (int j = 0; j < 3; j++) Jp[6 * q + j] = -(s0 * Jr[j] + s1 * Jr[3 + j] + s2 * Jr[6 + j]);
        }
    }
}
// Linearisation of one observation in the throughput kernels: the pixel factor from the view table of (slot, camera), the bearing
// factor from the pose / camera tables as in kernels.h
template <int FACTOR, bool WANT_J>
__device__ __forceinline__ void lm_linearize(const double* poseTab, const double* camTab, const double* vt, int n_cam, int sl, int cam,
                                             const double* m, const double* pw, double* r, double* Jp, double* Jl) {
    if (FACTOR == 0) pixel_factor_view<WANT_J>(vt + (sl * n_cam + cam) * LM_VT, poseTab + sl * POSE_TAB + 12, pw, m[0], m[1], r, Jp, Jl);
    else { const double* ct = camTab + cam * 17; angular_factor<WANT_J>(poseTab + sl * POSE_TAB, ct + 4, pw, m, ct[16], r, Jp, Jl); }
}

// Sum of 27 per-lane values over the wave: the first two levels exchange HALF of the values between lane pairs (xor 1, xor 2:
// 14 + 7 additions instead of 2 x 27), the seven values a lane is left with are summed over the lanes that share its position q
// in the quad (row_ror 4 / 8 inside the 16-lane row, then the permlane swaps across rows). A lane at position q returns, in
// v[0..6], the wave totals of the values e = i + 7 * (q >> 1) + 14 * (q & 1)   (e >= 27: zero).
__device__ __forceinline__ void lm_reduce27(const double* D, int ln, double* v) {
    const bool odd1 = ln & 1, odd2 = ln & 2;
    double t[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const double a = D[i], b = i + 14 < 27 ? D[i + 14] : 0.0;
        const double keep = odd1 ? b : a, send = odd1 ? a : b;
        t[i] = keep + dpp_f64<0xB1>(send);        // quad_perm [1,0,3,2]: the lane's xor-1 partner sends the half this lane keeps
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const double a = t[i], b = t[i + 7];
        const double keep = odd2 ? b : a, send = odd2 ? a : b;
        double x = keep + dpp_f64<0x4E>(send);    // quad_perm [2,3,0,1]: xor 2
        x += dpp_f64<0x124>(x);                   // row_ror:4  } the four quads of the row, same position
        x += dpp_f64<0x128>(x);                   // row_ror:8  }
        x = xor16_sum(x);
        v[i] = xor32_sum(x);
    }
}

// ---- K7 and the linearisation of the next step: one lane per landmark ---------------------------------------------------------
// Key-frame sums in a lane-per-landmark loop: consecutive landmarks observe the same key-frames, so at step j of the loop most
// lanes of a wave hold an observation of ONE key-frame. Every lane accumulates its 27 products (sum Jp^T Jp lower, sum Jp^T r) in
// registers under a wave-uniform key (the key-frame's slot in the tile); lanes with another key take another turn of the same
// step; when the key changes the wave's sums are reduced (lm_reduce27) and added to the tile's record in LDS: about as many
// reductions per wave as the wave's landmarks touch key-frames on a time-ordered map, more — never wrong — on a shuffled one.
// One workgroup (one wave) per SUB-BLOCK of LM_PASS_THREADS consecutive landmarks of a tile (lm_sub: tile | index in the tile): with a
// workgroup per tile the launch was 1.5 rounds of ~60 us workgroups on 2 waves / SIMD (a third of the chip idle in the second
// round); the sub-blocks' partial records (key-frame sums, cost totals) are summed by their consumers (k_build_obs, k_decide).
template <int FACTOR, bool INIT>
__global__ __launch_bounds__(LM_PASS_THREADS, 2) void k_lm_pass(DevPtrs P, int slot, int max_tile_kf, int sub_obs_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MS = FACTOR == 0 ? 2 : 3;
    const int2 sbi = ((const int2*)P.lm_sub)[blockIdx.x];   // work list: (tile, first sub-block, ...) — largest tiles first (host: LPT order)
    const int ti = sbi.x, sb_first = sbi.y;
    const long long rec_i = (long long)ti * P.lm_ksub + sb_first;  // this work item's partial records
    const Tile T = P.tiles[ti];
    const int sb_end = min(sb_first + P.lm_sub_per_item, (T.lmk1 - T.lmk0 + LM_PASS_THREADS - 1) / LM_PASS_THREADS);
    const int tid = threadIdx.x, ln = tid & 63;
    const LmState st = P.states[(long long)T.w * P.state_stride + slot];
    IterAcc* acc = P.acc + (long long)T.w * P.state_stride + slot;
    if (!INIT && sb_first == 0) zero_s_slice(P, T, ti);
    double* sacc = P.lm_sacc + ((long long)(slot & 1) * P.n_tiles * P.lm_ksub + rec_i) * 4;   // cand_cost | mcc | step_norm2 | cand_norm2
    if (st.done || (!INIT && acc->chol_fail)) {
        if (!INIT && tid == 0) { sacc[0] = 0.0; sacc[1] = 0.0; sacc[2] = 0.0; sacc[3] = 0.0; }
        return;
    }
    SADVIO_TS(3, 44);
    const int cur = st.cur, cbuf = INIT ? cur : 1 - cur;
    double* poseTab = (double*)smem;                                        // tables at x
    double* camTab = poseTab + (size_t)max_tile_kf * POSE_TAB;
    int* rowTab = (int*)(camTab + MAX_WIN_CAM * 17);
    char* sp = smem + tile_tables_bytes(max_tile_kf);
    double* candTab = (double*)sp; sp += INIT ? 0 : sizeof(double) * (size_t)max_tile_kf * POSE_TAB;   // tables at the candidate
    double* dpTab = (double*)sp; sp += INIT ? 0 : sizeof(double) * (size_t)max_tile_kf * 6;             // pose step of each listed key-frame
    double* Dacc = (double*)sp; sp += sizeof(double) * LM_DT_COST;                                        // the tile's key-frame sums
    double* s_meas = (double*)sp; sp += sizeof(double) * (size_t)MS * sub_obs_cap;
    int* s_ocs = (int*)sp; sp += sizeof(int) * (size_t)sub_obs_cap;                                       // camera (tile-local) | slot << 8
    double* vtX = (double*)sp; sp += FACTOR == 0 ? sizeof(double) * (size_t)max_tile_kf * T.n_cam * LM_VT : 0;             // view tables at x
    double* vtC = (double*)sp;                                                                                                // and at the candidate
    stage_tables(P, T, cur, poseTab, camTab, rowTab);
    if (!INIT) {
        const double* src = P.ptab + (long long)cbuf * P.ptab_stride;
        for (int i = tid; i < T.n_kf * POSE_TAB; i += blockDim.x) {
            const int k = i / POSE_TAB, e = i - POSE_TAB * k;
            candTab[i] = src[(long long)P.tile_kf[T.kf_off + k] * POSE_TAB + e];
        }
        const double* dp = P.delta + T.red_off;
        for (int i = tid; i < T.n_kf * 6; i += blockDim.x) {
            const int k = i / 6, e = i - 6 * k;
            const int fi = P.kf_fidx[P.tile_kf[T.kf_off + k]];
            dpTab[i] = fi < 0 ? 0.0 : dp[fi * T.dpf + e];
        }
    }
    for (int i = tid; i < LM_DT_COST; i += blockDim.x) Dacc[i] = 0.0;
    const double* linTab = INIT ? poseTab : candTab;     // where the last pass linearises
    const double* linVt = INIT ? vtX : vtC;
    if (FACTOR == 0) {
        __syncthreads();
        SADVIO_TS(3, 45);
        lm_stage_views(poseTab, camTab, T.n_kf, T.n_cam, vtX);
        if (!INIT) lm_stage_views(candTab, camTab, T.n_kf, T.n_cam, vtC);
    }
    const double* xl = P.xl + (long long)cur * P.xl_stride;
    double* xlc = P.xl + (long long)(1 - cur) * P.xl_stride;
    double* hg = P.lm_hg + (long long)cbuf * P.lm_hg_stride;
    double sn = 0.0, cn = 0.0, mcc = 0.0, cc = 0.0, fixed_part = 0.0, gmax_part = 0.0;
    double D[27];
#pragma unroll
    for (int i = 0; i < 27; i++) D[i] = 0.0;
    int cur_key = -1;                                    // wave-uniform: the key-frame slot the wave's D belongs to
    auto flush = [&]() {
        double v[7];
        lm_reduce27(D, ln, v);
        const int q = ln & 3;
        double* dst = Dacc + (rowTab[cur_key] / 6) * LM_DT_RANK + 7 * (q >> 1) + 14 * (q & 1);
        if (ln < 4) {
#pragma unroll
            for (int i = 0; i < 7; i++)
                if (i + 7 * (q >> 1) + 14 * (q & 1) < 27) atomic_add_f64(dst + i, v[i]);
        }
#pragma unroll
        for (int i = 0; i < 27; i++) D[i] = 0.0;
    };
    for (int sb = sb_first; sb < sb_end; sb++) {
        // the sub-block's observation constants: one coalesced copy into LDS
        const int l0 = T.lmk0 + sb * LM_PASS_THREADS, l1 = min(l0 + LM_PASS_THREADS, T.lmk1);
        const int ob0 = P.lmk_ob[l0], n_ob = P.lmk_oe[l1 - 1] - ob0;
        if (sb != sb_first) __syncthreads();   // the previous sub-block's readers are done
        for (int i = tid; i < n_ob; i += blockDim.x) {
            const long long o = ob0 + i;
            s_ocs[i] = (P.obs_cam[o] - T.cam_base) | ((int)P.obs_slot[o] << 8);
            if (FACTOR == 0) *(double2*)(s_meas + 2 * i) = *(const double2*)(P.obs_meas + 2 * o);
            else { s_meas[3 * i] = P.obs_meas[3 * o]; s_meas[3 * i + 1] = P.obs_meas[3 * o + 1]; s_meas[3 * i + 2] = P.obs_meas[3 * o + 2]; }
        }
        const int gl = l0 + tid;
        const bool have = gl < l1;
        int ob = 0, cnt = 0, lcode = 1;
        double p0[3] = {0.0, 0.0, 0.0}, x0[3] = {0.0, 0.0, 0.0};
        if (have) {
            ob = P.lmk_ob[gl] - ob0; cnt = P.lmk_oe[gl] - P.lmk_ob[gl];
            lcode = P.lmk_const ? P.lmk_const[gl] : 0;
#pragma unroll
            for (int i = 0; i < 3; i++) { p0[i] = P.lmk_p[3 * (long long)gl + i]; x0[i] = xl[3 * (long long)gl + i]; }
        }
        __syncthreads();
        SADVIO_TS(3, 46);
        auto get = [&](int i, int& sl, int& cam, double* m) {
            const int v = s_ocs[i];
            sl = v >> 8; cam = v & 255;
            if (FACTOR == 0) { const double2 mm = *(const double2*)(s_meas + 2 * i); m[0] = mm.x; m[1] = mm.y; m[2] = 0.0; }
            else { m[0] = s_meas[3 * i]; m[1] = s_meas[3 * i + 1]; m[2] = s_meas[3 * i + 2]; }
        };
        double pc[3] = {p0[0] + x0[0], p0[1] + x0[1], p0[2] + x0[2]};   // INIT: the point itself
        if (!INIT) {
            // ---- pass 2: back-substitution at x. Sums over the landmark's residual blocks that are in the program (u = Jp dp):
            //   H = sum Jl^T Jl, g = sum Jl^T r, t = sum Jl^T (r + u), a1 = sum u . r, a2 = sum u . u
            const double pw[3] = {pc[0], pc[1], pc[2]};
            double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, t[3] = {0, 0, 0}, a1 = 0.0, a2 = 0.0;
            for (int j = 0; j < cnt; j++) {
                int sl, cam; double m[3];
                get(ob + j, sl, cam, m);
                const int row = rowTab[sl];
                if (row < 0 && lcode == 1) continue;   // a block of constant parameters
                double r[2], Jp[12], Jl[6];
                lm_linearize<FACTOR, true>(poseTab, camTab, vtX, T.n_cam, sl, cam, m, pw, r, Jp, Jl);
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
            const bool active = have && lcode == 0 && cnt > 0;
            if (active) {
                const double s[3] = {P.s_lmk[3 * (long long)gl], P.s_lmk[3 * (long long)gl + 1], P.s_lmk[3 * (long long)gl + 2]};
                double Li[6], u[3], v[3];
                lm_damped_inverse(P, H, s, st.radius, Li);
                li_vec(Li, t, u);          // delta_l = -M^-1 t = -Li^T (Li t)
                li_tvec(Li, u, v);
                d0 = -v[0]; d1 = -v[1]; d2 = -v[2];
                sn += d0 * d0 + d1 * d1 + d2 * d2;
            }
            const double c0 = x0[0] + d0, c1 = x0[1] + d1, c2 = x0[2] + d2;
            if (have) { xlc[3 * (long long)gl] = c0; xlc[3 * (long long)gl + 1] = c1; xlc[3 * (long long)gl + 2] = c2; }
            if (active) cn += c0 * c0 + c1 * c1 + c2 * c2;
            // model cost change -sum m . (r + m / 2), m = u + Jl delta:
            //   sum m . r = a1 + delta . g;  sum m . m = a2 + 2 delta . (t - g) + delta^T H delta
            {
                const double dg = d0 * g[0] + d1 * g[1] + d2 * g[2];
                const double dc = d0 * (t[0] - g[0]) + d1 * (t[1] - g[1]) + d2 * (t[2] - g[2]);
                const double dHd = d0 * (H[0] * d0 + H[1] * d1 + H[2] * d2) + d1 * (H[1] * d0 + H[3] * d1 + H[4] * d2) + d2 * (H[2] * d0 + H[4] * d1 + H[5] * d2);
                mcc += -(a1 + dg) - 0.5 * (a2 + 2.0 * dc + dHd);
            }
            pc[0] = p0[0] + c0; pc[1] = p0[1] + c1; pc[2] = p0[2] + c2;
        }
        SADVIO_TS(3, 47);
        // ---- pass 3 (INIT: the only one): linearise at the candidate. Wave-uniform trip count: the key-frame sums are wave-level
        double Hn[6] = {0, 0, 0, 0, 0, 0}, gn[3] = {0, 0, 0};
        for (int j = 0; j < T.kmax; j++) {
            const bool act = j < cnt;
            int sl = 0, cam = 0, row = -1;
            double m[3] = {0.0, 0.0, 0.0};
            if (act) { get(ob + j, sl, cam, m); row = rowTab[sl]; }
            const bool fixed_blk = row < 0 && lcode == 1;       // a block of constant parameters: only the opening pass needs its cost
            double r[2] = {0.0, 0.0}, Jp[12], Jl[6];
#pragma unroll
            for (int i = 0; i < 12; i++) Jp[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) Jl[i] = 0.0;
            if (act && (INIT || !fixed_blk)) lm_linearize<FACTOR, true>(linTab, camTab, linVt, T.n_cam, sl, cam, m, pc, r, Jp, Jl);
            const double c = r[0] * r[0] + r[1] * r[1];
            if (fixed_blk) fixed_part += c; else cc += c;
            if (act && lcode == 0) {
                Hn[0] += Jl[0] * Jl[0] + Jl[3] * Jl[3]; Hn[1] += Jl[0] * Jl[1] + Jl[3] * Jl[4]; Hn[2] += Jl[0] * Jl[2] + Jl[3] * Jl[5];
                Hn[3] += Jl[1] * Jl[1] + Jl[4] * Jl[4]; Hn[4] += Jl[1] * Jl[2] + Jl[4] * Jl[5]; Hn[5] += Jl[2] * Jl[2] + Jl[5] * Jl[5];
                gn[0] += Jl[0] * r[0] + Jl[3] * r[1]; gn[1] += Jl[1] * r[0] + Jl[4] * r[1]; gn[2] += Jl[2] * r[0] + Jl[5] * r[1];
            }
            // key-frame sums: this step's observations by key, the wave's current key first
            const bool has = act && row >= 0;
            unsigned long long todo = __ballot(has);
            while (todo) {
                int k = cur_key;
                if (!(__ballot(has && sl == cur_key) & todo)) k = __builtin_amdgcn_readlane(sl, __ffsll((long long)todo) - 1);
                if (k != cur_key) { if (cur_key >= 0) flush(); cur_key = k; }
                const bool mine = has && sl == k;
                if (mine) {
                    int e = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int b = 0; b <= a; b++) D[e++] += Jp[a] * Jp[b] + Jp[6 + a] * Jp[6 + b];
                    }
#pragma unroll
                    for (int a = 0; a < 6; a++) D[21 + a] += Jp[a] * r[0] + Jp[6 + a] * r[1];
                }
                todo &= ~__ballot(mine);
            }
        }
        if (have) {
            const bool active = lcode == 0 && cnt > 0;
            double* dst = hg + (long long)gl * LM_HG;
#pragma unroll
            for (int i = 0; i < 6; i++) dst[i] = active ? Hn[i] : 0.0;
#pragma unroll
            for (int i = 0; i < 3; i++) dst[6 + i] = active ? gn[i] : 0.0;
            if (active) gmax_part = fmax(gmax_part, fmax(fabs(gn[0]), fmax(fabs(gn[1]), fabs(gn[2]))));
            if (INIT) {   // the landmark's Jacobi scale is fixed at the first linearisation
                const bool js = active && P.o.jacobi_scaling;
                P.s_lmk[3 * (long long)gl] = js ? 1.0 / (1.0 + sqrt(Hn[0])) : 1.0;
                P.s_lmk[3 * (long long)gl + 1] = js ? 1.0 / (1.0 + sqrt(Hn[3])) : 1.0;
                P.s_lmk[3 * (long long)gl + 2] = js ? 1.0 / (1.0 + sqrt(Hn[5])) : 1.0;
            }
        }
    }
    if (cur_key >= 0) flush();
    SADVIO_TS(3, 48);
    sn = wave_sum(sn); cn = wave_sum(cn); mcc = wave_sum(mcc); cc = wave_sum(cc); fixed_part = wave_sum(fixed_part); gmax_part = wave_max(gmax_part);
    __syncthreads();   // the key-frame sums are in Dacc
    double* rec = P.lm_dt + (long long)cbuf * P.lm_dt_stride + rec_i * LM_DT;
    for (int i = tid; i < T.n_free * LM_DT_RANK; i += blockDim.x) rec[i] = Dacc[i];
    if (tid == 0) {
        if (!INIT) {
            sacc[0] = cc; sacc[1] = mcc; sacc[2] = sn; sacc[3] = cn;
            fixed_part = P.lm_dt[(long long)cur * P.lm_dt_stride + rec_i * LM_DT + LM_DT_COST + 1];   // the constant blocks' cost does not move
        }
        rec[LM_DT_COST] = cc; rec[LM_DT_COST + 1] = fixed_part; rec[LM_DT_COST + 2] = gmax_part;
    }
    SADVIO_TS(3, 49);
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

// ---- K5: the reduced system of a tile from flat observations ---------------------------------------------------------------------
// Per chunk (<= 12 landmarks, <= 64 observations, lane = observation) the wave's strip [row][38] holds, in column
// 3 * landmark + c, Z = sum_a Jp_a^T (Jl_a L) in the rows of the observing key-frames and w in a spare row. One
// v_mfma_f64_16x16x4_f64 pass with A = B = the strip gives strip strip^T: the tile's rows are Z Z^T = E M^-1 E^T (to be
// subtracted from S) and the spare row is Z w = E M^-1 g_l (to be subtracted from the reduced gradient). The two cameras of a
// key-frame (adjacent lanes, the same rows) are summed with DPP before the store; no LDS atomics except one add of the
// accumulators into the tile at the end. The key-frame blocks (sum Jp^T Jp, sum Jp^T r at x) are the tile's lm_dt record.
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
    double* dsum = gT + Nt + (Nt & 1);         // the tile's key-frame sums and cost totals at x (LM_DT doubles)
    double* vt = dsum + LM_DT;                 // view tables of the pixel factor (lm_stage_views)
    stage_tables(P, T, st.cur, poseTab, camTab, rowTab);
    // the tile's key-frame sums at x: the partial records of its sub-blocks (k_lm_pass), summed into LDS
    const double* dt = P.lm_dt + (long long)st.cur * P.lm_dt_stride + (long long)ti * P.lm_ksub * LM_DT;
    const int n_sub = (T.lmk1 - T.lmk0 + LM_PASS_THREADS - 1) / LM_PASS_THREADS;
    for (int i = tid; i < tri_n + Nt; i += blockDim.x) Stile[i] = 0.0;
    for (int i = tid; i < LM_DT; i += blockDim.x) {
        double v = 0.0;
        if (i == LM_DT_COST + 2) { for (int q = 0; q < n_sub; q++) v = fmax(v, dt[q * LM_DT + i]); }
        else for (int q = 0; q < n_sub; q++) v += dt[q * LM_DT + i];
        dsum[i] = v;
    }
    __syncthreads();   // tables, the zeroed tile, the summed record
    if (FACTOR == 0) lm_stage_views(poseTab, camTab, T.n_kf, T.n_cam, vt);
    // the key-frame blocks of the tile: sum Jp^T Jp into the diagonal blocks, sum Jp^T r into the gradient
    for (int i = tid; i < T.n_free * 27; i += blockDim.x) {
        const int q = i / 27, e = i - 27 * q;
        const double v = dsum[q * LM_DT_RANK + e];
        if (e < 21) {
            int a = 0, b = e;
            while (b >= a + 1) { b -= a + 1; a++; }
            Stile[tri(6 * q + a, 6 * q + b)] = v;
        } else gT[6 * q + e - 21] = v;
    }
    if (tid == 0) {   // the linearisation totals k_solve sums (cost at x, the constant blocks' cost, the landmark gradients' max)
        TileAcc* ta = P.tacc + (long long)(slot & 1) * P.n_tiles + ti;
        ta->lin_cost = dsum[LM_DT_COST]; ta->fixed_cost = dsum[LM_DT_COST + 1]; ta->gmax = dsum[LM_DT_COST + 2];
    }
    __syncthreads();
    SADVIO_TS(3, 34);
    const double* xl = P.xl + (long long)st.cur * P.xl_stride;
    double* Zb = strips + wv * strip_doubles;
    typedef lm_d4 d4;
    d4 accZ[3] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}};
    const int lr = ln & 15, lk = ln >> 4;
    const int nt16 = (Nt + 15) >> 4;
    const int rows_used = 16 * nt16;          // the strip rows this tile touches; Nt <= rows_used - 2
    const int w_row = rows_used - 1;          // spare row: w = L^-1 g_l
    // level 1 = the observation's fields, level 2 = its landmark's position and H_ll | g_l | Jacobi scale record (lm_hg, left by
    // k_lm_pass). Both levels of chunk i + 1 are requested while chunk i is computed: a global load is ~2 us on a busy chip
    // (in-kernel stamps: 4 - 5 of the 8.8 us a chunk took were the two dependent levels), three waves per SIMD do not hide that.
    struct In1 { int ob0, ob1, lm0, nlm, sl, cam, ls; double m[3]; };
    struct In2 { double pw[3], H[6], g[3], s[3]; int lcode; };
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
    const double* hgx = P.lm_hg + (long long)st.cur * P.lm_hg_stride;
    auto load2 = [&](const In1& a, In2& b) {
        b.lcode = 1;
#pragma unroll
        for (int i = 0; i < 3; i++) { b.pw[i] = 0.0; b.g[i] = 0.0; b.s[i] = 1.0; }
#pragma unroll
        for (int i = 0; i < 6; i++) b.H[i] = 0.0;
        if (a.ob0 + ln < a.ob1) {
            const long long gl = a.lm0 + a.ls;
            b.lcode = P.lmk_const ? P.lmk_const[gl] : 0;
#pragma unroll
            for (int i = 0; i < 3; i++) { b.pw[i] = P.lmk_p[3 * gl + i] + xl[3 * gl + i]; b.s[i] = P.s_lmk[3 * gl + i]; }
            const double* R = hgx + gl * LM_HG;
#pragma unroll
            for (int i = 0; i < 6; i++) b.H[i] = R[i];
#pragma unroll
            for (int i = 0; i < 3; i++) b.g[i] = R[6 + i];
        }
    };
    In1 A1, N1;
    In2 A2, N2;
    if (T.chunk0 + wv < T.chunk1) { load1(T.chunk0 + wv, N1); load2(N1, N2); }
    for (int ch = T.chunk0 + wv; ch < T.chunk1; ch += BUILD_WAVES) {
        A1 = N1; A2 = N2;
        const bool more = ch + BUILD_WAVES < T.chunk1;
        if (more) load1(ch + BUILD_WAVES, N1);
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
        // the landmark's damping: Li = L^-1 of the damped block M = L L^T and w = Li g_l: with W = Jl Li^T the landmark's Schur term
        // is E M^-1 E^T = Z Z^T, Z = sum_a Jp_a^T W_a, and its part of the reduced gradient E M^-1 g_l = Z w. Every observation
        // lane of the landmark repeats it (the lanes would idle otherwise); no per-landmark record travels through HBM
        double Lw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (lfree) {
            lm_damped_inverse(P, A2.H, A2.s, st.radius, Lw);
            li_vec(Lw, A2.g, Lw + 6);
        }
        if (lfree) {
            row = rowTab[A1.sl];
            if (row >= 0) {   // a constant key-frame has no rows in the reduced system
                double r[2], Jp[12], Jl[6];
                lm_linearize<FACTOR, true>(poseTab, camTab, vt, T.n_cam, A1.sl, A1.cam - T.cam_base, A1.m, A2.pw, r, Jp, Jl);
                double W[6];   // W = Jl Li^T (2 x 3)
                li_row(Lw, Jl[0], Jl[1], Jl[2], W);
                li_row(Lw, Jl[3], Jl[4], Jl[5], W + 3);
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
            d[0] = Lw[6]; d[1] = Lw[7]; d[2] = Lw[8];
        }
        wave_lds_fence();
        SADVIO_TS(3, 36);
        if (more) load2(N1, N2);   // level 2 of the next chunk: in flight during the MFMA pass
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
    for (int i = tid; i < Nt; i += blockDim.x) {
        if (gT[i] != 0.0) atomic_add_f64(&P.gred[T.red_off + growTab[i]], gT[i]);
        // the un-reduced pose gradient and the diagonal of H_pp (Jacobi scale, LM diagonal) only see the key-frame blocks
        const int q = i / 6, a = i - 6 * q;
        const double gf = dsum[q * LM_DT_RANK + 21 + a], hd = dsum[q * LM_DT_RANK + a * (a + 1) / 2 + a];
        if (gf != 0.0) atomic_add_f64(&P.gfull[T.red_off + growTab[i]], gf);
        if (hd != 0.0) atomic_add_f64(&P.hdiag[T.red_off + growTab[i]], hd);
    }
    SADVIO_TS(3, 42);
}

}  // namespace sadvio
