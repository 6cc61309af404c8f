// ba_capi.hip — host driver + C ABI (include/sadvio_ba.h) of the MI355X BA backend.
//
// Replaces, behind the reference's optimizer boundary, the body of AOptimizer::localMapBA /
// localMapVIOptimization from `ceres::Solve` on (AOptimizer.cpp:326,388): the windows are uploaded
// once (set_windows), the whole LM solve is enqueued on the handle's stream without host round
// trips, and deltas are read back for the adapter to apply (AOptimizer.cpp:329-340).
// There is no CPU fallback: without a gfx950 device every compute call fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sadvio_ba.h"
#include "kernels.h"
#include "dense_chol.h"
#include "marg_kernels.h"
#include "lm_kernels.h"
#include "viinit_kernels.h"

using namespace sadvio;

namespace {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            h->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return SADVIO_E_HIP;                                                              \
        }                                                                                     \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        if (count <= n && p && !view) return hipSuccess;
        if (p && !view) (void)hipFree(p);
        view = false;
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    bool view = false;  // non-owning window into another buffer
    void set_view(T* ptr, size_t count) { if (p && !view) (void)hipFree(p); p = ptr; n = count; view = true; }
    void release() { if (p && !view) (void)hipFree(p); p = nullptr; n = 0; view = false; }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(view, o.view); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
};

struct KernelClass {
    const char* name;
    double total_ms = 0;
    long long launches = 0;
};

struct HostWin {
    WinDev d;
    std::vector<int64_t> kf_id, lmk_id;
    int hb_lmk = 0;  // max distance (in free key-frame index) between two key-frames observing one landmark
};

}  // namespace

// RCCL (all-reduce of the reduced system over xGMI), loaded on first use: single-GPU solves never touch it.
struct NcclId { char internal[128]; };
struct RcclLib {
    void* lib = nullptr;
    void* comm = nullptr;
    int (*get_id)(NcclId*) = nullptr;
    int (*init_rank)(void**, int, NcclId, int) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*destroy)(void*) = nullptr;
    const char* (*err_string)(int) = nullptr;
    int (*comm_count)(void*, int*) = nullptr;
    int (*comm_user_rank)(void*, int*) = nullptr;
    int (*comm_cu_device)(void*, int*) = nullptr;
};

// Deep copy of a caller's window (set_windows) + the observation arrays actually tiled: pose-to-landmark NFR factors
// whose landmark can be eliminated are appended to the landmark's observation list as two pseudo-observations
// (rows 0-1 and row 2 of the 3-row factor), so that they ride the ordinary Schur elimination.
struct SrcWin {
    sadvio_flat_window v{};   // view into the vectors below
    std::vector<int64_t> kf_id, lmk_id;
    std::vector<double> kf_T, kf_vel, kf_ba, kf_bg, cam_K, cam_T, cam_sigma, lmk_p, obs_meas;
    std::vector<uint8_t> kf_const, lmk_const;
    std::vector<int32_t> lmk_obs_ptr, obs_kf, obs_cam;
    // augmented observation list (what build_layout tiles) and its map to the caller's observation index (-1 = pseudo)
    std::vector<int32_t> a_ptr, a_kf, a_cam, a_src;
    std::vector<double> a_meas;
    // cameras with identical (K, T_s_f, sigma) are stored once: SaDVIO has one ImageSensor object per (frame, camera),
    // i.e. 2 x N_kf table entries that are all copies of the rig's two cameras
    std::vector<double> u_cam_K, u_cam_T, u_cam_sigma;
    std::vector<int32_t> u_obs_cam;
    std::vector<int> cam_map;   // caller's camera index -> stored camera index
};

// Host -> device uploads of one layout build are packed into ONE pinned staging buffer, copied with one
// hipMemcpyAsync and scattered to their destinations by one kernel: ~25 small pageable copies cost ~15 us each.
struct UploadItem { unsigned long long dst; unsigned long long off; unsigned long long bytes; };
__global__ void k_scatter_uploads(const char* stage, const UploadItem* items, int n_items) {
    for (int it = blockIdx.y; it < n_items; it += gridDim.y) {
        const UploadItem u = items[it];
        char* dst = (char*)u.dst;
        const char* src = stage + u.off;
        const unsigned long long words = u.bytes >> 3;  // staging offsets and device allocations are 8-byte aligned
        for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * blockDim.x)
            ((unsigned long long*)dst)[i] = ((const unsigned long long*)src)[i];
        if (blockIdx.x == 0 && threadIdx.x < (u.bytes & 7)) dst[(words << 3) + threadIdx.x] = src[(words << 3) + threadIdx.x];
    }
}
struct UploadBatch {
    std::vector<UploadItem> items;
    char* pinned = nullptr; size_t pinned_cap = 0, used = 0;   // sources are packed straight into pinned memory
    char* dev = nullptr; size_t dev_cap = 0;
    bool failed = false;
    // flush() does not wait: the staged copy + scatter are stream work like the kernels that read their output. The pinned buffer is
    // only touched again (next add / grow) after the event recorded behind the copy has fired — by then it normally has
    hipEvent_t ev = nullptr;
    bool in_flight = false;
    void wait() { if (in_flight) { (void)hipEventSynchronize(ev); in_flight = false; } }
    void reset() { wait(); items.clear(); used = 0; failed = false; }   // drop what an earlier, failed layout build left queued
    void reserve(size_t bytes) {
        if (bytes <= pinned_cap) return;
        wait();
        char* np = nullptr;
        const size_t cap = bytes + bytes / 2 + 4096;
        if (hipHostMalloc((void**)&np, cap, hipHostMallocDefault) != hipSuccess) { failed = true; return; }
        if (pinned) { memcpy(np, pinned, used); (void)hipHostFree(pinned); }
        pinned = np; pinned_cap = cap;
    }
    void add(void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        wait();
        const size_t off = (used + 7) & ~(size_t)7;
        reserve(off + bytes);
        if (failed) return;
        memcpy(pinned + off, src, bytes);
        used = off + bytes;
        items.push_back({(unsigned long long)dst, (unsigned long long)off, (unsigned long long)bytes});
    }
    hipError_t flush(hipStream_t stream) {
        if (failed) { failed = false; items.clear(); used = 0; return hipErrorOutOfMemory; }
        if (items.empty()) return hipSuccess;
        const size_t data_bytes = (used + 7) & ~(size_t)7;
        const size_t total = data_bytes + items.size() * sizeof(UploadItem);
        reserve(total);
        if (failed) { failed = false; items.clear(); used = 0; return hipErrorOutOfMemory; }
        hipError_t e = hipSuccess;
        // every error return drops the queue: callers that do not reset() afterwards (marginalize, sparsify) must not re-send it
        auto drop = [&](hipError_t err) { items.clear(); used = 0; return err; };
        if (!ev && (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return drop(e);
        if (dev_cap < total) {
            if (dev) { (void)hipStreamSynchronize(stream); (void)hipFree(dev); }   // an earlier scatter may still read it
            dev = nullptr; dev_cap = 0;
            if ((e = hipMalloc((void**)&dev, total + total / 2)) != hipSuccess) return drop(e);
            dev_cap = total + total / 2;
        }
        memcpy(pinned + data_bytes, items.data(), items.size() * sizeof(UploadItem));
        if ((e = hipMemcpyAsync(dev, pinned, total, hipMemcpyHostToDevice, stream)) != hipSuccess) return drop(e);
        hipLaunchKernelGGL(k_scatter_uploads, dim3(64, (unsigned)std::min<size_t>(items.size(), 64)), dim3(256), 0, stream, dev, (const UploadItem*)(dev + data_bytes), (int)items.size());
        e = hipEventRecord(ev, stream);
        in_flight = e == hipSuccess;
        if (!in_flight) e = hipStreamSynchronize(stream);
        items.clear(); used = 0;
        return e;
    }
    ~UploadBatch() { wait(); if (ev) (void)hipEventDestroy(ev); if (pinned) (void)hipHostFree(pinned); if (dev) (void)hipFree(dev); }
};

struct DensePriorHost {
    int n_full = 0, n = 0, kf_keep = -1, kf_col = 0;
    bool resident = false;      // J, r0 = the handle's prior (PriorState), copied device to device
    unsigned long long serial = 0;   // ... as it was when set_dense_prior named it
    std::vector<double> J, r0;
    std::vector<int> lmk_index, lmk_col;
};

// The handle's marginalisation prior: what the reference keeps in `_marginalization_last` inside the optimizer between
// marginalize() and the next window solve / the next marginalize() (AOptimizer.h:88-90, …Analytic.cpp:627-660). Device
// resident; the variables its columns refer to are named by the caller per window (it owns the id bookkeeping).
struct PriorState {
    bool valid = false;
    int n_full = 0, n = 0, form = 0, cut_mode = 0;   // cut_mode: the SADVIO_EIG_CUT_* it was built with (sparsify applies the same cut)
    DevBuf<double> J, r0;        // n_full x n row-major packed, n_full
    DevBuf<double> Z;            // n_full x n with Z^T Z = Sigma_k (built by the first sparsify of this prior)
    bool z_valid = false;
    DevBuf<int> step_of;         // Cholesky form: pivot step of every column
    DevBuf<double> H, g;         // J^T J = Ak and -J^T r0 = bk as the marginalisation that built the prior had them (full-rank Cholesky form only):
    bool hg_valid = false;       // the next marginalize / the next window's dense prior take them instead of re-forming J^T J (n^3 flops)
    unsigned long long serial = 0;   // bumped whenever the prior changes: a window that attached it checks it is still the same one
};

// Work buffers of marginalize / sparsify, kept in the handle: both run once per key-frame, and ~20 hipMalloc / hipFree pairs
// per call cost more than the kernels between them.
struct TriLevel { int first, count, max_m, max_n; };
struct MargScratch {
    DevBuf<double> A, b, G, V, ev, Vs, Ainv, T, Ak, bk, newJ, newr, lastJ, lastr, L, Tb, Zt, S, mi, jsel, lam, wtmp, Hl;
    DevBuf<int> ditems, flag, sel, lastcol, piv_of, lc, piv_mm;
    DevBuf<MargSmall> small;
    DevBuf<NfrSpecC> spec;
    std::vector<int> lcol, items, items_l, col;
    std::vector<double> hev, hS;
    struct TriPlan { DevBuf<TriNode> nodes; std::vector<TriLevel> levels; int leaves = 0; };
    std::map<int, TriPlan> tri_plans;   // node tables of the triangular inverse, by padded size (m of Amm and n of the prior alternate)
};

struct LineSetHost {   // deep copy of a sadvio_line_set
    std::vector<int64_t> id;
    std::vector<double> T, model, meas;
    std::vector<unsigned char> is_const;
    std::vector<int> ptr, obs_kf, obs_cam;
    int n() const { return (int)id.size(); }
};

// Host-side work arrays of build_layout, kept between calls: a sliding-window back end calls set_windows once per key-frame,
// and ~2 MB of fresh std::vectors per call are ~500 page faults (more than the layout arithmetic itself).
struct LayoutScratch {
    std::vector<double> kf_T0, kf_vel, kf_ba, kf_bg, cam_K, cam_T, cam_isig, lmk_p, obs_meas;
    std::vector<int> kf_fidx, lmk_ob, lmk_oe, obs_kf, obs_cam, tile_kf, tile_row, pkf, pcam, run_max, idx, mark, add, kfs, slot_of, chunk_ob, chunk_lm, perm;
    std::vector<unsigned char> lmk_const, obs_slot, obs_lslot;
    std::vector<char> held;
};

// The diagnostic switches of DESIGN.md 4 (environment), read ONCE when the handle is created: none is needed in production, and none is
// looked up again on a per-key-frame path (round 4 called getenv 39 times across set_windows / marginalize / solve).
struct EnvCfg {
    int debug = 0;
    int lm = -1, pf_wg = -1;            // -1: not set
    int tile_rounds = 0, lm_subs = 0, band_c = 0;   // 0: not set
    double jacobi_tol = 1e-14;
    bool marg_last_small = false, marg_eig_mm = false, marg_pivoted = false, marg_unpivoted = false, pchol_swap = false, pchol_strict = false,
         jacobi_b4 = false, jacobi_plain = false, no_lpt = false, no_pre = false, no_fork = false, no_bcr = false, wd_old = false, wd_nola = false, wd_r3 = false,
         wd_back1 = false, imu_items = false;
    void read() {
        auto on = [](const char* k) { return getenv(k) != nullptr; };
        auto num = [](const char* k, int unset) { const char* e = getenv(k); return e ? atoi(e) : unset; };
        debug = num("SADVIO_DEBUG", 0); lm = num("SADVIO_LM", -1); pf_wg = num("SADVIO_PF_WG", -1);
        tile_rounds = num("SADVIO_TILE_ROUNDS", 0); lm_subs = num("SADVIO_LM_SUBS", 0); band_c = num("SADVIO_BAND_C", 0);
        if (const char* e = getenv("SADVIO_JACOBI_TOL")) jacobi_tol = atof(e);
        marg_last_small = on("SADVIO_MARG_LAST_SMALL"); marg_eig_mm = on("SADVIO_MARG_EIG_MM"); marg_pivoted = on("SADVIO_MARG_PIVOTED");
        marg_unpivoted = on("SADVIO_MARG_UNPIVOTED"); pchol_swap = on("SADVIO_PCHOL_SWAP"); pchol_strict = on("SADVIO_PCHOL_STRICT");
        jacobi_b4 = on("SADVIO_JACOBI_B4"); jacobi_plain = on("SADVIO_JACOBI_PLAIN"); no_lpt = on("SADVIO_NO_LPT"); no_pre = on("SADVIO_NO_PRE"); no_fork = on("SADVIO_NO_FORK");
        no_bcr = on("SADVIO_NO_BCR"); wd_old = on("SADVIO_WD_OLD"); wd_nola = on("SADVIO_WD_NOLA"); wd_r3 = on("SADVIO_WD_R3"); wd_back1 = on("SADVIO_WD_BACK1");
        imu_items = on("SADVIO_IMU_ITEMS");   // A/B: the IMU pairs' entries through k_solve's item loop (the pre-0.5 path) on one device too
    }
};

struct sadvio_ba_handle {
    EnvCfg env;
    sadvio_ba_config cfg{};
    LayoutScratch ls;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // host mirrors
    std::vector<HostWin> wins;
    std::vector<Tile> tiles;
    std::vector<PriorDev> priors;
    std::vector<int> obs_perm;  // device observation position -> caller's observation index (within window)
    std::vector<std::vector<PriorDev>> priors_per_win;
    std::vector<std::vector<ImuDev>> imus_per_win;
    std::vector<ImuDev> imus;
    int n_kf_tot = 0, n_cam_tot = 0, n_lmk_tot = 0, n_obs_tot = 0, np_tot = 0;
    long long s_tot = 0;
    int factor_type = 0;
    int max_n_kf = 0, max_npose = 0, max_np = 0, n_big = 0;
    DevBuf<int> d_big_info;
    DevBuf<double> d_big_M;     // inverse diagonal blocks of the wide-panel dense solver, 96 x 96 per 96 columns (+ the factors' tiles)
    DevBuf<double> d_big_Lx;    // its out-of-place panels
    DevBuf<double> d_big_linv;  // inverse pivot blocks of the banded solver, N * NB doubles per out-of-LDS window
    DevBuf<double> d_coll_band; // band-packed copy of the reduced system for the sharded all-reduce
    DevBuf<double> d_bcr;       // block-cyclic-reduction workspace of the long banded systems
    DevBuf<double> d_big_mid;   // the two Schur complements on the middle block of the twisted banded factorisation
    bool uploaded = false, solved = false;
    bool defer = false, pending = false;   // begin_update .. commit_update: set_* calls only record, ONE layout build + staged upload at commit
    UploadBatch up;   // pending host -> device uploads of the current layout build
    // window sharded over several GPUs: collective hook (user callback or the built-in RCCL one)
    int world = 1, rank = 0;
    sadvio_allreduce_fn coll_fn = nullptr;
    void* coll_ctx = nullptr;
    RcclLib rccl;
    long long red_total = 0;  // doubles in [S | gred | gfull | hdiag | rank_b], the buffer of the per-step all-reduce
    DevBuf<double> d_rank_b, d_rank_s;
    // dense marginalisation priors (host copies, one per window) and the layout they induce
    std::vector<DensePriorHost> dprior_per_win;
    PriorState prior;   // the handle's own prior (sadvio_ba_marginalize leaves it here)
    MargScratch mg;
    std::vector<SrcWin> src;                       // caller windows (deep copies)
    std::vector<std::vector<char>> sp_elim;        // per window, per sparse factor: handled as pseudo-observations
    std::vector<int> n_obs_user;                   // caller's observation count per window
    std::vector<std::vector<sadvio_sparse_prior>> sparse_per_win;
    std::vector<LineSetHost> lines_per_win;        // linexd landmarks
    DevBuf<LineDev> d_lines;
    DevBuf<LineObsDev> d_lobs;
    DevBuf<double> d_xline, d_line_scratch;
    int n_line_tot = 0, n_lobs_tot = 0;
    DevBuf<SparseDev> d_sparse;
    DevBuf<int> d_sp_list;
    int n_sp_list = 0;   // sparse prior factors evaluated by sparse_factor_eval (all windows)
    size_t n_sparse_tot = 0;
    DevBuf<double> d_sp_scratch;
    std::vector<unsigned char> h_lmk_const_user;  // as given by the caller
    std::vector<int> h_lmk_ob, h_lmk_oe, h_kf_fidx, h_obs_kf;
    bool user_lmk_const = false;
    int n_kept = 0;
    DevBuf<int> d_lmk_red, d_kept_obs, d_dp_ints;
    DevBuf<double> d_dp_data;
    int slots_cap = 0;
    int last_slots = 0;
    // device buffers
    DevBuf<WinDev> d_win;
    DevBuf<Tile> d_tiles;
    DevBuf<double> d_kf_T0, d_xp, d_xv, d_xba, d_xbg, d_kf_vel, d_kf_ba, d_kf_bg;
    DevBuf<int> d_kf_fidx;
    DevBuf<double> d_cam_K, d_cam_T, d_cam_isig;
    DevBuf<double> d_lmk_p, d_xl, d_s_lmk;
    DevBuf<unsigned char> d_lmk_const;
    DevBuf<int> d_lmk_ob, d_lmk_oe, d_obs_kf, d_obs_cam, d_tile_kf, d_tile_row;
    DevBuf<int> d_pre_lane, d_pre_kf;     // first-round packets of the latency kernels (kernels.h: DevPtrs::pre_lane), few-tile submissions only
    bool pre_ok = false, pre_dirty = false;
    DevBuf<int> d_rank_col; DevBuf<double> d_rank_x;       // refine_rank_by_eigenvalue (guarded calls only): pivot column per step, the solved vectors
    DevBuf<unsigned char> d_obs_slot, d_obs_lslot;
    DevBuf<int> d_chunk_ob, d_chunk_lm, d_tile_perm;   // chunk tables of the throughput kernels (lm_kernels.h)
    DevBuf<int> d_jac_ints;               // pivoting / rank of the Cholesky-preconditioned Jacobi
    DevBuf<double> d_jac_dbl;             // its remaining diagonal + threshold
    DevBuf<double> d_lm_hg, d_lm_dt, d_lm_sacc;
    DevBuf<int> d_lm_sub;                 // work list of k_lm_pass (tile, sub-block), see DevPtrs
    int lm_n_sub = 0, lm_ksub = 1, lm_sub_per_item = 8;   // sub-blocks per work item of k_lm_pass (8 = the whole tile: MAX tile = 512 landmarks)   // throughput path: elimination records, per-landmark H_ll | g_l and per-tile key-frame sums (both per delta buffer)
    int lm_max_cam = 1;
    int hidden_eig_count = 0;             // sparsify: priors of full rank by their pivots whose inverse showed an eigenvalue that may lie below the cut (SADVIO_DEBUG prints it)
    int marg_stats[4] = {0, 0, 0, 0};     // Cholesky-form marginalisations: calls | took the unpivoted route | tried it and fell back | calls whose rank the eigenvalue refinement lowered
    int lm_sub_obs = 0;                   // most observations of LM_PASS_THREADS consecutive landmarks of a tile (LDS staging of k_lm_pass)
    bool gemm_run4 = false;               // a tile on the MFMA path holds runs of 3 - 4 observations on one key-frame (k_build<.., RARE = true> only)
    bool lm_ok = false;                   // every tile is on the MFMA path and chunked: k_build_obs / k_lm_pass may run
    long long lm_landmarks = 0;
    DevBuf<double> d_ptab;
    int max_tile_kf = 1, max_tile_free = 0, max_gemm_free = 0;
    DevBuf<double> d_obs_meas;
    DevBuf<PriorDev> d_priors;
    DevBuf<double> d_prior_lin;   // [2][n_prior][PRIOR_LIN], see DevPtrs::prior_lin
    DevBuf<ImuDev> d_imus;
    DevBuf<double> d_imu_scratch;
    DevBuf<double> d_S, d_gred, d_gfull, d_hdiag, d_delta, d_s_pose;
    DevBuf<LmState> d_states;
    DevBuf<double> d_trace;
    DevBuf<long long> d_tstart;
    DevBuf<IterAcc> d_acc;
    DevBuf<FinalRec> d_final;
    FinalRec* h_final = nullptr;  // pinned
    double* h_deltas = nullptr; size_t h_deltas_cap = 0; bool deltas_cached = false;   // pinned copy of BOTH delta buffers, fetched by the first get_deltas after a solve
    size_t h_final_n = 0;
    std::vector<FinalRec> fin;    // last solve's records
    DevBuf<TileAcc> d_tacc;
    DevBuf<long long> d_dbg;
    // hipGraph of one complete solve (all slots), re-captured whenever the launch parameters change
    hipGraphExec_t graph_exec = nullptr;
    std::vector<unsigned char> graph_key;
    DevBuf<double> d_probe;
    bool has_lmk_const = false;
    // profiling
    std::vector<KernelClass> kclasses;
    hipStream_t side = nullptr;            // IMU factor evaluation runs here, concurrently with k_build / k_backsub
    hipEvent_t ev_fork = nullptr, ev_lin = nullptr, ev_solved = nullptr, ev_cost = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<std::pair<int, int>> ev_used;  // (class, pool index)
    size_t ev_next = 0;
};

namespace {

int kclass_id(sadvio_ba_handle* h, const char* name) {
    for (size_t i = 0; i < h->kclasses.size(); i++)
        if (!strcmp(h->kclasses[i].name, name)) return (int)i;
    KernelClass k; k.name = name;
    h->kclasses.push_back(k);
    return (int)h->kclasses.size() - 1;
}

struct ScopedTimer {
    sadvio_ba_handle* h;
    int pool = -1;
    ScopedTimer(sadvio_ba_handle* h_, const char* name) : h(h_) {
        if (!h->cfg.profile_kernels) return;
        if (h->ev_next == h->ev_pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            h->ev_pool.push_back({a, b});
        }
        pool = (int)h->ev_next++;
        h->ev_used.push_back({kclass_id(h, name), pool});
        (void)hipEventRecord(h->ev_pool[pool].first, h->stream);
    }
    ~ScopedTimer() {
        if (pool >= 0) (void)hipEventRecord(h->ev_pool[pool].second, h->stream);
    }
};

void collect_timers(sadvio_ba_handle* h) {
    for (auto& u : h->ev_used) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev_pool[u.second].first, h->ev_pool[u.second].second) == hipSuccess) {
            h->kclasses[u.first].total_ms += ms;
            h->kclasses[u.first].launches += 1;
        }
    }
    h->ev_used.clear();
    h->ev_next = 0;
}

DevPtrs make_ptrs(sadvio_ba_handle* h, const SolveOpts& o, int state_stride) {
    DevPtrs P{};
    P.win = h->d_win.p; P.tiles = h->d_tiles.p;
    P.kf_T0 = h->d_kf_T0.p; P.kf_fidx = h->d_kf_fidx.p;
    P.xp = h->d_xp.p; P.xv = h->d_xv.p; P.xba = h->d_xba.p; P.xbg = h->d_xbg.p;
    P.xp_stride = 6LL * h->n_kf_tot; P.xv_stride = 3LL * h->n_kf_tot; P.xl_stride = 3LL * h->n_lmk_tot;
    P.kf_vel = h->d_kf_vel.p; P.kf_ba = h->d_kf_ba.p; P.kf_bg = h->d_kf_bg.p;
    P.cam_K = h->d_cam_K.p; P.cam_T = h->d_cam_T.p; P.cam_isig = h->d_cam_isig.p;
    P.lmk_p = h->d_lmk_p.p; P.xl = h->d_xl.p; P.s_lmk = h->d_s_lmk.p;
    P.lmk_const = h->has_lmk_const ? h->d_lmk_const.p : nullptr;
    P.lmk_ob = h->d_lmk_ob.p; P.lmk_oe = h->d_lmk_oe.p;
    P.obs_kf = h->d_obs_kf.p; P.obs_cam = h->d_obs_cam.p; P.obs_meas = h->d_obs_meas.p;
    P.obs_slot = h->d_obs_slot.p; P.tile_kf = h->d_tile_kf.p; P.tile_row = h->d_tile_row.p;
    P.pre_lane = h->pre_ok ? (const int4*)h->d_pre_lane.p : nullptr; P.pre_kf = h->pre_ok ? (const int2*)h->d_pre_kf.p : nullptr;
    P.ptab = h->d_ptab.p; P.ptab_stride = (long long)POSE_TAB * h->n_kf_tot;
    P.priors = h->d_priors.p; P.prior_lin = h->d_prior_lin.p; P.prior_lin_stride = (long long)h->priors.size() * PRIOR_LIN; P.n_prior_tot = (int)h->priors.size();
    P.imus = h->d_imus.p; P.imu_scratch = h->d_imu_scratch.p; P.imu_scratch_stride = (long long)h->imus.size() * IMU_ROW;
    P.S = h->d_S.p; P.gred = h->d_gred.p; P.gfull = h->d_gfull.p; P.hdiag = h->d_hdiag.p;
    P.delta = h->d_delta.p; P.s_pose = h->d_s_pose.p;
    P.dbg_ts = h->d_dbg.p;
    P.trace = h->d_trace.p; P.t_start = h->d_tstart.p;
    P.states = h->d_states.p; P.acc = h->d_acc.p; P.tacc = h->d_tacc.p; P.n_tiles = (int)h->tiles.size();
    P.state_stride = state_stride;
    P.final_out = h->h_final ? h->h_final : h->d_final.p;   // the final records go straight to pinned host memory (device-visible): no copy after the last kernel
    P.big_info = h->d_big_info.p;
    P.world = h->world; P.rank = h->rank; P.rank_b = h->d_rank_b.p; P.rank_s = h->d_rank_s.p;
    P.lmk_red = h->d_lmk_red.p; P.kept_obs = h->d_kept_obs.p; P.n_kept = h->n_kept;
    P.dp_data = h->d_dp_data.p; P.dp_ints = h->d_dp_ints.p;
    P.sparse = h->d_sparse.p; P.sp_scratch = h->d_sp_scratch.p; P.sp_list = h->d_sp_list.p;
    P.sp_scratch_stride = (long long)std::max<size_t>(h->n_sparse_tot, 1) * SPARSE_J; P.n_imu_tot = (int)h->imus.size(); P.n_sp_list = h->n_sp_list;
    P.chunk_ob = h->d_chunk_ob.p; P.chunk_lm = h->d_chunk_lm.p; P.tile_perm = h->d_tile_perm.p; P.obs_lslot = h->d_obs_lslot.p;
    P.lm_hg = h->d_lm_hg.p; P.lm_hg_stride = (long long)LM_HG * std::max(h->n_lmk_tot, 1);
    P.lm_dt = h->d_lm_dt.p; P.lm_dt_stride = (long long)LM_DT * std::max<long long>((long long)h->tiles.size(), 1) * h->lm_ksub;
    P.lm_sub = h->d_lm_sub.p; P.lm_ksub = h->lm_ksub; P.lm_sub_per_item = h->lm_sub_per_item; P.lm_sacc = h->lm_ok ? h->d_lm_sacc.p : nullptr;
    P.lines = h->d_lines.p; P.lobs = h->d_lobs.p; P.xline = h->d_xline.p; P.line_scratch = h->d_line_scratch.p;
    P.xline_stride = 6LL * h->n_line_tot;
    P.n_xp = (long long)h->d_xp.n; P.n_xv = (long long)h->d_xv.n; P.n_xl = (long long)h->d_xl.n;
    P.n_win = (int)h->wins.size();
    P.debug = h->env.debug;
    P.o = o;
    return P;
}

// J^T of a dense prior, once per upload (J is constant during the solve); H = J^T J is a k_mgemm launch (FP64 matrix cores:
// the one-thread-per-entry loop this replaces took 0.92 ms at n = 915, more than the layout build itself).
__global__ void k_dense_prior_prepare(const double* J, double* Jt, int nf, int n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long long)nf * n) {
        const int i = (int)(idx / n), a = (int)(idx - (long long)i * n);
        Jt[(size_t)a * nf + i] = J[idx];
    }
}

// H = the symmetric matrix whose lower triangle is A's (the marginalisation's Ak, read like Eigen reads it)
__global__ void k_sym_from_lower(const double* __restrict__ A, int n, double* __restrict__ H) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx - (long long)i * n);
    H[idx] = i >= j ? A[idx] : A[(size_t)j * n + i];
}

// Layout of the reduced systems of all windows: [free key-frames (dpf each) | prior-kept landmarks (3 each)].
// Called by set_windows and again by set_dense_prior (kept landmarks enlarge the reduced system).
int layout_reduced(sadvio_ba_handle* h) {
    const int n_windows = (int)h->wins.size();
    // a resident prior that changed after it was attached is refused BEFORE anything is queued in h->up: a non-fatal return with
    // items in the batch would scatter them, stale, with the next successful flush (the device buffers are grow-only)
    for (int w = 0; w < n_windows && w < (int)h->dprior_per_win.size(); w++) {
        const DensePriorHost& D = h->dprior_per_win[w];
        if (D.n_full > 0 && D.resident && (!h->prior.valid || h->prior.serial != D.serial || h->prior.n_full != D.n_full || h->prior.n != D.n)) {
            h->up.reset();
            h->err = "the handle's prior changed after set_dense_prior(SADVIO_PRIOR_RESIDENT) attached it to a window: attach it again"; return SADVIO_E_STATE;
        }
    }
    int red_b = 0; long long s_b = 0;
    h->max_np = 0; h->n_big = 0;
    std::vector<int> lmk_red(std::max(h->n_lmk_tot, 1), -1);
    std::vector<unsigned char> lmk_const = h->h_lmk_const_user;
    std::vector<int> kept, dp_ints;
    std::vector<SparseDev> sparse;
    std::vector<int> sp_list;
    std::vector<LineDev> lines;
    std::vector<LineObsDev> lobs;
    long long dp_total = 0;   // doubles of d_dp_data: per window [J | J^T | J^T J | r0 | dx | r | cost slot]
    bool any_red = false;
    struct Prep { long long off; int nf, n, w; };
    std::vector<Prep> preps;
    for (int w = 0; w < n_windows; w++) {
        WinDev& d = h->wins[w].d;
        const DensePriorHost& D = h->dprior_per_win[w];
        int n_red = 0;
        d.dp_n_full = d.dp_n = 0; d.dp_off = 0; d.dp_int_off = 0;
        d.kept_begin = (int)kept.size() / 3;
        if (D.n_full > 0) {
            const int n = D.n, nf = D.n_full;
            std::vector<int> kind(n, -1), index(n, 0), col(n, -1);
            if (D.kf_keep >= 0) {
                const int g = d.kf_base + D.kf_keep, fi = h->h_kf_fidx[g];
                for (int q = 0; q < 15; q++) {
                    const int a = D.kf_col + q;
                    if (q < 6) { kind[a] = 0; index[a] = 6 * g + q; }
                    else { kind[a] = 1 + (q - 6) / 3; index[a] = 3 * g + (q - 6) % 3; }
                    col[a] = (fi >= 0 && q < d.dpf) ? fi * d.dpf + q : -1;
                }
            }
            for (size_t i = 0; i < D.lmk_index.size(); i++) {
                if (D.lmk_col[i] < 0) continue;
                const int gl = d.lmk_base + D.lmk_index[i];
                const bool is_const = lmk_const[gl] == 1;
                if (!is_const) {
                    lmk_red[gl] = d.dpf * d.n_free_kf + 3 * n_red; n_red++;
                    lmk_const[gl] = 2; any_red = true;
                    for (int o = h->h_lmk_ob[gl]; o < h->h_lmk_oe[gl]; o++) { kept.push_back(o); kept.push_back(gl); kept.push_back(w); }
                }
                for (int a = 0; a < 3; a++) {
                    kind[D.lmk_col[i] + a] = 4; index[D.lmk_col[i] + a] = 3 * gl + a;
                    col[D.lmk_col[i] + a] = is_const ? -1 : lmk_red[gl] + a;
                }
            }
            d.dp_n_full = nf; d.dp_n = n;
            d.dp_int_off = (int)dp_ints.size();
            dp_ints.insert(dp_ints.end(), kind.begin(), kind.end());
            dp_ints.insert(dp_ints.end(), index.begin(), index.end());
            dp_ints.insert(dp_ints.end(), col.begin(), col.end());
            d.dp_off = dp_total;
            preps.push_back({d.dp_off, nf, n, w});
            dp_total += (long long)nf * n + (long long)n * nf + (long long)n * n + nf + n + nf + 2 + 3LL * ((nf + 3) / 4);   // J, Jt, H (device-filled), r0, dx, r scratch, cost slot, -, row-block partial sums (sharded windows)
            dp_total += dp_total & 1;
        }
        // landmarks touched by sparse prior factors stay in the reduced system as well
        d.sp_begin = (int)sparse.size();
        d.spl_begin = (int)sp_list.size();
        for (size_t sk = 0; sk < h->sparse_per_win[w].size(); sk++) {
            const sadvio_sparse_prior& s = h->sparse_per_win[w][sk];
            SparseDev o{};
            o.type = s.type; o.win = w;
            const bool elim = w < (int)h->sp_elim.size() && sk < h->sp_elim[w].size() && h->sp_elim[w][sk];
            if (elim) {
                // rides the Schur elimination as two pseudo-observations of its landmark: only its constants are needed
                o.type = 4; o.kf = d.kf_base + s.kf; o.lmk0 = d.lmk_base + s.lmk0; o.lmk1 = -1;
                memcpy(o.delta, s.delta, 24); memcpy(o.W, s.sqrt_inf, sizeof(o.W));
                sparse.push_back(o);
                continue;
            }
            o.kf = s.kf >= 0 ? d.kf_base + s.kf : -1;
            const bool rel = s.type == SADVIO_SPARSE_RELATIVE_POSE;
            if (rel) { o.type = 5; o.kf2 = d.kf_base + s.kf_b; }   // internal type 4 is the pseudo-observation form above
            const int ls[2] = {(s.type == SADVIO_SPARSE_IMU_PRIOR || rel) ? -1 : s.lmk0, s.type == SADVIO_SPARSE_LMK_TO_LMK ? s.lmk1 : -1};
            int gls[2] = {-1, -1};
            for (int q = 0; q < 2; q++) {
                if (ls[q] < 0) continue;
                const int gl = d.lmk_base + ls[q];
                gls[q] = gl;
                if (lmk_const[gl] == 1 || lmk_red[gl] >= 0) continue;
                lmk_red[gl] = d.dpf * d.n_free_kf + 3 * n_red; n_red++;
                lmk_const[gl] = 2; any_red = true;
                for (int ob = h->h_lmk_ob[gl]; ob < h->h_lmk_oe[gl]; ob++) { kept.push_back(ob); kept.push_back(gl); kept.push_back(w); }
            }
            o.lmk0 = gls[0]; o.lmk1 = gls[1];
            memcpy(o.T_prior, s.T_prior, sizeof(o.T_prior)); memcpy(o.v_prior, s.v_prior, 24); memcpy(o.ba_prior, s.ba_prior, 24);
            memcpy(o.bg_prior, s.bg_prior, 24); memcpy(o.delta, s.delta, 24); memcpy(o.W, s.sqrt_inf, sizeof(o.W));
            sp_list.push_back((int)sparse.size());
            sparse.push_back(o);
        }
        d.sp_end = (int)sparse.size();
        d.spl_end = (int)sp_list.size();
        d.kept_end = (int)kept.size() / 3;
        d.n_red = n_red;
        d.Np = d.n_free_kf * d.dpf + 3 * n_red;
        // linexd landmarks: 6 columns each after the kept landmarks
        d.line_begin = (int)lines.size(); d.lobs_begin = (int)lobs.size();
        if (w < (int)h->lines_per_win.size()) {
            const LineSetHost& LS = h->lines_per_win[w];
            for (int l = 0; l < LS.n(); l++) {
                LineDev o{};
                memcpy(o.T, &LS.T[12 * (size_t)l], 96); memcpy(o.model, &LS.model[6 * (size_t)l], 48);
                o.win = w; o.col = -1;
                if (!(LS.is_const.size() && LS.is_const[l])) { o.col = d.Np; d.Np += 6; }
                const int ms = d.factor_type == SADVIO_FACTOR_PIXEL ? 4 : 6;
                for (int ob = LS.ptr[l]; ob < LS.ptr[l + 1]; ob++) {
                    LineObsDev q{};
                    q.line = (int)lines.size(); q.kf = d.kf_base + LS.obs_kf[ob]; q.cam = d.cam_base + h->src[w].cam_map[LS.obs_cam[ob]]; q.win = w;
                    memcpy(q.meas, &LS.meas[(size_t)ms * ob], sizeof(double) * ms);
                    lobs.push_back(q);
                }
                lines.push_back(o);
            }
        }
        d.line_end = (int)lines.size(); d.lobs_end = (int)lobs.size();
        // reduced systems that fit LDS are kept as a packed lower triangle (16-byte aligned); larger ones as a
        // full row-major matrix (lower triangle used) that the library factorisation works on in place
        d.ld = d.Np > MAX_LDS_NP ? d.Np : 0;
        d.S_off = s_b; d.red_off = red_b;
        red_b += d.Np; s_b += d.ld ? (((long long)d.Np * d.Np + 1) & ~1LL) : (long long)c16_size(d.Np);   // LDS-sized systems: the tile-packed image of chol16.h
        if (d.ld) h->n_big++; else h->max_np = std::max(h->max_np, d.Np);
        for (int ti = d.tile_begin; ti < d.tile_end; ti++) {
            Tile& t = h->tiles[ti];
            t.Np = d.Np; t.red_off = d.red_off; t.S_off = d.S_off; t.ld = d.ld;
        }
    }
    h->np_tot = red_b; h->s_tot = s_b;
    h->n_kept = (int)kept.size() / 3;
    h->has_lmk_const = h->user_lmk_const || any_red;
    // one allocation [S | gred | gfull | hdiag | rank_b]: a window sharded over several GPUs all-reduces it whole
    const long long nrb = (long long)n_windows * h->world * 4;
    h->red_total = s_b + 3LL * red_b + nrb;
    HIP_TRY(h->d_S.alloc((size_t)std::max<long long>(h->red_total, 1)));
    h->d_gred.set_view(h->d_S.p + s_b, (size_t)red_b); h->d_gfull.set_view(h->d_gred.p + red_b, (size_t)red_b);
    h->d_hdiag.set_view(h->d_gfull.p + red_b, (size_t)red_b); h->d_rank_b.set_view(h->d_hdiag.p + red_b, (size_t)nrb);
    HIP_TRY(h->d_rank_s.alloc((size_t)nrb));
    HIP_TRY(h->d_delta.alloc((size_t)std::max(red_b, 1))); HIP_TRY(h->d_s_pose.alloc((size_t)std::max(red_b, 1)));
    HIP_TRY(hipMemsetAsync(h->d_S.p, 0, sizeof(double) * (size_t)std::max<long long>(h->red_total, 1), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_rank_s.p, 0, sizeof(double) * (size_t)nrb, h->stream));
    HIP_TRY(h->d_sparse.alloc(std::max<size_t>(sparse.size(), 1))); HIP_TRY(h->d_sp_scratch.alloc(2 * std::max<size_t>(sparse.size(), 1) * SPARSE_J)); h->n_sparse_tot = sparse.size();
    h->up.add(h->d_sparse.p, sparse.data(), sparse.size() * sizeof(SparseDev));
    HIP_TRY(h->d_sp_list.alloc(std::max<size_t>(sp_list.size(), 1)));
    h->n_sp_list = (int)sp_list.size();
    h->up.add(h->d_sp_list.p, sp_list.data(), sp_list.size() * sizeof(int));
    h->n_line_tot = (int)lines.size(); h->n_lobs_tot = (int)lobs.size();
    HIP_TRY(h->d_lines.alloc(std::max<size_t>(lines.size(), 1))); HIP_TRY(h->d_lobs.alloc(std::max<size_t>(lobs.size(), 1)));
    HIP_TRY(h->d_xline.alloc(std::max<size_t>(12 * lines.size(), 1))); HIP_TRY(h->d_line_scratch.alloc(std::max<size_t>(lobs.size(), 1) * LINE_ROW));
    h->up.add(h->d_lines.p, lines.data(), lines.size() * sizeof(LineDev));
    h->up.add(h->d_lobs.p, lobs.data(), lobs.size() * sizeof(LineObsDev));
    if (kept.empty()) kept.assign(3, 0);
    if (dp_ints.empty()) dp_ints.push_back(0);
    HIP_TRY(h->d_tiles.alloc(h->tiles.size())); HIP_TRY(h->d_lmk_red.alloc(lmk_red.size())); HIP_TRY(h->d_lmk_const.alloc(lmk_const.size()));
    HIP_TRY(h->d_kept_obs.alloc(kept.size())); HIP_TRY(h->d_dp_ints.alloc(dp_ints.size())); HIP_TRY(h->d_dp_data.alloc((size_t)std::max<long long>(dp_total, 1)));
#define UP(dst, src) h->up.add((dst).p, (src).data(), (src).size() * sizeof((src)[0]))
    UP(h->d_tiles, h->tiles); UP(h->d_lmk_red, lmk_red); UP(h->d_lmk_const, lmk_const); UP(h->d_kept_obs, kept);
    UP(h->d_dp_ints, dp_ints);
#undef UP
    // the dense priors' data never exists as one host array: scratch parts are zeroed on the device, J and r0 come from the
    // caller's copy (staged upload) or from the handle's prior (device to device, no PCIe traffic)
    if (!preps.empty()) HIP_TRY(hipMemsetAsync(h->d_dp_data.p, 0, sizeof(double) * (size_t)dp_total, h->stream));
    for (const Prep& pr : preps) {
        const DensePriorHost& D = h->dprior_per_win[pr.w];
        double* J = h->d_dp_data.p + pr.off;
        double* r0 = J + 2 * (size_t)pr.nf * pr.n + (size_t)pr.n * pr.n;
        if (D.resident) {
            if (!h->prior.valid || h->prior.serial != D.serial || h->prior.n_full != pr.nf || h->prior.n != pr.n) {
                h->err = "the handle's prior changed after set_dense_prior(SADVIO_PRIOR_RESIDENT) attached it to a window: attach it again"; return SADVIO_E_STATE;
            }
            HIP_TRY(hipMemcpyAsync(J, h->prior.J.p, sizeof(double) * (size_t)pr.nf * pr.n, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(r0, h->prior.r0.p, sizeof(double) * (size_t)pr.nf, hipMemcpyDeviceToDevice, h->stream));
        } else {
            h->up.add(J, D.J.data(), sizeof(double) * (size_t)pr.nf * pr.n);
            h->up.add(r0, D.r0.data(), sizeof(double) * (size_t)pr.nf);
        }
    }
    if (!preps.empty()) HIP_TRY(h->up.flush(h->stream));  // the prepare kernels read J on the device
    for (const Prep& pr : preps) {
        double* J = h->d_dp_data.p + pr.off;
        double* Jt = J + (size_t)pr.nf * pr.n;
        double* H = Jt + (size_t)pr.n * pr.nf;
        const long long items = (long long)pr.nf * pr.n;
        hipLaunchKernelGGL(k_dense_prior_prepare, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, h->stream, J, Jt, pr.nf, pr.n);
        if (h->dprior_per_win[pr.w].resident && h->prior.hg_valid && h->prior.n == pr.n && !h->env.marg_last_small)
            hipLaunchKernelGGL(k_sym_from_lower, dim3((unsigned)(((long long)pr.n * pr.n + 255) / 256)), dim3(256), 0, h->stream, h->prior.H.p, pr.n, H);   // H = Ak of the marginalisation
        else
        hipLaunchKernelGGL(k_mgemm, dim3((pr.n + 63) / 64, (pr.n + 63) / 64), dim3(256), 0, h->stream, H, (long long)pr.n, J, 1LL, (long long)pr.n, J, (long long)pr.n, 1LL,
                           pr.n, pr.n, pr.nf, 1.0, 0.0);
    }
    return SADVIO_OK;
}

int upload_priors(sadvio_ba_handle* h) {
    h->priors.clear();
    for (size_t w = 0; w < h->wins.size(); w++) {
        h->wins[w].d.prior_begin = (int)h->priors.size();
        for (auto& p : h->priors_per_win[w]) h->priors.push_back(p);
        h->wins[w].d.prior_end = (int)h->priors.size();
    }
    HIP_TRY(h->d_priors.alloc(h->priors.size()));
    HIP_TRY(h->d_prior_lin.alloc(2 * h->priors.size() * (size_t)PRIOR_LIN));
    if (!h->priors.empty())
        h->up.add(h->d_priors.p, h->priors.data(), h->priors.size() * sizeof(PriorDev));
    h->imus.clear();
    for (size_t w = 0; w < h->wins.size(); w++) {
        h->wins[w].d.imu_begin = (int)h->imus.size();
        for (auto& f : h->imus_per_win[w]) { h->imus.push_back(f); h->imus.back().win = (int)w; }
        h->wins[w].d.imu_end = (int)h->imus.size();
    }
    HIP_TRY(h->d_imus.alloc(h->imus.size()));
    HIP_TRY(h->d_imu_scratch.alloc(2 * h->imus.size() * (size_t)IMU_ROW));
    if (!h->imus.empty())
        h->up.add(h->d_imus.p, h->imus.data(), h->imus.size() * sizeof(ImuDev));
    std::vector<WinDev> wd(h->wins.size());
    for (size_t w = 0; w < h->wins.size(); w++) wd[w] = h->wins[w].d;
    h->up.add(h->d_win.p, wd.data(), wd.size() * sizeof(WinDev));
    HIP_TRY(h->up.flush(h->stream));  // one staged copy + scatter for everything queued since the layout build began
    if (h->pre_ok && h->pre_dirty) {
        hipLaunchKernelGGL(k_pre_packets, dim3((unsigned)h->tiles.size()), dim3(BUILD_THREADS), 0, h->stream, h->d_tiles.p, h->d_lmk_ob.p, h->d_lmk_oe.p, h->d_tile_kf.p,
                           h->d_kf_fidx.p, (int4*)h->d_pre_lane.p, (int2*)h->d_pre_kf.p);
        h->pre_dirty = false;
    }
    return SADVIO_OK;
}

}  // namespace

extern "C" {

int sadvio_ba_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strstr(p.gcnArchName, "gfx950")) ok++;
    }
    return ok;
}

void sadvio_ba_default_options(sadvio_solve_options* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->max_num_iterations = 20;   // AOptimizer.cpp:319
    o->function_tolerance = 1e-3; // AOptimizer.cpp:322
    // everything the reference leaves unset: Ceres Solver 2.2.0 defaults
    o->jacobi_scaling = 1;
    o->max_num_consecutive_invalid_steps = 5;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->min_relative_decrease = 1e-3;
}

int sadvio_ba_create(const sadvio_ba_config* cfg, sadvio_ba_handle** out) {
    if (!out) return SADVIO_E_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return SADVIO_E_NO_DEVICE;
    int dev = cfg ? cfg->device : 0;
    if (dev < 0 || dev >= n) return SADVIO_E_INVALID_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SADVIO_E_HIP;
    if (!strstr(prop.gcnArchName, "gfx950")) return SADVIO_E_NO_DEVICE;  // kernels are built for gfx950 only
    if (hipSetDevice(dev) != hipSuccess) return SADVIO_E_HIP;
    auto* h = new sadvio_ba_handle();
    h->env.read();
    if (cfg) h->cfg = *cfg;
    h->device = dev;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return SADVIO_E_HIP; }
    if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) h->side = nullptr;   // optional: without it everything is serial
    for (hipEvent_t* e : {&h->ev_fork, &h->ev_lin, &h->ev_solved, &h->ev_cost})
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { *e = nullptr; if (h->side) { (void)hipStreamDestroy(h->side); h->side = nullptr; } }
    *out = h;
    return SADVIO_OK;
}

void sadvio_ba_destroy(sadvio_ba_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); }
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->rccl.comm) (void)h->rccl.destroy(h->rccl.comm);
    if (h->h_final) (void)hipHostFree(h->h_final);
    if (h->h_deltas) (void)hipHostFree(h->h_deltas);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    for (hipEvent_t e : {h->ev_fork, h->ev_lin, h->ev_solved, h->ev_cost}) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (auto& e : h->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    h->d_win.release(); h->d_tiles.release(); h->d_kf_T0.release(); h->d_xp.release(); h->d_xv.release();
    h->d_xba.release(); h->d_xbg.release(); h->d_kf_vel.release(); h->d_kf_ba.release(); h->d_kf_bg.release();
    h->d_kf_fidx.release(); h->d_cam_K.release(); h->d_cam_T.release(); h->d_cam_isig.release();
    h->d_lmk_p.release(); h->d_xl.release(); h->d_s_lmk.release(); h->d_lmk_const.release();
    h->d_lmk_ob.release(); h->d_lmk_oe.release(); h->d_obs_kf.release(); h->d_obs_cam.release();
    h->d_obs_meas.release(); h->d_priors.release(); h->d_prior_lin.release(); h->d_S.release(); h->d_rank_s.release(); h->d_gred.release(); h->d_gfull.release();
    h->d_hdiag.release(); h->d_delta.release(); h->d_s_pose.release(); h->d_states.release(); h->d_trace.release(); h->d_tstart.release(); h->d_acc.release();
    h->d_probe.release(); h->d_tile_kf.release(); h->d_tile_row.release(); h->d_obs_slot.release(); h->d_ptab.release(); h->d_tacc.release(); h->d_dbg.release(); h->d_imus.release(); h->d_imu_scratch.release();
    delete h;
}

// (Re)build the device layout from the stored caller windows + the current factor lists: concatenation, the
// per-landmark observation order, pseudo-observations of eliminable pose-to-landmark factors, tiles, reduced layout.
static int build_layout(sadvio_ba_handle* h) {
    const bool dbg_t = (h->env.debug & 8192) != 0;
    auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (dbg_t) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[sadvio dbg] build_layout %-14s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_start).count()); t_start = t; } };
    const int n_windows = (int)h->src.size();
    HIP_TRY(hipSetDevice(h->device));
    h->solved = false;
    h->up.reset();
    h->wins.assign(n_windows, HostWin());
    h->tiles.clear();
    h->sp_elim.assign(n_windows, {});
    h->n_obs_user.assign(n_windows, 0);
    // which sparse factors ride the Schur elimination as pseudo-observations: PoseToLandmark factors whose landmark is
    // free and not held in the reduced system for another reason (dense prior, landmark prior / landmark chain factor)
    std::vector<sadvio_flat_window> views(n_windows);
    int sp_global = 0;
    for (int w = 0; w < n_windows; w++) {
        SrcWin& S = h->src[w];
        const auto& sp = h->sparse_per_win[w];
        h->sp_elim[w].assign(sp.size(), 0);
        h->n_obs_user[w] = S.v.n_obs;
        views[w] = S.v;
        {
            std::vector<int> cmap(S.v.n_cam, -1);
            S.u_cam_K.clear(); S.u_cam_T.clear(); S.u_cam_sigma.clear();
            for (int c = 0; c < S.v.n_cam; c++) {
                const double sg = S.cam_sigma.empty() ? 1.0 : S.cam_sigma[c];
                const int nu = (int)S.u_cam_sigma.size();
                for (int u = 0; u < nu && cmap[c] < 0; u++)
                    if (!memcmp(&S.u_cam_K[4 * u], &S.cam_K[4 * c], 32) && !memcmp(&S.u_cam_T[12 * u], &S.cam_T[12 * c], 96) && S.u_cam_sigma[u] == sg) cmap[c] = u;
                if (cmap[c] < 0) {
                    cmap[c] = nu;
                    S.u_cam_K.insert(S.u_cam_K.end(), &S.cam_K[4 * c], &S.cam_K[4 * c] + 4);
                    S.u_cam_T.insert(S.u_cam_T.end(), &S.cam_T[12 * c], &S.cam_T[12 * c] + 12);
                    S.u_cam_sigma.push_back(sg);
                }
            }
            S.u_obs_cam.resize(S.obs_cam.size());
            for (size_t o = 0; o < S.obs_cam.size(); o++) S.u_obs_cam[o] = cmap[S.obs_cam[o]];
            S.cam_map = cmap;
            views[w].n_cam = (int32_t)S.u_cam_sigma.size();
            views[w].cam_K = S.u_cam_K.data(); views[w].cam_T_s_f = S.u_cam_T.data(); views[w].cam_sigma = S.u_cam_sigma.data();
            views[w].obs_cam = S.u_obs_cam.data();
        }
        if (sp.empty()) { S.a_src.clear(); continue; }   // no sparse factors: nothing rides the elimination as a pseudo-observation
        auto& held = h->ls.held;
        held.assign(std::max(S.v.n_lmk, 1), 0);
        const DensePriorHost& D = h->dprior_per_win[w];
        for (size_t i = 0; i < D.lmk_index.size(); i++) if (D.n_full > 0 && D.lmk_col[i] >= 0) held[D.lmk_index[i]] = 1;
        for (const auto& s : sp) {
            if (s.type == SADVIO_SPARSE_LMK_PRIOR) held[s.lmk0] = 1;
            if (s.type == SADVIO_SPARSE_LMK_TO_LMK) { held[s.lmk0] = 1; held[s.lmk1] = 1; }
        }
        std::vector<std::vector<int>> extra(S.v.n_lmk);  // per landmark: global sparse-factor indices to append
        bool any = false;
        for (size_t k = 0; k < sp.size(); k++) {
            const auto& s = sp[k];
            if (s.type != SADVIO_SPARSE_POSE_TO_LMK || held[s.lmk0]) continue;
            if (!S.lmk_const.empty() && S.lmk_const[s.lmk0]) continue;
            h->sp_elim[w][k] = 1;
            extra[s.lmk0].push_back(sp_global + (int)k);
            any = true;
        }
        sp_global += (int)sp.size();
        if (any) {
            const int ms = S.v.factor_type == SADVIO_FACTOR_PIXEL ? 2 : 3;
            S.a_ptr.assign(1, 0); S.a_kf.clear(); S.a_cam.clear(); S.a_src.clear(); S.a_meas.clear();
            for (int l = 0; l < S.v.n_lmk; l++) {
                for (int o = S.lmk_obs_ptr[l]; o < S.lmk_obs_ptr[l + 1]; o++) {
                    S.a_kf.push_back(S.obs_kf[o]); S.a_cam.push_back(S.u_obs_cam[o]); S.a_src.push_back(o);
                    for (int q = 0; q < ms; q++) S.a_meas.push_back(S.obs_meas[(size_t)ms * o + q]);
                }
                for (int gfi : extra[l])
                    for (int half = 0; half < 2; half++) {
                        const auto& s = sp[gfi - (sp_global - (int)sp.size())];
                        S.a_kf.push_back(s.kf); S.a_cam.push_back(-1 - (2 * gfi + half)); S.a_src.push_back(-1);
                        for (int q = 0; q < ms; q++) S.a_meas.push_back(0.0);
                    }
                S.a_ptr.push_back((int32_t)S.a_kf.size());
            }
            views[w].n_obs = (int32_t)S.a_kf.size();
            views[w].lmk_obs_ptr = S.a_ptr.data(); views[w].obs_kf = S.a_kf.data(); views[w].obs_cam = S.a_cam.data();
            views[w].obs_meas = S.a_meas.data();
        } else {
            S.a_src.clear();
        }
    }
    lap("views");
    const sadvio_flat_window* wins = views.data();
    h->factor_type = wins[0].factor_type;
    int kf_b = 0, cam_b = 0, lmk_b = 0, obs_b = 0;
    h->max_n_kf = h->max_npose = h->max_np = 0; h->n_big = 0;
    h->has_lmk_const = false;
    for (int w = 0; w < n_windows; w++) {
        const sadvio_flat_window& F = wins[w];
        // a pose-graph window (relative-pose factors only) has no cameras, landmarks or observations
        if (F.n_kf <= 0 || F.n_cam < 0 || F.n_lmk < 0 || F.n_obs < 0 || !F.kf_T_f_w || (F.n_cam > 0 && (!F.cam_K || !F.cam_T_s_f)) ||
            (F.n_obs > 0 && F.n_cam == 0) ||
            (F.n_lmk > 0 && (!F.lmk_p || !F.lmk_obs_ptr)) || (F.n_obs > 0 && (!F.obs_kf || !F.obs_cam || !F.obs_meas))) {
            h->err = "set_windows: missing array in window " + std::to_string(w);
            return SADVIO_E_INVALID_ARG;
        }
        if (F.factor_type != h->factor_type || (F.factor_type != SADVIO_FACTOR_PIXEL && F.factor_type != SADVIO_FACTOR_ANGULAR)) {
            h->err = "set_windows: all windows of a batch must share one factor_type";
            return SADVIO_E_INVALID_ARG;
        }
        if (F.n_lmk > 0 && (F.lmk_obs_ptr[0] != 0 || F.lmk_obs_ptr[F.n_lmk] != F.n_obs)) {
            h->err = "set_windows: lmk_obs_ptr is not a CSR over n_obs";
            return SADVIO_E_INVALID_ARG;
        }
        if (F.lmk_const) h->has_lmk_const = true;
        HostWin& H = h->wins[w];
        WinDev& d = H.d;
        memset(&d, 0, sizeof(d));
        d.n_kf = F.n_kf; d.n_cam = F.n_cam; d.n_lmk = F.n_lmk; d.n_obs = F.n_obs;
        d.kf_base = kf_b; d.cam_base = cam_b; d.lmk_base = lmk_b; d.obs_base = obs_b;
        d.factor_type = F.factor_type; d.has_imu = F.has_imu;
        d.dpf = F.has_imu ? 15 : 6;
        int nfree = 0;
        for (int k = 0; k < F.n_kf; k++)
            if (!(F.kf_const && F.kf_const[k])) nfree++;
        d.n_free_kf = nfree;
        d.Npose = nfree * 6;
        d.Np = nfree * d.dpf;
        H.kf_id.assign(F.n_kf, 0); H.lmk_id.assign(F.n_lmk, 0);
        for (int k = 0; k < F.n_kf; k++) H.kf_id[k] = F.kf_id ? F.kf_id[k] : k;
        for (int l = 0; l < F.n_lmk; l++) H.lmk_id[l] = F.lmk_id ? F.lmk_id[l] : l;
        kf_b += F.n_kf; cam_b += F.n_cam; lmk_b += F.n_lmk; obs_b += F.n_obs;
        h->max_n_kf = std::max(h->max_n_kf, F.n_kf);
        h->max_npose = std::max(h->max_npose, d.Npose);
    }
    h->n_kf_tot = kf_b; h->n_cam_tot = cam_b; h->n_lmk_tot = lmk_b; h->n_obs_tot = obs_b;

    lap("  validate");
    // concatenate
    LayoutScratch& ls = h->ls;
    auto& kf_T0 = ls.kf_T0; auto& kf_vel = ls.kf_vel; auto& kf_ba = ls.kf_ba; auto& kf_bg = ls.kf_bg; auto& kf_fidx = ls.kf_fidx;
    auto& cam_K = ls.cam_K; auto& cam_T = ls.cam_T; auto& cam_isig = ls.cam_isig; auto& lmk_p = ls.lmk_p; auto& lmk_const = ls.lmk_const;
    auto& lmk_ob = ls.lmk_ob; auto& lmk_oe = ls.lmk_oe; auto& obs_kf = ls.obs_kf; auto& obs_cam = ls.obs_cam; auto& obs_meas = ls.obs_meas;
    auto& tile_kf = ls.tile_kf; auto& tile_row = ls.tile_row; auto& obs_slot = ls.obs_slot;
    kf_T0.resize(12 * (size_t)kf_b); kf_vel.assign(3 * (size_t)kf_b, 0.0); kf_ba.assign(3 * (size_t)kf_b, 0.0); kf_bg.assign(3 * (size_t)kf_b, 0.0);
    kf_fidx.resize(kf_b);
    cam_K.resize(4 * (size_t)cam_b); cam_T.resize(12 * (size_t)cam_b); cam_isig.resize(cam_b);
    lmk_p.resize(3 * (size_t)lmk_b);
    lmk_const.assign(std::max(lmk_b, 1), 0);
    lmk_ob.resize(std::max(lmk_b, 1)); lmk_oe.resize(std::max(lmk_b, 1)); obs_kf.resize(std::max(obs_b, 1)); obs_cam.resize(std::max(obs_b, 1));
    const int ms = h->factor_type == SADVIO_FACTOR_PIXEL ? 2 : 3;
    obs_meas.resize((size_t)ms * std::max(obs_b, 1));
    tile_kf.clear(); tile_row.clear();
    obs_slot.assign(std::max(obs_b, 1), 0);
    h->obs_perm.assign(std::max(obs_b, 1), 0);
    h->max_tile_kf = 1; h->max_tile_free = 0; h->max_gemm_free = 0; h->gemm_run4 = false;
    for (int w = 0; w < n_windows; w++) {
        const sadvio_flat_window& F = wins[w];
        WinDev& d = h->wins[w].d;
        memcpy(&kf_T0[12 * (size_t)d.kf_base], F.kf_T_f_w, sizeof(double) * 12 * F.n_kf);
        if (F.kf_vel) memcpy(&kf_vel[3 * (size_t)d.kf_base], F.kf_vel, sizeof(double) * 3 * F.n_kf);
        if (F.kf_ba) memcpy(&kf_ba[3 * (size_t)d.kf_base], F.kf_ba, sizeof(double) * 3 * F.n_kf);
        if (F.kf_bg) memcpy(&kf_bg[3 * (size_t)d.kf_base], F.kf_bg, sizeof(double) * 3 * F.n_kf);
        int fi = 0;
        for (int k = 0; k < F.n_kf; k++) kf_fidx[d.kf_base + k] = (F.kf_const && F.kf_const[k]) ? -1 : fi++;
        memcpy(&cam_K[4 * (size_t)d.cam_base], F.cam_K, sizeof(double) * 4 * F.n_cam);
        memcpy(&cam_T[12 * (size_t)d.cam_base], F.cam_T_s_f, sizeof(double) * 12 * F.n_cam);
        for (int c = 0; c < F.n_cam; c++) cam_isig[d.cam_base + c] = 1.0 / (F.cam_sigma ? F.cam_sigma[c] : 1.0);
        if (F.n_lmk) memcpy(&lmk_p[3 * (size_t)d.lmk_base], F.lmk_p, sizeof(double) * 3 * F.n_lmk);
        for (int l = 0; l < F.n_lmk; l++) {
            lmk_const[d.lmk_base + l] = F.lmk_const ? F.lmk_const[l] : 0;
            lmk_ob[d.lmk_base + l] = d.obs_base + F.lmk_obs_ptr[l];
            lmk_oe[d.lmk_base + l] = d.obs_base + F.lmk_obs_ptr[l + 1];
            if (F.lmk_obs_ptr[l + 1] < F.lmk_obs_ptr[l]) { h->err = "set_windows: CSR not monotone"; return SADVIO_E_INVALID_ARG; }
        }
        lap("  concat");
        // Observations of a landmark are stored sorted by key-frame (stable), so that the (at most two) cameras
        // of one key-frame sit in adjacent lanes; the landmark / key-frame order of the window is untouched and
        // obs_perm maps device position -> caller position for the per-observation probe.
        auto& pkf = ls.pkf; auto& pcam = ls.pcam; auto& run_max = ls.run_max; auto& idx = ls.idx;
        pkf.resize(std::max(F.n_obs, 1)); pcam.resize(std::max(F.n_obs, 1));
        run_max.assign(std::max(F.n_lmk, 1), 0);
        const bool has_asrc = !h->src[w].a_src.empty();
        const int32_t* asrc = has_asrc ? h->src[w].a_src.data() : nullptr;
        // first pass (integers only): is every landmark's list already key-frame sorted (what a flattening in frame order
        // produces)? Its longest same-key-frame run either way.
        bool all_sorted = true;
        int hb_win = 0;   // largest spread of free key-frame indices one landmark couples (half bandwidth of the reduced system)
        const int* fidx_w = kf_fidx.data() + d.kf_base;
        for (int l = 0; l < F.n_lmk; l++) {
            const int o0 = F.lmk_obs_ptr[l], o1 = F.lmk_obs_ptr[l + 1];
            int run = 0, rm = 0, prev = -1, lo = 1 << 30, hi = -1;
            for (int o = o0; o < o1; o++) {
                const int kf = F.obs_kf[o];
                if (kf < prev) { all_sorted = false; }
                run = (kf == prev) ? run + 1 : 1;
                rm = std::max(rm, run);
                prev = kf;
                const int fi = fidx_w[kf];
                if (fi >= 0) { lo = std::min(lo, fi); hi = std::max(hi, fi); }
            }
            run_max[l] = rm;
            if (hi >= 0) hb_win = std::max(hb_win, hi - lo);
        }
        h->wins[w].hb_lmk = hb_win;
        if (all_sorted) {
            // bulk path: the device order IS the caller's order — whole-array copies
            const int kb = d.kf_base, cb = d.cam_base, ob = d.obs_base, n = F.n_obs;
            for (int o = 0; o < n; o++) { pkf[o] = F.obs_kf[o]; obs_kf[ob + o] = kb + F.obs_kf[o]; }
            for (int o = 0; o < n; o++) { const int c = F.obs_cam[o]; pcam[o] = c; obs_cam[ob + o] = c < 0 ? c : cb + c; }
            if (has_asrc) for (int o = 0; o < n; o++) h->obs_perm[ob + o] = asrc[o];
            else for (int o = 0; o < n; o++) h->obs_perm[ob + o] = o;
            if (n) memcpy(&obs_meas[(size_t)ms * ob], F.obs_meas, sizeof(double) * (size_t)ms * n);
        } else
        for (int l = 0; l < F.n_lmk; l++) {
            const int o0 = F.lmk_obs_ptr[l], o1 = F.lmk_obs_ptr[l + 1], k_n = o1 - o0;
            // tracks are short: insertion sort (stable) on packed keys (key-frame << 8 | position) held in a local array —
            // no indirection through the caller's arrays inside the sort; longer tracks take the index sort
            int key[64];
            const int32_t* okf = F.obs_kf + o0;
            if (k_n <= 64 && F.n_kf < (1 << 22)) {
                for (int k = 0; k < k_n; k++) key[k] = (okf[k] << 8) | k;
                for (int a = 1; a < k_n; a++) {
                    const int v = key[a];
                    int b = a - 1;
                    while (b >= 0 && key[b] > v) { key[b + 1] = key[b]; b--; }
                    key[b + 1] = v;
                }
            } else {
                idx.resize(k_n);
                for (int k = 0; k < k_n; k++) idx[k] = k;
                std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return okf[a] < okf[b]; });
            }
            int run = 0, rm = 0, prev = -1;
            const int kb = d.kf_base, cb = d.cam_base, ob = d.obs_base;
            for (int k = 0; k < k_n; k++) {
                const int rel = k_n <= 64 && F.n_kf < (1 << 22) ? (key[k] & 255) : idx[k];
                const int src = o0 + rel, dst = o0 + k;
                const int kf = okf[rel], c = F.obs_cam[src];
                pkf[dst] = kf; pcam[dst] = c;
                h->obs_perm[ob + dst] = has_asrc ? asrc[src] : src;  // -1: pseudo-observation
                obs_kf[ob + dst] = kb + kf;
                obs_cam[ob + dst] = c < 0 ? c : cb + c;
                const double* m = F.obs_meas + (size_t)ms * src;
                double* md = &obs_meas[(size_t)ms * (ob + dst)];
                md[0] = m[0]; md[1] = m[1]; if (ms == 3) md[2] = m[2];
                run = (kf == prev) ? run + 1 : 1;
                rm = std::max(rm, run);
                prev = kf;
            }
            run_max[l] = rm;
        }
        lap("  sort+permute");
        lap("  half-bandwidth");
        // tiles: runs of consecutive landmarks. Every landmark gets a group of G lanes (G = pow2 >= the
        // tile's largest observation count); a workgroup of BUILD_WAVES waves holds BUILD_WAVES * 64 / G
        // landmarks per round. A tile is cut when its key-frame list would exceed the LDS tile capacity.
        if (F.n_cam > MAX_WIN_CAM) { h->err = "set_windows: more than 8 distinct cameras per window"; return SADVIO_E_INVALID_ARG; }
        // rounds per tile: one for a single window (most workgroups = lowest latency); a large batch gets fewer, larger tiles
        // (~2048 = 4 per resident workgroup slot) so that table staging, merge and flush are paid once per several rounds
        int tile_rounds = 1;
        {
            long long l_tot = 0;
            for (int ww = 0; ww < n_windows; ww++) l_tot += views[ww].n_lmk;
            tile_rounds = (int)std::min<long long>(16, std::max<long long>(1, (l_tot + 32LL * 2048 - 1) / (32LL * 2048)));
            if (h->env.tile_rounds > 0) tile_rounds = h->env.tile_rounds;
        }
        d.tile_begin = (int)h->tiles.size();
        {
            int l = 0;
            auto& mark = ls.mark; auto& add = ls.add; auto& kfs = ls.kfs; auto& slot_of = ls.slot_of;
            mark.assign(F.n_kf, -1); slot_of.assign(F.n_kf, -1);
            while (l < F.n_lmk || (int)h->tiles.size() == d.tile_begin) {
                Tile t{};
                t.w = w; t.lmk0 = d.lmk_base + l; t.kmax = 1; t.G = 8;
                t.dpf = d.dpf;  // Np, red_off, S_off, ld: layout_reduced
                t.cam_base = d.cam_base; t.n_cam = F.n_cam;
                t.first_of_window = ((int)h->tiles.size() == d.tile_begin) ? 1 : 0;
                kfs.clear();
                int nfree = 0, tile_run_max = 0;
                const int l_begin = l;
                while (l < F.n_lmk) {
                    const int k = F.lmk_obs_ptr[l + 1] - F.lmk_obs_ptr[l];
                    if (k > MAX_LMK_OBS) { h->err = "set_windows: a landmark has more than 64 observations"; return SADVIO_E_INVALID_ARG; }
                    int G = t.G;
                    while (G < k) G <<= 1;
                    const int cap = tile_rounds * BUILD_WAVES * (64 / G);  // landmarks per tile (tile_rounds rounds per wave)
                    if (l - l_begin + 1 > cap && l > l_begin) break;
                    // key-frames this landmark would add
                    add.clear();
                    int add_free = 0;
                    for (int o = F.lmk_obs_ptr[l]; o < F.lmk_obs_ptr[l + 1]; o++) {
                        const int kf = pkf[o];
                        if (mark[kf] != (int)h->tiles.size()) {
                            mark[kf] = (int)h->tiles.size();
                            add.push_back(kf);
                            if (!(F.kf_const && F.kf_const[kf])) add_free++;
                        }
                    }
                    const bool fits_hard = (int)(kfs.size() + add.size()) <= MAX_TILE_KF && nfree + add_free <= MAX_TILE_FREE_KF;
                    // soft limit: keep tiles on the MFMA path (<= MAX_GEMM_FREE_KF free key-frames) whenever a cut achieves it
                    const bool fits = fits_hard && (nfree + add_free <= MAX_GEMM_FREE_KF || l == l_begin);
                    if (!fits && l > l_begin) {
                        for (int kf : add) mark[kf] = -1;  // roll back
                        break;
                    }
                    for (int kf : add) kfs.push_back(kf);
                    nfree += add_free;
                    t.G = G;
                    t.kmax = std::max(t.kmax, k);
                    tile_run_max = std::max(tile_run_max, run_max[l]);
                    l++;
                    if (!fits_hard) break;  // a single landmark exceeding the capacity: global-atomics tile
                }
                t.lmk1 = d.lmk_base + l;
                std::sort(kfs.begin(), kfs.end());
                t.lds_mode = ((int)kfs.size() <= MAX_TILE_KF && nfree <= MAX_TILE_FREE_KF) ? ((nfree <= MAX_GEMM_FREE_KF && tile_run_max <= (has_asrc ? 4 : 2) && t.G == 8) ? 2 : 1) : 0;
                if (t.lds_mode == 2 && tile_run_max > 2) h->gemm_run4 = true;   // needs the RARE variant of k_build (pseudo-observations: it is taken)
                if (t.lds_mode == 2) h->max_gemm_free = std::max(h->max_gemm_free, nfree);
                if ((int)kfs.size() > 64) { h->err = "set_windows: a landmark is observed from more than 64 key-frames"; return SADVIO_E_INVALID_ARG; }
                t.kf_off = (int)tile_kf.size(); t.n_kf = (int)kfs.size(); t.n_free = t.lds_mode ? nfree : 0;
                int rank = 0;
                for (size_t i = 0; i < kfs.size(); i++) {
                    const int kf = kfs[i];
                    slot_of[kf] = (int)i;
                    tile_kf.push_back(d.kf_base + kf);
                    const bool is_const = F.kf_const && F.kf_const[kf];
                    if (is_const) tile_row.push_back(-1);
                    else if (t.lds_mode) tile_row.push_back(6 * rank++);
                    else tile_row.push_back(6 * kf_fidx[d.kf_base + kf]);  // global mode: 6 * free index of the window
                }
                for (int ll = l_begin; ll < l; ll++)
                    for (int o = F.lmk_obs_ptr[ll]; o < F.lmk_obs_ptr[ll + 1]; o++)
                        obs_slot[d.obs_base + o] = (unsigned char)slot_of[pkf[o]];
                h->max_tile_kf = std::max(h->max_tile_kf, t.n_kf);
                h->max_tile_free = std::max(h->max_tile_free, t.n_free);
                h->tiles.push_back(t);
                if (F.n_lmk == 0) break;
            }
        }
        d.tile_end = (int)h->tiles.size();
        for (int ti = d.tile_begin; ti < d.tile_end; ti++) { h->tiles[ti].win_tile0 = d.tile_begin; h->tiles[ti].win_ntiles = d.tile_end - d.tile_begin; }
    }
    lap("concat+tiles");
    // chunk tables of the throughput kernels: a tile's consecutive landmarks in chunks of <= LM_CHUNK landmarks and <= 64
    // observations; obs_lslot = index of the observation's landmark inside its chunk
    auto& chunk_ob = ls.chunk_ob; auto& chunk_lm = ls.chunk_lm;   // chunk starts + one sentinel (landmarks and observations are globally consecutive)
    auto& obs_lslot = ls.obs_lslot;
    chunk_ob.clear(); chunk_lm.clear();
    obs_lslot.assign(std::max(obs_b, 1), 0);
    // the tables cost host time (a second, sorted copy of the observation constants): only built where the throughput path can run
    bool want_lm = lmk_b >= 65536;
    if (h->env.lm >= 0) want_lm = h->env.lm != 0;
    h->lm_ok = want_lm && !h->tiles.empty();
    h->lm_landmarks = 0;
    h->lm_sub_obs = 0;
    for (auto& t : h->tiles) {
        if (!want_lm) { t.chunk0 = t.chunk1 = 0; continue; }
        t.chunk0 = (int)chunk_lm.size();
        if (t.lds_mode != 2) h->lm_ok = false;
        int l = t.lmk0;
        while (l < t.lmk1) {
            chunk_lm.push_back(l); chunk_ob.push_back(lmk_ob[l]);
            int nl = 0, no = 0;
            while (l < t.lmk1 && nl < LM_CHUNK && no + (lmk_oe[l] - lmk_ob[l]) <= 64) {
                for (int o = lmk_ob[l]; o < lmk_oe[l]; o++) obs_lslot[o] = (unsigned char)nl;
                no += lmk_oe[l] - lmk_ob[l]; nl++; l++;
            }
            if (nl == 0) { h->lm_ok = false; l++; }   // a landmark with more than 64 observations (never on the MFMA path)
        }
        t.chunk1 = (int)chunk_lm.size();
        h->lm_landmarks += t.lmk1 - t.lmk0;
    }
    chunk_lm.push_back(lmk_b); chunk_ob.push_back(obs_b);
    if (want_lm) {
        // k_lm_pass stages the observation constants of LM_PASS_THREADS consecutive landmarks of a tile in LDS: the largest such block
        for (const auto& t : h->tiles)
            for (int l0 = t.lmk0; l0 < t.lmk1; l0 += LM_PASS_THREADS) {
                const int l1 = std::min(l0 + LM_PASS_THREADS, t.lmk1);
                h->lm_sub_obs = std::max(h->lm_sub_obs, lmk_oe[l1 - 1] - lmk_ob[l0]);
            }
        h->lm_sub_obs = (h->lm_sub_obs + 3) & ~3;
        h->lm_max_cam = 1;
        for (const auto& t : h->tiles) h->lm_max_cam = std::max(h->lm_max_cam, t.n_cam);
        HIP_TRY(h->d_lm_hg.alloc(2 * (size_t)LM_HG * std::max(lmk_b, 1)));
        h->lm_ksub = 1;
        for (const auto& t : h->tiles) h->lm_ksub = std::max(h->lm_ksub, (t.lmk1 - t.lmk0 + LM_PASS_THREADS - 1) / LM_PASS_THREADS);
        const size_t n_rec = std::max<size_t>(h->tiles.size(), 1) * h->lm_ksub;
        HIP_TRY(h->d_lm_dt.alloc(2 * (size_t)LM_DT * n_rec));
        HIP_TRY(h->d_lm_sacc.alloc(2 * 4 * n_rec));
        HIP_TRY(hipMemsetAsync(h->d_lm_sacc.p, 0, sizeof(double) * 2 * 4 * n_rec, h->stream));   // slots of sub-blocks that do not exist stay zero
        // k_build_obs sums a tile's key-frame record over EVERY sub-block slot, k_lm_pass writes one slot per work item: the slots
        // nobody writes must be zero, and the buffer is grow-only (a re-layout with another tiling would leave stale records there)
        HIP_TRY(hipMemsetAsync(h->d_lm_dt.p, 0, sizeof(double) * 2 * (size_t)LM_DT * n_rec, h->stream));
    }
    HIP_TRY(h->d_chunk_ob.alloc(chunk_ob.size())); HIP_TRY(h->d_chunk_lm.alloc(chunk_lm.size())); HIP_TRY(h->d_obs_lslot.alloc(obs_lslot.size()));
    h->up.add(h->d_chunk_ob.p, chunk_ob.data(), chunk_ob.size() * sizeof(int));
    h->up.add(h->d_chunk_lm.p, chunk_lm.data(), chunk_lm.size() * sizeof(int));
    h->up.add(h->d_obs_lslot.p, obs_lslot.data(), obs_lslot.size());
    {
        // launch order of the throughput kernels: longest tiles first (LPT), so that the last workgroups to start are short ones
        auto& perm = ls.perm;
        perm.resize(h->tiles.size());
        for (size_t i = 0; i < perm.size(); i++) perm[i] = (int)i;
        if (want_lm && !h->env.no_lpt)
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
                return h->tiles[a].chunk1 - h->tiles[a].chunk0 > h->tiles[b].chunk1 - h->tiles[b].chunk0; });
        if (h->env.debug && want_lm && !perm.empty())
            fprintf(stderr, "[sadvio dbg] chunks per tile: largest %d, median %d, smallest %d\n", h->tiles[perm.front()].chunk1 - h->tiles[perm.front()].chunk0,
                    h->tiles[perm[perm.size() / 2]].chunk1 - h->tiles[perm[perm.size() / 2]].chunk0, h->tiles[perm.back()].chunk1 - h->tiles[perm.back()].chunk0);
        HIP_TRY(h->d_tile_perm.alloc(std::max<size_t>(perm.size(), 1)));
        h->up.add(h->d_tile_perm.p, perm.data(), perm.size() * sizeof(int));
        // work list of k_lm_pass: the sub-blocks (LM_PASS_THREADS landmarks) of every tile, in the same order
        std::vector<int> sub;
        if (h->env.lm_subs > 0) h->lm_sub_per_item = h->env.lm_subs;
        if (want_lm)
            for (int ti : perm) {
                const Tile& t = h->tiles[ti];
                for (int q = 0, l0 = t.lmk0; l0 < t.lmk1; l0 += LM_PASS_THREADS * h->lm_sub_per_item, q += h->lm_sub_per_item) { sub.push_back(ti); sub.push_back(q); }
            }
        h->lm_n_sub = (int)sub.size() / 2;
        HIP_TRY(h->d_lm_sub.alloc(std::max<size_t>(sub.size(), 2)));
        h->up.add(h->d_lm_sub.p, sub.data(), sub.size() * sizeof(int));
    }
    if (h->env.debug) {
        int hist[32] = {0}, modes[3] = {0};
        for (auto& t : h->tiles) { hist[std::min(t.n_free, 31)]++; modes[t.lds_mode]++; }
        fprintf(stderr, "[sadvio dbg] %zu tiles, modes global/atomic/gemm = %d/%d/%d, max_tile_kf %d, n_free histogram:", h->tiles.size(), modes[0], modes[1], modes[2], h->max_tile_kf);
        for (int i = 0; i < 32; i++) if (hist[i]) fprintf(stderr, " %d:%d", i, hist[i]);
        fprintf(stderr, "\n");
    }
    HIP_TRY(h->d_win.alloc(n_windows)); HIP_TRY(h->d_tacc.alloc(2 * h->tiles.size()));
    if (tile_kf.empty()) { tile_kf.push_back(0); tile_row.push_back(-1); }
    HIP_TRY(h->d_tile_kf.alloc(tile_kf.size())); HIP_TRY(h->d_tile_row.alloc(tile_row.size()));
    HIP_TRY(h->d_obs_slot.alloc(obs_slot.size())); HIP_TRY(h->d_ptab.alloc(2 * (size_t)POSE_TAB * kf_b));
    HIP_TRY(h->d_kf_T0.alloc(kf_T0.size())); HIP_TRY(h->d_kf_fidx.alloc(kf_fidx.size()));
    HIP_TRY(h->d_xp.alloc(2 * 6 * (size_t)kf_b)); HIP_TRY(h->d_xv.alloc(2 * 3 * (size_t)kf_b));
    HIP_TRY(h->d_xba.alloc(2 * 3 * (size_t)kf_b)); HIP_TRY(h->d_xbg.alloc(2 * 3 * (size_t)kf_b));
    HIP_TRY(h->d_kf_vel.alloc(kf_vel.size())); HIP_TRY(h->d_kf_ba.alloc(kf_ba.size())); HIP_TRY(h->d_kf_bg.alloc(kf_bg.size()));
    HIP_TRY(h->d_cam_K.alloc(cam_K.size())); HIP_TRY(h->d_cam_T.alloc(cam_T.size())); HIP_TRY(h->d_cam_isig.alloc(cam_isig.size()));
    HIP_TRY(h->d_lmk_p.alloc(lmk_p.size())); HIP_TRY(h->d_xl.alloc(2 * 3 * (size_t)std::max(lmk_b, 1)));
    HIP_TRY(h->d_s_lmk.alloc(3 * (size_t)std::max(lmk_b, 1)));
    HIP_TRY(h->d_lmk_ob.alloc(lmk_ob.size())); HIP_TRY(h->d_lmk_oe.alloc(lmk_oe.size()));
    HIP_TRY(h->d_obs_kf.alloc(obs_kf.size())); HIP_TRY(h->d_obs_cam.alloc(obs_cam.size())); HIP_TRY(h->d_obs_meas.alloc(obs_meas.size()));
#define UP(dst, src) h->up.add((dst).p, (src).data(), (src).size() * sizeof((src)[0]))
    UP(h->d_kf_T0, kf_T0); UP(h->d_kf_fidx, kf_fidx); UP(h->d_kf_vel, kf_vel);
    UP(h->d_kf_ba, kf_ba); UP(h->d_kf_bg, kf_bg); UP(h->d_cam_K, cam_K); UP(h->d_cam_T, cam_T); UP(h->d_cam_isig, cam_isig);
    if (lmk_b) { UP(h->d_lmk_p, lmk_p); }
    UP(h->d_lmk_ob, lmk_ob); UP(h->d_lmk_oe, lmk_oe); UP(h->d_obs_kf, obs_kf);
    UP(h->d_obs_cam, obs_cam); UP(h->d_obs_meas, obs_meas); UP(h->d_tile_kf, tile_kf); UP(h->d_tile_row, tile_row); UP(h->d_obs_slot, obs_slot);
    // First-round packets (round 6): what a lane of k_build / k_backsub needs to address the inputs of its tile's FIRST landmark round,
    // laid out by (tile, lane) so that the loads hang on blockIdx alone: | landmark | observation (-1: none) | observations of the
    // landmark + valid flag | first observation of the landmark |, and per tile the first PRE_KF key-frames of its list with their free
    // index. Without them the kernel's opening is a chain tile record -> CSR range / key-frame list -> observation / pose table
    // (three dependent round trips of ~1 us each on a single window); with them two. Few-tile submissions only (the single-window /
    // small-batch regime the latency kernels serve; 4 KB per tile).
    h->pre_ok = !h->tiles.empty() && h->tiles.size() <= PRE_MAX_TILES && !h->env.no_pre;
    if (h->pre_ok) {   // built on the device behind the upload (k_pre_packets, upload_priors): 1 MB of host stores + PCIe otherwise
        HIP_TRY(h->d_pre_lane.alloc((size_t)4 * (BUILD_THREADS / 8) * h->tiles.size())); HIP_TRY(h->d_pre_kf.alloc((size_t)2 * PRE_KF * h->tiles.size()));
        h->pre_dirty = true;
    }
#undef UP
    h->h_lmk_const_user = lmk_const; h->user_lmk_const = h->has_lmk_const;
    h->h_lmk_ob = lmk_ob; h->h_lmk_oe = lmk_oe; h->h_kf_fidx = kf_fidx; h->h_obs_kf = obs_kf;
    lap("alloc+queue");
    int rc = layout_reduced(h);
    if (rc != SADVIO_OK) return rc;
    lap("layout_reduced");
    rc = upload_priors(h);  // also uploads the window descriptors and synchronises (host vectors go out of scope)
    if (rc != SADVIO_OK) return rc;
    lap("flush");
    h->uploaded = true;
    for (auto& k : h->kclasses) { k.total_ms = 0; k.launches = 0; }
    return SADVIO_OK;
}

int sadvio_ba_set_windows(sadvio_ba_handle* h, int32_t n_windows, const sadvio_flat_window* wins) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (n_windows <= 0 || !wins) { h->err = "set_windows: no windows"; return SADVIO_E_INVALID_ARG; }
    h->uploaded = false; h->solved = false;
    for (int w = 0; w < n_windows; w++) {
        const sadvio_flat_window& F = wins[w];
        // a pose-graph window (relative-pose factors only) has no cameras, landmarks or observations
        if (F.n_kf <= 0 || F.n_cam < 0 || F.n_lmk < 0 || F.n_obs < 0 || !F.kf_T_f_w || (F.n_cam > 0 && (!F.cam_K || !F.cam_T_s_f)) ||
            (F.n_obs > 0 && F.n_cam == 0) ||
            (F.n_lmk > 0 && (!F.lmk_p || !F.lmk_obs_ptr)) || (F.n_obs > 0 && (!F.obs_kf || !F.obs_cam || !F.obs_meas))) {
            h->err = "set_windows: missing array in window " + std::to_string(w);
            return SADVIO_E_INVALID_ARG;
        }
        if (F.n_lmk > 0 && (F.lmk_obs_ptr[0] != 0 || F.lmk_obs_ptr[F.n_lmk] != F.n_obs)) {
            h->err = "set_windows: lmk_obs_ptr is not a CSR over n_obs";
            return SADVIO_E_INVALID_ARG;
        }
        for (int l = 0; l < F.n_lmk; l++)
            if (F.lmk_obs_ptr[l + 1] < F.lmk_obs_ptr[l]) { h->err = "set_windows: CSR not monotone"; return SADVIO_E_INVALID_ARG; }
        for (int o = 0; o < F.n_obs; o++)
            if (F.obs_kf[o] < 0 || F.obs_kf[o] >= F.n_kf || F.obs_cam[o] < 0 || F.obs_cam[o] >= F.n_cam) {
                h->err = "set_windows: observation index out of range";
                return SADVIO_E_INVALID_ARG;
            }
    }
    // deep copies: later set_* calls rebuild the layout without the caller's buffers
    if ((int)h->src.size() != n_windows) h->src.assign(n_windows, SrcWin());   // same batch size: the copies below reuse their capacity
    for (int w = 0; w < n_windows; w++) {
        const sadvio_flat_window& F = wins[w];
        SrcWin& S = h->src[w];
        S.cam_sigma.clear(); S.kf_id.clear(); S.kf_const.clear(); S.kf_vel.clear(); S.kf_ba.clear(); S.kf_bg.clear();
        S.lmk_p.clear(); S.lmk_id.clear(); S.lmk_const.clear(); S.obs_kf.clear(); S.obs_cam.clear(); S.obs_meas.clear();
        S.a_src.clear();
        const int ms = F.factor_type == SADVIO_FACTOR_PIXEL ? 2 : 3;
        S.kf_T.assign(F.kf_T_f_w, F.kf_T_f_w + 12 * (size_t)F.n_kf);
        S.cam_K.assign(F.cam_K, F.cam_K + 4 * (size_t)F.n_cam); S.cam_T.assign(F.cam_T_s_f, F.cam_T_s_f + 12 * (size_t)F.n_cam);
        if (F.cam_sigma) S.cam_sigma.assign(F.cam_sigma, F.cam_sigma + F.n_cam);
        if (F.kf_id) S.kf_id.assign(F.kf_id, F.kf_id + F.n_kf);
        if (F.kf_const) S.kf_const.assign(F.kf_const, F.kf_const + F.n_kf);
        if (F.kf_vel) S.kf_vel.assign(F.kf_vel, F.kf_vel + 3 * (size_t)F.n_kf);
        if (F.kf_ba) S.kf_ba.assign(F.kf_ba, F.kf_ba + 3 * (size_t)F.n_kf);
        if (F.kf_bg) S.kf_bg.assign(F.kf_bg, F.kf_bg + 3 * (size_t)F.n_kf);
        if (F.n_lmk) { S.lmk_p.assign(F.lmk_p, F.lmk_p + 3 * (size_t)F.n_lmk); S.lmk_obs_ptr.assign(F.lmk_obs_ptr, F.lmk_obs_ptr + F.n_lmk + 1); }
        else S.lmk_obs_ptr.assign(1, 0);
        if (F.lmk_id) S.lmk_id.assign(F.lmk_id, F.lmk_id + F.n_lmk);
        if (F.lmk_const) S.lmk_const.assign(F.lmk_const, F.lmk_const + F.n_lmk);
        if (F.n_obs) {
            S.obs_kf.assign(F.obs_kf, F.obs_kf + F.n_obs); S.obs_cam.assign(F.obs_cam, F.obs_cam + F.n_obs);
            S.obs_meas.assign(F.obs_meas, F.obs_meas + (size_t)ms * F.n_obs);
        }
        S.v = F;
        S.v.kf_T_f_w = S.kf_T.data(); S.v.cam_K = S.cam_K.data(); S.v.cam_T_s_f = S.cam_T.data();
        S.v.cam_sigma = F.cam_sigma ? S.cam_sigma.data() : nullptr;
        S.v.kf_id = F.kf_id ? S.kf_id.data() : nullptr; S.v.kf_const = F.kf_const ? S.kf_const.data() : nullptr;
        S.v.kf_vel = F.kf_vel ? S.kf_vel.data() : nullptr; S.v.kf_ba = F.kf_ba ? S.kf_ba.data() : nullptr; S.v.kf_bg = F.kf_bg ? S.kf_bg.data() : nullptr;
        S.v.lmk_p = S.lmk_p.data(); S.v.lmk_obs_ptr = S.lmk_obs_ptr.data();
        S.v.lmk_id = F.lmk_id ? S.lmk_id.data() : nullptr; S.v.lmk_const = F.lmk_const ? S.lmk_const.data() : nullptr;
        S.v.obs_kf = S.obs_kf.data(); S.v.obs_cam = S.obs_cam.data(); S.v.obs_meas = S.obs_meas.data();
    }
    h->priors_per_win.assign(n_windows, {});
    h->imus_per_win.assign(n_windows, {});
    h->dprior_per_win.assign(n_windows, {});
    h->sparse_per_win.assign(n_windows, {});
    h->lines_per_win.assign(n_windows, {});
    if (h->defer) {
        // the factor setters that follow validate against the windows' sizes and convert indices with their offsets in the batch:
        // the same numbers build_layout derives at commit
        h->wins.assign(n_windows, HostWin());
        int kf_b = 0, lmk_b = 0, obs_b = 0;
        for (int w = 0; w < n_windows; w++) {
            const sadvio_flat_window& F = wins[w];
            WinDev& d = h->wins[w].d;
            memset(&d, 0, sizeof(d));
            d.n_kf = F.n_kf; d.n_cam = F.n_cam; d.n_lmk = F.n_lmk; d.n_obs = F.n_obs;
            d.kf_base = kf_b; d.lmk_base = lmk_b; d.obs_base = obs_b;
            d.factor_type = F.factor_type; d.has_imu = F.has_imu; d.dpf = F.has_imu ? 15 : 6;
            kf_b += F.n_kf; lmk_b += F.n_lmk; obs_b += F.n_obs;
        }
        h->uploaded = true; h->pending = true;
        return SADVIO_OK;
    }
    return build_layout(h);
}

int sadvio_ba_begin_update(sadvio_ba_handle* h) {
    if (!h) return SADVIO_E_INVALID_ARG;
    h->defer = true; h->pending = false;
    return SADVIO_OK;
}

int sadvio_ba_commit_update(sadvio_ba_handle* h) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->defer) { h->err = "commit_update without begin_update"; return SADVIO_E_STATE; }
    h->defer = false;
    if (!h->pending) return SADVIO_OK;
    h->pending = false;
    HIP_TRY(hipSetDevice(h->device));
    h->uploaded = false;   // the deferred set_windows left stub window records: a failed build must not leave them usable
    return build_layout(h);
}

int sadvio_ba_set_lines(sadvio_ba_handle* h, int32_t w, const sadvio_line_set* L) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "set_lines before set_windows"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size()) { h->err = "set_lines: window out of range"; return SADVIO_E_INVALID_ARG; }
    const int n = L ? L->n_line : 0;
    if (h->world > 1 && n > 0) { h->err = "set_lines: not supported on a window sharded over several GPUs"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const WinDev& d = h->wins[w].d;
    LineSetHost H;
    if (n > 0) {
        if (n < 0 || L->n_obs < 0 || !L->line_T_w_l || !L->line_model || !L->line_obs_ptr || (L->n_obs > 0 && (!L->obs_kf || !L->obs_cam || !L->obs_meas))) {
            h->err = "set_lines: missing array"; return SADVIO_E_INVALID_ARG;
        }
        if (L->line_obs_ptr[0] != 0 || L->line_obs_ptr[n] != L->n_obs) { h->err = "set_lines: line_obs_ptr is not a CSR over n_obs"; return SADVIO_E_INVALID_ARG; }
        for (int l = 0; l < n; l++) if (L->line_obs_ptr[l + 1] < L->line_obs_ptr[l]) { h->err = "set_lines: CSR not monotone"; return SADVIO_E_INVALID_ARG; }
        for (int o = 0; o < L->n_obs; o++)
            if (L->obs_kf[o] < 0 || L->obs_kf[o] >= d.n_kf || L->obs_cam[o] < 0 || L->obs_cam[o] >= h->src[w].v.n_cam) {   // the caller's camera count (cam_map, built with the layout, has as many entries)
                h->err = "set_lines: observation index out of range"; return SADVIO_E_INVALID_ARG;
            }
        const int ms = d.factor_type == SADVIO_FACTOR_PIXEL ? 4 : 6;
        H.id.resize(n);
        for (int l = 0; l < n; l++) H.id[l] = L->line_id ? L->line_id[l] : l;
        H.T.assign(L->line_T_w_l, L->line_T_w_l + 12 * (size_t)n); H.model.assign(L->line_model, L->line_model + 6 * (size_t)n);
        if (L->line_const) H.is_const.assign(L->line_const, L->line_const + n);
        H.ptr.assign(L->line_obs_ptr, L->line_obs_ptr + n + 1);
        if (L->n_obs) {
            H.obs_kf.assign(L->obs_kf, L->obs_kf + L->n_obs); H.obs_cam.assign(L->obs_cam, L->obs_cam + L->n_obs);
            H.meas.assign(L->obs_meas, L->obs_meas + (size_t)ms * L->n_obs);
        }
    }
    h->lines_per_win[w] = std::move(H);
    h->solved = false;
    if (h->defer) { h->pending = true; return SADVIO_OK; }
    h->up.reset();
    int rc = layout_reduced(h);   // the lines enlarge the reduced system; the landmark tiles are unchanged
    if (rc != SADVIO_OK) return rc;
    if (h->defer) { h->pending = true; return SADVIO_OK; }
    return upload_priors(h);
}

int sadvio_ba_get_line_deltas(sadvio_ba_handle* h, int32_t w, double* line_delta6) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->solved) { h->err = "get_line_deltas before solve"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size() || !line_delta6) { h->err = "get_line_deltas: bad argument"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const WinDev& d = h->wins[w].d;
    const int n = d.line_end - d.line_begin;
    if (n > 0) HIP_TRY(hipMemcpy(line_delta6, h->d_xline.p + (size_t)h->fin[w].s.cur * 6 * h->n_line_tot + 6 * (size_t)d.line_begin, sizeof(double) * 6 * n, hipMemcpyDeviceToHost));
    return SADVIO_OK;
}

int sadvio_ba_set_pose_priors(sadvio_ba_handle* h, int32_t w, int32_t n, const sadvio_pose_prior* pr) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "set_pose_priors before set_windows"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size() || n < 0 || (n > 0 && !pr)) { h->err = "set_pose_priors: bad argument"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    auto& v = h->priors_per_win[w];
    v.clear();
    for (int i = 0; i < n; i++) {
        if (pr[i].kf < 0 || pr[i].kf >= h->wins[w].d.n_kf) { h->err = "set_pose_priors: kf out of range"; return SADVIO_E_INVALID_ARG; }
        PriorDev d{};
        d.kf = h->wins[w].d.kf_base + pr[i].kf;
        memcpy(d.T_prior, pr[i].T_prior, sizeof(d.T_prior));
        memcpy(d.inf, pr[i].inf_diag, sizeof(d.inf));
        v.push_back(d);
    }
    if (h->defer) { h->pending = true; return SADVIO_OK; }
    h->up.reset();
    return upload_priors(h);
}

// 9x9 square-root information W = L^T with L L^T = cov^-1 (residuals.hpp:151-154): Gauss-Jordan inverse with
// partial pivoting + Cholesky, on the host, once per factor.
static bool imu_sqrt_information(const double* cov, double* W) {
    double A[81], I[81];
    memcpy(A, cov, sizeof(A));
    memset(I, 0, sizeof(I));
    for (int i = 0; i < 9; i++) I[i * 9 + i] = 1.0;
    for (int c = 0; c < 9; c++) {
        int piv = c;
        double best = fabs(A[c * 9 + c]);
        for (int r = c + 1; r < 9; r++)
            if (fabs(A[r * 9 + c]) > best) { best = fabs(A[r * 9 + c]); piv = r; }
        if (best == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 9; j++) { std::swap(A[c * 9 + j], A[piv * 9 + j]); std::swap(I[c * 9 + j], I[piv * 9 + j]); }
        double d = 1.0 / A[c * 9 + c];
        for (int j = 0; j < 9; j++) { A[c * 9 + j] *= d; I[c * 9 + j] *= d; }
        for (int r = 0; r < 9; r++) {
            if (r == c) continue;
            double f = A[r * 9 + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 9; j++) { A[r * 9 + j] -= f * A[c * 9 + j]; I[r * 9 + j] -= f * I[c * 9 + j]; }
        }
    }
    double L[81];
    memset(L, 0, sizeof(L));
    for (int j = 0; j < 9; j++) {
        double s = I[j * 9 + j];
        for (int k = 0; k < j; k++) s -= L[j * 9 + k] * L[j * 9 + k];
        if (!(s > 0.0)) return false;
        double d = sqrt(s);
        L[j * 9 + j] = d;
        for (int i = j + 1; i < 9; i++) {
            double t = I[i * 9 + j];
            for (int k = 0; k < j; k++) t -= L[i * 9 + k] * L[j * 9 + k];
            L[i * 9 + j] = t / d;
        }
    }
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) W[i * 9 + j] = L[j * 9 + i];
    return true;
}

int sadvio_ba_set_imu_factors(sadvio_ba_handle* h, int32_t w, int32_t n, const sadvio_imu_factor* fs) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "set_imu_factors before set_windows"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size() || n < 0 || (n > 0 && !fs)) { h->err = "set_imu_factors: bad argument"; return SADVIO_E_INVALID_ARG; }
    const WinDev& d = h->wins[w].d;
    if (n > 0 && !d.has_imu) { h->err = "set_imu_factors: the window was uploaded with has_imu = 0"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    auto& v = h->imus_per_win[w];
    v.clear();
    for (int k = 0; k < n; k++) {
        const sadvio_imu_factor& f = fs[k];
        if (f.kf_i < 0 || f.kf_i >= d.n_kf || f.kf_j < 0 || f.kf_j >= d.n_kf || f.kf_i == f.kf_j || !(f.dt > 0)) {
            h->err = "set_imu_factors: key-frame index / dt out of range"; return SADVIO_E_INVALID_ARG;
        }
        ImuDev o{};
        o.kf_i = d.kf_base + f.kf_i; o.kf_j = d.kf_base + f.kf_j; o.dt = f.dt;
        memcpy(o.dR, f.delta_R, sizeof(o.dR)); memcpy(o.dv, f.delta_v, sizeof(o.dv)); memcpy(o.dp, f.delta_p, sizeof(o.dp));
        memcpy(o.J_dR_bg, f.J_dR_bg, 72); memcpy(o.J_dv_ba, f.J_dv_ba, 72); memcpy(o.J_dv_bg, f.J_dv_bg, 72);
        memcpy(o.J_dp_ba, f.J_dp_ba, 72); memcpy(o.J_dp_bg, f.J_dp_bg, 72);
        if (!imu_sqrt_information(f.cov, o.W)) { h->err = "set_imu_factors: covariance is not positive definite"; return SADVIO_E_INVALID_ARG; }
        o.sa = 1.0 / sqrt(f.dt * f.bacc_noise * f.bacc_noise);
        o.sg = 1.0 / sqrt(f.dt * f.bgyr_noise * f.bgyr_noise);
        v.push_back(o);
    }
    if (h->defer) { h->pending = true; return SADVIO_OK; }
    h->up.reset();
    return upload_priors(h);
}

int sadvio_ba_set_dense_prior(sadvio_ba_handle* h, int32_t w, int32_t n_full, int32_t n, const double* J, const double* r0,
                              int32_t kf_keep, int32_t kf_col, int32_t n_keep, const int32_t* lmk_index, const int32_t* lmk_col) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "set_dense_prior before set_windows"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size() || (n_full < 0 && n_full != SADVIO_PRIOR_RESIDENT) || (n < 0 && n_full != SADVIO_PRIOR_RESIDENT) || n_keep < 0) { h->err = "set_dense_prior: bad argument"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    // a window sharded over several GPUs carries a dense prior too (round 5): every rank holds the prior and its variables — the kept
    // frame is replicated anyway, the kept landmarks are in every rank's window (with their observations on rank 0 only:
    // sadvio_amd/sharding.py) — rank 0 adds J^T J / J^T r to the all-reduced system, every rank evaluates the cost with the same bits
    const WinDev& d = h->wins[w].d;
    DensePriorHost D;
    const bool resident = n_full == SADVIO_PRIOR_RESIDENT;
    if (resident) {
        if (J || r0) { h->err = "set_dense_prior: SADVIO_PRIOR_RESIDENT takes J = r0 = NULL"; return SADVIO_E_INVALID_ARG; }
        if (!h->prior.valid) { h->err = "set_dense_prior: SADVIO_PRIOR_RESIDENT but the handle holds no prior"; return SADVIO_E_STATE; }
        if (h->world > 1) { h->err = "set_dense_prior: not supported on a window sharded over several GPUs"; return SADVIO_E_INVALID_ARG; }
        n_full = h->prior.n_full; n = h->prior.n;
    }
    if (n_full > 0) {
        if ((!resident && (!J || !r0)) || n <= 0 || (n_keep > 0 && (!lmk_index || !lmk_col))) { h->err = "set_dense_prior: missing array"; return SADVIO_E_INVALID_ARG; }
        if (kf_keep >= d.n_kf || (kf_keep >= 0 && (kf_col < 0 || kf_col + 15 > n))) { h->err = "set_dense_prior: kept key-frame block out of range"; return SADVIO_E_INVALID_ARG; }
        std::vector<char> used(n, 0);
        if (kf_keep >= 0) for (int q = 0; q < 15; q++) used[kf_col + q] = 1;
        for (int i = 0; i < n_keep; i++) {
            if (lmk_col[i] < 0) continue;
            if (lmk_index[i] < 0 || lmk_index[i] >= d.n_lmk || lmk_col[i] + 3 > n) { h->err = "set_dense_prior: kept landmark out of range"; return SADVIO_E_INVALID_ARG; }
            for (int a = 0; a < 3; a++) {
                if (used[lmk_col[i] + a]) { h->err = "set_dense_prior: overlapping column blocks"; return SADVIO_E_INVALID_ARG; }
                used[lmk_col[i] + a] = 1;
            }
            // sharded window (sadvio_ba.h): the kept landmarks' observations live on rank 0 ONLY — every rank adds its kept rows to the
            // all-reduced system, so observations present on two ranks would be counted twice without any error
            if (h->world > 1 && h->rank != 0) {
                const int32_t* op = h->src[w].v.lmk_obs_ptr;
                if (op && op[lmk_index[i] + 1] != op[lmk_index[i]]) {
                    h->err = "set_dense_prior: on a sharded window the kept landmarks carry their observations on rank 0 only (landmark " + std::to_string(lmk_index[i]) + " has some on rank " + std::to_string(h->rank) + ")";
                    return SADVIO_E_INVALID_ARG;
                }
            }
        }
        D.n_full = n_full; D.n = n; D.kf_keep = kf_keep; D.kf_col = kf_col;
        D.resident = resident; D.serial = h->prior.serial;
        if (!resident) { D.J.assign(J, J + (size_t)n_full * n); D.r0.assign(r0, r0 + n_full); }
        D.lmk_index.assign(lmk_index, lmk_index + n_keep); D.lmk_col.assign(lmk_col, lmk_col + n_keep);
    }
    h->dprior_per_win[w] = std::move(D);
    if (h->defer) { h->pending = true; return SADVIO_OK; }
    if (!h->sparse_per_win[w].empty()) return build_layout(h);  // which sparse factors are eliminable may change
    h->up.reset();
    int rc = layout_reduced(h);
    if (rc != SADVIO_OK) return rc;
    if (h->defer) { h->pending = true; return SADVIO_OK; }
    return upload_priors(h);
}

int sadvio_ba_set_sparse_priors(sadvio_ba_handle* h, int32_t w, int32_t n, const sadvio_sparse_prior* f) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "set_sparse_priors before set_windows"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size() || n < 0 || (n > 0 && !f)) { h->err = "set_sparse_priors: bad argument"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const WinDev& d = h->wins[w].d;
    for (int i = 0; i < n; i++) {
        const sadvio_sparse_prior& s = f[i];
        // A window sharded over several GPUs carries the SPARSIFIED VIO prior (what sparsifyVIO produces): IMUPriordx is a
        // pose-only factor every rank evaluates identically after the all-reduce, a PoseToLandmarkFactor on a free landmark
        // rides the elimination of the rank that owns the landmark. Factors that hold landmarks in the reduced system
        // (Landmark3DPrior, landmark chains, relative poses of other layouts) would need those landmarks on every rank.
        if (h->world > 1 && !(s.type == SADVIO_SPARSE_IMU_PRIOR || (s.type == SADVIO_SPARSE_POSE_TO_LMK && s.lmk0 >= 0 && s.lmk0 < d.n_lmk &&
                                                                     !(h->src[w].v.lmk_const && h->src[w].v.lmk_const[s.lmk0])))) {
            h->err = "set_sparse_priors: a window sharded over several GPUs takes IMU-prior and pose-to-landmark factors (on free landmarks of this rank) only";
            return SADVIO_E_INVALID_ARG;
        }
        const bool rel = s.type == SADVIO_SPARSE_RELATIVE_POSE;
        if (rel && (s.kf_b < 0 || s.kf_b >= d.n_kf || s.kf_b == s.kf || d.dpf != 6)) {
            h->err = "set_sparse_priors: relative-pose factor " + std::to_string(i) + " has a bad key-frame (or the window carries IMU states)";
            return SADVIO_E_INVALID_ARG;
        }
        const bool need_kf = s.type == SADVIO_SPARSE_IMU_PRIOR || s.type == SADVIO_SPARSE_POSE_TO_LMK || rel;
        const bool need_l0 = s.type != SADVIO_SPARSE_IMU_PRIOR && !rel, need_l1 = s.type == SADVIO_SPARSE_LMK_TO_LMK;
        if (s.type < 0 || s.type > 4 || (need_kf && (s.kf < 0 || s.kf >= d.n_kf)) || (need_l0 && (s.lmk0 < 0 || s.lmk0 >= d.n_lmk)) ||
            (need_l1 && (s.lmk1 < 0 || s.lmk1 >= d.n_lmk || s.lmk1 == s.lmk0))) {
            h->err = "set_sparse_priors: factor " + std::to_string(i) + " has a bad type or index";
            return SADVIO_E_INVALID_ARG;
        }
    }
    h->sparse_per_win[w].assign(f, f + n);
    if (h->defer) { h->pending = true; return SADVIO_OK; }
    return build_layout(h);  // eliminable pose-to-landmark factors become pseudo-observations: the tiles change
}

namespace {
// Diagonally pivoted Cholesky S = G^T G without data movement (marg_kernels.h: k_pchol_panel_np / k_pchol_syrk_full), in place on
// the n x n scratch S (destroyed); G (n x n) receives the factor's rows by ORIGINAL column index, h->d_jac_ints[0..n) the pivot
// step of every index (-1 = never chosen). tau >= 0: stop at pivots <= tau * max diagonal; tau < 0: at pivots <= -tau
// (absolute). Returns the rank (number of pivots taken), negative on a HIP error. One host synchronisation (the rank).
int run_pchol(sadvio_ba_handle* h, double* S, int n, double* G, double tau, bool allow_swap = true) {
    if (h->d_jac_ints.alloc((size_t)n + 8) != hipSuccess) return -1;
    int* piv = h->d_jac_ints.p; int* rank_d = piv + n;
    if (h->d_jac_dbl.alloc(2 * (size_t)n + 8) != hipSuccess) return -1;
    double* dg = h->d_jac_dbl.p; double* dctl = dg + 2 * (size_t)n;      // remaining diagonal | original diagonal | tau
    if (hipMemsetAsync(rank_d, 0, sizeof(int) * 8, h->stream) != hipSuccess) return -1;
    if (hipMemsetAsync(rank_d, 0xff, sizeof(int), h->stream) != hipSuccess) return -1;   // -1: still factorising
    const bool swap_pchol = allow_swap && h->env.pchol_swap;   // the data-moving version (kept for comparison; the eigen path only:
                                                                                      // the Cholesky-form routes read the factor by original column index)
    if (swap_pchol) {
        for (int k0 = 0; k0 < n; k0 += PCH_NB) {
            hipLaunchKernelGGL(k_pchol_panel, dim3(1), dim3(PCH_THREADS), 0, h->stream, S, n, G, piv, dg, rank_d, dctl, k0, tau);
            const int m = n - (k0 + PCH_NB);
            if (m > 0) hipLaunchKernelGGL(k_pchol_syrk, dim3((m + 63) / 64, (m + 63) / 64), dim3(256), 0, h->stream, S, n, G, rank_d, k0);
        }
    } else {
        if (hipMemsetAsync(piv, 0xff, sizeof(int) * (size_t)n, h->stream) != hipSuccess) return -1;   // done[i] = -1
        const unsigned gt = (unsigned)((n + 63) / 64);
        if (!h->env.pchol_strict) {
            // relaxed pivoting (marg_kernels.h: k_pchol_panel_rx): a panel picks its pivots up front; the first row of a panel is device state
            const double safe = 1024.0 * n * 2.220446049250313e-16;
            const int nb = n <= PCH_THREADS ? 32 : 16;
            int launched = 0, r = -1;
            for (int round = 0; round < 64 && r < 0; round++) {
                const int np = round == 0 ? (n + nb - 1) / nb + 8 : 4;   // threshold pivoting leaves some panels partly filled; a launch after the end returns at once
                for (int pnl = 0; pnl < np; pnl++, launched++) {
                    if (n <= PCH_THREADS) {
                        hipLaunchKernelGGL((k_pchol_panel_rx<1, 32>), dim3(1), dim3(PCH_THREADS), 0, h->stream, S, n, G, piv, dg, rank_d, dctl, launched == 0 ? 1 : 0, tau, safe);
                        hipLaunchKernelGGL(k_pchol_syrk_mma<32>, dim3(gt, gt), dim3(256), 0, h->stream, S, n, G, rank_d);
                    } else {
                        hipLaunchKernelGGL((k_pchol_panel_rx<2, 16>), dim3(1), dim3(PCH_THREADS), 0, h->stream, S, n, G, piv, dg, rank_d, dctl, launched == 0 ? 1 : 0, tau, safe);
                        hipLaunchKernelGGL(k_pchol_syrk_mma<16>, dim3(gt, gt), dim3(256), 0, h->stream, S, n, G, rank_d);
                    }
                }
                if (hipMemcpyAsync(&r, rank_d, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1;
            }
            if ((h->env.debug & 16384)) {
                int c8[8];
                if (hipMemcpy(c8, rank_d, sizeof(c8), hipMemcpyDeviceToHost) == hipSuccess)
                    fprintf(stderr, "[sadvio dbg] relaxed pivoted cholesky n %d rank %d: %d panels launched, %d ran (%d strict), %d candidates skipped\n", n, r, launched, c8[3] + 1, c8[4], c8[5]);
            }
            return r < 0 ? -1 : r;
        } else if (n <= PCH_THREADS) {
            for (int k0 = 0; k0 < n; k0 += 32) {
                hipLaunchKernelGGL((k_pchol_panel_np<1, 32>), dim3(1), dim3(PCH_THREADS), 0, h->stream, S, n, G, piv, dg, rank_d, dctl, k0, tau);
                if (k0 + 32 < n) hipLaunchKernelGGL(k_pchol_syrk_full<32>, dim3(gt, gt), dim3(256), 0, h->stream, S, n, G, rank_d, k0);
            }
        } else {
            for (int k0 = 0; k0 < n; k0 += 16) {
                hipLaunchKernelGGL((k_pchol_panel_np<2, 16>), dim3(1), dim3(PCH_THREADS), 0, h->stream, S, n, G, piv, dg, rank_d, dctl, k0, tau);
                if (k0 + 16 < n) hipLaunchKernelGGL(k_pchol_syrk_full<16>, dim3(gt, gt), dim3(256), 0, h->stream, S, n, G, rank_d, k0);
            }
        }
    }
    int r = 0;
    if (hipMemcpyAsync(&r, rank_d, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    if (r < 0) r = n;
    return r;
}

// Block one-sided Jacobi on the r rows (length n, packed) of G until they are mutually orthogonal: the rows converge to
// sqrt(lambda_i) u_i^T of G^T G. Returns the number of sweeps (negative = HIP error). One host synchronisation per sweep.
int run_jacobi_rows(sadvio_ba_handle* h, double* G, int r, int n, int* flag) {
    const bool b4 = h->env.jacobi_b4 || n > JM_MAXN;   // the 4-row VALU version (large n; kept for comparison)
    const int ldx = jm_ldx(n);
    const size_t jm_lds = (size_t)JM2 * ldx * sizeof(double);
    if (!b4 && hipFuncSetAttribute((const void*)k_jacobi_mma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)jm_lds) != hipSuccess) return -1;
    const int jb = b4 ? JB : JM;
    const int nb = (r + jb - 1) / jb, nbpad = nb + (nb & 1);
    const double jtol = h->env.jacobi_tol;   // |g_p . g_q| <= jtol |g_p| |g_q| ends a pair
    int sweeps = 0;
    long long* jts = nullptr;   // phase timestamps of one launch (SADVIO_KERNEL_TS builds, SADVIO_DEBUG & 4096)
    if ((h->env.debug & 4096) && n > 500 && h->d_dbg.alloc(128) == hipSuccess) jts = h->d_dbg.p + 44;
    for (; sweeps < 40 && nbpad >= 2; sweeps++) {
        if (hipMemsetAsync(flag, 0, sizeof(int), h->stream) != hipSuccess) return -1;
        for (int st = 0; st < nbpad - 1; st++) {
            long long* ts = sweeps == 0 && st == 3 ? jts : nullptr;
            if (!b4) hipLaunchKernelGGL(k_jacobi_mma, dim3(nbpad / 2), dim3(JAC_THREADS), jm_lds, h->stream, G, r, n, ldx, nbpad, st, jtol, flag, ts);
            else if (n <= 4 * JAC_THREADS) hipLaunchKernelGGL(k_jacobi_block<4>, dim3(nbpad / 2), dim3(JAC_THREADS), 0, h->stream, G, r, n, nbpad, st, jtol, flag);
            else hipLaunchKernelGGL(k_jacobi_block<8>, dim3(nbpad / 2), dim3(JAC_THREADS), 0, h->stream, G, r, n, nbpad, st, jtol, flag);
        }
        int f = 0;
        if (hipMemcpyAsync(&f, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1;
        if ((h->env.debug & 16384)) fprintf(stderr, "[sadvio dbg] block jacobi n %d rank %d sweep %d block pairs rotated %d\n", n, r, sweeps, f);
        if (!f) { sweeps++; break; }
    }
    if (jts) {
        long long t8[8];
        if (hipMemcpy(t8, jts, sizeof(t8), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[sadvio dbg] k_jacobi_mma phases (us, cumulative): gram | update loads issued | barrier | M | check | inner sweep | end:");
            for (int i = 1; i < 8; i++) fprintf(stderr, " %.2f", (t8[i] - t8[0]) * 0.01);
            fprintf(stderr, "\n");
        }
    }
    return sweeps;
}

// pivot tolerance of the rank-revealing Cholesky for an eigenvalue-cut mode: the noise floor's pivots end at 4 n eps of the
// largest one (the null space of a marginalisation prior sits exactly there); the reference's absolute 1e-12 (Marginalization::
// _eps) becomes a pivot floor of 1e-12 / n — a remaining eigenvalue above 1e-12 keeps the remaining trace, hence the largest
// remaining diagonal entry, above it.
double pchol_tau(int n, int eig_cut_mode) {
    return eig_cut_mode == SADVIO_EIG_CUT_NOISE_FLOOR ? 4.0 * n * 2.220446049250313e-16 : -1e-12 / std::max(n, 1);
}

// Unpivoted Cholesky of the symmetric positive definite n x n matrix whose lower triangle sits in V (leading dimension n; destroyed)
// by the wide-panel solver of the dense reduced systems (dense_chol.h: k_wchol_diag16 + k_wchol_step, one launch per 96 columns), the
// vector y riding along as its right-hand side (-> L^-1 y). Lx (n x n): the panels; Ltw: ceil(n / 96) * (WD_LT + 6 * 256) doubles for
// the diagonal blocks' tiles and their re-inverted diagonal tiles; A0 (leading dimension ld0): the original matrix, for the pivot test
// (k_wfac_diag). Returns 1 = every pivot is safely positive (the factor is in Lx / Ltw, packed by k_wfac_pack), 0 = not (the caller
// takes the rank-revealing route), -1 = HIP error. One host synchronisation.
int run_wfac(sadvio_ba_handle* h, double* V, int n, double* y, double* Lx, double* Ltw, const double* A0, long long ld0, double tau_rel, double* dmax, int* info) {
    const int nsteps = (n + WD - 1) / WD;
    double* Ld = Ltw + (size_t)nsteps * WD_LT;
    if (hipMemsetAsync(info, 0, sizeof(int) * 2, h->stream) != hipSuccess) return -1;
    const size_t lds_st = sizeof(double) * wdstep_lds_doubles() + 64;
    (void)hipFuncSetAttribute((const void*)k_wchol_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_st);
    (void)hipFuncSetAttribute((const void*)k_wchol_diag16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * wd16_lds_doubles() + 64));
    hipLaunchKernelGGL(k_wchol_diag16, dim3(1), dim3(SOLVE_THREADS), sizeof(double) * wd16_lds_doubles() + 64, h->stream, V, (long long)n, y, (double*)nullptr, n, 0, info, (const int*)nullptr, Ltw);
    for (int st = 0; st + 1 < nsteps; st++) {
        const int c0 = st * WD, mrem = n - (c0 + WD);
        const int nt = (mrem + CH_TS - 1) / CH_TS;
        hipLaunchKernelGGL(k_wchol_step, dim3(nt * (nt + 1) / 2 + 2), dim3(SOLVE_THREADS), lds_st, h->stream, V, (long long)n, Lx, y, Ltw + (size_t)st * WD_LT, Ltw + (size_t)(st + 1) * WD_LT,
                           (double*)nullptr, (double*)nullptr, n, c0, info, (const int*)nullptr, (long long*)nullptr);
    }
    hipLaunchKernelGGL(k_diag_max, dim3(1), dim3(256), 0, h->stream, A0, ld0, n, dmax);
    hipLaunchKernelGGL(k_wfac_diag, dim3(nsteps * WD_T), dim3(64), 0, h->stream, Ltw, n, A0, ld0, tau_rel, dmax, 1024.0 * n * 2.220446049250313e-16, Ld, info + 1);
    int st2[2] = {0, 0};
    if (hipMemcpyAsync(st2, info, sizeof(st2), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    return st2[0] == 0 && st2[1] == 0 ? 1 : 0;
}
size_t wfac_scratch_doubles(int n) { return (size_t)((n + WD - 1) / WD) * (WD_LT + WD_T * 256) + 8; }

// one-sided Jacobi eigen-decomposition of the symmetric n x n block at A (leading dimension lda): G, V (n x n each)
// and ev (n) are device buffers; returns the number of sweeps (negative = HIP error)
int run_jacobi(sadvio_ba_handle* h, const double* A, long long lda, int n, int lower_only, double* G, double* V, double* ev, int* flag, int eig_cut_mode) {
    const long long nn = (long long)n * n;
    // Cholesky-preconditioned block Jacobi (marg_kernels.h): sym(A) -> V (scratch), pivoted Cholesky V -> G = L^T, block
    // one-sided Jacobi sweeps on the rows of G, then eigen-pairs from the rows -> V, ev
    if (n >= 32 && n <= PCH_MAXN && !h->env.jacobi_plain) {
        hipLaunchKernelGGL(k_jacobi_init, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, h->stream, A, lda, n, V, G, lower_only);
        const int r = run_pchol(h, V, n, G, pchol_tau(n, eig_cut_mode));
        if (r < 0) return -1;
        const int sweeps = run_jacobi_rows(h, G, r, n, flag);
        if (sweeps < 0) return -1;
        const bool swap_pchol = h->env.pchol_swap;
        hipLaunchKernelGGL(k_eig_from_rows, dim3(n), dim3(JAC_THREADS), 0, h->stream, G, swap_pchol ? h->d_jac_ints.p : (const int*)nullptr, h->d_jac_ints.p + n, n, V, ev);
        return sweeps;
    }
    hipLaunchKernelGGL(k_jacobi_init, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, h->stream, A, lda, n, G, V, lower_only);
    const int npad = n + (n & 1);
    // noise floor of the column norms (see k_jacobi_floor); flag[2..3] = max norm bits, flag[4..5] = the floor
    unsigned long long* amax = (unsigned long long*)(flag + 2);
    double* floor2 = (double*)(flag + 4);
    if (hipMemsetAsync(flag, 0, 8 * sizeof(int), h->stream) != hipSuccess) return -1;
    if (n > 0 && eig_cut_mode == SADVIO_EIG_CUT_NOISE_FLOOR) {   // the reference's cut sits below that floor: every pair keeps rotating
        hipLaunchKernelGGL(k_jacobi_floor, dim3(n), dim3(JAC_THREADS), 0, h->stream, G, n, amax, floor2, 0);
        hipLaunchKernelGGL(k_jacobi_floor, dim3(1), dim3(JAC_THREADS), 0, h->stream, G, n, amax, floor2, 1);
    }
    int sweeps = 0;
    for (; sweeps < 40 && npad >= 2; sweeps++) {
        if (hipMemsetAsync(flag, 0, sizeof(int), h->stream) != hipSuccess) return -1;
        for (int s = 0; s < npad - 1; s++)
            hipLaunchKernelGGL(k_jacobi_step, dim3(npad / 2), dim3(JAC_THREADS), 0, h->stream, G, V, n, npad, s, 1e-14, flag, floor2);
        int f = 0;
        if (hipMemcpyAsync(&f, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
        if ((h->env.debug & 16384)) fprintf(stderr, "[sadvio dbg] jacobi n %d sweep %d rotations %d\n", n, sweeps, f);
        if (!f) { sweeps++; break; }
    }
    hipLaunchKernelGGL(k_jacobi_eigenvalues, dim3(n), dim3(JAC_THREADS), 0, h->stream, G, V, n, ev);
    return sweeps;
}

// The reference cuts by EIGENVALUE (lambda > 1e-12, marginalization.cpp:318-342, marginalization.hpp:58); the rank-revealing Cholesky
// cuts by pivot, at the floor 1e-12 / n that never drops an eigenvalue above the cut — and therefore keeps directions whose
// eigenvalue lies below it (lambda_min <= the last pivot d <= ~n lambda_min). For the trailing pivots inside that band the small
// eigenvalues are evaluated the way the eigenvalue test means them: with G in pivot order (upper triangular) and x_s = G^-1 e_s for the
// k trailing steps, the k smallest eigenvalues of A = G^T G are, to O(d / gap) relative, the reciprocals of the eigenvalues of X^T X
// (A^-1 = G^-1 G^-T is dominated by those columns; k = 1: lambda = d / (1 + |w|^2), the Rayleigh quotient of the near-null vector) —
// relatively accurate where an eigen-decomposition of A in double precision only returns noise of size eps |A|. Rows whose eigenvalue
// is <= 1e-12 are dropped from the end. The solves run on the device (k_rank_backsub), the k x k eigenproblem on the host, on guarded calls only (a trailing pivot below
// RANK_GUARD x 1e-12: one call in 25 in the sliding sequences). Returns the refined rank, -1 on a HIP error.
constexpr double RANK_GUARD = 1e4;
int refine_rank_by_eigenvalue(sadvio_ba_handle* h, const double* G, int n1, int nf) {
    constexpr int KMAX = 8;              // trailing pivots looked at
    const int nl = std::min(nf, KMAX);
    std::vector<int> step_of(n1);
    if (hipMemcpyAsync(step_of.data(), h->d_jac_ints.p, sizeof(int) * (size_t)n1, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
    std::vector<double> last((size_t)nl * n1);      // the last rank rows: their diagonal entries are the trailing pivots
    if (hipMemcpyAsync(last.data(), G + (size_t)(nf - nl) * n1, sizeof(double) * (size_t)nl * n1, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    std::vector<int> col_of(nf, -1);                  // pivot column of every step
    for (int c = 0; c < n1; c++) if (step_of[c] >= 0 && step_of[c] < nf) col_of[step_of[c]] = c;
    for (int s = 0; s < nf; s++) if (col_of[s] < 0) return nf;    // (cannot happen: every step has its column)
    auto gd = [&](int s) { return last[(size_t)(s - (nf - nl)) * n1 + col_of[s]]; };   // diagonal of the factor in pivot order, s >= nf - nl
    const double d_last = gd(nf - 1) * gd(nf - 1);
    if (h->env.debug) fprintf(stderr, "[sadvio dbg] rank refinement: last pivot %.3e (rank %d of %d)\n", d_last, nf, n1 - 1);
    if (!(d_last <= RANK_GUARD * 1e-12)) return nf;
    int k = 0;
    while (k < nl) { const double dv = gd(nf - 1 - k); if (dv * dv <= RANK_GUARD * 1e-12) k++; else break; }
    // guarded: x_q = G^-1 e_s, s = nf - 1 - q, by k_rank_backsub on the device (one workgroup per vector; 0.29 ms with the read-back at n = 915, k = 4)
    const auto tq0 = std::chrono::steady_clock::now();
    if (h->d_rank_col.alloc((size_t)nf) != hipSuccess || h->d_rank_x.alloc((size_t)k * nf) != hipSuccess) return -1;
    if (hipMemcpyAsync(h->d_rank_col.p, col_of.data(), sizeof(int) * (size_t)nf, hipMemcpyHostToDevice, h->stream) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_rank_backsub, dim3(k), dim3(RB_THREADS), 0, h->stream, G, n1, h->d_rank_col.p, nf - 1, nf, h->d_rank_x.p);   // one workgroup per vector
    std::vector<double> Xall((size_t)k * nf);
    if (hipMemcpyAsync(Xall.data(), h->d_rank_x.p, sizeof(double) * (size_t)k * nf, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    std::vector<std::vector<double>> X(k, std::vector<double>(nf, 0.0));
    for (int q = 0; q < k; q++) std::copy(Xall.begin() + (size_t)q * nf, Xall.begin() + (size_t)q * nf + (nf - q), X[q].begin());   // rows above s = nf - 1 - q are not written: x is zero there
    const auto tq1 = std::chrono::steady_clock::now();
    // eigenvalues of the k x k Gram matrix X^T X (cyclic Jacobi); lambda_small(A) = 1 / them
    std::vector<double> B((size_t)k * k);
    for (int a = 0; a < k; a++) for (int b = 0; b < k; b++) { double acc = 0.0; for (int i = 0; i < nf; i++) acc += X[a][i] * X[b][i]; B[(size_t)a * k + b] = acc; }
    for (int sweep = 0; sweep < 30 && k > 1; sweep++) {
        double off = 0.0;
        for (int a = 0; a < k; a++) for (int b = a + 1; b < k; b++) {
            const double apq = B[(size_t)a * k + b];
            off += apq * apq;
            if (apq == 0.0) continue;
            const double th = (B[(size_t)b * k + b] - B[(size_t)a * k + a]) / (2.0 * apq);
            const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0)), c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int i = 0; i < k; i++) { const double u = B[(size_t)i * k + a], v = B[(size_t)i * k + b]; B[(size_t)i * k + a] = c * u - sn * v; B[(size_t)i * k + b] = sn * u + c * v; }
            for (int i = 0; i < k; i++) { const double u = B[(size_t)a * k + i], v = B[(size_t)b * k + i]; B[(size_t)a * k + i] = c * u - sn * v; B[(size_t)b * k + i] = sn * u + c * v; }
        }
        if (off <= 1e-30 * B[0] * B[0]) break;
    }
    int drop = 0;
    for (int a = 0; a < k; a++) if (!(1.0 / B[(size_t)a * k + a] > 1e-12)) drop++;
    if (h->env.debug) fprintf(stderr, "[sadvio dbg] rank refinement: %d vector(s), device back-substitution + read-back %.3f ms, host %.3f ms\n", k, std::chrono::duration<double, std::milli>(tq1 - tq0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq1).count());
    if ((h->env.debug & 16384) || drop) {
        h->marg_stats[3] += drop ? 1 : 0;
        if (h->env.debug) {
            fprintf(stderr, "[sadvio dbg] rank refinement: %d trailing pivot(s) below %.0e, eigenvalue estimates", k, RANK_GUARD * 1e-12);
            for (int a = 0; a < k; a++) fprintf(stderr, " %.3e", 1.0 / B[(size_t)a * k + a]);
            fprintf(stderr, " -> %d dropped (rank %d of %d)\n", drop, nf - drop, n1 - 1);
        }
    }
    if (drop) {   // the packing kernels read the rank from the device (k_marg_pack_chol)
        const int r2 = nf - drop;
        if (hipMemcpy(h->d_jac_ints.p + n1, &r2, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
    }
    return nf - drop;
}

// eigenvalue cut of the pseudo-inverse / rank-revealing decomposition (sadvio_ba.h: SADVIO_EIG_CUT_*): the reference's absolute
// 1e-12 (marginalization.hpp:58, applied at marginalization.cpp:237,322), or that constant with the rounding-noise floor
// n eps lambda_max (see oracle/marg.c)
double marg_cut(const std::vector<double>& ev, int eig_cut_mode) {
    if (eig_cut_mode != SADVIO_EIG_CUT_NOISE_FLOOR) return 1e-12;
    double mx = 0.0;
    for (double v : ev) mx = std::max(mx, std::fabs(v));
    return std::max(1e-12, (double)ev.size() * 2.220446049250313e-16 * mx);
}

// exp_so3((a, b, 0)) row-major (geometry.h:131-147: first order below 1e-9)
void host_exp_so3(double a, double b, double* R) {
    const double th = std::sqrt(a * a + b * b);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (th < 1e-9) { const double S[9] = {0, 0, b, 0, 0, -a, -b, a, 0}; for (int i = 0; i < 9; i++) R[i] = I[i] + S[i]; return; }
    const double x = a / th, y = b / th;
    const double S[9] = {0, 0, y, 0, 0, -x, -y, x, 0};
    double S2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S2[3 * i + j] = S[3 * i] * S[j] + S[3 * i + 1] * S[3 + j] + S[3 * i + 2] * S[6 + j];
    for (int i = 0; i < 9; i++) R[i] = I[i] + (1.0 - std::cos(th)) * S2[i] + std::sin(th) * S[i];
}
bool make_imu_dev(const sadvio_imu_factor& f, int kf_base, ImuDev& o) {
    o.kf_i = kf_base + f.kf_i; o.kf_j = kf_base + f.kf_j; o.dt = f.dt;
    memcpy(o.dR, f.delta_R, sizeof(o.dR)); memcpy(o.dv, f.delta_v, sizeof(o.dv)); memcpy(o.dp, f.delta_p, sizeof(o.dp));
    memcpy(o.J_dR_bg, f.J_dR_bg, 72); memcpy(o.J_dv_ba, f.J_dv_ba, 72); memcpy(o.J_dv_bg, f.J_dv_bg, 72);
    memcpy(o.J_dp_ba, f.J_dp_ba, 72); memcpy(o.J_dp_bg, f.J_dp_bg, 72);
    if (!imu_sqrt_information(f.cov, o.W)) return false;
    o.sa = 1.0 / sqrt(f.dt * f.bacc_noise * f.bacc_noise);
    o.sg = 1.0 / sqrt(f.dt * f.bgyr_noise * f.bgyr_noise);
    o.win = 0; o.pad = 0;
    return true;
}

inline void launch_mgemm(sadvio_ba_handle* h, double* C, long long ldc, const double* A, long long sai, long long sak, const double* B, long long sbk,
                         long long sbj, int M, int N, int K, double alpha, double beta) {
    hipLaunchKernelGGL(k_mgemm, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, h->stream, C, ldc, A, sai, sak, B, sbk, sbj, M, N, K, alpha, beta);
}
}  // namespace

namespace { int prior_build_Z(sadvio_ba_handle* h, const double* J, int nf, int n, int form, const int* step_of, double* Z, int cut_mode, double* trace_out = nullptr, bool guard = false); }

int sadvio_ba_marginalize(sadvio_ba_handle* h, int32_t w, const sadvio_marg_request* rq, sadvio_marg_result* res, int32_t* lmk_col_out,
                          double* J_out, double* r0_out) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "marginalize before set_windows"; return SADVIO_E_STATE; }
    if (h->defer) { h->err = "marginalize between begin_update and commit_update"; return SADVIO_E_STATE; }
    if (!rq || w < 0 || w >= (int)h->wins.size()) { h->err = "marginalize: bad argument"; return SADVIO_E_INVALID_ARG; }
    if (h->world > 1) { h->err = "marginalize: the window is sharded over several GPUs (each rank holds a landmark partition only)"; return SADVIO_E_INVALID_ARG; }
    const WinDev& d = h->wins[w].d;
    if (rq->kf_marg < 0 || rq->kf_marg >= d.n_kf || rq->kf_keep >= d.n_kf || rq->n_marg < 0 || rq->n_keep < 0 || rq->n_prior < 0 ||
        rq->n_prior > 4 || (rq->n_marg > 0 && !rq->lmk_marg) || (rq->n_keep > 0 && !rq->lmk_keep) || (rq->n_prior > 0 && !rq->priors) ||
        (rq->eig_cut_mode != SADVIO_EIG_CUT_REFERENCE && rq->eig_cut_mode != SADVIO_EIG_CUT_NOISE_FLOOR) ||
        (rq->prior_form != SADVIO_PRIOR_FORM_EIGEN && rq->prior_form != SADVIO_PRIOR_FORM_CHOLESKY)) {
        h->err = "marginalize: request out of range"; return SADVIO_E_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    MargScratch& M = h->mg;
    PriorState& PR = h->prior;
    // index layout, marginalization.cpp:38-113
    const int m = 6 + (rq->marg_has_imu ? 9 : 0) + 3 * rq->n_marg;
    const int n = (rq->kf_keep >= 0 ? 15 : 0) + 3 * rq->n_keep;
    const int N = m + n;
    std::vector<int>& lcol = M.lcol;
    lcol.assign(std::max(d.n_lmk, 1), -1);
    int idx = 6 + (rq->marg_has_imu ? 9 : 0);
    for (int k = 0; k < rq->n_marg; k++) {
        if (rq->lmk_marg[k] < 0 || rq->lmk_marg[k] >= d.n_lmk) { h->err = "marginalize: landmark index out of range"; return SADVIO_E_INVALID_ARG; }
        lcol[rq->lmk_marg[k]] = idx; idx += 3;
    }
    int kf_keep_col = -1;
    if (rq->kf_keep >= 0) { kf_keep_col = idx; idx += 15; }
    for (int k = 0; k < rq->n_keep; k++) {
        if (rq->lmk_keep[k] < 0 || rq->lmk_keep[k] >= d.n_lmk) { h->err = "marginalize: landmark index out of range"; return SADVIO_E_INVALID_ARG; }
        lcol[rq->lmk_keep[k]] = idx; idx += 3;
    }
    if (res) { res->m = m; res->n = n; res->n_full = 0; res->kf_col = kf_keep_col >= 0 ? kf_keep_col - m : -1; res->sweeps_mm = res->sweeps_k = 0; }
    if (lmk_col_out) for (int k = 0; k < rq->n_keep; k++) lmk_col_out[k] = lcol[rq->lmk_keep[k]] - m;
    if (n < 4) {   // the reference clears its prior state too (…Analytic.cpp:620-625)
        PR.valid = false; PR.z_valid = false; PR.hg_valid = false; PR.serial++;
        h->err = "marginalize: fewer than 4 kept columns, refused (marginalization.cpp:215-216)"; return SADVIO_E_REFUSED;
    }
    const bool chol_form = rq->prior_form == SADVIO_PRIOR_FORM_CHOLESKY && n + 1 <= PCH_MAXN;
    if (rq->prior_form == SADVIO_PRIOR_FORM_CHOLESKY && !chol_form) { h->err = "marginalize: the Cholesky form handles n < 2048"; return SADVIO_E_INVALID_ARG; }

    SolveOpts so{};
    DevPtrs P = make_ptrs(h, so, 2);
    const int big = std::max(m, n + 1);
    HIP_TRY(M.A.alloc((size_t)N * N)); HIP_TRY(M.b.alloc(N)); HIP_TRY(M.flag.alloc(8));
    HIP_TRY(M.G.alloc((size_t)big * big)); HIP_TRY(M.V.alloc((size_t)big * big)); HIP_TRY(M.ev.alloc(big)); HIP_TRY(M.Vs.alloc((size_t)big * big));
    HIP_TRY(M.Ainv.alloc((size_t)m * m)); HIP_TRY(M.T.alloc((size_t)n * m)); HIP_TRY(M.Ak.alloc((size_t)n * n)); HIP_TRY(M.bk.alloc(n));
    HIP_TRY(M.newJ.alloc((size_t)n * n)); HIP_TRY(M.newr.alloc(n));
    HIP_TRY(hipMemsetAsync(M.A.p, 0, sizeof(double) * (size_t)N * N, h->stream));
    HIP_TRY(hipMemsetAsync(M.b.p, 0, sizeof(double) * N, h->stream));
    // ---- host-side lists of the blocks, ONE staged upload -----------------------------------------------------------
    // reprojection factors of kept then marginalised landmarks seen from frame0
    std::vector<int>& it2 = M.items; std::vector<int>& itl = M.items_l;
    it2.clear(); itl.clear();
    for (int pass = 0; pass < 2; pass++) {
        const int cnt = pass == 0 ? rq->n_keep : rq->n_marg;
        const int32_t* list = pass == 0 ? rq->lmk_keep : rq->lmk_marg;
        for (int k = 0; k < cnt; k++) {
            const int gl = d.lmk_base + list[k];
            for (int o = h->h_lmk_ob[gl]; o < h->h_lmk_oe[gl]; o++)
                if (h->h_obs_kf[o] == d.kf_base + rq->kf_marg && h->obs_perm[o] >= 0) { it2.push_back(o); it2.push_back(lcol[list[k]]); itl.push_back(gl); }  // pseudo-observations excluded
        }
    }
    const int n_items = (int)itl.size();
    it2.insert(it2.end(), itl.begin(), itl.end());
    // IMU factor + bias factor, pose priors
    MargSmall S{};
    if (rq->imu && rq->kf_keep >= 0 && rq->marg_has_imu) {
        sadvio_imu_factor f = *rq->imu;
        f.kf_i = rq->kf_marg; f.kf_j = rq->kf_keep;
        if (!make_imu_dev(f, d.kf_base, S.imu)) { h->err = "marginalize: IMU covariance is not positive definite"; return SADVIO_E_INVALID_ARG; }
        S.has_imu = 1; S.kf_i = d.kf_base + rq->kf_marg; S.kf_j = d.kf_base + rq->kf_keep; S.kf_keep_col = kf_keep_col;
    }
    for (int k = 0; k < rq->n_prior; k++) {
        const sadvio_pose_prior& pr = rq->priors[k];
        const int base = pr.kf == rq->kf_marg ? 0 : (pr.kf == rq->kf_keep ? kf_keep_col : -1);
        if (base < 0) continue;
        const int q = S.n_prior++;
        S.prior_kf[q] = d.kf_base + pr.kf; S.prior_base[q] = base;
        memcpy(S.prior_T[q], pr.T_prior, sizeof(S.prior_T[q])); memcpy(S.prior_inf[q], pr.inf_diag, sizeof(S.prior_inf[q]));
    }
    // previous prior at zero deltas: the handle's (no upload) or the caller's arrays
    const double* lastJ = nullptr; const double* lastr = nullptr;
    int nl = 0, nfl = 0;
    if (rq->last_n_full == SADVIO_PRIOR_RESIDENT) {
        if (!PR.valid) { h->err = "marginalize: last_n_full = SADVIO_PRIOR_RESIDENT but the handle holds no prior"; return SADVIO_E_STATE; }
        nl = PR.n; nfl = PR.n_full; lastJ = PR.J.p; lastr = PR.r0.p;
    } else if (rq->last_n_full > 0) {
        nl = rq->last_n; nfl = rq->last_n_full;
        if (!rq->last_J || !rq->last_r0 || nl <= 0) { h->err = "marginalize: previous prior arrays missing"; return SADVIO_E_INVALID_ARG; }
    } else if (rq->last_n_full < 0) { h->err = "marginalize: last_n_full < 0"; return SADVIO_E_INVALID_ARG; }
    std::vector<int>& col = M.col;
    if (nfl > 0) {
        if (rq->last_n_keep > 0 && (!rq->last_lmk_index || !rq->last_lmk_col)) { h->err = "marginalize: previous prior landmark lists missing"; return SADVIO_E_INVALID_ARG; }
        if (rq->last_kf >= 0 && rq->last_kf_col < 0) { h->err = "marginalize: last_kf_col < 0"; return SADVIO_E_INVALID_ARG; }
        col.assign(nl, -1);
        if (rq->last_kf >= 0) {
            const int base = (rq->last_kf == rq->kf_marg) ? 0 : ((rq->last_kf == rq->kf_keep) ? kf_keep_col : -1);
            const int width = (rq->last_kf == rq->kf_marg) ? (rq->marg_has_imu ? 15 : 6) : 15;
            if (base >= 0) for (int a = 0; a < width && rq->last_kf_col + a < nl; a++) col[rq->last_kf_col + a] = base + a;
        }
        for (int k = 0; k < rq->last_n_keep; k++) {
            if (rq->last_lmk_col[k] < 0) continue;
            if (rq->last_lmk_col[k] + 3 > nl) { h->err = "marginalize: last_lmk_col exceeds the previous prior's columns"; return SADVIO_E_INVALID_ARG; }
            const int li = rq->last_lmk_index[k];
            if (li < 0 || li >= d.n_lmk) continue;
            const int lc = lcol[li];
            if (lc < 0) continue;
            for (int a = 0; a < 3; a++) col[rq->last_lmk_col[k] + a] = lc + a;
        }
        HIP_TRY(M.lastcol.alloc(nl));
        h->up.add(M.lastcol.p, col.data(), sizeof(int) * (size_t)nl);
        if (!lastJ) {
            HIP_TRY(M.lastJ.alloc((size_t)nfl * nl)); HIP_TRY(M.lastr.alloc(nfl));
            h->up.add(M.lastJ.p, rq->last_J, sizeof(double) * (size_t)nfl * nl);
            h->up.add(M.lastr.p, rq->last_r0, sizeof(double) * (size_t)nfl);
            lastJ = M.lastJ.p; lastr = M.lastr.p;
        }
    }
    if (n_items > 0) { HIP_TRY(M.ditems.alloc(it2.size())); h->up.add(M.ditems.p, it2.data(), it2.size() * sizeof(int)); }
    if (S.has_imu || S.n_prior) { HIP_TRY(M.small.alloc(1)); h->up.add(M.small.p, &S, sizeof(S)); }
    HIP_TRY(h->up.flush(h->stream));
    if (n_items > 0) {
        auto ko = h->factor_type == SADVIO_FACTOR_PIXEL ? k_marg_obs<0> : k_marg_obs<1>;
        hipLaunchKernelGGL(ko, dim3((n_items + 127) / 128), dim3(128), 0, h->stream, P, M.ditems.p, n_items, M.A.p, M.b.p, N);
    }
    if (S.has_imu || S.n_prior) hipLaunchKernelGGL(k_marg_small, dim3(1), dim3(64), 0, h->stream, P, M.small.p, M.A.p, M.b.p, N);
    if (nfl > 0) {
        const long long items = (long long)nl * nl;
        if (rq->last_n_full == SADVIO_PRIOR_RESIDENT && PR.hg_valid && PR.n == nl && !h->env.marg_last_small) {
            // the resident prior still carries the Ak / bk it was factorised from: J^T J and J^T r0 without touching J
            hipLaunchKernelGGL(k_marg_last_scatter_h, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, h->stream, PR.H.p, PR.g.p, M.lastcol.p, nl, M.A.p, M.b.p, N);
        } else if (nl >= 64 && !h->env.marg_last_small) {
            HIP_TRY(M.Hl.alloc((size_t)nl * nl));
            launch_mgemm(h, M.Hl.p, nl, lastJ, 1LL, (long long)nl, lastJ, (long long)nl, 1LL, nl, nl, nfl, 1.0, 0.0);      // H = J^T J
            hipLaunchKernelGGL(k_marg_last_scatter, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, h->stream, M.Hl.p, lastJ, lastr, M.lastcol.p, nfl, nl, M.A.p, M.b.p, N);
        } else
        hipLaunchKernelGGL(k_marg_last_prior, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, h->stream, lastJ, lastr, M.lastcol.p, nfl, nl, M.A.p, M.b.p, N);
    }
    // ---- Schur complement with the pseudo-inverse of Amm (marginalization.cpp:234-248) ----------------------------------
    // A well-conditioned Amm (every pivot of the rank-revealing Cholesky above the noise floor: the usual case — frame0's states and
    // its lonely stereo landmarks are fully observed) has Amm^+ = Amm^-1 under either cut, and the inverse comes from the triangular
    // inverse of that factor (Z^T Z, k_tri_*): 0.3 ms instead of ~1.1 ms of Jacobi launches. Anything else takes the reference's
    // eigen-decomposition with the request's cut.
    std::vector<double>& hev = M.hev;
    hev.resize(std::max(m, n));
    bool mm_by_cholesky = false;
    HIP_TRY(M.wtmp.alloc((size_t)big + 16));
    // ... but only where the inverse itself PROVES that no eigenvalue of Amm lies below the reference's cut (round 6). Pivots bound
    // eigenvalues from above only: behind a rank-deficient previous prior frame0's block can carry an eigenvalue below 1e-12 under pivots
    // that all pass, and where the reference's pseudo-inverse zeroes that direction (marginalization.cpp:234-240) an inverse divides by it
    // — step 13 of the config-3-size dense sequence (previous prior 917 of 918): Ak lost 9 187 of information along the kept frame's
    // rotation, 1e-6 in the next solve's poses (tests/test_gpu_sliding_full_size.py; found by marginalising the device's own window
    // with the oracle, scripts/sliding_same_inputs_marg.py). The proof: lambda_min(Amm) >= 1 / trace(Amm^-1), and trace(Amm^-1) =
    // |Z|_F^2 of the triangular inverse the route forms anyway (m row norms, one read-back; 2.3e13 at that step, 1e3 .. 5e11 at the other
    // 24). A trace of 1e12 or more sends the call to the reference's eigen-decomposition with the request's cut. (Gating on the previous
    // prior's rank instead was measured too: it sends full-rank blocks through the eigen route as well, whose inverse agrees with the
    // oracle's to 4e-9 where the Cholesky inverse agrees to 1e-11 — the sequence's worst step 3e-7 instead of 1.3e-8.)
    bool mm_force_eig = false;
    double mm_trace = 0.0;
    auto mm_inverse_bounded = [&]() -> int {    // 1 = every eigenvalue of Amm above the reference's cut, 0 = not shown (NaN / inf included)
        return (rq->eig_cut_mode != SADVIO_EIG_CUT_REFERENCE || mm_trace < 1e12) ? 1 : 0;
    };
    if (m > 0 && !h->env.marg_eig_mm && !h->env.marg_pivoted && rq->eig_cut_mode == SADVIO_EIG_CUT_REFERENCE) {
        // Amm is positive definite whenever frame0 carries a prior or enough observations: unpivoted wide-panel factor first (run_wfac)
        const long long mm2 = (long long)m * m;
        HIP_TRY(M.Vs.alloc(std::max(wfac_scratch_doubles(m), (size_t)big * big)));
        hipLaunchKernelGGL(k_jacobi_init, dim3((unsigned)((mm2 + 255) / 256)), dim3(256), 0, h->stream, M.A.p, (long long)N, m, M.V.p, M.G.p, 0);
        HIP_TRY(hipMemsetAsync(M.wtmp.p + 8, 0, sizeof(double) * (size_t)m, h->stream));
        const int okf = run_wfac(h, M.V.p, m, M.wtmp.p + 8, M.Ainv.p, M.Vs.p, M.A.p, (long long)N, pchol_tau(m, SADVIO_EIG_CUT_REFERENCE), M.wtmp.p, M.flag.p);
        if (okf < 0) { h->err = "marginalize: HIP error in the unpivoted Cholesky"; return SADVIO_E_HIP; }
        if (okf == 1) {
            hipLaunchKernelGGL(k_wfac_pack, dim3((unsigned)((mm2 + 255) / 256)), dim3(256), 0, h->stream, M.Ainv.p, (long long)m, M.Vs.p, M.Vs.p + (size_t)((m + WD - 1) / WD) * WD_LT,
                               M.wtmp.p + 8, m, M.G.p, M.wtmp.p + 8);
            HIP_TRY(M.piv_mm.alloc(m));
            hipLaunchKernelGGL(k_iota, dim3((m + 255) / 256), dim3(256), 0, h->stream, M.piv_mm.p, m);
            const int rc = prior_build_Z(h, M.G.p, m, m, SADVIO_PRIOR_FORM_CHOLESKY, M.piv_mm.p, M.Vs.p, rq->eig_cut_mode, rq->eig_cut_mode == SADVIO_EIG_CUT_REFERENCE ? &mm_trace : nullptr);
            if (rc != SADVIO_OK) return rc;
            const int ok = mm_inverse_bounded();
            mm_by_cholesky = ok == 1;
            mm_force_eig = ok == 0;
        }
    }
    if (!mm_by_cholesky && !mm_force_eig && m >= 32 && m <= PCH_MAXN && !h->env.marg_eig_mm) {
        const long long mm2 = (long long)m * m;
        hipLaunchKernelGGL(k_jacobi_init, dim3((unsigned)((mm2 + 255) / 256)), dim3(256), 0, h->stream, M.A.p, (long long)N, m, M.V.p, M.G.p, 0);
        const int r = run_pchol(h, M.V.p, m, M.G.p, pchol_tau(m, SADVIO_EIG_CUT_NOISE_FLOOR), false);
        if (r < 0) { h->err = "marginalize: HIP error in the pivoted Cholesky"; return SADVIO_E_HIP; }
        if (r == m) {
            HIP_TRY(M.piv_mm.alloc(m));
            HIP_TRY(hipMemcpyAsync(M.piv_mm.p, h->d_jac_ints.p, sizeof(int) * (size_t)m, hipMemcpyDeviceToDevice, h->stream));
            const int rc = prior_build_Z(h, M.G.p, m, m, SADVIO_PRIOR_FORM_CHOLESKY, M.piv_mm.p, M.Vs.p, rq->eig_cut_mode, rq->eig_cut_mode == SADVIO_EIG_CUT_REFERENCE ? &mm_trace : nullptr);
            if (rc != SADVIO_OK) return rc;
            const int ok = mm_inverse_bounded();
            mm_by_cholesky = ok == 1;
        }
    }
    int sw = 0;
    if (!mm_by_cholesky) {
        sw = run_jacobi(h, M.A.p, N, m, 0, M.G.p, M.V.p, M.ev.p, M.flag.p, rq->eig_cut_mode);
        if (sw < 0) { h->err = "marginalize: HIP error in the eigen-solver"; return SADVIO_E_HIP; }
        HIP_TRY(hipMemcpyAsync(hev.data(), M.ev.p, sizeof(double) * m, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (res) res->sweeps_mm = sw;
    {
        std::vector<double> sel(m);
        if (!mm_by_cholesky) {
            hev.resize(m);
            const double cut = marg_cut(hev, rq->eig_cut_mode);
            for (int i = 0; i < m; i++) sel[i] = hev[i] > cut ? 1.0 / sqrt(hev[i]) : 0.0;
            HIP_TRY(hipMemcpyAsync(M.ev.p, sel.data(), sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
            const long long mm = (long long)m * m;
            hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, h->stream, M.V.p, M.ev.p, m, M.Vs.p);
        }
        // Ainv = Vs^T Vs (Vs = Lambda^-1/2 U^T, or Z = G^-T of the Cholesky route) ; T = Arm Ainv ; Ak = Arr - T Arm^T ; bk = brr - T bmm   (FP64 matrix cores)
        launch_mgemm(h, M.Ainv.p, m, M.Vs.p, 1LL, (long long)m, M.Vs.p, (long long)m, 1LL, m, m, m, 1.0, 0.0);
        launch_mgemm(h, M.T.p, m, M.A.p + (size_t)m * N, (long long)N, 1LL, M.Ainv.p, (long long)m, 1LL, n, m, m, 1.0, 0.0);
        HIP_TRY(hipMemcpy2DAsync(M.Ak.p, sizeof(double) * n, M.A.p + (size_t)m * N + m, sizeof(double) * N, sizeof(double) * n, n, hipMemcpyDeviceToDevice, h->stream));
        launch_mgemm(h, M.Ak.p, n, M.T.p, (long long)m, 1LL, M.A.p + (size_t)m * N, 1LL, (long long)N, n, n, m, -1.0, 1.0);
        hipLaunchKernelGGL(k_marg_bk, dim3((n + 127) / 128), dim3(128), 0, h->stream, M.T.p, M.b.p, n, m, M.bk.p);
        if (!mm_by_cholesky) HIP_TRY(hipStreamSynchronize(h->stream));   // `sel` (uploaded above) goes out of scope
    }
    int nf = 0;
    bool unpivoted_ok = false;
    if (chol_form) {
        // ---- Cholesky form: J = G with G^T G = Ak (rank-revealing, pivots cut like the eigenvalues), r0 = -G^-T bk as the
        // factor's extra column. No eigen-decomposition.
        const int n1 = n + 1;
        const long long nn1 = (long long)n1 * n1;
        bool unpivoted = false;
        h->marg_stats[0]++;
        // (without an earlier prior frame1's velocity / bias directions are only held relative to frame0's: Ak is rank deficient, the
        // attempt would be wasted; SADVIO_MARG_UNPIVOTED=1 tries it regardless)
        // Only under the reference's absolute cut: an unpivoted factorisation is not rank revealing (the pivot of the last index of a
        // dependent set is lambda / v_i^2, v = the null vector - any size), so the noise-floor mode, whose point is a reliable
        // numerical rank, always takes the pivoted route; under the absolute 1e-12 cut both routes keep every direction whose pivot is
        // positive, as the reference's eigenvalue test does.
        // ... and only behind a previous prior of FULL rank (round 6): a prior that dropped a direction hands its near-null direction on to
        // the next Ak, where the unpivoted pivots do not show it (measured, step 13 of the VIO dense sliding sequence: eigenvalue 1.1e-14 —
        // below the reference's cut — under pivots that all pass; profiles/r06_rank_arbiter.txt).
        if (!h->env.marg_pivoted && rq->eig_cut_mode == SADVIO_EIG_CUT_REFERENCE && ((rq->last_n_full != 0 && nfl == nl) || h->env.marg_unpivoted)) {
            // A prior that carries an earlier prior is normally of full rank: then the factor needs no pivoting and the wide-panel
            // solver of the dense reduced systems (dense_chol.h: k_wchol_diag16 + k_wchol_step, one launch per 96 columns, bk riding
            // along as its right-hand side) delivers L and z = L^-1 bk in a third of the pivoted factorisation's time. Every pivot is
            // tested afterwards (k_wfac_diag); one that is not safely positive sends the call to the rank-revealing route below.
            HIP_TRY(M.Vs.alloc(std::max(wfac_scratch_doubles(n), (size_t)big * big)));
            double* Ltw = M.Vs.p; double* Ld = Ltw + (size_t)((n + WD - 1) / WD) * WD_LT;
            HIP_TRY(hipMemcpyAsync(M.V.p, M.Ak.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(hipMemcpyAsync(M.newr.p, M.bk.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
            const int okf = run_wfac(h, M.V.p, n, M.newr.p, M.G.p, Ltw, M.Ak.p, (long long)n, pchol_tau(n, rq->eig_cut_mode), M.wtmp.p, M.flag.p);
            if (okf < 0) { h->err = "marginalize: HIP error in the unpivoted Cholesky"; return SADVIO_E_HIP; }
            if (okf != 1) h->marg_stats[2]++;
            if (okf == 1) {
                unpivoted = true; unpivoted_ok = true;
                h->marg_stats[1]++;
                nf = n;
                hipLaunchKernelGGL(k_wfac_pack, dim3((unsigned)(((long long)n * n + 255) / 256)), dim3(256), 0, h->stream, M.G.p, (long long)n, Ltw, Ld, M.newr.p, n, M.newJ.p, M.newr.p);
                HIP_TRY(PR.step_of.alloc(n));
                hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, h->stream, PR.step_of.p, n);
            }
        }
        if (!unpivoted) {
        hipLaunchKernelGGL(k_marg_aug_init, dim3((unsigned)((nn1 + 255) / 256)), dim3(256), 0, h->stream, M.Ak.p, M.bk.p, n, M.V.p);
        nf = run_pchol(h, M.V.p, n1, M.G.p, pchol_tau(n, rq->eig_cut_mode), false);
        if (nf < 0) { h->err = "marginalize: HIP error in the pivoted Cholesky"; return SADVIO_E_HIP; }
        if (nf > n) nf = n;
        if (rq->eig_cut_mode == SADVIO_EIG_CUT_REFERENCE && nf > 0) {
            const int rc = refine_rank_by_eigenvalue(h, M.G.p, n1, nf);
            if (rc < 0) { h->err = "marginalize: HIP error in the rank refinement"; return SADVIO_E_HIP; }
            nf = rc;
        }
        }
        if (!unpivoted && nf > 0) {
            const long long cnt = (long long)nf * n1;
            hipLaunchKernelGGL(k_marg_pack_chol, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, M.G.p, n, h->d_jac_ints.p + n1, M.newJ.p, M.newr.p);
            HIP_TRY(PR.step_of.alloc(n));
            HIP_TRY(hipMemcpyAsync(PR.step_of.p, h->d_jac_ints.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
        }
    } else {
        // ---- rank-revealing decomposition of Ak (lower triangle, as Eigen reads it), marginalization.cpp:318-342
        sw = run_jacobi(h, M.Ak.p, n, n, 1, M.G.p, M.V.p, M.ev.p, M.flag.p, rq->eig_cut_mode);
        if (sw < 0) { h->err = "marginalize: HIP error in the eigen-solver"; return SADVIO_E_HIP; }
        if (res) res->sweeps_k = sw;
        hev.resize(n);
        HIP_TRY(hipMemcpyAsync(hev.data(), M.ev.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::vector<int> sel_rows;
        {
            const double cut = marg_cut(hev, rq->eig_cut_mode);
            // ascending eigenvalue order, like Eigen::SelfAdjointEigenSolver (row order of J only)
            std::vector<int> order(n);
            for (int i = 0; i < n; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return hev[a] < hev[b]; });
            for (int i : order) if (hev[i] > cut) sel_rows.push_back(i);
        }
        nf = (int)sel_rows.size();
        if (nf > 0) {
            HIP_TRY(M.sel.alloc(nf));
            HIP_TRY(hipMemcpyAsync(M.sel.p, sel_rows.data(), sizeof(int) * nf, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_marg_prior, dim3(nf), dim3(JAC_THREADS), 0, h->stream, M.V.p, M.ev.p, M.sel.p, nf, n, M.bk.p, M.newJ.p, M.newr.p);
            HIP_TRY(hipStreamSynchronize(h->stream));   // sel_rows goes out of scope
        }
    }
    if (res) res->n_full = nf;
    // the new prior becomes the handle's (AOptimizer.h:88-90: _marginalization_last), the old one's buffers become scratch
    PR.J.swap(M.newJ); PR.r0.swap(M.newr);
    PR.serial++;
    PR.hg_valid = false;
    if (unpivoted_ok) { PR.H.swap(M.Ak); PR.g.swap(M.bk); PR.hg_valid = true; }   // (full rank: J^T J = Ak, J^T r0 = -bk to rounding)
    PR.valid = nf > 0; PR.z_valid = false; PR.n_full = nf; PR.n = n; PR.form = chol_form ? SADVIO_PRIOR_FORM_CHOLESKY : SADVIO_PRIOR_FORM_EIGEN; PR.cut_mode = rq->eig_cut_mode;
    if (nf > 0) {
        if (J_out) HIP_TRY(hipMemcpyAsync(J_out, PR.J.p, sizeof(double) * (size_t)nf * n, hipMemcpyDeviceToHost, h->stream));
        if (r0_out) HIP_TRY(hipMemcpyAsync(r0_out, PR.r0.p, sizeof(double) * nf, hipMemcpyDeviceToHost, h->stream));
    }
    // the prior stays on the device and everything that reads it is stream-ordered behind this call: only a read-back has to wait
    // (asynchronous contract, sadvio_ba.h: a fault of the tail kernels surfaces in the next call that waits on this handle's stream;
    // SADVIO_DEBUG != 0 or cfg.profile_kernels wait here so that it is attributed to marginalize)
    if ((nf > 0 && (J_out || r0_out)) || h->env.debug || h->cfg.profile_kernels) HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
    return SADVIO_OK;
}

int sadvio_ba_marg_stats(sadvio_ba_handle* h, int32_t* calls, int32_t* unpivoted, int32_t* fell_back) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (calls) *calls = h->marg_stats[0];
    if (unpivoted) *unpivoted = h->marg_stats[1];
    if (fell_back) *fell_back = h->marg_stats[2];
    return SADVIO_OK;
}

int sadvio_ba_get_prior(sadvio_ba_handle* h, sadvio_prior_info* info, double* J, double* r0) {
    if (!h) return SADVIO_E_INVALID_ARG;
    const PriorState& PR = h->prior;
    if (info) { info->valid = PR.valid ? 1 : 0; info->n_full = PR.valid ? PR.n_full : 0; info->n = PR.valid ? PR.n : 0; info->form = PR.form; }
    if (!PR.valid) { if (J || r0) { h->err = "get_prior: the handle holds no prior"; return SADVIO_E_STATE; } return SADVIO_OK; }
    HIP_TRY(hipSetDevice(h->device));
    if (J) HIP_TRY(hipMemcpyAsync(J, PR.J.p, sizeof(double) * (size_t)PR.n_full * PR.n, hipMemcpyDeviceToHost, h->stream));
    if (r0) HIP_TRY(hipMemcpyAsync(r0, PR.r0.p, sizeof(double) * PR.n_full, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SADVIO_OK;
}

int sadvio_ba_set_prior(sadvio_ba_handle* h, int32_t n_full, int32_t n, int32_t form, const double* J, const double* r0) {
    if (!h) return SADVIO_E_INVALID_ARG;
    PriorState& PR = h->prior;
    PR.serial++;
    PR.hg_valid = false;
    if (n_full <= 0) { PR.valid = false; PR.z_valid = false; return SADVIO_OK; }
    if (n <= 0 || !J || !r0 || form != SADVIO_PRIOR_FORM_EIGEN) {   // a Cholesky-form prior carries its pivot order: only the device produces one
        h->err = "set_prior: needs J, r0 in the eigen form (orthogonal rows)"; return SADVIO_E_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(PR.J.alloc((size_t)n_full * n)); HIP_TRY(PR.r0.alloc(n_full));
    HIP_TRY(hipMemcpyAsync(PR.J.p, J, sizeof(double) * (size_t)n_full * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(PR.r0.p, r0, sizeof(double) * n_full, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    PR.valid = true; PR.z_valid = false; PR.n_full = n_full; PR.n = n; PR.form = form; PR.cut_mode = SADVIO_EIG_CUT_REFERENCE;
    return SADVIO_OK;
}

namespace {
// host-side post-processing of the tiny (3x3 / 15x15) NFR covariances
void host_sym_eig(const double* Ain, int n, double* ev, double* V) {
    double A[225];
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-60 || off <= 1e-32 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) { const double a = A[k * n + p], b = A[k * n + q]; A[k * n + p] = c * a - s * b; A[k * n + q] = s * a + c * b; }
                for (int k = 0; k < n; k++) { const double a = A[p * n + k], b = A[q * n + k]; A[p * n + k] = c * a - s * b; A[q * n + k] = s * a + c * b; }
                for (int k = 0; k < n; k++) { const double a = V[k * n + p], b = V[k * n + q]; V[k * n + p] = c * a - s * b; V[k * n + q] = s * a + c * b; }
            }
    }
    for (int i = 0; i < n; i++) ev[i] = A[i * n + i];
}

bool host_inverse(const double* A, int n, double* Ai) {
    std::vector<double> M((size_t)n * 2 * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { M[i * 2 * n + j] = A[i * n + j]; M[i * 2 * n + n + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(M[r * 2 * n + c]) > std::fabs(M[p * 2 * n + c])) p = r;
        if (M[p * 2 * n + c] == 0.0) return false;
        if (p != c) for (int j = 0; j < 2 * n; j++) std::swap(M[c * 2 * n + j], M[p * 2 * n + j]);
        const double d = 1.0 / M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] *= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double f = M[r * 2 * n + c];
            if (f != 0.0) for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Ai[i * n + j] = M[i * 2 * n + n + j];
    return true;
}

// symmetric square root of the information of an NFR factor from its covariance (marginalization.cpp:379-385 /
// :482-487): VIO inverts first and keeps eigenvalues > 1e-12, VO inverts the eigenvalues > 1e-12
bool nfr_sqrt_info(const double* S, int rows, bool invert_first, double* W) {
    double M[225], ev[15], V[225];
    if (invert_first) { if (!host_inverse(S, rows, M)) return false; }
    else memcpy(M, S, sizeof(double) * rows * rows);
    for (int i = 0; i < rows; i++) for (int j = 0; j < i; j++) { const double s = 0.5 * (M[i * rows + j] + M[j * rows + i]); M[i * rows + j] = M[j * rows + i] = s; }
    host_sym_eig(M, rows, ev, V);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < rows; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) {
                const double e = ev[k] > 1e-12 ? (invert_first ? ev[k] : 1.0 / ev[k]) : 0.0;
                s += V[i * rows + k] * std::sqrt(e) * V[j * rows + k];
            }
            W[i * rows + j] = s;
        }
    return true;
}
}  // namespace

int sadvio_ba_marginalize_relative(sadvio_ba_handle* h, int32_t w, int32_t kf_a, int32_t kf_b, int32_t eig_cut_mode, double* inf36, double* Ak144) {
    if (!h || !inf36) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "marginalize_relative before set_windows"; return SADVIO_E_STATE; }
    if (h->defer) { h->err = "marginalize_relative between begin_update and commit_update"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size()) { h->err = "marginalize_relative: window out of range"; return SADVIO_E_INVALID_ARG; }
    if (eig_cut_mode != SADVIO_EIG_CUT_REFERENCE && eig_cut_mode != SADVIO_EIG_CUT_NOISE_FLOOR) { h->err = "marginalize_relative: bad eig_cut_mode"; return SADVIO_E_INVALID_ARG; }
    if (h->world > 1) { h->err = "marginalize_relative: the window is sharded over several GPUs (each rank holds a landmark partition only)"; return SADVIO_E_INVALID_ARG; }
    const WinDev& d = h->wins[w].d;
    if (kf_a < 0 || kf_a >= d.n_kf || kf_b < 0 || kf_b >= d.n_kf || kf_a == kf_b) { h->err = "marginalize_relative: bad key-frame index"; return SADVIO_E_INVALID_ARG; }
    if (d.has_imu) { h->err = "marginalize_relative: frames with IMU states are not supported (the reference's own column layout for them is inconsistent, BundleAdjustmentCERESAnalytic.cpp:705-737)"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    // preMarginalizeRelative (marginalization.cpp:532-588): a landmark of frame a is entered once per feature it has in frame b
    const int ga = d.kf_base + kf_a, gb = d.kf_base + kf_b;
    std::vector<int> items;
    int m = 0;
    for (int l = 0; l < d.n_lmk; l++) {
        const int gl = d.lmk_base + l;
        int ca = 0, cb = 0;
        for (int o = h->h_lmk_ob[gl]; o < h->h_lmk_oe[gl]; o++) {
            if (h->obs_perm[o] < 0) continue;      // pseudo-observation of a sparse prior factor
            ca += h->h_obs_kf[o] == ga; cb += h->h_obs_kf[o] == gb;
        }
        if (ca > 0 && cb > 0) { items.push_back(gl); items.push_back(cb); m += 3 * cb; }
    }
    const int n_items = (int)items.size() / 2;
    if (n_items == 0) { h->err = "marginalize_relative: the two key-frames share no landmark"; return SADVIO_E_REFUSED; }
    DevBuf<int> ditems; DevBuf<double> dscr, dAk, dJ; DevBuf<unsigned long long> dmax;
    HIP_TRY(ditems.alloc(items.size())); HIP_TRY(dscr.alloc((size_t)n_items * RELM_ROW)); HIP_TRY(dAk.alloc(144)); HIP_TRY(dJ.alloc(72)); HIP_TRY(dmax.alloc(1));
    HIP_TRY(hipMemcpyAsync(ditems.p, items.data(), items.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(dAk.p, 0, 144 * sizeof(double), h->stream)); HIP_TRY(hipMemsetAsync(dmax.p, 0, 8, h->stream));
    SolveOpts o{};
    DevPtrs P = make_ptrs(h, o, 1);
    auto kl = h->factor_type == SADVIO_FACTOR_PIXEL ? k_relmarg_lmk<0> : k_relmarg_lmk<1>;
    hipLaunchKernelGGL(kl, dim3((n_items + 63) / 64), dim3(64), 0, h->stream, P, ditems.p, n_items, ga, gb, dscr.p, dAk.p, dmax.p);
    hipLaunchKernelGGL(k_relmarg_apply, dim3((n_items + 63) / 64), dim3(64), 0, h->stream, dscr.p, n_items, m, dmax.p, dAk.p, eig_cut_mode == SADVIO_EIG_CUT_NOISE_FLOOR ? 1 : 0);
    hipLaunchKernelGGL(k_relmarg_jac, dim3(1), dim3(64), 0, h->stream, P, ga, gb, dJ.p);
    HIP_TRY(hipGetLastError());
    double Ak[144], J[72];
    HIP_TRY(hipMemcpyAsync(Ak, dAk.p, sizeof(Ak), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(J, dJ.p, sizeof(J), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (Ak144) memcpy(Ak144, Ak, sizeof(Ak));
    // rankReveallingDecomposition (Eigen reads the lower triangle) -> Sigma_k = U diag(1 / lambda) U^T (marginalization.cpp:255-262)
    double As[144], ev[12], V[144], Sk[144];
    for (int i = 0; i < 12; i++) for (int j = 0; j <= i; j++) As[12 * i + j] = As[12 * j + i] = Ak[12 * i + j];
    host_sym_eig(As, 12, ev, V);
    double mx = 0.0;
    for (int k = 0; k < 12; k++) mx = std::max(mx, std::fabs(ev[k]));
    // SADVIO_EIG_CUT_NOISE_FLOOR: the floor of the Schur complement = a sum over the marginalised landmarks (see oracle/marg.c):
    // the gauge null space of Ak computes to ~ eps * lambda_max * n_items; SADVIO_EIG_CUT_REFERENCE: the reference's absolute 1e-12
    const double cut = eig_cut_mode == SADVIO_EIG_CUT_NOISE_FLOOR ? std::max(1e-12, 12 * 2.220446049250313e-16 * mx * (2.0 + n_items)) : 1e-12;
    memset(Sk, 0, sizeof(Sk));
    for (int k = 0; k < 12; k++) {
        if (!(ev[k] > cut)) continue;
        const double iv = 1.0 / ev[k];
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) Sk[12 * i + j] += V[12 * i + k] * iv * V[12 * j + k];
    }
    double JS[72], cov[36];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 12; j++) { double s2 = 0; for (int k = 0; k < 12; k++) s2 += J[12 * i + k] * Sk[12 * k + j]; JS[12 * i + j] = s2; }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { double s2 = 0; for (int k = 0; k < 12; k++) s2 += JS[12 * i + k] * J[12 * j + k]; cov[6 * i + j] = s2; }
    if (!host_inverse(cov, 6, inf36)) { h->err = "marginalize_relative: singular covariance of the relative pose"; return SADVIO_E_REFUSED; }
    return SADVIO_OK;
}

namespace {
// Z (n_full x n) with Z^T Z = Sigma_k = pseudo-inverse of the prior's information, for the NFR covariances of sparsify:
// eigen form: rows J_c / lambda_c; Cholesky form of full rank: the triangular inverse of G (k_tri_*: recursive halving on the
// matrix cores); a rank-deficient Cholesky-form prior is first orthogonalised by the block Jacobi (its rows then ARE the eigen form).
int prior_build_Z(sadvio_ba_handle* h, const double* J, int nf, int n, int form, const int* step_of, double* Z, int cut_mode, double* trace_out, bool guard) {
    MargScratch& M = h->mg;
    bool hidden = false;    // guard: the inverse shows an eigenvalue that may lie below the reference's cut -> pseudo-inverse by orthogonalised rows
    if (form == SADVIO_PRIOR_FORM_CHOLESKY && nf == n) {
        const int npad = (n + 31) / 32 * 32;
        HIP_TRY(M.L.alloc((size_t)npad * npad)); HIP_TRY(M.Tb.alloc((size_t)npad * npad)); HIP_TRY(M.piv_of.alloc(n));
        MargScratch::TriPlan& TP = M.tri_plans[npad];
        if (TP.leaves == 0) {
            // node table of the recursion over [0, npad): leaves of <= 32 rows, inner nodes grouped by height
            std::vector<std::vector<TriNode>> lev;
            std::vector<TriNode> leaves;
            struct Rec { static int go(int lo, int hi, std::vector<std::vector<TriNode>>& lev, std::vector<TriNode>& leaves) {
                if (hi - lo <= 32) { leaves.push_back({lo, lo, hi, 0}); return 0; }
                const int blocks = (hi - lo + 31) / 32, mid = lo + 32 * ((blocks + 1) / 2);
                const int hl = go(lo, mid, lev, leaves), hr = go(mid, hi, lev, leaves);
                const int ht = std::max(hl, hr) + 1;
                if ((int)lev.size() < ht) lev.resize(ht);
                lev[ht - 1].push_back({lo, mid, hi, 0});
                return ht;
            } };
            Rec::go(0, npad, lev, leaves);
            std::vector<TriNode> all(leaves);
            TP.levels.clear();
            for (auto& l : lev) {
                int mm = 0, mn = 0;
                for (auto& nd : l) { mm = std::max(mm, nd.hi - nd.mid); mn = std::max(mn, nd.mid - nd.lo); }
                TP.levels.push_back({(int)all.size(), (int)l.size(), mm, mn});
                all.insert(all.end(), l.begin(), l.end());
            }
            TP.leaves = (int)leaves.size();
            HIP_TRY(TP.nodes.alloc(all.size()));
            h->up.add(TP.nodes.p, all.data(), all.size() * sizeof(TriNode));
            HIP_TRY(h->up.flush(h->stream));
        }
        hipLaunchKernelGGL(k_tri_gather, dim3((n + 255) / 256), dim3(256), 0, h->stream, J, n, step_of, M.piv_of.p, npad, M.L.p, 0);
        const long long np2 = (long long)npad * npad;
        hipLaunchKernelGGL(k_tri_gather, dim3((unsigned)((np2 + 255) / 256)), dim3(256), 0, h->stream, J, n, step_of, M.piv_of.p, npad, M.L.p, 1);
        hipLaunchKernelGGL(k_tri_leaf, dim3(TP.leaves), dim3(64), 0, h->stream, M.L.p, npad, TP.nodes.p);
        for (const auto& lv : TP.levels) {
            const dim3 grid((lv.max_n + 63) / 64, (lv.max_m + 63) / 64, lv.count);
            hipLaunchKernelGGL(k_tri_level, grid, dim3(256), 0, h->stream, M.L.p, M.Tb.p, npad, TP.nodes.p + lv.first, 0);
            hipLaunchKernelGGL(k_tri_level, grid, dim3(256), 0, h->stream, M.L.p, M.Tb.p, npad, TP.nodes.p + lv.first, 1);
        }
        const long long nn = (long long)n * n;
        hipLaunchKernelGGL(k_tri_scatter, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, h->stream, M.L.p, npad, n, step_of, Z);
        if (trace_out || (guard && cut_mode == SADVIO_EIG_CUT_REFERENCE)) {
            // trace(A^-1) = |Z|_F^2 bounds the smallest eigenvalue of A = G^T G from below (lambda_min >= 1 / trace): pivots that all pass do
            // not (they bound eigenvalues from above). Callers: Amm's pseudo-inverse in marginalize (trace_out), Sigma_k of sparsify (guard)
            HIP_TRY(M.lam.alloc((size_t)n + 2));
            hipLaunchKernelGGL(k_row_norm2, dim3(n), dim3(JAC_THREADS), 0, h->stream, Z, n, n, M.lam.p);
            std::vector<double> rn(n);
            HIP_TRY(hipMemcpyAsync(rn.data(), M.lam.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            double tr = 0.0;
            for (int i = 0; i < n; i++) tr += rn[i];
            if (h->env.debug & 16384) fprintf(stderr, "[sadvio dbg] triangular inverse: trace(A^-1) %.3e (n %d): lambda_min >= %.3e\n", tr, n, 1.0 / tr);
            if (trace_out) *trace_out = tr;
            hidden = guard && cut_mode == SADVIO_EIG_CUT_REFERENCE && !(tr < 1e12);
            if (hidden) h->hidden_eig_count++;
        }
        if (!hidden) return SADVIO_OK;
    }
    const double* rows = J;
    if (form == SADVIO_PRIOR_FORM_CHOLESKY) {   // rank-deficient (or an eigenvalue that may lie below the cut): orthogonalise a copy of the rows
        HIP_TRY(M.G.alloc((size_t)nf * n)); HIP_TRY(M.flag.alloc(8));
        HIP_TRY(hipMemcpyAsync(M.G.p, J, sizeof(double) * (size_t)nf * n, hipMemcpyDeviceToDevice, h->stream));
        if (run_jacobi_rows(h, M.G.p, nf, n, M.flag.p) < 0) { h->err = "sparsify: HIP error in the eigen-solver"; return SADVIO_E_HIP; }
        rows = M.G.p;
    }
    HIP_TRY(M.lam.alloc((size_t)nf + 2));
    hipLaunchKernelGGL(k_row_norm2, dim3(nf), dim3(JAC_THREADS), 0, h->stream, rows, nf, n, M.lam.p);
    hipLaunchKernelGGL(k_z_cut, dim3(1), dim3(256), 0, h->stream, M.lam.p, nf, cut_mode == SADVIO_EIG_CUT_NOISE_FLOOR ? 1 : 0, M.lam.p + nf);
    hipLaunchKernelGGL(k_z_from_eig, dim3(nf), dim3(JAC_THREADS), 0, h->stream, rows, n, M.lam.p, M.lam.p + nf, Z);
    return SADVIO_OK;
}
}  // namespace

int sadvio_ba_sparsify(sadvio_ba_handle* h, int32_t w, int32_t vio, int32_t nf, int32_t n, const double* J, int32_t kf_keep,
                       int32_t kf_col, int32_t n_keep, const int32_t* lmk_index, const int32_t* lmk_col, int32_t* n_out,
                       sadvio_sparse_prior* out) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (n_out) *n_out = 0;
    if (!h->uploaded) { h->err = "sparsify before set_windows"; return SADVIO_E_STATE; }
    if (h->defer) { h->err = "sparsify between begin_update and commit_update"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size() || !n_out || !out || n_keep < 0 || (n_keep > 0 && (!lmk_index || !lmk_col))) { h->err = "sparsify: bad argument"; return SADVIO_E_INVALID_ARG; }
    if (h->world > 1) { h->err = "sparsify: the window is sharded over several GPUs (linearisation values of a landmark partition only)"; return SADVIO_E_INVALID_ARG; }
    PriorState& PR = h->prior;
    MargScratch& M = h->mg;
    const bool resident = J == nullptr;
    if (resident) {
        if (!PR.valid) { h->err = "sparsify: J = NULL but the handle holds no prior"; return SADVIO_E_STATE; }
        nf = PR.n_full; n = PR.n;
    }
    if (n <= 0 || nf <= 0) { h->err = "sparsify: empty prior"; return SADVIO_E_REFUSED; }
    const HostWin& HW = h->wins[w];
    const WinDev& d = HW.d;
    const SrcWin& SW = h->src[w];
    if (vio && (kf_keep < 0 || kf_keep >= d.n_kf || kf_col < 0 || kf_col + 15 > n)) { h->err = "sparsify: kept key-frame out of range"; return SADVIO_E_INVALID_ARG; }
    for (int k = 0; k < n_keep; k++)
        if (lmk_col[k] >= 0 && (lmk_index[k] < 0 || lmk_index[k] >= d.n_lmk || lmk_col[k] + 3 > n)) { h->err = "sparsify: kept landmark out of range"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    // linearisation values of the window: the handle's deep copy of the caller's arrays (no read-back)
    double T[12], v3[3] = {0, 0, 0}, ba3[3] = {0, 0, 0}, bg3[3] = {0, 0, 0};
    if (vio) {
        memcpy(T, &SW.kf_T[12 * (size_t)kf_keep], sizeof(T));
        if (!SW.kf_vel.empty()) memcpy(v3, &SW.kf_vel[3 * (size_t)kf_keep], 24);
        if (!SW.kf_ba.empty()) memcpy(ba3, &SW.kf_ba[3 * (size_t)kf_keep], 24);
        if (!SW.kf_bg.empty()) memcpy(bg3, &SW.kf_bg[3 * (size_t)kf_keep], 24);
    }
    const double* lp = SW.lmk_p.data();
    // the prior's rows and Z with Z^T Z = Sigma_k on the device
    const double* dJ = nullptr; const double* dZ = nullptr;
    if (resident) {
        dJ = PR.J.p;
        if (!PR.z_valid) {
            HIP_TRY(PR.Z.alloc((size_t)nf * n));
            const int rc = prior_build_Z(h, PR.J.p, nf, n, PR.form, PR.step_of.p, PR.Z.p, PR.cut_mode, nullptr, true);   // guard: Sigma_k = the pseudo-inverse the reference takes (marginalization.cpp:255-262)
            if (rc != SADVIO_OK) return rc;
            PR.z_valid = true;
        }
        dZ = PR.Z.p;
    } else {
        HIP_TRY(M.lastJ.alloc((size_t)nf * n)); HIP_TRY(M.Zt.alloc((size_t)nf * n));
        HIP_TRY(hipMemcpyAsync(M.lastJ.p, J, sizeof(double) * (size_t)nf * n, hipMemcpyHostToDevice, h->stream));
        const int rc = prior_build_Z(h, M.lastJ.p, nf, n, SADVIO_PRIOR_FORM_EIGEN, nullptr, M.Zt.p, SADVIO_EIG_CUT_REFERENCE);
        if (rc != SADVIO_OK) return rc;
        dJ = M.lastJ.p; dZ = M.Zt.p;
    }
    std::vector<NfrSpecC> specs;
    std::vector<double> jsel(4 * 225, 0.0);
    std::vector<int> kept;  // positions k with lmk_col >= 0
    for (int k = 0; k < n_keep; k++) if (lmk_col[k] >= 0) kept.push_back(k);
    std::vector<int> order;  // VO: chain order (indices into kept)
    int out_off = 0;
    auto push = [&](int rows, int cols, int js) { NfrSpecC s{}; s.rows = rows; s.cols = cols; s.jsel = js; s.out_off = out_off; out_off += rows * rows; specs.push_back(s); return &specs.back(); };
    if (vio) {
        double tsk[9] = {0, -T[11], T[10], T[11], 0, -T[9], -T[10], T[9], 0}, Rt[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[3 * i + k] * tsk[3 * k + j]; Rt[3 * i + j] = s; }
        double* J0 = &jsel[0];       // IMUPriordx selector 15 x 15 (marginalization.cpp:366-378)
        for (int a = 0; a < 15; a++) J0[a * 15 + a] = 1.0;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { J0[i * 15 + j] = T[3 * i + j]; J0[i * 15 + 3 + j] = T[3 * i + j]; J0[(3 + i) * 15 + 3 + j] = T[3 * i + j]; }
        double* J1 = &jsel[225];     // PoseToLandmarkFactor selector 3 x 9: [R | -R [t]x | R] on (landmark, rotation, translation)
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { J1[i * 9 + j] = T[3 * i + j]; J1[i * 9 + 3 + j] = -Rt[3 * i + j]; J1[i * 9 + 6 + j] = T[3 * i + j]; }
        NfrSpecC* f = push(15, 15, 0);
        for (int a = 0; a < 15; a++) f->cidx[a] = kf_col + a;
        for (int k : kept) {
            NfrSpecC* s = push(3, 9, 1);
            s->fin = 1;   // the kernel returns the factor's information square root (nfr_sqrt_info3), not its covariance
            for (int a = 0; a < 3; a++) { s->cidx[a] = lmk_col[k] + a; s->cidx[3 + a] = kf_col + a; s->cidx[6 + a] = kf_col + 3 + a; }
        }
    } else {
        const int K = (int)kept.size();
        if (K < 2) { h->err = "sparsify: fewer than two kept landmarks"; return SADVIO_E_REFUSED; }
        std::vector<int> lc(K);
        for (int a = 0; a < K; a++) lc[a] = lmk_col[kept[a]];
        HIP_TRY(M.lc.alloc(K)); HIP_TRY(M.mi.alloc((size_t)K * K));
        HIP_TRY(hipMemcpyAsync(M.lc.p, lc.data(), sizeof(int) * K, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemsetAsync(M.mi.p, 0, sizeof(double) * (size_t)K * K, h->stream));
        hipLaunchKernelGGL(k_nfr_trace, dim3((K * K + 255) / 256), dim3(256), 0, h->stream, dJ, nf, n, M.lc.p, K, M.mi.p);
        std::vector<double> mi((size_t)K * K);
        HIP_TRY(hipMemcpyAsync(mi.data(), M.mi.p, sizeof(double) * (size_t)K * K, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        // greedy chain (marginalization.cpp:432-456); Eigen's maxCoeff visits a column-major matrix column by column
        int mr = 0, mc = 0; double best = -1;
        for (int j = 0; j < K; j++) for (int i = 0; i < K; i++) if (mi[(size_t)i * K + j] > best) { best = mi[(size_t)i * K + j]; mr = i; mc = j; }
        order.push_back(mr); order.push_back(mc);
        for (int i = 0; i < K; i++) { mi[(size_t)i * K + mr] = 0; mi[(size_t)mr * K + i] = 0; mi[(size_t)i * K + mc] = 0; }
        int cur = mc;
        for (;;) {
            int bc = 0; double bv = mi[(size_t)cur * K];
            for (int j = 1; j < K; j++) if (mi[(size_t)cur * K + j] > bv) { bv = mi[(size_t)cur * K + j]; bc = j; }
            if (bv == 0) break;
            order.push_back(bc);
            for (int j = 0; j < K; j++) mi[(size_t)cur * K + j] = 0;
            for (int i = 0; i < K; i++) mi[(size_t)i * K + bc] = 0;
            cur = bc;
        }
        double* J2 = &jsel[2 * 225];   // identity 3 x 3
        double* J3 = &jsel[3 * 225];   // [I -I] 3 x 6
        for (int q = 0; q < 3; q++) { J2[q * 3 + q] = 1.0; J3[q * 6 + q] = 1.0; J3[q * 6 + 3 + q] = -1.0; }
        // covariance of every ordered landmark (entropy root, unary factor) then of every chain link
        for (int a : order) {
            NfrSpecC* s = push(3, 3, 2);
            for (int q = 0; q < 3; q++) s->cidx[q] = lmk_col[kept[a]] + q;
        }
        for (size_t k = 0; k + 1 < order.size(); k++) {
            NfrSpecC* s = push(3, 6, 3);
            for (int q = 0; q < 3; q++) { s->cidx[q] = lmk_col[kept[order[k]]] + q; s->cidx[3 + q] = lmk_col[kept[order[k + 1]]] + q; }
        }
    }
    const int ns = (int)specs.size();
    HIP_TRY(M.spec.alloc(ns)); HIP_TRY(M.jsel.alloc(4 * 225)); HIP_TRY(M.S.alloc((size_t)std::max(out_off, 1)));
    h->up.add(M.spec.p, specs.data(), sizeof(NfrSpecC) * (size_t)ns);
    h->up.add(M.jsel.p, jsel.data(), sizeof(double) * jsel.size());
    HIP_TRY(h->up.flush(h->stream));
    int first3 = 0;
    if (vio) {
        // the one 15-row factor (IMUPriordx) as two matrix-core products — W = Jsel Z[:, kf]^T (15 x nf), cov = W W^T — instead of
        // 120 LDS atomics per row of Z from one workgroup (measured 1.0 ms of the 1.6 ms call)
        HIP_TRY(M.T.alloc((size_t)15 * nf));
        launch_mgemm(h, M.T.p, nf, M.jsel.p, 15LL, 1LL, dZ + kf_col, 1LL, (long long)n, 15, nf, 15, 1.0, 0.0);
        launch_mgemm(h, M.S.p + specs[0].out_off, 15, M.T.p, (long long)nf, 1LL, M.T.p, 1LL, (long long)nf, 15, 15, nf, 1.0, 0.0);
        first3 = 1;
    }
    if (ns > first3) hipLaunchKernelGGL(k_nfr_cov_z, dim3(ns - first3), dim3(JAC_THREADS), 0, h->stream, dZ, nf, n, M.spec.p + first3, M.jsel.p, M.S.p);
    std::vector<double>& S = M.hS;
    S.resize((size_t)out_off);
    HIP_TRY(hipMemcpyAsync(S.data(), M.S.p, sizeof(double) * (size_t)out_off, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipGetLastError());
    int cnt = 0;
    auto fail = [&]() { h->err = "sparsify: singular factor covariance"; return SADVIO_E_NOT_USABLE; };
    if (vio) {
        sadvio_sparse_prior* o = out + cnt++;
        memset(o, 0, sizeof(*o));
        o->type = SADVIO_SPARSE_IMU_PRIOR; o->kf = kf_keep; o->lmk0 = o->lmk1 = -1;
        memcpy(o->T_prior, T, sizeof(T)); memcpy(o->v_prior, v3, 24); memcpy(o->ba_prior, ba3, 24); memcpy(o->bg_prior, bg3, 24);
        if (!nfr_sqrt_info(&S[specs[0].out_off], 15, true, o->sqrt_inf)) return fail();
        for (size_t i = 0; i < kept.size(); i++) {
            const int k = kept[i];
            o = out + cnt++;
            memset(o, 0, sizeof(*o));
            o->type = SADVIO_SPARSE_POSE_TO_LMK; o->kf = kf_keep; o->lmk0 = lmk_index[k]; o->lmk1 = -1;
            const double* p = &lp[3 * (size_t)lmk_index[k]];
            for (int a = 0; a < 3; a++) o->delta[a] = T[3 * a] * p[0] + T[3 * a + 1] * p[1] + T[3 * a + 2] * p[2] + T[9 + a];
            const double* Wd = &S[specs[i + 1].out_off];    // taken on the device
            for (int a = 0; a < 9; a++) { if (!std::isfinite(Wd[a])) return fail(); o->sqrt_inf[a] = Wd[a]; }
        }
    } else {
        const int no = (int)order.size();
        int root = 0; double best_det = 0;
        for (int k = 0; k < no; k++) {
            const double* s = &S[specs[k].out_off];
            const double det = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) + s[2] * (s[3] * s[7] - s[4] * s[6]);
            if (k == 0 || det < best_det) { best_det = det; root = k; }
        }
        sadvio_sparse_prior* o = out + cnt++;
        memset(o, 0, sizeof(*o));
        const int lr = lmk_index[kept[order[root]]];
        o->type = SADVIO_SPARSE_LMK_PRIOR; o->kf = -1; o->lmk0 = lr; o->lmk1 = -1;
        memcpy(o->delta, &lp[3 * (size_t)lr], 24);
        if (!nfr_sqrt_info(&S[specs[root].out_off], 3, false, o->sqrt_inf)) return fail();
        for (int k = 0; k + 1 < no; k++) {
            const int la = lmk_index[kept[order[k]]], lb = lmk_index[kept[order[k + 1]]];
            o = out + cnt++;
            memset(o, 0, sizeof(*o));
            o->type = SADVIO_SPARSE_LMK_TO_LMK; o->kf = -1; o->lmk0 = la; o->lmk1 = lb;
            for (int a = 0; a < 3; a++) o->delta[a] = lp[3 * (size_t)la + a] - lp[3 * (size_t)lb + a];
            if (!nfr_sqrt_info(&S[specs[no + k].out_off], 3, false, o->sqrt_inf)) return fail();
        }
    }
    *n_out = cnt;
    return SADVIO_OK;
}

int sadvio_ba_set_collective(sadvio_ba_handle* h, int32_t rank, int32_t world, sadvio_allreduce_fn fn, void* ctx) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) { h->err = "set_collective: bad rank / world / callback"; return SADVIO_E_INVALID_ARG; }
    if (h->uploaded) { h->err = "set_collective must precede set_windows (the buffer layout depends on the world size)"; return SADVIO_E_STATE; }
    h->rank = rank; h->world = world; h->coll_fn = fn; h->coll_ctx = ctx;
    return SADVIO_OK;
}

namespace {
static std::string load_rccl(RcclLib& R) {
    if (R.lib) return "";
    R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!R.lib) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!R.lib) return std::string("dlopen librccl: ") + dlerror();
    R.get_id = (int (*)(NcclId*))dlsym(R.lib, "ncclGetUniqueId");
    R.init_rank = (int (*)(void**, int, NcclId, int))dlsym(R.lib, "ncclCommInitRank");
    R.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(R.lib, "ncclAllReduce");
    R.destroy = (int (*)(void*))dlsym(R.lib, "ncclCommDestroy");
    R.err_string = (const char* (*)(int))dlsym(R.lib, "ncclGetErrorString");
    R.comm_count = (int (*)(void*, int*))dlsym(R.lib, "ncclCommCount");
    R.comm_user_rank = (int (*)(void*, int*))dlsym(R.lib, "ncclCommUserRank");
    R.comm_cu_device = (int (*)(void*, int*))dlsym(R.lib, "ncclCommCuDevice");
    if (!R.get_id || !R.init_rank || !R.all_reduce || !R.destroy) return "missing RCCL symbol";
    return "";
}
int rccl_allreduce(void* ctx, double* buf, int64_t count, void* stream) {
    sadvio_ba_handle* h = (sadvio_ba_handle*)ctx;
    // ncclDouble = 8, ncclSum = 0; in place
    return h->rccl.all_reduce(buf, buf, (size_t)count, 8, 0, h->rccl.comm, (hipStream_t)stream);
}
}  // namespace

int sadvio_ba_rccl_unique_id(void* id128) {
    if (!id128) return SADVIO_E_INVALID_ARG;
    RcclLib R;
    const std::string e = load_rccl(R);
    if (!e.empty()) return SADVIO_E_RCCL;
    NcclId id;
    if (R.get_id(&id) != 0) return SADVIO_E_RCCL;
    memcpy(id128, &id, sizeof(id));
    return SADVIO_OK;  // the library stays loaded (process lifetime)
}

int sadvio_ba_comm_init_rccl(sadvio_ba_handle* h, int32_t rank, int32_t world, const void* id128) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!id128 || world < 1 || rank < 0 || rank >= world) { h->err = "comm_init_rccl: bad argument"; return SADVIO_E_INVALID_ARG; }
    if (h->uploaded) { h->err = "comm_init_rccl must precede set_windows"; return SADVIO_E_STATE; }
    HIP_TRY(hipSetDevice(h->device));
    const std::string e = load_rccl(h->rccl);
    if (!e.empty()) { h->err = "comm_init_rccl: " + e; return SADVIO_E_RCCL; }
    NcclId id;
    memcpy(&id, id128, sizeof(id));
    const int rc = h->rccl.init_rank(&h->rccl.comm, world, id, rank);
    if (rc != 0) {
        h->err = std::string("comm_init_rccl: ncclCommInitRank: ") + (h->rccl.err_string ? h->rccl.err_string(rc) : "error");
        h->rccl.comm = nullptr;
        return SADVIO_E_RCCL;
    }
    h->rank = rank; h->world = world; h->coll_fn = rccl_allreduce; h->coll_ctx = h;
    return SADVIO_OK;
}

int sadvio_ba_comm_info(sadvio_ba_handle* h, int32_t* nranks, int32_t* rank, int32_t* device, int32_t* is_rccl) {
    if (!h) return SADVIO_E_INVALID_ARG;
    int n = h->world, r = h->rank, d = h->device;
    const bool rccl = h->rccl.comm != nullptr;
    if (rccl) {   // what the COMMUNICATOR says, not what it was asked for
        if (!h->rccl.comm_count || !h->rccl.comm_user_rank || h->rccl.comm_count(h->rccl.comm, &n) != 0 || h->rccl.comm_user_rank(h->rccl.comm, &r) != 0) {
            h->err = "comm_info: ncclCommCount / ncclCommUserRank failed"; return SADVIO_E_RCCL;
        }
        if (h->rccl.comm_cu_device) (void)h->rccl.comm_cu_device(h->rccl.comm, &d);
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    if (is_rccl) *is_rccl = rccl ? 1 : 0;
    return SADVIO_OK;
}

int sadvio_ba_solve(sadvio_ba_handle* h, const sadvio_solve_options* opts, sadvio_solve_summary* summaries) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "solve before set_windows"; return SADVIO_E_STATE; }
    if (h->defer) { h->err = "solve between begin_update and commit_update"; return SADVIO_E_STATE; }
    sadvio_solve_options defo;
    if (!opts) { sadvio_ba_default_options(&defo); opts = &defo; }
    if (opts->max_num_iterations < 0 || opts->max_num_iterations > 1000) { h->err = "solve: max_num_iterations out of range"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    SolveOpts o{};
    o.max_num_iterations = opts->max_num_iterations; o.jacobi_scaling = opts->jacobi_scaling;
    o.max_num_consecutive_invalid_steps = opts->max_num_consecutive_invalid_steps;
    o.function_tolerance = opts->function_tolerance; o.gradient_tolerance = opts->gradient_tolerance;
    o.parameter_tolerance = opts->parameter_tolerance; o.initial_radius = opts->initial_trust_region_radius;
    o.max_radius = opts->max_trust_region_radius; o.min_radius = opts->min_trust_region_radius;
    o.min_lm_diagonal = opts->min_lm_diagonal; o.max_lm_diagonal = opts->max_lm_diagonal;
    o.min_relative_decrease = opts->min_relative_decrease;
    o.huber_a = opts->huber_a;
    // wall_clock64: 100 MHz. Not applied to a window sharded over several GPUs: the ranks' clocks would disagree on the slot
    o.max_time_ticks = (opts->max_solver_time_in_seconds > 0.0 && h->world == 1) ? opts->max_solver_time_in_seconds * 1e8 : 0.0;
    if (!(o.huber_a >= 0.0)) { h->err = "solve: huber_a must be >= 0"; return SADVIO_E_INVALID_ARG; }
    const int n_win = (int)h->wins.size();
    // Slot s (s = 0 .. slots-1) is one step attempt; with max_num_iterations = 0 Ceres still evaluates
    // iteration 0, so at least one slot is always run and the final decision is taken by k_final.
    const int slots = std::max(1, o.max_num_iterations);
    const int stride = slots + 2;
    HIP_TRY(h->d_dbg.alloc(128));
    HIP_TRY(h->d_states.alloc((size_t)n_win * stride));
    HIP_TRY(h->d_trace.alloc((size_t)n_win * stride * 8));
    HIP_TRY(h->d_tstart.alloc(1));
    HIP_TRY(h->d_acc.alloc((size_t)n_win * stride));
    HIP_TRY(h->d_final.alloc((size_t)n_win));
    HIP_TRY(h->d_big_info.alloc((size_t)n_win));
    if (h->h_final_n < (size_t)n_win) {
        if (h->h_final) (void)hipHostFree(h->h_final);
        h->h_final = nullptr; h->h_final_n = 0;
        HIP_TRY(hipHostMalloc((void**)&h->h_final, sizeof(FinalRec) * (size_t)n_win, hipHostMallocDefault));
        h->h_final_n = (size_t)n_win;
    }
    DevPtrs P = make_ptrs(h, o, stride);
    const int n_tiles = (int)h->tiles.size();
    // with many tiles, re-summing all partials in every k_build workgroup costs more than one tiny launch per slot
    // The decision of a slot is re-derived by every k_build workgroup from its OWN window's tile partials (read by all 256 threads
    // with every load in flight: the cost does not depend on how many windows the batch has), as long as a window has at most
    // 4 * BUILD_THREADS tiles (the canonical summation order of wave_sum_backsub_partials); a separate k_decide launch per slot
    // only for larger windows (configs 4 / 5) and for the throughput kernels, which read the decided state.
    int max_win_tiles = 0;
    for (int w = 0; w < n_win; w++) max_win_tiles = std::max(max_win_tiles, h->wins[w].d.tile_end - h->wins[w].d.tile_begin);
    P.decide_kernel = max_win_tiles > 4 * BUILD_THREADS ? 1 : 0;
    // a sharded window keeps the item loop of k_solve: every rank must leave it with the same bits (plain adds, one factor at a time)
    P.imu_direct = (!h->coll_fn && P.world == 1 && !h->env.imu_items) ? 1 : 0;
    const int mtk = h->max_tile_kf;
    const size_t nt = 6 * (size_t)h->max_tile_free;
    int Rp = 16 * ((6 * h->max_gemm_free + 15) / 16);                          // padded rows of the Y / E strips
    int strip_doubles = std::max(STAGE_VALS * 64, 2 * Rp * (32 + 2));          // per wave: Y | E strips, later the wave's copy of the tile
    size_t lds_build = tile_tables_bytes(mtk) + sizeof(double) * ((size_t)BUILD_WAVES * strip_doubles + nt * (nt + 1) / 2 + 3 * nt +
                                                                   0) + 16;
    if (lds_build > 160 * 1024 && h->max_gemm_free > 0) {
        // a window mixing short tracks with very long ones: the MFMA strips + the large atomic tile do not fit
        // together; run every tile on the ds_add_f64 path instead
        for (auto& t : h->tiles) if (t.lds_mode == 2) t.lds_mode = 1;
        HIP_TRY(hipMemcpyAsync(h->d_tiles.p, h->tiles.data(), h->tiles.size() * sizeof(Tile), hipMemcpyHostToDevice, h->stream));
        h->max_gemm_free = 0;
        Rp = 0; strip_doubles = STAGE_VALS * 64;
        lds_build = tile_tables_bytes(mtk) + sizeof(double) * ((size_t)BUILD_WAVES * strip_doubles + nt * (nt + 1) / 2 + 3 * nt +
                                                               0) + 16;
    }
    if (h->env.debug) fprintf(stderr, "[sadvio dbg] lds_build %zu B, Rp %d, strip_doubles %d, max_tile_kf %d, max_tile_free %d, tiles %d\n", lds_build, Rp, strip_doubles, mtk, h->max_tile_free, n_tiles);
    const size_t lds_back = tile_tables_bytes(mtk) + sizeof(double) * ((size_t)mtk * 18) + 16;
    // k_solve<0>: tile-packed image + y / gf / hd / xs + the chol16 exchange areas
    const size_t npq = (size_t)h->max_np;
    const size_t lds_solve = sizeof(double) * ((size_t)c16_size((int)npq) + 4 * npq + 1 + C16_WORK + 16 * (size_t)c16_blocks((int)npq + 1) + SOLVE_KFC * SOLVE_KFC_STRIDE) + 64;
    // robust loss or prior-kept landmarks in the batch: the kernels carrying those (rare) paths
    bool any_pseudo = false;
    for (const auto& v : h->sp_elim) for (char e : v) any_pseudo |= e != 0;
    // (kept landmarks: by their reduced columns, not by their observations — on a sharded window the ranks other than 0 hold them without any)
    bool any_kept_lmk = false;
    for (int w = 0; w < n_win; w++) any_kept_lmk |= h->wins[w].d.n_red > 0;
    const bool rare = o.huber_a > 0.0 || h->n_kept > 0 || any_kept_lmk || any_pseudo || h->gemm_run4;
    const bool pix = h->factor_type == SADVIO_FACTOR_PIXEL;
    // IMU factor pairs and listed sparse-prior factors ride k_build (linearisation) and k_backsub (candidate cost) as extra workgroups
    // when the submission is a window or two: the inlined linearisation leaves those variants of k_build one workgroup per CU, which
    // a batch of VIO windows would pay for; there the evaluation runs as kernels of its own on the same stream (k_pf_eval)
    const bool have_pf = !h->imus.empty() || h->n_sp_list > 0;
    bool with_imu = have_pf && n_tiles <= 3 * 256;
    if (h->env.pf_wg >= 0) with_imu = have_pf && h->env.pf_wg != 0;
    auto kb = with_imu ? (pix ? (rare ? k_build<0, true, true> : k_build<0, false, true>) : (rare ? k_build<1, true, true> : k_build<1, false, true>))
                       : (pix ? (rare ? k_build<0, true, false> : k_build<0, false, false>) : (rare ? k_build<1, true, false> : k_build<1, false, false>));
    auto kk = with_imu ? (pix ? (rare ? k_backsub<0, true, true> : k_backsub<0, false, true>) : (rare ? k_backsub<1, true, true> : k_backsub<1, false, true>))
                       : (pix ? (rare ? k_backsub<0, true, false> : k_backsub<0, false, false>) : (rare ? k_backsub<1, true, false> : k_backsub<1, false, false>));
    auto kbk = h->factor_type == SADVIO_FACTOR_PIXEL ? k_build_kept<0> : k_build_kept<1>;
    // large plain batches: the throughput kernels of lm_kernels.h (SADVIO_LM=1 / 0 forces / forbids them, for tests and A/B runs)
    bool use_lm = h->lm_ok && !rare && !h->coll_fn && h->world == 1 && h->lm_landmarks >= 65536;
    if (h->env.lm >= 0) use_lm = h->lm_ok && !rare && !h->coll_fn && h->world == 1 && h->env.lm != 0;
    auto kbo = pix ? k_build_obs<0> : k_build_obs<1>;
    auto kps = pix ? k_lm_pass<0, false> : k_lm_pass<1, false>;
    auto kps0 = pix ? k_lm_pass<0, true> : k_lm_pass<1, true>;
    const size_t lds_views = pix ? sizeof(double) * (size_t)mtk * h->lm_max_cam * LM_VT : 0;   // view tables of the pixel factor
    const size_t lds_bobs = tile_tables_bytes(mtk) + sizeof(double) * ((size_t)BUILD_WAVES * Rp * LM_KS + nt * (nt + 1) / 2 + nt + 1 + LM_DT) + lds_views + 16;
    // k_lm_pass: tables at x (+ at the candidate, + the pose steps), the tile's key-frame sums, the staged observation constants
    const size_t lds_pass0 = tile_tables_bytes(mtk) + sizeof(double) * LM_DT_COST + (size_t)h->lm_sub_obs * ((pix ? 2 : 3) * sizeof(double) + sizeof(int)) + lds_views + 16;
    const size_t lds_pass = lds_pass0 + sizeof(double) * (size_t)mtk * (POSE_TAB + 6) + lds_views;
    if (!use_lm) P.lm_sacc = nullptr;   // k_decide sums the tiles' k_backsub partials
    if (use_lm) {
        P.decide_kernel = 1;   // the kernels read the decided state of their slot
        HIP_TRY(hipFuncSetAttribute((const void*)kbo, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bobs));
        HIP_TRY(hipFuncSetAttribute((const void*)kps, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pass));
        HIP_TRY(hipFuncSetAttribute((const void*)kps0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pass0));
    }
    HIP_TRY(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_build));
    HIP_TRY(hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_back));
    bool extras = false;  // any pose-only factor family beyond PosePriordx in the batch?
    for (int w = 0; w < n_win; w++) {
        const WinDev& d = h->wins[w].d;
        if (d.imu_end > d.imu_begin || d.sp_end > d.sp_begin || d.dp_n_full > 0 || d.dpf == 15 || d.lobs_end > d.lobs_begin || d.line_end > d.line_begin) extras = true;   // dpf 15: padded pivots live in the EXTRAS kernel
    }
    auto ks0 = extras ? k_solve<0, true> : k_solve<0, false>;
    auto ks1 = extras ? k_solve<1, true> : k_solve<1, false>;
    auto ks2 = extras ? k_solve<2, true> : k_solve<2, false>;
    HIP_TRY(hipFuncSetAttribute((const void*)ks0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_solve));
    const int reset_blocks = (int)std::min<long long>(1024, std::max<long long>(1, (P.n_xl + P.n_xp + 255) / 256));
    // rows below a block column of S that can be non-zero: (block half-bandwidth + 1) * dpf from the co-visibility
    // structure and the IMU pairs; a dense prior fills the kept-landmark block, so those windows are dense
    std::vector<int> big_bw(n_win, 0);
    for (int w = 0; w < n_win; w++) {
        const WinDev& d = h->wins[w].d;
        if (!d.ld) continue;
        int hb = h->wins[w].hb_lmk;
        for (const ImuDev& f : h->imus_per_win[w]) {
            const int fi = h->h_kf_fidx[f.kf_i], fj = h->h_kf_fidx[f.kf_j];
            if (fi >= 0 && fj >= 0) hb = std::max(hb, std::abs(fi - fj));
        }
        if (w < (int)h->sparse_per_win.size())
            for (const sadvio_sparse_prior& sp : h->sparse_per_win[w])
                if (sp.type == SADVIO_SPARSE_RELATIVE_POSE) {     // a relative-pose factor couples its two key-frames
                    const int fi = h->h_kf_fidx[d.kf_base + sp.kf], fj = h->h_kf_fidx[d.kf_base + sp.kf_b];
                    if (fi >= 0 && fj >= 0) hb = std::max(hb, std::abs(fi - fj));
                }
        big_bw[w] = (d.n_red > 0 || d.dp_n_full > 0 || d.line_end > d.line_begin) ? d.Np : std::min(d.Np, (hb + 1) * d.dpf);
    }
    if (h->coll_fn && h->world > 1 && h->n_big) {
        // every rank must factor the all-reduced S with the same (largest) bandwidth: gather the local ones
        std::vector<double> slots((size_t)n_win * h->world * 4, 0.0);
        for (int w = 0; w < n_win; w++) slots[((size_t)w * h->world + h->rank) * 4] = (double)big_bw[w];
        HIP_TRY(hipMemcpyAsync(h->d_rank_s.p, slots.data(), slots.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (h->coll_fn(h->coll_ctx, h->d_rank_s.p, (int64_t)slots.size(), (void*)h->stream) != 0) { h->err = "solve: all-reduce failed"; return SADVIO_E_RCCL; }
        HIP_TRY(hipMemcpyAsync(slots.data(), h->d_rank_s.p, slots.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int w = 0; w < n_win; w++)
            for (int r = 0; r < h->world; r++) big_bw[w] = std::max(big_bw[w], (int)slots[((size_t)w * h->world + r) * 4]);
    }
    std::vector<long long> big_linv_off(n_win, 0), big_M_off(n_win, 0), big_mid_off(n_win, 0), big_Lx_off(n_win, 0);
    {
        long long totm = 0;
        for (int w = 0; w < n_win; w++)
            if (h->wins[w].d.ld) { const long long b = std::min(big_bw[w], MAX_LDS_NP); big_mid_off[w] = totm; totm += 2 * (b * (b + 1) / 2 + b); }
        HIP_TRY(h->d_big_mid.alloc((size_t)std::max<long long>(totm, 1)));
        // workspace of the block cyclic reduction (one banded window with >= 22 diagonal blocks): allocated here, never
        // inside the (possibly captured) launch sequence
        if (n_win == 1 && h->wins[0].d.ld && h->wins[0].d.dpf == 6) {
            const size_t b = (size_t)big_bw[0], K = ((size_t)h->wins[0].d.Np + b - 1) / std::max<size_t>(b, 1);
            if (b > 0 && K >= 22 && 2 * b <= (size_t)MAX_LDS_NP - 1) HIP_TRY(h->d_bcr.alloc(K * (8 * b * b + b * (b + 1) / 2 + (b / 6) * 36 + 5 * b)));
        }
        long long totM = 0;
        for (int w = 0; w < n_win; w++) if (h->wins[w].d.ld) { big_M_off[w] = totM; totM += (long long)((h->wins[w].d.Np + WD - 1) / WD) * (WD * WD + WD_LT); }   // M | the factors' tiles
        HIP_TRY(h->d_big_M.alloc((size_t)std::max<long long>(totM, 1)));
        long long totL = 0;      // out-of-place panels of the dense wide-panel solver (k_wchol_step)
        for (int w = 0; w < n_win; w++) {
            const WinDev& d = h->wins[w].d;
            const int nbp = d.dpf == 6 ? 6 : 5;
            if (d.ld && !(big_bw[w] < d.Np && big_bw[w] + nbp <= MAX_LDS_NP) && d.Np >= 2 * WD) { big_Lx_off[w] = totL; totL += (long long)d.Np * d.ld; }
        }
        HIP_TRY(h->d_big_Lx.alloc((size_t)std::max<long long>(totL, 1)));
        long long tot = 0;
        for (int w = 0; w < n_win; w++) if (h->wins[w].d.ld) { big_linv_off[w] = tot; tot += 6LL * h->wins[w].d.Np; }
        HIP_TRY(h->d_big_linv.alloc((size_t)std::max<long long>(tot, 1)));
    }
    int dp_max_nf = 0, dp_max_n = 0;
    for (int w = 0; w < n_win; w++) { dp_max_nf = std::max(dp_max_nf, h->wins[w].d.dp_n_full); dp_max_n = std::max(dp_max_n, h->wins[w].d.dp_n); }
    bool coll_failed = false;
    auto enqueue = [&]() {
        {   // zero deltas / accumulators + initial LM state, and (extra blocks) the pose tables / prior records at x = 0
            ScopedTimer t(h, "k_reset");
            const int table_blocks = (std::max(h->n_kf_tot, (int)h->priors.size()) + 63) / 64;
            hipLaunchKernelGGL(k_reset, dim3(reset_blocks + table_blocks), dim3(256), 0, h->stream, P, reset_blocks, h->n_kf_tot);
        }
        const int n_imu_all = (int)h->imus.size();
        // IMU factor pairs and the listed sparse-prior factors ride the tile kernels as extra workgroups (kernels.h: pose_factor_eval).
        // Line observations are still evaluated on a side stream: the linearisation next to k_build, the candidate cost next to
        // k_backsub (fork / join with events; parallel branches of the captured graph)
        const int n_pf = n_imu_all + h->n_sp_list;
        const int n_lo = h->n_lobs_tot;
        const bool fork = n_lo > 0 && h->side && !h->cfg.profile_kernels && !h->coll_fn && !h->env.no_fork;
        for (int s = 0; s < slots; s++) {
            if (fork) {
                (void)hipEventRecord(h->ev_fork, h->stream);
                (void)hipStreamWaitEvent(h->side, h->ev_fork, 0);
                if (n_lo) hipLaunchKernelGGL(k_line_eval<true>, dim3(n_lo), dim3(64), 0, h->side, P, s, 1);
                (void)hipEventRecord(h->ev_lin, h->side);
            }
            if (use_lm) {
                // the opening pass linearises at x (H_ll, g_l per landmark, key-frame sums per tile); later slots get them from the
                // candidate pass of the slot before
                if (s == 0) { ScopedTimer t(h, "k_lm_pass0"); hipLaunchKernelGGL(kps0, dim3(h->lm_n_sub), dim3(LM_PASS_THREADS), lds_pass0, h->stream, P, s, mtk, h->lm_sub_obs); }
                { ScopedTimer t(h, "k_build_obs"); hipLaunchKernelGGL(kbo, dim3(n_tiles), dim3(BUILD_THREADS), lds_bobs, h->stream, P, s, mtk, Rp); }
            } else
            { ScopedTimer t(h, "k_build"); hipLaunchKernelGGL(kb, dim3(n_tiles + (with_imu ? n_pf : 0)), dim3(BUILD_THREADS), lds_build, h->stream, P, s, mtk, strip_doubles, Rp); }
            if (n_pf && (use_lm || !with_imu)) { ScopedTimer t(h, "k_pf_lin"); hipLaunchKernelGGL(k_pf_eval<false>, dim3(n_pf), dim3(BUILD_THREADS), 0, h->stream, P, s); }
            if (h->n_kept) { ScopedTimer t(h, "k_build_kept"); hipLaunchKernelGGL(kbk, dim3((h->n_kept + 127) / 128), dim3(128), 0, h->stream, P, s); }
            if (dp_max_nf > 0) {
                ScopedTimer t(h, "k_prior_r+gh");
                const int colb = (dp_max_n + 3) / 4;
                hipLaunchKernelGGL(k_prior_r, dim3((dp_max_nf + 3) / 4, n_win), dim3(256), 0, h->stream, P, s);
                hipLaunchKernelGGL(k_prior_gh, dim3(colb + (unsigned)(((long long)dp_max_n * dp_max_n + 255) / 256), n_win), dim3(256), 0, h->stream, P, s, colb);
            }
            if (h->coll_fn) {
                // the window spans devices: gather the per-rank partial sums and all-reduce the reduced system
                { ScopedTimer t(h, "k_rank_partials"); hipLaunchKernelGGL(k_rank_partials, dim3(n_win), dim3(64), 0, h->stream, P, s, 0); }
                ScopedTimer t(h, "allreduce_reduced_system");
                const WinDev& d0 = h->wins[0].d;
                if (n_win == 1 && d0.ld && big_bw[0] < d0.Np && d0.S_off == 0) {
                    // one banded window spanning the devices: only the band of S travels (dense_chol.h: k_band_pack)
                    const long long nbd = (long long)d0.Np * big_bw[0], tail = h->red_total - (long long)d0.Np * d0.ld;
                    if (h->d_coll_band.alloc((size_t)(nbd + tail)) != hipSuccess) coll_failed = true;
                    else {
                        const int pb = (int)std::min<long long>((nbd + tail + 255) / 256, 2048);
                        hipLaunchKernelGGL(k_band_pack, dim3(pb), dim3(256), 0, h->stream, h->d_S.p, (long long)d0.ld, d0.Np, big_bw[0], tail, h->d_coll_band.p, 0);
                        if (h->coll_fn(h->coll_ctx, h->d_coll_band.p, (int64_t)(nbd + tail), (void*)h->stream) != 0) coll_failed = true;
                        hipLaunchKernelGGL(k_band_pack, dim3(pb), dim3(256), 0, h->stream, h->d_S.p, (long long)d0.ld, d0.Np, big_bw[0], tail, h->d_coll_band.p, 1);
                    }
                } else if (h->coll_fn(h->coll_ctx, h->d_S.p, (int64_t)h->red_total, (void*)h->stream) != 0) coll_failed = true;
            }
            if (fork) (void)hipStreamWaitEvent(h->stream, h->ev_lin, 0);
            else {
                if (n_lo) { ScopedTimer t(h, "k_line_lin"); hipLaunchKernelGGL(k_line_eval<true>, dim3(n_lo), dim3(64), 0, h->stream, P, s, 0); }
            }
            if (h->n_big < n_win) { ScopedTimer t(h, "k_solve"); hipLaunchKernelGGL(ks0, dim3(n_win), dim3(SOLVE_THREADS), lds_solve, h->stream, P, s); }
            if (h->n_big) {
                { ScopedTimer t(h, "k_solve_front"); hipLaunchKernelGGL(ks1, dim3(n_win), dim3(SOLVE_THREADS), 64, h->stream, P, s); }
                {
                    // Cholesky + solve of S, gred in place (dense_chol.h): banded systems by one sliding-window launch,
                    // dense ones (a dense prior fills the kept-landmark block) by two launches per 32 columns
                    ScopedTimer t(h, "reduced_cholesky_solve");
                    for (int w = 0; w < n_win; w++) {
                        const WinDev& d = h->wins[w].d;
                        if (!d.ld) continue;
                        double* Sw = h->d_S.p + d.S_off;
                        double* yw = h->d_gred.p + d.red_off;
                        int* info = h->d_big_info.p + w;
                        const int* skip = (const int*)((const char*)(h->d_states.p + (size_t)w * stride + s) + offsetof(LmState, done));
                        const int N = d.Np, bw = big_bw[w];
                        const int nb = d.dpf == 6 ? 6 : 5;
                        if (bw < N && bw + nb <= MAX_LDS_NP) {
                            // block-banded system: one workgroup slides an LDS window down the band (dense_chol.h)
                            // with the update trimmed to the band a step costs the same in any window: the largest window that fits
                            // amortises the per-window carry / load / store best (SADVIO_BAND_C overrides, for measurements)
                            int C = std::max(nb, (MAX_LDS_NP - bw) / nb * nb);
                            if (h->env.band_c > 0) C = std::max(nb, std::min(C, h->env.band_c / nb * nb));
                            const int Rmax = bw + C;
                            const size_t lds = sizeof(double) * ((size_t)(Rmax + 2) * 6 + (size_t)(Rmax + 1) * (Rmax + 2) / 2 + 2 * (size_t)Rmax +
                                                                 (size_t)(Rmax / nb + 1) * nb * nb) + 64;
                            auto kbs = d.dpf == 6 ? k_band_solve<6> : k_band_solve<5>;
                            (void)hipFuncSetAttribute((const void*)kbs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                            long long* dbg = (P.debug & 4096) && s == 3 ? h->d_dbg.p + 44 : nullptr;
                            double* lv = h->d_big_linv.p + big_linv_off[w];
                            const int Kb = (N + bw - 1) / bw;
                            if (nb == 6 && Kb >= 22 && 2 * bw <= MAX_LDS_NP - 1 && n_win == 1 && !h->env.no_bcr) {   // below ~22 blocks the twisted solver wins (measured)
                                // very long band: block cyclic reduction over the bw x bw blocks, log2(K) levels (dense_chol.h)
                                const size_t b = (size_t)bw, bb = b * b, K = (size_t)Kb;
                                const size_t per = 8 * bb + b * (b + 1) / 2 + (b / 6) * 36 + 5 * b;
                                BcrPtrs B{};
                                double* q = h->d_bcr.p;
                                B.D = q; q += K * bb; B.E = q; q += K * bb; B.Wp = q; q += K * bb; B.Wn = q; q += K * bb;
                                B.Ul = q; q += K * bb; B.Ur = q; q += K * bb;
                                B.Lp = q; q += K * (b * (b + 1) / 2); B.linv = q; q += K * (b / 6) * 36;
                                B.g = q; q += K * b; B.yv = q; q += K * b; B.gl = q; q += K * b; B.gr = q; q += K * b; B.X = q; q += K * b;
                                B.K = Kb; B.b = bw; B.N = N;
                                const int R2 = 2 * bw;
                                const size_t lds_e = sizeof(double) * ((size_t)(R2 + 2) * 6 + (size_t)(R2 + 1) * (R2 + 2) / 2 + 2 * (size_t)R2 + (size_t)(bw / 6 + 1) * 36) + 64;
                                const size_t lds_c = sizeof(double) * 2 * b * (b + 1);
                                const size_t lds_r = lds_e + sizeof(double) * (size_t)(bw / 6 + 1) * 36;   // up to two blocks, all 2 b / 6 pivot blocks kept
                                const size_t lds_b = sizeof(double) * (b * (b + 1) / 2 + 4 * b + (b / 6) * 36);
                                (void)hipFuncSetAttribute((const void*)k_bcr_elim<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e);
                                (void)hipFuncSetAttribute((const void*)k_bcr_combine, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
                                (void)hipFuncSetAttribute((const void*)k_bcr_root<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
                                (void)hipFuncSetAttribute((const void*)k_bcr_back<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
                                hipLaunchKernelGGL(k_bcr_extract, dim3(Kb), dim3(256), 0, h->stream, Sw, (long long)d.ld, yw, B, skip, info);
                                int smax = 0, s_root = 0;
                                for (int sst = 1; sst < Kb; sst *= 2) {
                                    const int na = (Kb + sst - 1) / sst;
                                    if (na == 2) { s_root = sst; break; }   // two blocks left: solved together by k_bcr_root
                                    hipLaunchKernelGGL(k_bcr_elim<6>, dim3(na / 2, 2), dim3(SOLVE_THREADS), lds_e, h->stream, B, sst, info, skip);
                                    hipLaunchKernelGGL(k_bcr_combine, dim3(na), dim3(512), lds_c, h->stream, B, sst, info, skip);
                                    smax = sst;
                                }
                                hipLaunchKernelGGL(k_bcr_root<6>, dim3(1), dim3(SOLVE_THREADS), lds_r, h->stream, B, s_root, info, skip);
                                for (int sst = smax; sst >= 1; sst /= 2) {
                                    const int na = (Kb + sst - 1) / sst;
                                    hipLaunchKernelGGL(k_bcr_back<6>, dim3(na / 2), dim3(256), lds_b, h->stream, B, sst, info, skip);
                                }
                                hipLaunchKernelGGL(k_bcr_writeback, dim3((N + 255) / 256), dim3(256), 0, h->stream, B, yw, info, skip);
                                if (dbg) fprintf(stderr, "[sadvio dbg] block cyclic reduction N %d bw %d K %d\n", N, bw, Kb);
                            } else if (N - bw >= 4 * C) {
                                // long band: twisted factorisation, both ends at once (dense_chol.h)
                                const int M = (N - bw) / 2 / nb * nb;
                                double* md = h->d_big_mid.p + big_mid_off[w];
                                auto kbm = d.dpf == 6 ? k_band_mid<6> : k_band_mid<5>;
                                const size_t lds_m = sizeof(double) * ((size_t)(bw + 2) * 6 + (size_t)(bw + 1) * (bw + 2) / 2 + 2 * (size_t)bw + (size_t)(bw / nb + 1) * nb * nb) + 64;
                                (void)hipFuncSetAttribute((const void*)kbm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
                                if (dbg) fprintf(stderr, "[sadvio dbg] twisted band solve N %d bw %d C %d M %d\n", N, bw, C, M);
                                hipLaunchKernelGGL(kbs, dim3(2), dim3(SOLVE_THREADS), lds, h->stream, Sw, (long long)d.ld, yw, lv, N, bw, C, info, skip, dbg, M, 0, md);
                                hipLaunchKernelGGL(kbm, dim3(1), dim3(SOLVE_THREADS), lds_m, h->stream, Sw, (long long)d.ld, yw, N, bw, M, md, info, skip);
                                hipLaunchKernelGGL(kbs, dim3(2), dim3(SOLVE_THREADS), lds, h->stream, Sw, (long long)d.ld, yw, lv, N, bw, C, info, skip, dbg, M, 1, md);
                            } else {
                                hipLaunchKernelGGL(kbs, dim3(1), dim3(SOLVE_THREADS), lds, h->stream, Sw, (long long)d.ld, yw, lv, N, bw, C, info, skip, dbg, -1, 0, (double*)nullptr);
                            }
                            continue;
                        }
                        if (N >= 2 * WD) {
                            // dense system: 96-column panels through the LDS solver + MFMA panel / update kernels
                            double* Mw = h->d_big_M.p + big_M_off[w];
                            const size_t lds_d = sizeof(double) * ((size_t)(WD + 2) * 6 + (size_t)(WD + 1) * (WD + 2) / 2 + 2 * WD + 2 * WD_NBK * 36 + (size_t)WD * (WD + 1)) + 64;
                            const size_t lds_t = sizeof(double) * (size_t)(CH_TS + WD) * WDS;
                            const size_t lds_s = sizeof(double) * 2 * (size_t)CH_TS * WDS;
                            (void)hipFuncSetAttribute((const void*)k_wchol_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d);
                            (void)hipFuncSetAttribute((const void*)k_wchol_diag16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * wd16_lds_doubles() + 64));
                            const bool wd16 = !h->env.wd_old;   // the 96 x 96 diagonal blocks on 16 x 16 MFMA tiles (chol16.h); the 6-column LDS solver for A/B runs
                            (void)hipFuncSetAttribute((const void*)k_wchol_trsm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
                            (void)hipFuncSetAttribute((const void*)k_wchol_syrk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
                            (void)hipFuncSetAttribute((const void*)k_wchol_backsolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * WD * WDS));
                            const bool wdla = wd16 && !h->env.wd_nola;   // look-ahead: the next diagonal block inside the trailing-update launch
                            const size_t lds_la = std::max(sizeof(double) * wdla_lds_doubles() + 64, lds_s);
                            (void)hipFuncSetAttribute((const void*)k_wchol_syrk_la, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_la);
                            (void)hipFuncSetAttribute((const void*)k_wchol_trsm8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t);
                            const bool wdstep = wdla && !h->env.wd_r3;     // one launch per panel (round 4); SADVIO_WD_R3 = the trsm8 + syrk_la loop
                            if (wdstep) {
                                const int nsteps = (N + WD - 1) / WD;
                                double* Ltw = Mw + (size_t)nsteps * WD * WD;
                                double* Lx = h->d_big_Lx.p + big_Lx_off[w];
                                const size_t lds_st = sizeof(double) * wdstep_lds_doubles() + 64;
                                (void)hipFuncSetAttribute((const void*)k_wchol_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_st);
                                (void)hipFuncSetAttribute((const void*)k_wchol_backstep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * WD * WDS));
                                hipLaunchKernelGGL(k_wchol_diag16, dim3(1), dim3(SOLVE_THREADS), sizeof(double) * wd16_lds_doubles() + 64, h->stream, Sw, (long long)d.ld, yw, (double*)nullptr, N, 0, info, skip, Ltw);
                                for (int st = 0; st + 1 < nsteps; st++) {
                                    const int c0 = st * WD, m = N - (c0 + WD);
                                    const int nt = (m + CH_TS - 1) / CH_TS;
                                    hipLaunchKernelGGL(k_wchol_step, dim3(nt * (nt + 1) / 2 + 2), dim3(SOLVE_THREADS), lds_st, h->stream, Sw, (long long)d.ld, Lx, yw,
                                                       Ltw + (size_t)st * WD_LT, Ltw + (size_t)(st + 1) * WD_LT, Mw + (size_t)st * WD * WD, Mw + (size_t)(st + 1) * WD * WD, N, c0, info, skip,
                                                       (P.debug & 4096) && s == 3 && st == 1 ? h->d_dbg.p + 106 : (long long*)nullptr);
                                }
                                for (int bs = nsteps - 1; bs >= 0; bs--)
                                    hipLaunchKernelGGL(k_wchol_backstep, dim3(1 + (bs + 1 < nsteps ? (bs * WD + 63) / 64 : 0)), dim3(SOLVE_THREADS), sizeof(double) * WD * WDS, h->stream,
                                                       Lx, (long long)d.ld, yw, Mw, N, bs, info, skip);
                                continue;
                            }
                            int st = 0;
                            for (int c0 = 0; c0 < N; c0 += WD, st++) {
                                if (wdla) {
                                    if (c0 == 0) hipLaunchKernelGGL(k_wchol_diag16, dim3(1), dim3(SOLVE_THREADS), sizeof(double) * wd16_lds_doubles() + 64, h->stream, Sw, (long long)d.ld, yw, Mw, N, 0, info, skip);
                                    const int m = N - (c0 + WD);
                                    if (m > 0) {
                                        const int nt = (m + CH_TS - 1) / CH_TS;
                                        hipLaunchKernelGGL(k_wchol_trsm8, dim3(nt), dim3(SOLVE_THREADS), lds_t, h->stream, Sw, (long long)d.ld, Mw + (size_t)st * WD * WD, N, c0, info, skip);
                                        hipLaunchKernelGGL(k_wchol_syrk_la, dim3(nt * (nt + 1) / 2 + (m + SOLVE_THREADS - 1) / SOLVE_THREADS + 1), dim3(SOLVE_THREADS), lds_la, h->stream,
                                                           Sw, (long long)d.ld, yw, Mw + (size_t)(st + 1) * WD * WD, N, c0, info, skip,
                                                           (P.debug & 4096) && s == 3 && c0 == WD ? h->d_dbg.p + 90 : (long long*)nullptr);
                                    }
                                    continue;
                                }
                                if (wd16) hipLaunchKernelGGL(k_wchol_diag16, dim3(1), dim3(SOLVE_THREADS), sizeof(double) * wd16_lds_doubles() + 64, h->stream, Sw, (long long)d.ld, yw, Mw + (size_t)st * WD * WD, N, c0, info, skip);
                                else
                                hipLaunchKernelGGL(k_wchol_diag, dim3(1), dim3(SOLVE_THREADS), lds_d, h->stream, Sw, (long long)d.ld, yw, Mw + (size_t)st * WD * WD, N, c0, info, skip);
                                const int m = N - (c0 + WD);
                                if (m > 0) {
                                    const int nt = (m + CH_TS - 1) / CH_TS;
                                    hipLaunchKernelGGL(k_wchol_trsm, dim3(nt), dim3(CH_THREADS), lds_t, h->stream, Sw, (long long)d.ld, Mw + (size_t)st * WD * WD, N, c0, info, skip);
                                    hipLaunchKernelGGL(k_wchol_syrk, dim3(nt * (nt + 1) / 2 + (m + CH_THREADS - 1) / CH_THREADS), dim3(CH_THREADS), lds_s, h->stream, Sw, (long long)d.ld, yw, N, c0, info, skip);
                                }
                            }
                            if (wdla && !h->env.wd_back1) {
                                // back-substitution: one launch per super-step (dense_chol.h: k_wchol_backstep)
                                (void)hipFuncSetAttribute((const void*)k_wchol_backstep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * WD * WDS));
                                const int nsteps = (N + WD - 1) / WD;
                                for (int bs = nsteps - 1; bs >= 0; bs--)
                                    hipLaunchKernelGGL(k_wchol_backstep, dim3(1 + (bs + 1 < nsteps ? (bs * WD + 63) / 64 : 0)), dim3(SOLVE_THREADS), sizeof(double) * WD * WDS, h->stream,
                                                       Sw, (long long)d.ld, yw, Mw, N, bs, info, skip);
                            } else
                            hipLaunchKernelGGL(k_wchol_backsolve, dim3(1), dim3(SOLVE_THREADS), sizeof(double) * WD * WDS, h->stream, Sw, (long long)d.ld, yw, Mw, N, info, skip);
                            continue;
                        }
                        for (int k0 = 0; k0 < N; k0 += CH_NB) {
                            const int nb = std::min(CH_NB, N - k0), s0 = k0 + nb;
                            const int rows_end = std::min(N, s0 + bw), m = rows_end - s0;
                            hipLaunchKernelGGL(k_chol_panel, dim3((m + 1 + CH_THREADS - 1) / CH_THREADS), dim3(CH_THREADS), 0, h->stream,
                                               Sw, (long long)d.ld, yw, N, k0, rows_end, info, skip, (P.debug & 4096) && s == 3 && k0 == 64 ? h->d_dbg.p + 44 : nullptr);
                            if (m > 0) {
                                const int nt = (m + CH_TS - 1) / CH_TS;
                                hipLaunchKernelGGL(k_chol_update, dim3(nt * (nt + 1) / 2 + (m + CH_THREADS - 1) / CH_THREADS), dim3(CH_THREADS), 0,
                                                   h->stream, Sw, (long long)d.ld, yw, N, k0, rows_end, info, skip);
                            }
                        }
                        hipLaunchKernelGGL(k_chol_backsolve, dim3(1), dim3(CH_THREADS), 0, h->stream, Sw, (long long)d.ld, yw, N, bw, info, skip);
                    }
                }
                { ScopedTimer t(h, "k_solve_back"); hipLaunchKernelGGL(ks2, dim3(n_win), dim3(SOLVE_THREADS), 64, h->stream, P, s); }
                for (int w = 0; w < n_win; w++) {
                    const WinDev& d = h->wins[w].d;
                    if (!d.ld) continue;
                    if (big_bw[w] < d.Np && big_bw[w] + (d.dpf == 6 ? 6 : 5) <= MAX_LDS_NP)   // banded: only the band was written
                        hipLaunchKernelGGL(k_band_zero, dim3((unsigned)std::min<long long>(((long long)d.Np * big_bw[w] + 255) / 256, 4096)), dim3(256), 0, h->stream,
                                           h->d_S.p + d.S_off, (long long)d.ld, d.Np, big_bw[w]);
                    else (void)hipMemsetAsync(h->d_S.p + d.S_off, 0, sizeof(double) * (size_t)d.Np * d.Np, h->stream);
                }
            }
            if (fork) {
                (void)hipEventRecord(h->ev_solved, h->stream);
                (void)hipStreamWaitEvent(h->side, h->ev_solved, 0);
                if (n_lo) hipLaunchKernelGGL(k_line_eval<false>, dim3(n_lo), dim3(64), 0, h->side, P, s, 0);
                (void)hipEventRecord(h->ev_cost, h->side);
            } else {
                if (n_lo) { ScopedTimer t(h, "k_line_cost"); hipLaunchKernelGGL(k_line_eval<false>, dim3(n_lo), dim3(64), 0, h->stream, P, s, 0); }
            }
            if (dp_max_nf > 0) { ScopedTimer t(h, "k_prior_m"); hipLaunchKernelGGL(k_prior_m, dim3((dp_max_nf + 3) / 4, n_win), dim3(256), 0, h->stream, P, s); }
            if (n_pf && (use_lm || !with_imu)) { ScopedTimer t(h, "k_pf_cost"); hipLaunchKernelGGL(k_pf_eval<true>, dim3(n_pf), dim3(64), 0, h->stream, P, s); }
            if (use_lm) {
                ScopedTimer t(h, "k_lm_pass"); hipLaunchKernelGGL(kps, dim3(h->lm_n_sub), dim3(LM_PASS_THREADS), lds_pass, h->stream, P, s, mtk, h->lm_sub_obs);
            }
            else
            { ScopedTimer t(h, "k_backsub"); hipLaunchKernelGGL(kk, dim3(n_tiles + (with_imu ? n_pf : 0)), dim3(BUILD_THREADS), lds_back, h->stream, P, s, mtk); }
            if (fork) (void)hipStreamWaitEvent(h->stream, h->ev_cost, 0);
            if (h->coll_fn) {
                { ScopedTimer t(h, "k_rank_partials"); hipLaunchKernelGGL(k_rank_partials, dim3(n_win), dim3(64), 0, h->stream, P, s, 1); }
                ScopedTimer t(h, "allreduce_step_partials");
                if (h->coll_fn(h->coll_ctx, h->d_rank_s.p, (int64_t)n_win * h->world * 4, (void*)h->stream) != 0) coll_failed = true;
            }
            if (P.decide_kernel) { ScopedTimer t(h, "k_decide"); hipLaunchKernelGGL(k_decide, dim3(n_win), dim3(max_win_tiles > 4 * BUILD_THREADS ? 1024 : 64), 0, h->stream, P, s, 0); }
        }
        { ScopedTimer t(h, "k_final"); hipLaunchKernelGGL(k_decide, dim3(n_win), dim3(64), 0, h->stream, P, slots - 1, 1); }
    };
    if (h->cfg.use_graph && !h->cfg.profile_kernels && !h->coll_fn) {
        // the whole <= 20-iteration solve is one graph launch; the key covers every kernel argument
        std::vector<int> lay;  // layout-dependent launch parameters of the out-of-LDS windows
        for (int w = 0; w < n_win; w++) { lay.push_back(h->wins[w].d.Np); lay.push_back(h->wins[w].d.ld); lay.push_back(big_bw[w]); }
        lay.push_back(h->n_kept); lay.push_back(h->lm_sub_obs); lay.push_back(h->lm_n_sub); lay.push_back(h->lm_max_cam); lay.push_back(h->n_lobs_tot); lay.push_back(h->n_line_tot); lay.push_back(h->n_big); lay.push_back(dp_max_nf); lay.push_back(dp_max_n);
        std::vector<unsigned char> key(sizeof(DevPtrs) + 8 * sizeof(int) + 3 * sizeof(size_t) + lay.size() * sizeof(int));
        unsigned char* kp = key.data();
        memcpy(kp + sizeof(DevPtrs) + 8 * sizeof(int) + 3 * sizeof(size_t), lay.data(), lay.size() * sizeof(int));
        memcpy(kp, &P, sizeof(DevPtrs)); kp += sizeof(DevPtrs);
        // launch-shape switches read from the environment inside enqueue() are part of the key too: a handle that already captured a
        // graph must not replay it when an A/B switch changes
        const int env_bits = (int)with_imu + 2 * (int)h->env.no_fork + 4 * (int)h->env.wd_nola + 8 * (int)h->env.wd_back1 +
                             16 * (int)h->env.wd_old + 32 * (int)h->env.no_bcr + 128 * (int)h->env.no_lpt +
                             256 * (int)h->env.wd_r3;
        const int ints[8] = {slots, n_tiles, n_win, mtk, strip_doubles, Rp, h->n_kf_tot, h->factor_type + 2 * (int)extras + 4 * (int)rare + 8 * (int)use_lm + 16 * env_bits};  // P (incl. decide_kernel) is part of the key
        memcpy(kp, ints, sizeof(ints)); kp += sizeof(ints);
        const size_t szs[3] = {lds_build, lds_back, lds_solve};
        memcpy(kp, szs, sizeof(szs));
        if (!h->graph_exec || key != h->graph_key) {
            if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
            enqueue();
            HIP_TRY(hipStreamEndCapture(h->stream, &g));
            HIP_TRY(hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            h->graph_key = key;
        }
        HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
    } else {
        enqueue();
    }
    HIP_TRY(hipGetLastError());
    if (coll_failed) { h->err = "solve: the all-reduce of the reduced system failed"; return SADVIO_E_RCCL; }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->cfg.profile_kernels) collect_timers(h);
    if ((h->env.debug & 4096)) {
        long long ts[128];
        if (hipMemcpy(ts, h->d_dbg.p, sizeof(ts), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[sadvio dbg] phase dt (us):");
            for (int i = 1; i < 16; i++) fprintf(stderr, " %d:%.2f", i, (ts[i] - ts[0]) * 0.01);
            fprintf(stderr, "  shader clock %.3f GHz\n[sadvio dbg] k_build:", (double)(ts[21] - ts[20]) / ((ts[15] - ts[0]) * 10.0));
            for (int i = 33; i < 43; i++) fprintf(stderr, " %d:%.2f", i, (ts[i] - ts[32]) * 0.01);
            fprintf(stderr, "\n[sadvio dbg] first IMU pair of k_build, us since its start (residual + Jacobian on lane 0 | decision | W J | entries + adds): %.2f %.2f %.2f %.2f, start %.2f us after tile 0",
                    (ts[57] - ts[56]) * 0.01, (ts[58] - ts[56]) * 0.01, (ts[59] - ts[56]) * 0.01, (ts[60] - ts[56]) * 0.01, (ts[56] - ts[32]) * 0.01);
            fprintf(stderr, "\n[sadvio dbg] look-ahead workgroup of k_wchol_syrk_la, us since its start (loads landed, X slab in LDS | tiles updated | factored + inverted):");
            fprintf(stderr, " %.2f %.2f %.2f", (ts[91] - ts[90]) * 0.01, (ts[94] - ts[90]) * 0.01, (ts[95] - ts[90]) * 0.01);
            fprintf(stderr, " | waves after their tiles:"); for (int i = 98; i < 106; i++) fprintf(stderr, " %.2f", (ts[i] - ts[90]) * 0.01);
            fprintf(stderr, "\n[sadvio dbg] k_wchol_step (panel 1), us since the workgroup's start: look-ahead (operands in LDS | substituted | block updated | factored) %.2f %.2f %.2f %.2f",
                    (ts[107] - ts[106]) * 0.01, (ts[108] - ts[106]) * 0.01, (ts[109] - ts[106]) * 0.01, (ts[110] - ts[106]) * 0.01);
            fprintf(stderr, " | tile workgroup 7, %.2f us after it (operands | substituted | end) %.2f %.2f %.2f | inverse workgroup, %.2f us after it: %.2f",
                    (ts[114] - ts[106]) * 0.01, (ts[115] - ts[114]) * 0.01, (ts[116] - ts[114]) * 0.01, (ts[117] - ts[114]) * 0.01, (ts[120] - ts[106]) * 0.01, (ts[121] - ts[120]) * 0.01);
            fprintf(stderr, "\n[sadvio dbg] chol16 cycles since its first barrier (panel | trailing + next pivot, per block column):");
            for (int i = 65; i < 81; i++) fprintf(stderr, " %lld", ts[i] - ts[64]);
            fprintf(stderr, " | end %lld", ts[84] - ts[64]);
            fprintf(stderr, "\n[sadvio dbg] k_chol_panel / k_band_solve (fwd window 2: carry fresh chol store | fwd end | bwd window 2: load below steps | bwd end):");
            for (int i = 45; i < 55; i++) fprintf(stderr, " %d:%.2f", i - 44, (ts[i] - ts[44]) * 0.01);
            fprintf(stderr, "\n");
        }
    }
    h->fin.assign(h->h_final, h->h_final + n_win);
    h->deltas_cached = false;
    h->last_slots = slots;
    h->solved = true;
    int rc = SADVIO_OK;
    for (int w = 0; w < n_win; w++) {
        const LmState& s = h->fin[w].s;
        if (summaries) {
            sadvio_solve_summary& S = summaries[w];
            S.iterations = s.iter; S.num_successful_steps = s.n_success; S.num_unsuccessful_steps = s.n_unsuccess;
            S.termination = s.termination; S.initial_cost = s.initial_cost; S.final_cost = s.x_cost;
            S.fixed_cost = 0.5 * h->fin[w].fixed_cost; S.final_radius = s.radius;
        }
        if (s.termination == SADVIO_TERM_FAILURE) rc = SADVIO_E_NOT_USABLE;
    }
    // remember which buffer holds x for each window
    (void)h->d_probe.alloc(1);
    return rc;
}

int sadvio_ba_get_deltas(sadvio_ba_handle* h, int32_t w, double* pose, double* lmk, double* dv, double* dba, double* dbg) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->solved) { h->err = "get_deltas before solve"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size()) { h->err = "get_deltas: window out of range"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const WinDev& d = h->wins[w].d;
    const int cur = h->fin[w].s.cur;
    // One read-back per solve: both buffers of every delta array go to a pinned host buffer with asynchronous copies and ONE
    // synchronisation (a pageable hipMemcpy per array and window costs 30 - 80 us each); every get_deltas of this solve is then
    // a host memcpy. Layout: xp [2][6 n_kf] | xl [2][3 n_lmk] | xv | xba | xbg [2][3 n_kf] each.
    const size_t nk = (size_t)h->n_kf_tot, nl = (size_t)h->n_lmk_tot;
    const size_t o_xp = 0, o_xl = o_xp + 12 * nk, o_xv = o_xl + 6 * nl, o_xba = o_xv + 6 * nk, o_xbg = o_xba + 6 * nk, total = o_xbg + 6 * nk;
    if (!h->deltas_cached) {
        if (h->h_deltas_cap < total) {
            if (h->h_deltas) (void)hipHostFree(h->h_deltas);
            h->h_deltas = nullptr; h->h_deltas_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&h->h_deltas, sizeof(double) * (total + total / 2), hipHostMallocDefault));
            h->h_deltas_cap = total + total / 2;
        }
        bool any_imu = false;
        for (const auto& hw : h->wins) any_imu |= hw.d.has_imu != 0;
        HIP_TRY(hipMemcpyAsync(h->h_deltas + o_xp, h->d_xp.p, sizeof(double) * 12 * nk, hipMemcpyDeviceToHost, h->stream));
        if (nl) HIP_TRY(hipMemcpyAsync(h->h_deltas + o_xl, h->d_xl.p, sizeof(double) * 6 * nl, hipMemcpyDeviceToHost, h->stream));
        if (any_imu) {
            HIP_TRY(hipMemcpyAsync(h->h_deltas + o_xv, h->d_xv.p, sizeof(double) * 6 * nk, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipMemcpyAsync(h->h_deltas + o_xba, h->d_xba.p, sizeof(double) * 6 * nk, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipMemcpyAsync(h->h_deltas + o_xbg, h->d_xbg.p, sizeof(double) * 6 * nk, hipMemcpyDeviceToHost, h->stream));
        } else {
            memset(h->h_deltas + o_xv, 0, sizeof(double) * 18 * nk);   // windows without IMU states: their deltas are never touched (zero)
        }
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->deltas_cached = true;
    }
    if (pose) memcpy(pose, h->h_deltas + o_xp + (size_t)cur * 6 * nk + 6 * (size_t)d.kf_base, sizeof(double) * 6 * d.n_kf);
    if (lmk && d.n_lmk) memcpy(lmk, h->h_deltas + o_xl + (size_t)cur * 3 * nl + 3 * (size_t)d.lmk_base, sizeof(double) * 3 * d.n_lmk);
    double* outs[3] = {dv, dba, dbg};
    const size_t offs[3] = {o_xv, o_xba, o_xbg};
    for (int q = 0; q < 3; q++)
        if (outs[q]) memcpy(outs[q], h->h_deltas + offs[q] + (size_t)cur * 3 * nk + 3 * (size_t)d.kf_base, sizeof(double) * 3 * d.n_kf);
    return SADVIO_OK;
}

int sadvio_ba_get_trace(sadvio_ba_handle* h, int32_t w, int32_t cap_rows, double* rows8, int32_t* n_rows) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->solved) { h->err = "get_trace before solve"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size() || cap_rows < 0 || (cap_rows > 0 && !rows8)) { h->err = "get_trace: bad argument"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const int stride = h->last_slots + 2;
    const int n = std::min(h->fin[w].s.iter + 1, stride);
    if (n_rows) *n_rows = n;
    const int m = std::min(n, cap_rows);
    if (m > 0) HIP_TRY(hipMemcpy(rows8, h->d_trace.p + (size_t)w * stride * 8, sizeof(double) * 8 * (size_t)m, hipMemcpyDeviceToHost));
    return SADVIO_OK;
}

int sadvio_ba_get_ids(sadvio_ba_handle* h, int32_t w, int64_t* kf_id, int64_t* lmk_id) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (w < 0 || w >= (int)h->wins.size()) { h->err = "get_ids: window out of range"; return SADVIO_E_INVALID_ARG; }
    if (kf_id) memcpy(kf_id, h->wins[w].kf_id.data(), sizeof(int64_t) * h->wins[w].kf_id.size());
    if (lmk_id) memcpy(lmk_id, h->wins[w].lmk_id.data(), sizeof(int64_t) * h->wins[w].lmk_id.size());
    return SADVIO_OK;
}

int sadvio_ba_linearize(sadvio_ba_handle* h, int32_t w, const double* pose_delta6, const double* lmk_delta3, double* r2,
                        double* Jp12, double* Jl6) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "linearize before set_windows"; return SADVIO_E_STATE; }
    if (h->defer) { h->err = "linearize between begin_update and commit_update"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size()) { h->err = "linearize: window out of range"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const WinDev& d = h->wins[w].d;
    HIP_TRY(hipMemsetAsync(h->d_xp.p, 0, sizeof(double) * h->d_xp.n, h->stream));
    HIP_TRY(hipMemsetAsync(h->d_xl.p, 0, sizeof(double) * h->d_xl.n, h->stream));
    if (pose_delta6) HIP_TRY(hipMemcpyAsync(h->d_xp.p + 6 * (size_t)d.kf_base, pose_delta6, sizeof(double) * 6 * d.n_kf, hipMemcpyHostToDevice, h->stream));
    if (lmk_delta3 && d.n_lmk) HIP_TRY(hipMemcpyAsync(h->d_xl.p + 3 * (size_t)d.lmk_base, lmk_delta3, sizeof(double) * 3 * d.n_lmk, hipMemcpyHostToDevice, h->stream));
    if (d.n_obs == 0) { HIP_TRY(hipStreamSynchronize(h->stream)); return SADVIO_OK; }
    HIP_TRY(h->d_probe.alloc(20 * (size_t)d.n_obs));
    SolveOpts o{};
    DevPtrs P = make_ptrs(h, o, 1);
    double* pr = h->d_probe.p; double* pj = pr + 2 * (size_t)d.n_obs; double* pl = pj + 12 * (size_t)d.n_obs;
    int blocks = (d.n_obs + 255) / 256;
    if (h->factor_type == SADVIO_FACTOR_PIXEL) hipLaunchKernelGGL(k_linearize_probe<0>, dim3(blocks), dim3(256), 0, h->stream, P, w, pr, pj, pl);
    else hipLaunchKernelGGL(k_linearize_probe<1>, dim3(blocks), dim3(256), 0, h->stream, P, w, pr, pj, pl);
    HIP_TRY(hipGetLastError());
    std::vector<double> hb(20 * (size_t)d.n_obs);
    HIP_TRY(hipMemcpyAsync(hb.data(), pr, sizeof(double) * 20 * (size_t)d.n_obs, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int a = 0; a < d.n_obs; a++) {
        const int src = h->obs_perm[d.obs_base + a];  // caller's index of the observation stored at position a
        if (src < 0) continue;                          // pseudo-observation
        if (r2) memcpy(r2 + 2 * (size_t)src, &hb[2 * (size_t)a], 16);
        if (Jp12) memcpy(Jp12 + 12 * (size_t)src, &hb[2 * (size_t)d.n_obs + 12 * (size_t)a], 96);
        if (Jl6) memcpy(Jl6 + 6 * (size_t)src, &hb[14 * (size_t)d.n_obs + 6 * (size_t)a], 48);
    }
    h->solved = false;
    return SADVIO_OK;
}

int sadvio_ba_vi_init(sadvio_ba_handle* h, const sadvio_viinit_problem* pb, const sadvio_solve_options* opts, sadvio_solve_summary* sum,
                      sadvio_viinit_result* res, double* dv3) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!pb || !opts || pb->n_frames < 0 || pb->n_factors < 0 || (pb->n_frames > 0 && (!pb->T_f_w || !pb->vel)) || (pb->n_factors > 0 && !pb->factors)) {
        h->err = "vi_init: bad argument"; return SADVIO_E_INVALID_ARG;
    }
    if (pb->n_frames > VIINIT_MAX_FRAMES) { h->err = "vi_init: more than 48 frames"; return SADVIO_E_INVALID_ARG; }
    if (pb->optim_bias && !(pb->sigma_dba > 0.0 && pb->sigma_dbg > 0.0)) { h->err = "vi_init: bias prior sigmas must be positive"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const int n = pb->n_frames, nf = pb->n_factors;
    // program layout: r_wi | velocity deltas of the frames a factor touches (frame order) | dba dbg | lambda
    std::vector<int> vcol(std::max(n, 1), -1);
    std::vector<ImuDev> fd(std::max(nf, 1));
    for (int k = 0; k < nf; k++) {
        const sadvio_imu_factor& f = pb->factors[k];
        if (f.kf_i < 0 || f.kf_i >= n || f.kf_j < 0 || f.kf_j >= n || f.kf_i == f.kf_j) { h->err = "vi_init: factor frame index out of range"; return SADVIO_E_INVALID_ARG; }
        if (!make_imu_dev(f, 0, fd[k])) { h->err = "vi_init: IMU covariance is not positive definite"; return SADVIO_E_INVALID_ARG; }
        vcol[f.kf_i] = vcol[f.kf_j] = 0;
    }
    int D = 2;
    for (int i = 0; i < n; i++) if (vcol[i] == 0) { vcol[i] = D; D += 3; }
    ViInitDev P{};
    P.n_frames = n; P.n_factors = nf; P.optim_bias = pb->optim_bias ? 1 : 0;
    P.c_ba = P.c_bg = P.c_l = -1;
    if (pb->optim_bias) { P.c_ba = D; P.c_bg = D + 3; D += 6; P.isig_ba = 1.0 / pb->sigma_dba; P.isig_bg = 1.0 / pb->sigma_dbg; }
    if (pb->optim_scale) { P.c_l = D; D += 1; }
    P.D = D;
    std::vector<double> out((size_t)D + 8, 0.0);
    sadvio_solve_summary S{};
    if (nf == 0) S.termination = SADVIO_TERM_GRADIENT_TOL;   // empty program: Ceres returns at once
    else {
        P.o.max_num_iterations = opts->max_num_iterations; P.o.jacobi_scaling = opts->jacobi_scaling;
        P.o.max_num_consecutive_invalid_steps = opts->max_num_consecutive_invalid_steps;
        P.o.function_tolerance = opts->function_tolerance; P.o.gradient_tolerance = opts->gradient_tolerance;
        P.o.parameter_tolerance = opts->parameter_tolerance; P.o.initial_radius = opts->initial_trust_region_radius;
        P.o.max_radius = opts->max_trust_region_radius; P.o.min_radius = opts->min_trust_region_radius;
        P.o.min_lm_diagonal = opts->min_lm_diagonal; P.o.max_lm_diagonal = opts->max_lm_diagonal; P.o.min_relative_decrease = opts->min_relative_decrease;
        // one allocation: header | T | vel | out | scratch | factors | vcol
        const size_t n_d = 12 * (size_t)n + 3 * (size_t)n + out.size() + 2 * (size_t)nf * VIINIT_FJ;
        const size_t bytes = sizeof(ViInitDev) + 8 * n_d + sizeof(ImuDev) * (size_t)nf + sizeof(int) * (size_t)n + 64;
        DevBuf<char> buf;
        HIP_TRY(buf.alloc(bytes));
        char* base = buf.p;
        double* dT = (double*)(base + ((sizeof(ViInitDev) + 15) & ~(size_t)15));
        double* dvel = dT + 12 * (size_t)n; double* dout = dvel + 3 * (size_t)n; double* dscr = dout + out.size();
        ImuDev* df = (ImuDev*)(dscr + 2 * (size_t)nf * VIINIT_FJ);
        int* dvc = (int*)(df + nf);
        P.T = dT; P.vel = dvel; P.out = dout; P.scratch = dscr; P.f = df; P.vcol = dvc;
        HIP_TRY(hipMemcpyAsync(base, &P, sizeof(P), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dT, pb->T_f_w, 96 * (size_t)n, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dvel, pb->vel, 24 * (size_t)n, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(df, fd.data(), sizeof(ImuDev) * (size_t)nf, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(dvc, vcol.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
        const size_t lds = 8 * ((size_t)(D + 1) * (D + 2) / 2 + 6 * (size_t)D);
        HIP_TRY(hipFuncSetAttribute((const void*)k_viinit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_viinit, dim3(1), dim3(VIINIT_THREADS), lds, h->stream, (const ViInitDev*)base);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out.data(), dout, 8 * out.size(), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        const double* s = out.data() + D;
        S.initial_cost = s[0]; S.final_cost = s[1]; S.final_radius = s[2]; S.iterations = (int)s[3]; S.termination = (int)s[4];
        S.num_successful_steps = (int)s[5]; S.num_unsuccessful_steps = (int)s[6];
    }
    if (sum) *sum = S;
    if (res) {
        memset(res, 0, sizeof(*res));
        res->r_wi[0] = out[0]; res->r_wi[1] = out[1];
        res->lambda = P.c_l >= 0 ? out[P.c_l] : 0.0;
        for (int a = 0; a < 3; a++) { res->dba[a] = P.c_ba >= 0 ? out[P.c_ba + a] : 0.0; res->dbg[a] = P.c_bg >= 0 ? out[P.c_bg + a] : 0.0; }
        host_exp_so3(res->r_wi[0], res->r_wi[1], res->R_w_i);
        res->scale = std::exp(res->lambda);
    }
    if (dv3)
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) dv3[3 * i + a] = vcol[i] >= 0 ? out[vcol[i] + a] : 0.0;
    return S.termination == SADVIO_TERM_FAILURE ? SADVIO_E_NOT_USABLE : SADVIO_OK;
}

int sadvio_ba_set_window(sadvio_ba_handle* h, const sadvio_flat_window* window) { return sadvio_ba_set_windows(h, 1, window); }

int sadvio_ba_landmark_chi2(sadvio_ba_handle* h, int32_t w, const double* pose_delta6, const double* lmk_delta3, const double* image_wh,
                            double pixel_sigma, double* avg_chi2, int32_t* inlier) {
    if (!h) return SADVIO_E_INVALID_ARG;
    if (!h->uploaded) { h->err = "landmark_chi2 before set_windows"; return SADVIO_E_STATE; }
    if (h->defer) { h->err = "landmark_chi2 between begin_update and commit_update"; return SADVIO_E_STATE; }
    if (w < 0 || w >= (int)h->wins.size()) { h->err = "landmark_chi2: window out of range"; return SADVIO_E_INVALID_ARG; }
    HIP_TRY(hipSetDevice(h->device));
    const WinDev& d = h->wins[w].d;
    if (d.n_lmk == 0) return SADVIO_OK;
    const SrcWin& S = h->src[w];
    // image bounds per stored camera (identical cameras are stored once; they must agree on the image size)
    std::vector<double> wh(2 * (size_t)d.n_cam, -1.0);
    for (int c = 0; c < S.v.n_cam; c++) {
        const int u = S.cam_map[c];
        const double bw = image_wh ? image_wh[2 * c] : 2.0 * S.cam_K[4 * (size_t)c + 2], bh = image_wh ? image_wh[2 * c + 1] : 2.0 * S.cam_K[4 * (size_t)c + 3];
        if (wh[2 * u] >= 0.0 && (wh[2 * u] != bw || wh[2 * u + 1] != bh)) { h->err = "landmark_chi2: cameras with identical (K, T_s_f, sigma) differ in image size"; return SADVIO_E_INVALID_ARG; }
        wh[2 * u] = bw; wh[2 * u + 1] = bh;
    }
    // deltas live in scratch (the solved state of the handle stays readable): [wh | out | xp | xl]
    const size_t n_wh = 2 * (size_t)(d.cam_base + d.n_cam), n_xp = 6 * (size_t)d.n_kf, n_xl = 3 * (size_t)d.n_lmk;
    HIP_TRY(h->d_probe.alloc(n_wh + 2 * (size_t)d.n_lmk + n_xp + n_xl));
    double* d_wh = h->d_probe.p; double* d_out = d_wh + n_wh; double* d_sxp = d_out + 2 * (size_t)d.n_lmk; double* d_sxl = d_sxp + n_xp;
    HIP_TRY(hipMemsetAsync(d_sxp, 0, sizeof(double) * (n_xp + n_xl), h->stream));
    if (pose_delta6) HIP_TRY(hipMemcpyAsync(d_sxp, pose_delta6, sizeof(double) * n_xp, hipMemcpyHostToDevice, h->stream));
    if (lmk_delta3) HIP_TRY(hipMemcpyAsync(d_sxl, lmk_delta3, sizeof(double) * n_xl, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_wh + 2 * (size_t)d.cam_base, wh.data(), sizeof(double) * wh.size(), hipMemcpyHostToDevice, h->stream));
    SolveOpts o{};
    DevPtrs P = make_ptrs(h, o, 1);
    P.xp = d_sxp - 6 * (ptrdiff_t)d.kf_base; P.xl = d_sxl - 3 * (ptrdiff_t)d.lmk_base;  // the kernel indexes globally
    const int blocks = (d.n_lmk + 63) / 64;
    const double isig = pixel_sigma > 0.0 ? 1.0 / pixel_sigma : (h->factor_type == SADVIO_FACTOR_PIXEL ? 0.0 : 1.0);  // 0: cam_isig
    if (h->factor_type == SADVIO_FACTOR_PIXEL) hipLaunchKernelGGL(k_lmk_chi2<0>, dim3(blocks), dim3(64), 0, h->stream, P, w, d_wh, isig, d_out);
    else hipLaunchKernelGGL(k_lmk_chi2<1>, dim3(blocks), dim3(64), 0, h->stream, P, w, d_wh, isig, d_out);
    HIP_TRY(hipGetLastError());
    std::vector<double> hb(2 * (size_t)d.n_lmk);
    HIP_TRY(hipMemcpyAsync(hb.data(), d_out, sizeof(double) * hb.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int l = 0; l < d.n_lmk; l++) {
        if (avg_chi2) avg_chi2[l] = hb[2 * (size_t)l];
        if (inlier) inlier[l] = (hb[2 * (size_t)l + 1] >= 2.0 && !(hb[2 * (size_t)l] > 2.0)) ? 1 : 0;
    }
    return SADVIO_OK;
}

int sadvio_ba_get_kernel_times(sadvio_ba_handle* h, int32_t cap, const char** names, double* avg_us, int64_t* launches) {
    if (!h) return SADVIO_E_INVALID_ARG;
    int n = 0;
    for (auto& k : h->kclasses) {
        if (n >= cap) break;
        if (names) names[n] = k.name;
        if (avg_us) avg_us[n] = k.launches ? 1e3 * k.total_ms / (double)k.launches : 0.0;
        if (launches) launches[n] = k.launches;
        n++;
    }
    return n;
}

const char* sadvio_ba_last_error(sadvio_ba_handle* h) { return h ? h->err.c_str() : "null handle"; }
const char* sadvio_ba_version(void) { return "sadvio-ba-mi355x 0.5 (gfx950)"; }

}  // extern "C"
