// sadvio_optimizer.hpp — C++17 host layer over the C ABI (include/sadvio_ba.h), header-only, no third-party types.
//
// Mirrors the solve entry points of isae::AOptimizer (cpp/include/isaeslam/optimizers/AOptimizer.h:22-45) — same
// names, same argument meaning, same return convention — on a plain-struct snapshot of what those methods read from
// isae::LocalMap / isae::Frame / isae::ALandmark. The reference's own classes need Eigen / Ceres / OpenCV (absent in
// this image); the adapter of INTEGRATION.md fills these structs from them and copies the results back.
// Everything the reference does around ceres::Solve is here: the inclusion rules of addResidualsLocalMap
// (BundleAdjustmentCERESAnalytic.cpp:197-314), the solver options of each entry point and the write-back
// (AOptimizer.cpp:329-340, 391-434).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <array>
#include <string>
#include <vector>

#include "sadvio_ba.h"
#include "sadvio_cameras.hpp"
#include "sadvio_io.hpp"

namespace sadvio {

struct Pose {                 // Eigen::Affine3d as 3x3 row-major + translation
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double t[3] = {0, 0, 0};
};

inline void exp_so3(const double* w, double* R) {   // geometry.h:131-147 (first order below 1e-9)
    const double a = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (a < 1e-9) {
        const double S[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        for (int i = 0; i < 9; i++) R[i] = I[i] + S[i];
        return;
    }
    const double x = w[0] / a, y = w[1] / a, z = w[2] / a;
    const double S[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    double S2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S2[3 * i + j] = S[3 * i] * S[j] + S[3 * i + 1] * S[3 + j] + S[3 * i + 2] * S[6 + j];
    for (int i = 0; i < 9; i++) R[i] = I[i] + (1.0 - std::cos(a)) * S2[i] + std::sin(a) * S[i];
}

// T <- T * (exp(w), t): frame write-back (AOptimizer.cpp:329-332, parametersBlock.hpp:34-37)
inline void apply_pose_delta(Pose& T, const double* d6) {
    double dR[9], R[9], t[3];
    exp_so3(d6, dR);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[3 * i + j] = T.R[3 * i] * dR[j] + T.R[3 * i + 1] * dR[3 + j] + T.R[3 * i + 2] * dR[6 + j];
        t[i] = T.R[3 * i] * d6[3] + T.R[3 * i + 1] * d6[4] + T.R[3 * i + 2] * d6[5] + T.t[i];
    }
    std::memcpy(T.R, R, sizeof(R)); std::memcpy(T.t, t, sizeof(t));
}

// IMU::biasDeltaCorrection (IMU.cpp:104-108): first-order update of a pre-integration for a bias change of its older frame
inline void bias_delta_correction(sadvio_imu_factor& f, const double* d_ba, const double* d_bg) {
    double w[3], E[9], R[9];
    for (int i = 0; i < 3; i++) {
        double dp = 0, dv = 0; w[i] = 0;
        for (int j = 0; j < 3; j++) {
            dp += f.J_dp_ba[3 * i + j] * d_ba[j] + f.J_dp_bg[3 * i + j] * d_bg[j];
            dv += f.J_dv_ba[3 * i + j] * d_ba[j] + f.J_dv_bg[3 * i + j] * d_bg[j];
            w[i] += f.J_dR_bg[3 * i + j] * d_bg[j];
        }
        f.delta_p[i] += dp; f.delta_v[i] += dv;
    }
    exp_so3(w, E);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = f.delta_R[3 * i] * E[j] + f.delta_R[3 * i + 1] * E[3 + j] + f.delta_R[3 * i + 2] * E[6 + j];
    std::memcpy(f.delta_R, R, sizeof(R));
}

struct CameraModel {          // one ImageSensor of a key-frame: K + frame -> sensor transform (+ the model's parameters)
    double fx, fy, cx, cy;
    Pose T_s_f;
    double width = 0, height = 0;   // image size used by ALandmark::sanityCheck (Camera.cpp:45); 0 = (2 cx, 2 cy)
    // non-pinhole models (sadvio_cameras.hpp) only work with the ANGULAR backend, as in the reference: the bearing comes
    // from the model's getRayCamera, the factor itself is model-free
    CameraKind kind = CameraKind::Pinhole;
    double rmax = 1, xi = 0, alpha = 0;
    bool distortion = false;
    double D[4] = {0, 0, 0, 0};
    CameraIntrinsics intrinsics() const {
        CameraIntrinsics k;
        k.kind = kind; k.fx = fx; k.fy = fy; k.cx = cx; k.cy = cy; k.rmax = rmax; k.xi = xi; k.alpha = alpha; k.distortion = distortion;
        for (int i = 0; i < 4; i++) k.D[i] = D[i];
        k.width = width > 0 ? width : 2.0 * cx; k.height = height > 0 ? height : 2.0 * cy;
        return k;
    }
};

struct FrameState {           // what the optimizer reads / writes of an isae::Frame (+ its IMU)
    int64_t id = 0;
    Pose T_f_w;
    bool is_keyframe = true;  // Frame::isKeyFrame: window / landmark solves only use features of key-frames (…Analytic.cpp:131,256)
    std::vector<CameraModel> cameras;
    bool has_prior = false;   // Frame::hasPrior / getPrior / getInfPrior (…Analytic.cpp:224-228)
    Pose T_prior;
    double inf_prior[6] = {0, 0, 0, 0, 0, 0};
    bool has_imu = false;
    double v[3] = {0, 0, 0}, ba[3] = {0, 0, 0}, bg[3] = {0, 0, 0};
    sadvio_imu_factor preint{};   // pre-integration towards the NEXT OLDER... see LocalMapSnapshot::imu_pairs
};

struct Feature {              // one AFeature of a landmark: which frame / camera saw it where
    int frame;                // index into LocalMapSnapshot::frames
    int camera;               // index into that frame's cameras
    double u, v;              // pixel measurement (pixel factor) — bearings are derived (Camera.cpp:15-25)
};

struct LandmarkState {
    int64_t id = 0;
    double p[3] = {0, 0, 0};  // T_w_l translation
    bool initialized = true, outlier = false;
    bool in_map = true;       // ALandmark::isInMap
    bool has_prior = false;   // ALandmark::hasPrior / setPrior (set by marginalize, marginalization.cpp:83)
    std::vector<Feature> features;
};

struct LineFeature {          // one AFeature of a linexd landmark: the two end points of the 2D segment (AFeature::getPoints)
    int frame, camera;
    double u0, v0, u1, v1;
};

struct LineLandmarkState {    // a "linexd" landmark: pose T_w_l whose x axis carries the segment (Line3D.h)
    int64_t id = 0;
    Pose T_w_l;
    double model[6] = {-0.5, 0, 0, 0.5, 0, 0};   // ModelLine3D end points as ReprojectionErrCeres_linexd_dx reads them: model3d_->getModel(), getScale() NOT applied
                                                 // (BundleAdjustmentCERESAnalytic.h:128,165; Line3D multiplies _scale by 100, Line3D.h:15, but the BA residual never reads it)
    bool initialized = true, outlier = false;
    std::vector<LineFeature> features;
};

struct ImuPair {              // IMUFactor + IMUBiasFactor between two consecutive key-frames (AOptimizer.cpp:55-92)
    int frame_i, frame_j;     // i = older
    sadvio_imu_factor f;      // kf_i / kf_j are filled by the optimizer
};

struct LocalMapSnapshot {     // LocalMap::getLastNFramesIn (newest first, amap.h:28-32) + getLandmarks()["pointxd"]
    std::vector<FrameState> frames;
    std::vector<LandmarkState> landmarks;
    std::vector<ImuPair> imu_pairs;
    std::vector<LineLandmarkState> lines;   // getLandmarks()["linexd"]
};

class HipOptimizer {
  public:
    explicit HipOptimizer(int device = 0, bool angular = false) : _angular(angular) {
        sadvio_ba_config cfg{device, 0, 1, 0};
        if (sadvio_ba_create(&cfg, &_h) != SADVIO_OK) throw std::runtime_error("sadvio: no usable gfx950 device");
    }
    ~HipOptimizer() { sadvio_ba_destroy(_h); }
    HipOptimizer(const HipOptimizer&) = delete;
    HipOptimizer& operator=(const HipOptimizer&) = delete;

    // AOptimizer::localMapBA (AOptimizer.cpp:299-350): always returns true; on a backend error the state is untouched.
    bool localMapBA(LocalMapSnapshot& map, size_t fixed_frame_number = 0) {
        sadvio_solve_options o; sadvio_ba_default_options(&o);           // LM, 20 iterations, f_tol 1e-3 (:315-323)
        solve(map, fixed_frame_number, false, o, /*all_frames_const*/ false, /*landmarks_const*/ false);
        return true;
    }
    // AOptimizer::localMapVIOptimization (:352-446)
    bool localMapVIOptimization(LocalMapSnapshot& map, size_t fixed_frame_number = 0) {
        sadvio_solve_options o; sadvio_ba_default_options(&o);
        solve(map, fixed_frame_number, true, o, false, false);
        return true;
    }
    // AOptimizer::landmarkOptimization (:98-150): poses constant, Huber(sqrt(1.345)), 10 iterations; a landmark is
    // written back only if it passes ALandmark::sanityCheck — evaluated, as in the reference, at the pose it had BEFORE
    // the solve (:124-141) — and is flagged outlier / inlier by that check
    bool landmarkOptimization(LocalMapSnapshot& map) {
        sadvio_solve_options o; sadvio_ba_default_options(&o);
        o.max_num_iterations = 10; o.huber_a = std::sqrt(1.345);
        solve(map, 0, false, o, true, false, true);
        return true;
    }
    // AOptimizer::singleFrameOptimization (:152-217): frame 0 free, landmarks constant, 5 iterations, no loss
    bool singleFrameOptimization(LocalMapSnapshot& map) {
        sadvio_solve_options o; sadvio_ba_default_options(&o);
        o.max_num_iterations = 5;
        solve(map, 0, false, o, false, true);
        return true;
    }
    // AOptimizer::singleFrameVIOptimization (:219-297): returns summary.IsSolutionUsable()
    bool singleFrameVIOptimization(LocalMapSnapshot& map) {
        sadvio_solve_options o; sadvio_ba_default_options(&o);
        o.max_num_iterations = 5; o.huber_a = std::sqrt(1.345);
        o.max_solver_time_in_seconds = 0.005;                                            // :254
        return solve(map, 0, true, o, false, true);
    }

    // AOptimizer::VIInit (:448-581): gravity direction, velocities and (optim_scale) the metric scale of a visual-only
    // map from the pre-integrated IMU factors between its key-frames; returns exp(lambda) and applies the result as
    // :526-562 — velocities += dv, T_f_w <- (R_f_w, s t_f_w) * (R_w_i, 0), priors re-anchored at 100, landmarks
    // p <- s R_w_i^T p. (The reference adds dba to Ba twice and never updates Bg, :529-530; both deltas are constant
    // zero in this problem, :472-476, so nothing is written.) R_w_i is row-major 3x3. On a backend error nothing is
    // written and 1.0 is returned.
    double VIInit(LocalMapSnapshot& map, double* R_w_i, bool optim_scale = false) {
        const int n = (int)map.frames.size();
        std::vector<double> T(12 * (size_t)n), vel(3 * (size_t)n), dv(3 * (size_t)n, 0.0);
        for (int i = 0; i < n; i++) {
            std::memcpy(&T[12 * (size_t)i], map.frames[i].T_f_w.R, 72); std::memcpy(&T[12 * (size_t)i + 9], map.frames[i].T_f_w.t, 24);
            std::memcpy(&vel[3 * (size_t)i], map.frames[i].v, 24);
        }
        std::vector<sadvio_imu_factor> fs;
        for (const ImuPair& p : map.imu_pairs) {                                          // :485-500
            if (p.frame_i == p.frame_j || p.frame_i < 0 || p.frame_j < 0 || p.frame_i >= n || p.frame_j >= n) continue;
            if (!map.frames[p.frame_i].has_imu || !map.frames[p.frame_j].has_imu) continue;
            sadvio_imu_factor f = p.f; f.kf_i = p.frame_i; f.kf_j = p.frame_j;
            fs.push_back(f);
        }
        sadvio_viinit_problem pb{};
        pb.n_frames = n; pb.n_factors = (int)fs.size(); pb.T_f_w = T.data(); pb.vel = vel.data(); pb.factors = fs.data();
        pb.optim_scale = optim_scale ? 1 : 0; pb.optim_bias = 0; pb.sigma_dba = pb.sigma_dbg = 1.0;
        sadvio_solve_options o; sadvio_ba_default_options(&o);
        o.max_num_iterations = 50;                                                        // :519-527
        sadvio_viinit_result r{};
        const int rc = sadvio_ba_vi_init(_h, &pb, &o, &_sum, &r, dv.data());
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (R_w_i) std::memcpy(R_w_i, I3, sizeof(I3));
        if (rc != SADVIO_OK) { _err = sadvio_ba_last_error(_h); return 1.0; }
        if (R_w_i) std::memcpy(R_w_i, r.R_w_i, sizeof(r.R_w_i));
        const double s = r.scale;
        for (int i = 0; i < n; i++) {
            FrameState& f = map.frames[i];
            if (f.has_imu) for (int a = 0; a < 3; a++) f.v[a] += dv[3 * (size_t)i + a];   // :526-531
            double Rn[9];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) Rn[3 * a + b] = f.T_f_w.R[3 * a] * r.R_w_i[b] + f.T_f_w.R[3 * a + 1] * r.R_w_i[3 + b] + f.T_f_w.R[3 * a + 2] * r.R_w_i[6 + b];
            std::memcpy(f.T_f_w.R, Rn, sizeof(Rn));
            for (int a = 0; a < 3; a++) f.T_f_w.t[a] *= s;                                 // :541-548
            if (f.has_prior) { f.T_prior = f.T_f_w; for (double& x : f.inf_prior) x = 100.0; }   // :550-552
        }
        for (LandmarkState& L : map.landmarks) {                                          // :555-562
            if (L.outlier) continue;
            double q[3];
            for (int a = 0; a < 3; a++) q[a] = s * (r.R_w_i[a] * L.p[0] + r.R_w_i[3 + a] * L.p[1] + r.R_w_i[6 + a] * L.p[2]);
            std::memcpy(L.p, q, sizeof(q));
        }
        return s;
    }

    // BundleAdjustmentCERESAnalytic::marginalize (…Analytic.cpp:431-663) with Marginalization::preMarginalize
    // (marginalization.cpp:23-143) restated on the snapshot: frame0 (index into map.frames) is marginalised into a prior
    // on frame1's states (VIO) and on the landmarks frame0 shares with the rest of the window. Returns false when the
    // Schur complement is refused (fewer than 4 kept columns, marginalization.cpp:215-216): the prior is then cleared,
    // as the reference does (…Analytic.cpp:620-625). The prior is held by this object (like _marginalization_last)
    // and added to the next localMapBA / localMapVIOptimization whose snapshot still contains its variables (by id).
    bool marginalize(LocalMapSnapshot& map, int frame0, int frame1, bool enable_sparsif) {
        const int nkf = (int)map.frames.size();
        if (frame0 < 0 || frame0 >= nkf || frame1 < 0 || frame1 >= nkf || frame0 == frame1) { _err = "marginalize: bad frame index"; return false; }
        Flat F;
        flatten(map, 0, map.frames[frame0].has_imu || map.frames[frame1].has_imu, false, false, F);
        // landmark selection (marginalization.cpp:50-88)
        std::vector<int32_t> keep, marg;
        std::vector<int> flat_of(map.landmarks.size(), -1);
        for (size_t k = 0; k < F.lmk_src.size(); k++) flat_of[F.lmk_src[k]] = (int)k;
        for (size_t l = 0; l < map.landmarks.size(); l++) {
            LandmarkState& L = map.landmarks[l];
            if (L.outlier || !L.in_map || !L.initialized) continue;
            bool in_frame0 = false, lonely = true;
            int num_cam = 0;
            for (const Feature& ft : L.features) {
                if (ft.frame != frame0) lonely = false; else { num_cam++; in_frame0 = true; }
            }
            if (!in_frame0) continue;                       // only the landmarks of frame0 are visited (:51)
            if (num_cam != 2 && !L.has_prior) continue;     // no full 3D information and no prior: ignored (:72-75)
            if (!lonely) { L.has_prior = true; keep.push_back(flat_of[l]); }
            else marg.push_back(flat_of[l]);
        }
        // resurrected landmarks of the previous prior (:116-139)
        bool discard_prior = false;
        if (_prior.valid)
            for (size_t q = 0; q < _prior.lmk_id.size() && !discard_prior; q++) {
                int idx = -1;
                for (size_t l = 0; l < map.landmarks.size(); l++) if (map.landmarks[l].id == _prior.lmk_id[q]) idx = (int)l;
                if (idx < 0 || flat_of[idx] < 0) { if (idx >= 0 && map.landmarks[idx].outlier) discard_prior = true; continue; }
                bool known = false;
                for (int32_t k : keep) known |= k == flat_of[idx];
                for (int32_t k : marg) known |= k == flat_of[idx];
                if (!known) keep.push_back(flat_of[idx]);
            }
        if (discard_prior) _prior = Prior();
        sadvio_marg_request rq{};
        rq.kf_marg = frame0;
        rq.marg_has_imu = map.frames[frame0].has_imu ? 1 : 0;
        rq.kf_keep = map.frames[frame1].has_imu ? frame1 : -1;                          // marginalization.cpp:99-104
        rq.n_marg = (int)marg.size(); rq.lmk_marg = marg.data();
        rq.n_keep = (int)keep.size(); rq.lmk_keep = keep.data();
        sadvio_imu_factor imu{};
        if (map.frames[frame0].has_imu && map.frames[frame1].has_imu)                  // …Analytic.cpp:449
            for (const ImuPair& p : map.imu_pairs)
                if (p.frame_i == frame0 && p.frame_j == frame1) { imu = p.f; rq.imu = &imu; }
        std::vector<sadvio_pose_prior> pri;
        // frame0's PosePriordx (…Analytic.cpp:605-617); the angular backend also folds in frame1's (Angular….cpp:682)
        for (const sadvio_pose_prior& p : F.priors) if (p.kf == frame0 || (_angular && p.kf == frame1)) pri.push_back(p);
        rq.n_prior = (int)pri.size(); rq.priors = pri.data();
        std::vector<int32_t> last_idx, last_col;
        // the previous prior is folded in whenever it kept landmarks (:573); its kept frame — VIO only — is frame0 now.
        // A VO prior (kf_col < 0) has no frame to match
        if (_prior.valid && !_prior.lmk_id.empty() && (_prior.kf_col < 0 || _prior.kf_id == map.frames[frame0].id)) {
            rq.last_n_full = SADVIO_PRIOR_RESIDENT;   // the prior never left the device (the handle holds it, AOptimizer.h:88-90)
            rq.last_kf = _prior.kf_col >= 0 ? frame0 : -1; rq.last_kf_col = std::max(_prior.kf_col, 0);
            for (size_t q = 0; q < _prior.lmk_id.size(); q++) {
                int fi = -1;
                for (size_t k = 0; k < F.lmk_id.size(); k++) if (F.lmk_id[k] == _prior.lmk_id[q]) fi = (int)k;
                last_idx.push_back(fi < 0 ? 0 : fi); last_col.push_back(fi < 0 ? -1 : _prior.lmk_col[q]);
            }
            rq.last_n_keep = (int)last_idx.size(); rq.last_lmk_index = last_idx.data(); rq.last_lmk_col = last_col.data();
        }
        rq.eig_cut_mode = _eig_cut_mode; rq.prior_form = _prior_form;
        std::vector<int32_t> lcol(std::max<size_t>(keep.size(), 1));
        sadvio_marg_result res{};
        if (F.non_pinhole_pixel) {   // the reference's own pixel factor has no Jacobian for these models (fisheye.cpp:352-405)
            _err = "pixel factor with a non-pinhole camera: use the angular backend (HipOptimizer(device, true))";
            return false;
        }
        int rc = upload(F);
        if (rc == SADVIO_OK) rc = sadvio_ba_marginalize(_h, 0, &rq, &res, lcol.data(), nullptr, nullptr);   // J, r0 stay on the device
        _prior = Prior(); _sparse.clear(); _sparse_lmk_id.clear();
        if (rc != SADVIO_OK) {
            if (rc != SADVIO_E_REFUSED) { _err = sadvio_ba_last_error(_h); sadvio_ba_set_prior(_h, 0, 0, 0, nullptr, nullptr); }   // no stale prior behind a failed call
            return false;
        }
        _prior.valid = res.n_full > 0; _prior.n_full = res.n_full; _prior.n = res.n;
        _prior.kf_id = rq.kf_keep >= 0 ? map.frames[frame1].id : -1; _prior.kf_col = res.kf_col;
        for (size_t k = 0; k < keep.size(); k++) { _prior.lmk_id.push_back(F.lmk_id[keep[k]]); _prior.lmk_col.push_back(lcol[k]); }
        if (enable_sparsif && keep.size() > 1) {
            std::vector<sadvio_sparse_prior> out(keep.size() + 1);
            int32_t n_out = 0;
            rc = sadvio_ba_sparsify(_h, 0, rq.kf_keep >= 0, 0, 0, nullptr /* the handle's prior */, rq.kf_keep, std::max(res.kf_col, 0),
                                    (int)keep.size(), keep.data(), lcol.data(), &n_out, out.data());
            if (rc == SADVIO_OK) {
                _sparse.assign(out.begin(), out.begin() + n_out);
                _sparse_kf_id = rq.kf_keep >= 0 ? map.frames[frame1].id : -1;
                for (const sadvio_sparse_prior& s : _sparse) { _sparse_lmk_id.push_back(s.lmk0 >= 0 ? F.lmk_id[s.lmk0] : -1); _sparse_lmk_id.push_back(s.lmk1 >= 0 ? F.lmk_id[s.lmk1] : -1); }
            }
        }
        return true;
    }
    // BundleAdjustmentCERESAnalytic::marginalizeRelative (…Analytic.cpp:665-809): the 6 x 6 information of the relative pose
    // T_frame0_frame1 recovered from the landmarks both frames observe (Schur complement on the two poses, then the NFR
    // covariance recovery through the Relative6DPose Jacobian). Returns false — and leaves a zero matrix, as the reference
    // returns Zero(12, 12) — when the Schur complement is refused. inf36 row-major; Ak144 (optional) = _marginalization->_Ak.
    bool marginalizeRelative(LocalMapSnapshot& map, int frame0, int frame1, double* inf36, double* Ak144 = nullptr) {
        const int nkf = (int)map.frames.size();
        for (int i = 0; i < 36; i++) inf36[i] = 0.0;
        if (frame0 < 0 || frame0 >= nkf || frame1 < 0 || frame1 >= nkf || frame0 == frame1) { _err = "marginalizeRelative: bad frame index"; return false; }
        Flat F;
        flatten(map, 0, false, false, false, F);
        if (F.non_pinhole_pixel) { _err = "pixel factor with a non-pinhole camera: use the angular backend"; return false; }
        double Ak[144];
        int rc = upload(F, false);
        if (rc == SADVIO_OK) rc = sadvio_ba_marginalize_relative(_h, 0, frame0, frame1, _rel_eig_cut_mode, inf36, Ak);
        if (rc != SADVIO_OK) { _err = sadvio_ba_last_error(_h); for (int i = 0; i < 36; i++) inf36[i] = 0.0; return false; }
        if (Ak144) std::memcpy(Ak144, Ak, sizeof(Ak));
        return true;
    }

    bool has_prior() const { return _prior.valid; }
    int prior_rows() const { return _prior.n_full; }
    int prior_cols() const { return _prior.n; }
    // n_full x n row-major, read back from the device on request (sadvio_ba_get_prior): the solve path never needs it on the host
    // Route counters of this optimizer's Cholesky-form marginalisations (sadvio_ba_marg_stats): {calls, unpivoted, fell_back}
    std::array<int32_t, 3> marg_stats() const {
        std::array<int32_t, 3> v{0, 0, 0};
        (void)sadvio_ba_marg_stats(_h, &v[0], &v[1], &v[2]);
        return v;
    }
    // Read-backs over PCIe (not for per-frame use); the buffers are sized from what the HANDLE reports, not from this object's copy
    // of the shape, so a divergence between the two cannot overflow them.
    std::vector<double> prior_J() const {
        sadvio_prior_info pi{};
        if (!_prior.valid || sadvio_ba_get_prior(_h, &pi, nullptr, nullptr) != SADVIO_OK || !pi.valid) return {};
        std::vector<double> J((size_t)pi.n_full * (size_t)pi.n);
        if (sadvio_ba_get_prior(_h, nullptr, J.data(), nullptr) != SADVIO_OK) J.clear();
        return J;
    }
    std::vector<double> prior_r0() const {
        sadvio_prior_info pi{};
        if (!_prior.valid || sadvio_ba_get_prior(_h, &pi, nullptr, nullptr) != SADVIO_OK || !pi.valid) return {};
        std::vector<double> r((size_t)pi.n_full);
        if (sadvio_ba_get_prior(_h, nullptr, nullptr, r.data()) != SADVIO_OK) r.clear();
        return r;
    }
    // Eigenvalue cut of marginalize / marginalizeRelative (SADVIO_EIG_CUT_*; default: the reference's absolute 1e-12,
    // marginalization.hpp:58) and form of the stored prior (SADVIO_PRIOR_FORM_*; default: the rank-revealing Cholesky factor —
    // same MarginalizationFactor cost / gradient / Gauss-Newton matrix as the reference's Lambda^1/2 U^T, 5 x faster to form;
    // SADVIO_PRIOR_FORM_EIGEN reproduces the reference's rows).
    void set_eig_cut_mode(int mode) { _eig_cut_mode = mode; }
    // marginalizeRelative keeps the noise floor by default: Ak of two poses has an EXACT 6-dimensional gauge null space whose
    // eigenvalues compute to +-1e-7 .. 1e-6; the reference's absolute 1e-12 inverts the positive ones (1 / lambda ~ 1e6) and the
    // recovered 6 x 6 'information' is then rounding noise (measured through this layer: trace 6.9e6, min diagonal -4.9e5) —
    // the reference's own function has no caller. SADVIO_EIG_CUT_REFERENCE reproduces it on request.
    void set_relative_eig_cut_mode(int mode) { _rel_eig_cut_mode = mode; }
    void set_prior_form(int form) { _prior_form = form; }
    const std::vector<int64_t>& prior_landmark_ids() const { return _prior.lmk_id; }
    const std::vector<int32_t>& prior_landmark_cols() const { return _prior.lmk_col; }
    size_t sparse_factor_count() const { return _sparse.size(); }
    // features the last flattened window left out because their camera index was outside the frame's sensor list (also warned on stderr)
    int skipped_bad_camera_features() const { return _skipped_bad_camera; }

    const sadvio_solve_summary& summary() const { return _sum; }
    const std::string& last_error() const { return _err; }
    // every window handed to the backend is also written to <dir>/window_NNNNNN.sadvio (include/sadvio_io.hpp)
    void set_dump_dir(const std::string& dir) { _dump_dir = dir; _dump_count = 0; }

  private:
    struct Flat {   // the flattened window + the vectors it points into
        sadvio_flat_window w{};
        std::vector<int64_t> kf_id, lmk_id;
        std::vector<double> kf_T, kf_v, kf_ba, kf_bg, cam_K, cam_T, cam_sigma, lmk_p, meas;
        std::vector<uint8_t> kf_const, lc;
        std::vector<int> cam_base, lmk_src;
        std::vector<double> cam_wh;       // image size per flat camera (sanityCheck)
        int n_non_kf_obs = 0;             // features skipped because their frame is not a key-frame
        int n_bad_camera = 0;             // features skipped because their camera index is not one of the frame's sensors
        bool any_non_pinhole = false, non_pinhole_pixel = false;
        std::vector<int32_t> ptr, obs_kf, obs_cam;
        std::vector<sadvio_pose_prior> priors;
        std::vector<sadvio_imu_factor> imus;
        // linexd landmarks (…Analytic.cpp:270-310)
        std::vector<int> line_src;
        std::vector<int64_t> line_id;
        std::vector<double> line_T, line_model, line_meas;
        std::vector<int32_t> line_ptr, line_obs_kf, line_obs_cam;
        sadvio_line_set lines{};
    };
    struct Prior {  // the dense prior kept between marginalize() and the next window solves, variables named by id
        bool valid = false;
        int n_full = 0, n = 0, kf_col = -1;
        int64_t kf_id = -1;
        std::vector<int64_t> lmk_id;   // J, r0 themselves live in the backend handle (device resident)
        std::vector<int32_t> lmk_col;
    };

    // inclusion rules of addResidualsLocalMap (BundleAdjustmentCERESAnalytic.cpp:197-314)
    void flatten(const LocalMapSnapshot& map, size_t fixed, bool vio, bool all_const, bool lmk_const, Flat& F, bool kf_only = true) const {
        const int nkf = (int)map.frames.size();
        F.kf_id.resize(nkf); F.kf_T.resize(12 * (size_t)nkf); F.kf_v.resize(3 * (size_t)nkf); F.kf_ba.resize(3 * (size_t)nkf);
        F.kf_bg.resize(3 * (size_t)nkf); F.kf_const.resize(nkf); F.cam_base.resize(nkf); F.ptr.assign(1, 0);
        for (int i = 0; i < nkf; i++) {
            const FrameState& f = map.frames[i];
            F.kf_id[i] = f.id;
            std::memcpy(&F.kf_T[12 * (size_t)i], f.T_f_w.R, 72); std::memcpy(&F.kf_T[12 * (size_t)i + 9], f.T_f_w.t, 24);
            std::memcpy(&F.kf_v[3 * (size_t)i], f.v, 24); std::memcpy(&F.kf_ba[3 * (size_t)i], f.ba, 24); std::memcpy(&F.kf_bg[3 * (size_t)i], f.bg, 24);
            F.kf_const[i] = all_const || (i > nkf - (int)fixed - 1);                     // …Analytic.cpp:219
            if (f.has_prior) {                                                           // :224-228
                sadvio_pose_prior p{};
                p.kf = i; std::memcpy(p.T_prior, f.T_prior.R, 72); std::memcpy(p.T_prior + 9, f.T_prior.t, 24);
                std::memcpy(p.inf_diag, f.inf_prior, 48);
                F.priors.push_back(p);
            }
            F.cam_base[i] = (int)F.cam_sigma.size();
            for (const CameraModel& c : f.cameras) {
                const double K[4] = {c.fx, c.fy, c.cx, c.cy};
                F.cam_K.insert(F.cam_K.end(), K, K + 4);
                F.cam_T.insert(F.cam_T.end(), c.T_s_f.R, c.T_s_f.R + 9); F.cam_T.insert(F.cam_T.end(), c.T_s_f.t, c.T_s_f.t + 3);
                F.cam_sigma.push_back(_angular ? 1.5 / (0.5 * (c.fx + c.fy)) : 1.0);    // …Analytic.h:46 / Angular….cpp:283
                F.cam_wh.push_back(c.width > 0 ? c.width : 2.0 * c.cx); F.cam_wh.push_back(c.height > 0 ? c.height : 2.0 * c.cy);
            }
        }
        for (int l = 0; l < (int)map.landmarks.size(); l++) {
            const LandmarkState& L = map.landmarks[l];
            if (!L.initialized || L.outlier) continue;                                   // :239
            F.lmk_src.push_back(l); F.lmk_id.push_back(L.id); F.lc.push_back(lmk_const ? 1 : 0);
            F.lmk_p.insert(F.lmk_p.end(), L.p, L.p + 3);
            for (const Feature& ft : L.features) {
                if (ft.frame < 0 || ft.frame >= nkf) continue;                           // :256-258 (frame not in the window)
                if (kf_only && !map.frames[ft.frame].is_keyframe) { F.n_non_kf_obs++; continue; }   // :131, :256
                if (ft.camera < 0 || ft.camera >= (int)map.frames[ft.frame].cameras.size()) { F.n_bad_camera++; continue; }   // a feature of a sensor the frame does not carry
                const CameraModel& c = map.frames[ft.frame].cameras[ft.camera];
                F.obs_kf.push_back(ft.frame); F.obs_cam.push_back(F.cam_base[ft.frame] + ft.camera);
                if (_angular) {                                                          // getRayCamera of the feature's camera model
                    double b[3] = {0.0, 0.0, 1.0};
                    ray_camera(c.intrinsics(), ft.u, ft.v, b);
                    F.meas.insert(F.meas.end(), {b[0], b[1], b[2]});
                } else {
                    if (c.kind != CameraKind::Pinhole) F.non_pinhole_pixel = true;           // no analytic pixel factor exists for it
                    F.meas.insert(F.meas.end(), {ft.u, ft.v});
                }
                if (c.kind != CameraKind::Pinhole) F.any_non_pinhole = true;
            }
            F.ptr.push_back((int32_t)F.obs_kf.size());
        }
        F.line_ptr.assign(1, 0);
        for (int l = 0; l < (int)map.lines.size(); l++) {                                // …Analytic.cpp:270-310
            const LineLandmarkState& L = map.lines[l];
            if (!L.initialized || L.outlier) continue;
            F.line_src.push_back(l); F.line_id.push_back(L.id);
            F.line_T.insert(F.line_T.end(), L.T_w_l.R, L.T_w_l.R + 9); F.line_T.insert(F.line_T.end(), L.T_w_l.t, L.T_w_l.t + 3);
            F.line_model.insert(F.line_model.end(), L.model, L.model + 6);
            for (const LineFeature& ft : L.features) {
                if (ft.frame < 0 || ft.frame >= nkf || !map.frames[ft.frame].is_keyframe) continue;   // :296-299
                if (ft.camera < 0 || ft.camera >= (int)map.frames[ft.frame].cameras.size()) { F.n_bad_camera++; continue; }
                const CameraModel& c = map.frames[ft.frame].cameras[ft.camera];
                F.line_obs_kf.push_back(ft.frame); F.line_obs_cam.push_back(F.cam_base[ft.frame] + ft.camera);
                if (_angular) {                                                          // feature->getBearingVectors()
                    double b0[3] = {0, 0, 1}, b1[3] = {0, 0, 1};
                    ray_camera(c.intrinsics(), ft.u0, ft.v0, b0); ray_camera(c.intrinsics(), ft.u1, ft.v1, b1);
                    F.line_meas.insert(F.line_meas.end(), {b0[0], b0[1], b0[2], b1[0], b1[1], b1[2]});
                } else {
                    if (c.kind != CameraKind::Pinhole) F.non_pinhole_pixel = true;
                    F.line_meas.insert(F.line_meas.end(), {ft.u0, ft.v0, ft.u1, ft.v1});
                }
            }
            F.line_ptr.push_back((int32_t)F.line_obs_kf.size());
        }
        F.lines.n_line = (int)F.line_src.size(); F.lines.n_obs = (int)F.line_obs_kf.size();
        F.lines.line_id = F.line_id.data(); F.lines.line_T_w_l = F.line_T.data(); F.lines.line_model = F.line_model.data();
        F.lines.line_const = nullptr; F.lines.line_obs_ptr = F.line_ptr.data();
        F.lines.obs_kf = F.line_obs_kf.data(); F.lines.obs_cam = F.line_obs_cam.data(); F.lines.obs_meas = F.line_meas.data();
        if (vio)
            for (const ImuPair& p : map.imu_pairs) {                                     // AOptimizer.cpp:69-72
                if (p.frame_i == p.frame_j || p.f.dt > 1.0) continue;
                sadvio_imu_factor f = p.f; f.kf_i = p.frame_i; f.kf_j = p.frame_j;
                F.imus.push_back(f);
            }
        sadvio_flat_window& w = F.w;
        w.n_kf = nkf; w.n_cam = (int)F.cam_sigma.size(); w.n_lmk = (int)F.lmk_src.size(); w.n_obs = (int)F.obs_kf.size();
        w.factor_type = _angular ? SADVIO_FACTOR_ANGULAR : SADVIO_FACTOR_PIXEL; w.has_imu = vio ? 1 : 0;
        w.kf_id = F.kf_id.data(); w.kf_T_f_w = F.kf_T.data(); w.kf_const = F.kf_const.data();
        w.kf_vel = F.kf_v.data(); w.kf_ba = F.kf_ba.data(); w.kf_bg = F.kf_bg.data();
        w.cam_K = F.cam_K.data(); w.cam_T_s_f = F.cam_T.data(); w.cam_sigma = F.cam_sigma.data();
        w.lmk_id = F.lmk_id.data(); w.lmk_p = F.lmk_p.data(); w.lmk_const = F.lc.data(); w.lmk_obs_ptr = F.ptr.data();
        w.obs_kf = F.obs_kf.data(); w.obs_cam = F.obs_cam.data(); w.obs_meas = F.meas.data();
    }

    // with_pose_priors: PosePriordx blocks are only added by addResidualsLocalMap (…Analytic.cpp:224-228) and marginalize
    // (:605-617); addSingleFrameResiduals / addLandmarkResiduals (:5-50, 102-150) add none
    int upload(const Flat& F, bool with_pose_priors = true, const Flat* prior_of = nullptr) {
        // a feature whose camera index is not one of its frame's sensors cannot be evaluated: it is skipped (the reference would
        // throw from .at()), and the count is REPORTED — a corrupted map must not solve silently with fewer observations
        _skipped_bad_camera = F.n_bad_camera;
        if (F.n_bad_camera) std::fprintf(stderr, "[sadvio] warning: %d feature(s) skipped: camera index outside the frame's sensor list\n", F.n_bad_camera);
        // one layout build for the window and all its factor lists (sadvio_ba_begin_update .. commit_update)
        int rc = sadvio_ba_begin_update(_h);
        if (rc == SADVIO_OK) rc = sadvio_ba_set_windows(_h, 1, &F.w);
        if (rc == SADVIO_OK && with_pose_priors) rc = sadvio_ba_set_pose_priors(_h, 0, (int)F.priors.size(), F.priors.data());
        if (rc == SADVIO_OK && !F.imus.empty()) rc = sadvio_ba_set_imu_factors(_h, 0, (int)F.imus.size(), F.imus.data());
        if (rc == SADVIO_OK && prior_of) rc = add_marginalization_prior(*prior_of);
        const int rc2 = sadvio_ba_commit_update(_h);
        return rc != SADVIO_OK ? rc : rc2;
    }

    // addMarginalizationResiduals (…Analytic.cpp:316-426): the stored prior on the variables still in the window
    int add_marginalization_prior(const Flat& F) {
        if (!_sparse.empty()) {
            std::vector<sadvio_sparse_prior> fs;
            for (size_t k = 0; k < _sparse.size(); k++) {
                sadvio_sparse_prior s = _sparse[k];
                auto find_l = [&](int64_t id) { for (size_t q = 0; q < F.lmk_id.size(); q++) if (F.lmk_id[q] == id) return (int)q; return -1; };
                auto find_k = [&](int64_t id) { for (size_t q = 0; q < F.kf_id.size(); q++) if (F.kf_id[q] == id) return (int)q; return -1; };
                if (s.kf >= 0) { s.kf = find_k(_sparse_kf_id); if (s.kf < 0) continue; }
                if (s.lmk0 >= 0) { s.lmk0 = find_l(_sparse_lmk_id[2 * k]); if (s.lmk0 < 0) continue; }
                if (s.lmk1 >= 0) { s.lmk1 = find_l(_sparse_lmk_id[2 * k + 1]); if (s.lmk1 < 0) continue; }
                fs.push_back(s);
            }
            return fs.empty() ? SADVIO_OK : sadvio_ba_set_sparse_priors(_h, 0, (int)fs.size(), fs.data());
        }
        if (!_prior.valid) return SADVIO_OK;
        int kf = -1;
        if (_prior.kf_col >= 0) {
            for (size_t q = 0; q < F.kf_id.size(); q++) if (F.kf_id[q] == _prior.kf_id) kf = (int)q;
            if (kf < 0) return SADVIO_OK;   // the kept frame left the window: nothing to attach the prior to
        }
        std::vector<int32_t> idx, col;
        for (size_t q = 0; q < _prior.lmk_id.size(); q++) {
            int fi = -1;
            for (size_t k = 0; k < F.lmk_id.size(); k++) if (F.lmk_id[k] == _prior.lmk_id[q]) fi = (int)k;
            idx.push_back(fi < 0 ? 0 : fi); col.push_back(fi < 0 ? -1 : _prior.lmk_col[q]);   // absent landmark: zero delta
        }
        return sadvio_ba_set_dense_prior(_h, 0, SADVIO_PRIOR_RESIDENT, 0, nullptr, nullptr, kf, std::max(_prior.kf_col, 0),
                                         (int)idx.size(), idx.data(), col.data());
    }

    // ALandmark::sanityCheck on the host for maps with non-pinhole cameras: every feature of the landmark (key-frame or
    // not) is projected with its own camera model (sadvio_cameras.hpp) at the landmark's CURRENT position
    void host_chi2_gate(const LocalMapSnapshot& map, const Flat& F, std::vector<int32_t>& inlier) const {
        for (size_t k = 0; k < F.lmk_src.size(); k++) {
            const LandmarkState& L = map.landmarks[F.lmk_src[k]];
            double sum = 0.0;
            int n = 0;
            for (const Feature& ft : L.features) {
                if (ft.frame < 0 || ft.frame >= (int)map.frames.size()) continue;
                const FrameState& fr = map.frames[ft.frame];
                if (ft.camera < 0 || ft.camera >= (int)fr.cameras.size()) continue;
                const CameraModel& c = fr.cameras[ft.camera];
                double pf[3], pc[3], u, v;
                for (int a = 0; a < 3; a++) pf[a] = fr.T_f_w.R[3 * a] * L.p[0] + fr.T_f_w.R[3 * a + 1] * L.p[1] + fr.T_f_w.R[3 * a + 2] * L.p[2] + fr.T_f_w.t[a];
                for (int a = 0; a < 3; a++) pc[a] = c.T_s_f.R[3 * a] * pf[0] + c.T_s_f.R[3 * a + 1] * pf[1] + c.T_s_f.R[3 * a + 2] * pf[2] + c.T_s_f.t[a];
                const bool ok = project_camera(c.intrinsics(), pc, u, v);
                sum += ok ? (u - ft.u) * (u - ft.u) + (v - ft.v) * (v - ft.v) : 1000.0;   // feature sigma = 1 px (AFeature2D.h:18)
                n++;
            }
            inlier[k] = (n >= 2 && !(sum / n > 2.0)) ? 1 : 0;
        }
    }

    bool solve(LocalMapSnapshot& map, size_t fixed, bool vio, const sadvio_solve_options& opt, bool all_const, bool lmk_const,
               bool chi2_gate = false) {
        const int nkf = (int)map.frames.size();
        Flat F;
        flatten(map, fixed, vio, all_const, lmk_const, F, !lmk_const);   // addSingleFrameResiduals has no key-frame test (:5-50)
        if (!_dump_dir.empty()) {   // replayable record of exactly what the backend is given (scripts/replay.py)
            char name[64];
            std::snprintf(name, sizeof(name), "/window_%06d.sadvio", _dump_count++);
            const std::string e = write_window(_dump_dir + name, F.w, (!all_const && !lmk_const) ? (int)F.priors.size() : 0, F.priors.data(), (int)F.imus.size(), F.imus.data());
            if (!e.empty()) _err = e;
        }
        if (F.non_pinhole_pixel) {   // the reference's own pixel factor has no Jacobian for these models (fisheye.cpp:352-405)
            _err = "pixel factor with a non-pinhole camera: use the angular backend (HipOptimizer(device, true))";
            return false;
        }
        const bool window_solve = !all_const && !lmk_const;
        int rc = upload(F, window_solve, window_solve ? &F : nullptr);   // the stored prior rides the same layout build
        const bool with_lines = window_solve && F.lines.n_line > 0;      // the line blocks are only built by addResidualsLocalMap
        if (rc == SADVIO_OK && with_lines) rc = sadvio_ba_set_lines(_h, 0, &F.lines);
        if (rc == SADVIO_OK) rc = sadvio_ba_solve(_h, &opt, &_sum);
        if (rc != SADVIO_OK && rc != SADVIO_E_NOT_USABLE) { _err = sadvio_ba_last_error(_h); return false; }   // state untouched
        if (rc == SADVIO_E_NOT_USABLE) return false;
        std::vector<double> dpose(6 * (size_t)nkf), dl(3 * (size_t)std::max(F.w.n_lmk, 1)), dv(3 * (size_t)nkf), dba(3 * (size_t)nkf), dbg(3 * (size_t)nkf);
        if (sadvio_ba_get_deltas(_h, 0, dpose.data(), dl.data(), dv.data(), dba.data(), dbg.data()) != SADVIO_OK) { _err = sadvio_ba_last_error(_h); return false; }
        for (int i = 0; i < nkf; i++) {                                                  // AOptimizer.cpp:329-332
            apply_pose_delta(map.frames[i].T_f_w, &dpose[6 * (size_t)i]);
            if (vio)                                                                     // :405-418
                for (int a = 0; a < 3; a++) {
                    map.frames[i].v[a] += dv[3 * (size_t)i + a]; map.frames[i].ba[a] += dba[3 * (size_t)i + a]; map.frames[i].bg[a] += dbg[3 * (size_t)i + a];
                }
        }
        // IMU::biasDeltaCorrection (IMU.cpp:104-108) of every pre-integration whose older frame got a bias update
        // (AOptimizer.cpp:421-434): localMapVIOptimization only — singleFrameVIOptimization does not do it (:262-296)
        if (vio && window_solve)
            for (ImuPair& p : map.imu_pairs) {
                if (p.frame_i < 0 || p.frame_i >= nkf || p.frame_j < 0 || p.frame_j >= nkf || !map.frames[p.frame_j].has_imu || !map.frames[p.frame_i].has_imu) continue;
                bias_delta_correction(p.f, &dba[3 * (size_t)p.frame_i], &dbg[3 * (size_t)p.frame_i]);
            }
        if (with_lines) {                                                                // AOptimizer.cpp:334-336: T_w_l <- T_w_l * (exp(w), t)
            std::vector<double> d6(6 * (size_t)F.lines.n_line);
            if (sadvio_ba_get_line_deltas(_h, 0, d6.data()) != SADVIO_OK) { _err = sadvio_ba_last_error(_h); return false; }
            for (size_t k = 0; k < F.line_src.size(); k++) apply_pose_delta(map.lines[F.line_src[k]].T_w_l, &d6[6 * k]);
        }
        std::vector<int32_t> inlier(F.lmk_src.size(), 1);
        if (chi2_gate && !F.lmk_src.empty()) {                                           // ALandmark.cpp:130-146
            const Flat* G = &F;
            Flat Fall;
            if (F.n_non_kf_obs) {   // sanityCheck walks ALL the features of the landmark, key-frame or not
                flatten(map, fixed, vio, all_const, lmk_const, Fall, false);
                if (sadvio_ba_set_windows(_h, 1, &Fall.w) != SADVIO_OK) { _err = sadvio_ba_last_error(_h); return false; }
                G = &Fall;
            }
            if (F.any_non_pinhole) host_chi2_gate(map, F, inlier);   // the device gate projects with K only
            else if (sadvio_ba_landmark_chi2(_h, 0, nullptr, nullptr, G->cam_wh.data(), 1.0, nullptr, inlier.data()) != SADVIO_OK) {
                _err = sadvio_ba_last_error(_h); return false;
            }
        }
        for (size_t k = 0; k < F.lmk_src.size(); k++) {                                  // :334-340
            LandmarkState& L = map.landmarks[F.lmk_src[k]];
            if (chi2_gate) L.outlier = !inlier[k];                                       // setOutlier / setInlier
            if (inlier[k]) for (int a = 0; a < 3; a++) L.p[a] += dl[3 * k + a];
        }
        return true;
    }

    std::string _dump_dir;
    int _dump_count = 0;
    int _skipped_bad_camera = 0;
    int _eig_cut_mode = SADVIO_EIG_CUT_REFERENCE, _rel_eig_cut_mode = SADVIO_EIG_CUT_NOISE_FLOOR, _prior_form = SADVIO_PRIOR_FORM_CHOLESKY;
    sadvio_ba_handle* _h = nullptr;
    sadvio_solve_summary _sum{};
    std::string _err;
    bool _angular;
    Prior _prior;
    std::vector<sadvio_sparse_prior> _sparse;   // the sparsified prior (indices of the window it was built on)
    std::vector<int64_t> _sparse_lmk_id;        // [2 k], [2 k + 1]: ids of lmk0 / lmk1 of factor k
    int64_t _sparse_kf_id = -1;
};

}  // namespace sadvio
