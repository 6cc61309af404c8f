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
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "sadvio_ba.h"

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

struct CameraModel {          // one ImageSensor of a key-frame: pinhole K + frame -> sensor transform
    double fx, fy, cx, cy;
    Pose T_s_f;
};

struct FrameState {           // what the optimizer reads / writes of an isae::Frame (+ its IMU)
    int64_t id = 0;
    Pose T_f_w;
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
    std::vector<Feature> features;
};

struct ImuPair {              // IMUFactor + IMUBiasFactor between two consecutive key-frames (AOptimizer.cpp:55-92)
    int frame_i, frame_j;     // i = older
    sadvio_imu_factor f;      // kf_i / kf_j are filled by the optimizer
};

struct LocalMapSnapshot {     // LocalMap::getLastNFramesIn (newest first, amap.h:28-32) + getLandmarks()["pointxd"]
    std::vector<FrameState> frames;
    std::vector<LandmarkState> landmarks;
    std::vector<ImuPair> imu_pairs;
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
    // AOptimizer::landmarkOptimization (:98-150): poses constant, Huber(sqrt(1.345)), 10 iterations
    bool landmarkOptimization(LocalMapSnapshot& map) {
        sadvio_solve_options o; sadvio_ba_default_options(&o);
        o.max_num_iterations = 10; o.huber_a = std::sqrt(1.345);
        solve(map, 0, false, o, true, false);
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
        return solve(map, 0, true, o, false, true);
    }

    const sadvio_solve_summary& summary() const { return _sum; }
    const std::string& last_error() const { return _err; }

  private:
    bool solve(LocalMapSnapshot& map, size_t fixed, bool vio, const sadvio_solve_options& opt, bool all_const, bool lmk_const) {
        const int nkf = (int)map.frames.size();
        std::vector<int64_t> kf_id(nkf), lmk_id;
        std::vector<double> kf_T(12 * (size_t)nkf), kf_v(3 * (size_t)nkf), kf_ba(3 * (size_t)nkf), kf_bg(3 * (size_t)nkf);
        std::vector<uint8_t> kf_const(nkf), lc;
        std::vector<double> cam_K, cam_T, cam_sigma;
        std::vector<int> cam_base(nkf);
        std::vector<sadvio_pose_prior> priors;
        for (int i = 0; i < nkf; i++) {
            const FrameState& f = map.frames[i];
            kf_id[i] = f.id;
            std::memcpy(&kf_T[12 * (size_t)i], f.T_f_w.R, 72); std::memcpy(&kf_T[12 * (size_t)i + 9], f.T_f_w.t, 24);
            std::memcpy(&kf_v[3 * (size_t)i], f.v, 24); std::memcpy(&kf_ba[3 * (size_t)i], f.ba, 24); std::memcpy(&kf_bg[3 * (size_t)i], f.bg, 24);
            kf_const[i] = all_const || (i > nkf - (int)fixed - 1);                       // …Analytic.cpp:219
            if (f.has_prior) {                                                           // :224-228
                sadvio_pose_prior p{};
                p.kf = i; std::memcpy(p.T_prior, f.T_prior.R, 72); std::memcpy(p.T_prior + 9, f.T_prior.t, 24);
                std::memcpy(p.inf_diag, f.inf_prior, 48);
                priors.push_back(p);
            }
            cam_base[i] = (int)cam_sigma.size();
            for (const CameraModel& c : f.cameras) {
                const double K[4] = {c.fx, c.fy, c.cx, c.cy};
                cam_K.insert(cam_K.end(), K, K + 4);
                cam_T.insert(cam_T.end(), c.T_s_f.R, c.T_s_f.R + 9); cam_T.insert(cam_T.end(), c.T_s_f.t, c.T_s_f.t + 3);
                cam_sigma.push_back(_angular ? 1.5 / (0.5 * (c.fx + c.fy)) : 1.0);      // …Analytic.h:46 / Angular….cpp:283
            }
        }
        std::vector<int> lmk_src;
        std::vector<double> lmk_p, meas;
        std::vector<int32_t> ptr{0}, obs_kf, obs_cam;
        for (int l = 0; l < (int)map.landmarks.size(); l++) {
            const LandmarkState& L = map.landmarks[l];
            if (!L.initialized || L.outlier) continue;                                   // :239
            lmk_src.push_back(l); lmk_id.push_back(L.id); lc.push_back(lmk_const ? 1 : 0);
            lmk_p.insert(lmk_p.end(), L.p, L.p + 3);
            for (const Feature& ft : L.features) {
                if (ft.frame < 0 || ft.frame >= nkf) continue;                           // :256-258 (frame not in the window)
                const CameraModel& c = map.frames[ft.frame].cameras[ft.camera];
                obs_kf.push_back(ft.frame); obs_cam.push_back(cam_base[ft.frame] + ft.camera);
                if (_angular) {                                                          // Camera.cpp:15-25: K^-1 [u v 1] normalised
                    double b[3] = {(ft.u - c.cx) / c.fx, (ft.v - c.cy) / c.fy, 1.0};
                    const double n = std::sqrt(b[0] * b[0] + b[1] * b[1] + 1.0);
                    meas.insert(meas.end(), {b[0] / n, b[1] / n, b[2] / n});
                } else meas.insert(meas.end(), {ft.u, ft.v});
            }
            ptr.push_back((int32_t)obs_kf.size());
        }
        sadvio_flat_window w{};
        w.n_kf = nkf; w.n_cam = (int)cam_sigma.size(); w.n_lmk = (int)lmk_src.size(); w.n_obs = (int)obs_kf.size();
        w.factor_type = _angular ? SADVIO_FACTOR_ANGULAR : SADVIO_FACTOR_PIXEL; w.has_imu = vio ? 1 : 0;
        w.kf_id = kf_id.data(); w.kf_T_f_w = kf_T.data(); w.kf_const = kf_const.data();
        w.kf_vel = kf_v.data(); w.kf_ba = kf_ba.data(); w.kf_bg = kf_bg.data();
        w.cam_K = cam_K.data(); w.cam_T_s_f = cam_T.data(); w.cam_sigma = cam_sigma.data();
        w.lmk_id = lmk_id.data(); w.lmk_p = lmk_p.data(); w.lmk_const = lc.data(); w.lmk_obs_ptr = ptr.data();
        w.obs_kf = obs_kf.data(); w.obs_cam = obs_cam.data(); w.obs_meas = meas.data();
        std::vector<sadvio_imu_factor> imus;
        if (vio)
            for (const ImuPair& p : map.imu_pairs) {                                     // AOptimizer.cpp:69-72
                if (p.frame_i == p.frame_j || p.f.dt > 1.0) continue;
                sadvio_imu_factor f = p.f; f.kf_i = p.frame_i; f.kf_j = p.frame_j;
                imus.push_back(f);
            }
        int rc = sadvio_ba_set_windows(_h, 1, &w);
        if (rc == SADVIO_OK) rc = sadvio_ba_set_pose_priors(_h, 0, (int)priors.size(), priors.data());
        if (rc == SADVIO_OK && !imus.empty()) rc = sadvio_ba_set_imu_factors(_h, 0, (int)imus.size(), imus.data());
        if (rc == SADVIO_OK) rc = sadvio_ba_solve(_h, &opt, &_sum);
        if (rc != SADVIO_OK && rc != SADVIO_E_NOT_USABLE) { _err = sadvio_ba_last_error(_h); return false; }   // state untouched
        if (rc == SADVIO_E_NOT_USABLE) return false;
        std::vector<double> dpose(6 * (size_t)nkf), dl(3 * (size_t)std::max(w.n_lmk, 1)), dv(3 * (size_t)nkf), dba(3 * (size_t)nkf), dbg(3 * (size_t)nkf);
        if (sadvio_ba_get_deltas(_h, 0, dpose.data(), dl.data(), dv.data(), dba.data(), dbg.data()) != SADVIO_OK) { _err = sadvio_ba_last_error(_h); return false; }
        for (int i = 0; i < nkf; i++) {                                                  // AOptimizer.cpp:329-332
            apply_pose_delta(map.frames[i].T_f_w, &dpose[6 * (size_t)i]);
            if (vio)                                                                     // :405-418
                for (int a = 0; a < 3; a++) {
                    map.frames[i].v[a] += dv[3 * (size_t)i + a]; map.frames[i].ba[a] += dba[3 * (size_t)i + a]; map.frames[i].bg[a] += dbg[3 * (size_t)i + a];
                }
        }
        for (size_t k = 0; k < lmk_src.size(); k++)                                      // :334-340
            for (int a = 0; a < 3; a++) map.landmarks[lmk_src[k]].p[a] += dl[3 * k + a];
        return true;
    }

    sadvio_ba_handle* _h = nullptr;
    sadvio_solve_summary _sum{};
    std::string _err;
    bool _angular;
};

}  // namespace sadvio
