// sadvio_io.hpp — on-disk format of a flattened window ("SADVIOW1"), header-only C++17, no dependencies.
//
// Purpose: dump the problem a live SaDVIO run hands to the optimizer — the sadvio_flat_window the
// adapter builds from isae::LocalMap (BundleAdjustmentCERESAnalytic.cpp:197-314) plus its pose priors and IMU factors —
// so that the identical inputs can be replayed through any backend (scripts/replay.py: GPU library and CPU oracle) without
// linking the reference on the GPU box. Little-endian, every array 8-byte aligned:
//   char[8]  "SADVIOW1"
//   int32[12] n_kf n_cam n_lmk n_obs factor_type has_imu n_prior n_imu has_lmk_const has_cam_sigma has_ids reserved
//   int64[n_kf] kf_id (has_ids) | f64[n_kf][12] kf_T_f_w | u8[n_kf] kf_const | f64[n_kf][3] kf_vel, kf_ba, kf_bg (has_imu)
//   f64[n_cam][4] cam_K | f64[n_cam][12] cam_T_s_f | f64[n_cam] cam_sigma (has_cam_sigma)
//   int64[n_lmk] lmk_id (has_ids) | f64[n_lmk][3] lmk_p | u8[n_lmk] lmk_const (has_lmk_const) | int32[n_lmk + 1] lmk_obs_ptr
//   int32[n_obs] obs_kf | int32[n_obs] obs_cam | f64[n_obs][2 | 3] obs_meas (pixel | bearing)
//   sadvio_pose_prior[n_prior] | sadvio_imu_factor[n_imu]      (the C structs of sadvio_ba.h, as laid out in memory)
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sadvio_ba.h"

namespace sadvio {

struct WindowFile {   // owning copy of a window read back from disk
    sadvio_flat_window w{};
    std::vector<sadvio_pose_prior> priors;
    std::vector<sadvio_imu_factor> imus;
    std::vector<int64_t> kf_id, lmk_id;
    std::vector<double> kf_T, kf_vel, kf_ba, kf_bg, cam_K, cam_T, cam_sigma, lmk_p, meas;
    std::vector<uint8_t> kf_const, lmk_const;
    std::vector<int32_t> ptr, obs_kf, obs_cam;
};

namespace io_detail {
inline bool put(std::FILE* f, const void* p, size_t bytes) {
    static const char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (bytes && std::fwrite(p, 1, bytes, f) != bytes) return false;
    const size_t pad = (8 - bytes % 8) % 8;
    return !pad || std::fwrite(zeros, 1, pad, f) == pad;
}
template <typename T>
inline bool get(std::FILE* f, std::vector<T>& v, size_t count) {
    v.resize(count);
    const size_t bytes = count * sizeof(T);
    if (bytes && std::fread(v.data(), 1, bytes, f) != bytes) return false;
    char skip[8];
    const size_t pad = (8 - bytes % 8) % 8;
    return !pad || std::fread(skip, 1, pad, f) == pad;
}
}  // namespace io_detail

// Returns "" on success, else what failed.
inline std::string write_window(const std::string& path, const sadvio_flat_window& w, int n_prior = 0, const sadvio_pose_prior* priors = nullptr,
                                int n_imu = 0, const sadvio_imu_factor* imus = nullptr) {
    using io_detail::put;
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return "cannot open " + path;
    const int md = w.factor_type == SADVIO_FACTOR_ANGULAR ? 3 : 2;
    const int32_t has_ids = (w.kf_id && w.lmk_id) ? 1 : 0;
    const int32_t hdr[12] = {w.n_kf, w.n_cam, w.n_lmk, w.n_obs, w.factor_type, w.has_imu, n_prior, n_imu, w.lmk_const ? 1 : 0, w.cam_sigma ? 1 : 0, has_ids, 0};
    bool ok = put(f, "SADVIOW1", 8) && put(f, hdr, sizeof(hdr));
    if (has_ids) ok = ok && put(f, w.kf_id, 8 * (size_t)w.n_kf);
    ok = ok && put(f, w.kf_T_f_w, 96 * (size_t)w.n_kf);
    std::vector<uint8_t> kc((size_t)w.n_kf, 0);
    if (w.kf_const) std::memcpy(kc.data(), w.kf_const, (size_t)w.n_kf);
    ok = ok && put(f, kc.data(), kc.size());
    if (w.has_imu) {
        std::vector<double> z(3 * (size_t)w.n_kf, 0.0);
        ok = ok && put(f, w.kf_vel ? w.kf_vel : z.data(), 24 * (size_t)w.n_kf) && put(f, w.kf_ba ? w.kf_ba : z.data(), 24 * (size_t)w.n_kf) &&
             put(f, w.kf_bg ? w.kf_bg : z.data(), 24 * (size_t)w.n_kf);
    }
    ok = ok && put(f, w.cam_K, 32 * (size_t)w.n_cam) && put(f, w.cam_T_s_f, 96 * (size_t)w.n_cam);
    if (w.cam_sigma) ok = ok && put(f, w.cam_sigma, 8 * (size_t)w.n_cam);
    if (has_ids) ok = ok && put(f, w.lmk_id, 8 * (size_t)w.n_lmk);
    ok = ok && put(f, w.lmk_p, 24 * (size_t)w.n_lmk);
    if (w.lmk_const) ok = ok && put(f, w.lmk_const, (size_t)w.n_lmk);
    ok = ok && put(f, w.lmk_obs_ptr, 4 * ((size_t)w.n_lmk + 1)) && put(f, w.obs_kf, 4 * (size_t)w.n_obs) && put(f, w.obs_cam, 4 * (size_t)w.n_obs) &&
         put(f, w.obs_meas, 8 * (size_t)md * (size_t)w.n_obs);
    ok = ok && put(f, priors, sizeof(sadvio_pose_prior) * (size_t)n_prior) && put(f, imus, sizeof(sadvio_imu_factor) * (size_t)n_imu);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? "" : "short write to " + path;
}

inline std::string read_window(const std::string& path, WindowFile& o) {
    using io_detail::get;
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    std::vector<char> magic;
    std::vector<int32_t> hdr;
    bool ok = get(f, magic, 8) && std::memcmp(magic.data(), "SADVIOW1", 8) == 0 && get(f, hdr, 12);
    if (!ok) { std::fclose(f); return "not a SADVIOW1 file: " + path; }
    sadvio_flat_window& w = o.w;
    w.n_kf = hdr[0]; w.n_cam = hdr[1]; w.n_lmk = hdr[2]; w.n_obs = hdr[3]; w.factor_type = hdr[4]; w.has_imu = hdr[5];
    if (w.n_kf < 0 || w.n_cam < 0 || w.n_lmk < 0 || w.n_obs < 0 || hdr[6] < 0 || hdr[7] < 0) { std::fclose(f); return "corrupt header: " + path; }
    const int md = w.factor_type == SADVIO_FACTOR_ANGULAR ? 3 : 2;
    if (hdr[10]) ok = ok && get(f, o.kf_id, (size_t)w.n_kf);
    ok = ok && get(f, o.kf_T, 12 * (size_t)w.n_kf) && get(f, o.kf_const, (size_t)w.n_kf);
    if (w.has_imu) ok = ok && get(f, o.kf_vel, 3 * (size_t)w.n_kf) && get(f, o.kf_ba, 3 * (size_t)w.n_kf) && get(f, o.kf_bg, 3 * (size_t)w.n_kf);
    ok = ok && get(f, o.cam_K, 4 * (size_t)w.n_cam) && get(f, o.cam_T, 12 * (size_t)w.n_cam);
    if (hdr[9]) ok = ok && get(f, o.cam_sigma, (size_t)w.n_cam);
    if (hdr[10]) ok = ok && get(f, o.lmk_id, (size_t)w.n_lmk);
    ok = ok && get(f, o.lmk_p, 3 * (size_t)w.n_lmk);
    if (hdr[8]) ok = ok && get(f, o.lmk_const, (size_t)w.n_lmk);
    ok = ok && get(f, o.ptr, (size_t)w.n_lmk + 1) && get(f, o.obs_kf, (size_t)w.n_obs) && get(f, o.obs_cam, (size_t)w.n_obs) &&
         get(f, o.meas, (size_t)md * (size_t)w.n_obs) && get(f, o.priors, (size_t)hdr[6]) && get(f, o.imus, (size_t)hdr[7]);
    std::fclose(f);
    if (!ok) return "truncated file: " + path;
    w.kf_id = hdr[10] ? o.kf_id.data() : nullptr; w.kf_T_f_w = o.kf_T.data(); w.kf_const = o.kf_const.data();
    w.kf_vel = w.has_imu ? o.kf_vel.data() : nullptr; w.kf_ba = w.has_imu ? o.kf_ba.data() : nullptr; w.kf_bg = w.has_imu ? o.kf_bg.data() : nullptr;
    w.cam_K = o.cam_K.data(); w.cam_T_s_f = o.cam_T.data(); w.cam_sigma = hdr[9] ? o.cam_sigma.data() : nullptr;
    w.lmk_id = hdr[10] ? o.lmk_id.data() : nullptr; w.lmk_p = o.lmk_p.data(); w.lmk_const = hdr[8] ? o.lmk_const.data() : nullptr;
    w.lmk_obs_ptr = o.ptr.data(); w.obs_kf = o.obs_kf.data(); w.obs_cam = o.obs_cam.data(); w.obs_meas = o.meas.data();
    return "";
}

}  // namespace sadvio
