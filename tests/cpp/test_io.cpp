// CPU check of include/sadvio_io.hpp: writes a small window with a prior and an IMU factor, reads it back, compares.
// Usage: test_io <path>   (the file stays on disk for the Python reader test). Exit code 0 = pass.
#include <cstdio>

#include "sadvio_io.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const int n_kf = 3, n_cam = 2, n_lmk = 4;
    int64_t kf_id[n_kf] = {7, 8, 9}, lmk_id[n_lmk] = {100, 101, 102, 103};
    double kf_T[12 * n_kf], cam_K[4 * n_cam], cam_T[12 * n_cam], cam_sigma[n_cam] = {1.0, 1.5}, lmk_p[3 * n_lmk];
    double vel[3 * n_kf], ba[3 * n_kf], bg[3 * n_kf];
    uint8_t kf_const[n_kf] = {0, 0, 1}, lmk_const[n_lmk] = {0, 1, 0, 0};
    for (int i = 0; i < 12 * n_kf; i++) kf_T[i] = 0.01 * i;
    for (int i = 0; i < 4 * n_cam; i++) cam_K[i] = 400.0 + i;
    for (int i = 0; i < 12 * n_cam; i++) cam_T[i] = -0.02 * i;
    for (int i = 0; i < 3 * n_lmk; i++) lmk_p[i] = 1.0 + 0.5 * i;
    for (int i = 0; i < 3 * n_kf; i++) { vel[i] = 0.1 * i; ba[i] = 0.001 * i; bg[i] = -0.001 * i; }
    int32_t ptr[n_lmk + 1] = {0, 2, 2, 5, 7}, obs_kf[7] = {0, 1, 0, 1, 2, 2, 0}, obs_cam[7] = {0, 1, 0, 0, 1, 1, 0};
    double meas[14];
    for (int i = 0; i < 14; i++) meas[i] = 10.0 * i + 0.25;
    sadvio_flat_window w{};
    w.n_kf = n_kf; w.n_cam = n_cam; w.n_lmk = n_lmk; w.n_obs = 7; w.factor_type = SADVIO_FACTOR_PIXEL; w.has_imu = 1;
    w.kf_id = kf_id; w.kf_T_f_w = kf_T; w.kf_const = kf_const; w.kf_vel = vel; w.kf_ba = ba; w.kf_bg = bg;
    w.cam_K = cam_K; w.cam_T_s_f = cam_T; w.cam_sigma = cam_sigma; w.lmk_id = lmk_id; w.lmk_p = lmk_p; w.lmk_const = lmk_const;
    w.lmk_obs_ptr = ptr; w.obs_kf = obs_kf; w.obs_cam = obs_cam; w.obs_meas = meas;
    sadvio_pose_prior pr{};
    pr.kf = 2;
    for (int i = 0; i < 12; i++) pr.T_prior[i] = 0.5 + i;
    for (int i = 0; i < 6; i++) pr.inf_diag[i] = 100.0;
    sadvio_imu_factor f{};
    f.kf_i = 1; f.kf_j = 0; f.dt = 0.25;
    for (int i = 0; i < 81; i++) f.cov[i] = (i % 10 == 0) ? 1e-4 * (1 + i) : 0.0;
    for (int i = 0; i < 9; i++) f.delta_R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    f.delta_v[2] = 2.45; f.delta_p[2] = 0.31; f.bacc_noise = 3e-3; f.bgyr_noise = 2e-5;
    std::string e = sadvio::write_window(argv[1], w, 1, &pr, 1, &f);
    if (!e.empty()) { std::printf("write: %s\n", e.c_str()); return 1; }
    sadvio::WindowFile r;
    e = sadvio::read_window(argv[1], r);
    if (!e.empty()) { std::printf("read: %s\n", e.c_str()); return 1; }
    bool ok = r.w.n_kf == n_kf && r.w.n_obs == 7 && r.w.has_imu == 1 && r.priors.size() == 1 && r.imus.size() == 1;
    ok = ok && !std::memcmp(r.w.kf_T_f_w, kf_T, sizeof(kf_T)) && !std::memcmp(r.w.obs_meas, meas, sizeof(meas)) && !std::memcmp(r.w.lmk_obs_ptr, ptr, sizeof(ptr));
    ok = ok && !std::memcmp(r.w.kf_const, kf_const, n_kf) && !std::memcmp(r.w.lmk_const, lmk_const, n_lmk) && !std::memcmp(r.w.lmk_id, lmk_id, sizeof(lmk_id));
    ok = ok && !std::memcmp(&r.priors[0], &pr, sizeof(pr)) && !std::memcmp(&r.imus[0], &f, sizeof(f)) && !std::memcmp(r.w.kf_bg, bg, sizeof(bg));
    sadvio::WindowFile bad;
    ok = ok && !sadvio::read_window(std::string(argv[1]) + ".missing", bad).empty();
    std::printf("%s\n", ok ? "PASSED" : "FAILED");
    return ok ? 0 : 1;
}
