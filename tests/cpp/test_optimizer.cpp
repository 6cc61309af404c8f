// C++ check of the host layer (include/sadvio_optimizer.hpp) against the reference's own solver-level acceptance
// style (imu_test.cpp:464-487: perturb, optimise, compare with the ground truth): a small stereo local map is built
// in plain structs, perturbed, solved through HipOptimizer::localMapBA / landmarkOptimization / singleFrameOptimization,
// and the recovered state is compared with the ground truth. Exit code 0 = pass. Needs a gfx950 device.
#include <array>
#include <cstdio>
#include <random>

#include "sadvio_optimizer.hpp"

using namespace sadvio;

static void project(const FrameState& f, int cam, const double* p, double& u, double& v) {
    const CameraModel& c = f.cameras[cam];
    double pf[3], ps[3];
    for (int i = 0; i < 3; i++) pf[i] = f.T_f_w.R[3 * i] * p[0] + f.T_f_w.R[3 * i + 1] * p[1] + f.T_f_w.R[3 * i + 2] * p[2] + f.T_f_w.t[i];
    for (int i = 0; i < 3; i++) ps[i] = c.T_s_f.R[3 * i] * pf[0] + c.T_s_f.R[3 * i + 1] * pf[1] + c.T_s_f.R[3 * i + 2] * pf[2] + c.T_s_f.t[i];
    u = c.fx * ps[0] / ps[2] + c.cx; v = c.fy * ps[1] / ps[2] + c.cy;
}

static LocalMapSnapshot make_map(std::mt19937& rng, int n_frames, int n_lmk) {
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    LocalMapSnapshot m;
    for (int i = 0; i < n_frames; i++) {                       // newest first: frame 0 is the furthest along +x
        FrameState f;
        f.id = 100 + i;
        f.T_f_w.t[0] = -0.3 * (n_frames - 1 - i);              // T_f_w = (I, -c): camera centre c = (0.3 k, 0, 0)
        CameraModel c0{458.654, 457.296, 367.215, 248.375, Pose()}, c1 = c0;
        c1.T_s_f.t[0] = -0.11;                                  // stereo baseline
        f.cameras = {c0, c1};
        m.frames.push_back(f);
    }
    m.frames.back().has_prior = true;                           // oldest frame: prior 100 * I (slamBiMono.cpp:17)
    m.frames.back().T_prior = m.frames.back().T_f_w;
    for (double& x : m.frames.back().inf_prior) x = 100.0;
    for (int l = 0; l < n_lmk; l++) {
        LandmarkState L;
        L.id = 5000 + l;
        L.p[0] = 2.0 * U(rng) + 0.3; L.p[1] = 1.2 * U(rng); L.p[2] = 4.0 + 2.0 * U(rng);
        for (int i = 0; i < n_frames; i++)
            for (int c = 0; c < 2; c++) {
                double u, v;
                project(m.frames[i], c, L.p, u, v);
                // well inside the reference's validity window [0, 2cx] x [0, 2cy] (Camera.cpp:131-133): a projection that
                // leaves it gets r = 0 with the Jacobian kept, and the landmark could not be pulled back
                if (u > 80 && u < 650 && v > 60 && v < 430) L.features.push_back({i, c, u, v});
            }
        if ((int)L.features.size() >= std::min(4, 2 * n_frames)) m.landmarks.push_back(L);
    }
    return m;
}

static double pose_err(const Pose& a, const Pose& b) {
    double e = 0;
    for (int i = 0; i < 9; i++) e = std::fmax(e, std::fabs(a.R[i] - b.R[i]));
    for (int i = 0; i < 3; i++) e = std::fmax(e, std::fabs(a.t[i] - b.t[i]));
    return e;
}

int main(int argc, char** argv) {
    std::mt19937 rng(20250404);
    std::normal_distribution<double> G(0.0, 1.0);
    int fails = 0;
    auto check = [&](bool ok, const char* what) { std::printf("%-70s %s\n", what, ok ? "ok" : "FAIL"); if (!ok) fails++; };
    HipOptimizer opt(0);

    // --- localMapBA: perturbed poses and landmarks return to the ground truth (noise-free measurements) ---
    {
        LocalMapSnapshot truth = make_map(rng, 5, 300), m = truth;
        for (size_t i = 0; i + 1 < m.frames.size(); i++) {
            double d[6] = {0.01 * G(rng), 0.01 * G(rng), 0.01 * G(rng), 0.03 * G(rng), 0.03 * G(rng), 0.03 * G(rng)};
            apply_pose_delta(m.frames[i].T_f_w, d);
        }
        for (auto& L : m.landmarks) for (double& x : L.p) x += 0.05 * G(rng);
        m.landmarks[3].outlier = true;                          // excluded from the problem, must stay untouched (…Analytic.cpp:239)
        const double p3 = m.landmarks[3].p[0];
        sadvio_solve_options unused; (void)unused;
        if (argc > 1) opt.set_dump_dir(argv[1]);
        for (int rep = 0; rep < 3; rep++) check(opt.localMapBA(m, 1), "localMapBA returns true");
        opt.set_dump_dir("");
        double worst = 0;
        for (size_t i = 0; i < m.frames.size(); i++) worst = std::fmax(worst, pose_err(m.frames[i].T_f_w, truth.frames[i].T_f_w));
        std::printf("   last_error: '%s'\n", opt.last_error().c_str());
        std::printf("   pose err %.3e, summary: it %d term %d cost %.3e -> %.3e\n", worst, opt.summary().iterations, opt.summary().termination, opt.summary().initial_cost, opt.summary().final_cost);
        check(worst < 1e-5, "localMapBA: poses recovered to 1e-5");
        double wl = 0;
        for (size_t l = 0; l < m.landmarks.size(); l++) if (l != 3) for (int a = 0; a < 3; a++) wl = std::fmax(wl, std::fabs(m.landmarks[l].p[a] - truth.landmarks[l].p[a]));
        { int worst_l = -1; double ww = 0; for (size_t l = 0; l < m.landmarks.size(); l++) if (l != 3) for (int a = 0; a < 3; a++) if (std::fabs(m.landmarks[l].p[a] - truth.landmarks[l].p[a]) > ww) { ww = std::fabs(m.landmarks[l].p[a] - truth.landmarks[l].p[a]); worst_l = (int)l; }
          std::printf("   lmk err %.3e at landmark %d with %zu features, p = %.3f %.3f %.3f\n", wl, worst_l, m.landmarks[worst_l].features.size(), truth.landmarks[worst_l].p[0], truth.landmarks[worst_l].p[1], truth.landmarks[worst_l].p[2]); }
        check(wl < 1e-4, "localMapBA: landmarks recovered to 1e-4");
        check(m.landmarks[3].p[0] == p3, "localMapBA: outlier landmark untouched");
        check(pose_err(m.frames.back().T_f_w, truth.frames.back().T_f_w) == 0.0, "localMapBA: fixed oldest frame untouched");
    }
    // --- landmarkOptimization: poses constant; landmarks that pass the chi2 gate at their OLD pose move back, the
    //     others are flagged outlier and left where they were (AOptimizer.cpp:124-141, ALandmark.cpp:130-146) ---
    {
        LocalMapSnapshot truth = make_map(rng, 4, 200), m = truth;
        for (auto& L : m.landmarks) for (double& x : L.p) x += 0.002 * G(rng);   // ~0.5 px: inside the Huber inlier region
        m.landmarks[5].p[0] += 0.2;                                               // ~20 px off: fails the 95 % chi2 test
        m.frames[0].is_keyframe = false;                                          // its features leave the residuals, not the gate
        const LocalMapSnapshot before = m;
        check(opt.landmarkOptimization(m), "landmarkOptimization returns true");
        double wl = 0, wp = 0, moved_out = 0;
        int n_in = 0, n_out = 0;
        for (size_t l = 0; l < m.landmarks.size(); l++) {
            if (m.landmarks[l].outlier) { n_out++; for (int a = 0; a < 3; a++) moved_out = std::fmax(moved_out, std::fabs(m.landmarks[l].p[a] - before.landmarks[l].p[a])); }
            else { n_in++; for (int a = 0; a < 3; a++) wl = std::fmax(wl, std::fabs(m.landmarks[l].p[a] - truth.landmarks[l].p[a])); }
        }
        for (size_t i = 0; i < m.frames.size(); i++) wp = std::fmax(wp, pose_err(m.frames[i].T_f_w, truth.frames[i].T_f_w));
        std::printf("   inliers %d outliers %d lmk err %.3e pose err %.3e it %d cost %.3e -> %.3e\n", n_in, n_out, wl, wp, opt.summary().iterations, opt.summary().initial_cost, opt.summary().final_cost);
        check(wl < 1e-3 && wp == 0.0, "landmarkOptimization: inlier landmarks recovered, frames untouched");
        check(m.landmarks[5].outlier && moved_out == 0.0, "landmarkOptimization: chi2-rejected landmarks flagged and untouched");
        check(n_in > 4 * n_out, "landmarkOptimization: most landmarks pass the gate");
    }
    // --- singleFrameOptimization: one frame against constant landmarks ---
    {
        LocalMapSnapshot truth = make_map(rng, 1, 300), m = truth;
        m.frames[0].has_prior = false;
        double d[6] = {0.02, -0.01, 0.015, 0.05, -0.04, 0.03};
        apply_pose_delta(m.frames[0].T_f_w, d);
        opt.singleFrameOptimization(m); opt.singleFrameOptimization(m);
        std::printf("   last_error: '%s'\n", opt.last_error().c_str());
        std::printf("   pose err %.3e it %d cost %.3e -> %.3e\n", pose_err(m.frames[0].T_f_w, truth.frames[0].T_f_w), opt.summary().iterations, opt.summary().initial_cost, opt.summary().final_cost);
        check(pose_err(m.frames[0].T_f_w, truth.frames[0].T_f_w) < 1e-5, "singleFrameOptimization: pose recovered to 1e-5");
        double wl = 0;
        for (size_t l = 0; l < m.landmarks.size(); l++) for (int a = 0; a < 3; a++) wl = std::fmax(wl, std::fabs(m.landmarks[l].p[a] - truth.landmarks[l].p[a]));
        check(wl == 0.0, "singleFrameOptimization: landmarks untouched");
    }
    // --- singleFrameOptimization ignores Frame::hasPrior: addSingleFrameResiduals adds no PosePriordx (…Analytic.cpp:5-50),
    //     although slamMono.cpp:18 puts a prior on every frame ---
    {
        LocalMapSnapshot truth = make_map(rng, 1, 300), m = truth;
        double d[6] = {0.02, -0.01, 0.015, 0.05, -0.04, 0.03};
        apply_pose_delta(m.frames[0].T_f_w, d);
        m.frames[0].has_prior = true; m.frames[0].T_prior = m.frames[0].T_f_w;   // a strong prior at the WRONG pose
        for (double& x : m.frames[0].inf_prior) x = 1e4;
        opt.singleFrameOptimization(m); opt.singleFrameOptimization(m);
        check(pose_err(m.frames[0].T_f_w, truth.frames[0].T_f_w) < 1e-5, "singleFrameOptimization: pose prior of the frame is not part of the problem");
    }
    // --- two marginalisations in a row in VO mode: the second prior folds the first one in (…Analytic.cpp:573-603; the
    //     previous prior has no kept frame in VO) ---
    {
        LocalMapSnapshot base = make_map(rng, 6, 300);
        auto drop_last = [](LocalMapSnapshot& s) {
            const int gone = (int)s.frames.size() - 1;
            s.frames.pop_back();
            for (auto& L : s.landmarks) {
                std::vector<Feature> kept;
                for (const Feature& ft : L.features) if (ft.frame != gone) kept.push_back(ft);
                L.features = kept;
            }
        };
        auto info_of = [](const HipOptimizer& o, int64_t id) {   // trace of the 3x3 information block J^T J of a kept landmark
            const auto& ids = o.prior_landmark_ids(); const auto& cols = o.prior_landmark_cols(); const auto& J = o.prior_J();
            const int n = o.prior_cols(), nf = o.prior_rows();
            for (size_t q = 0; q < ids.size(); q++)
                if (ids[q] == id && cols[q] >= 0) {
                    double tr = 0;
                    for (int r = 0; r < nf; r++) for (int a = 0; a < 3; a++) tr += J[(size_t)r * n + cols[q] + a] * J[(size_t)r * n + cols[q] + a];
                    return tr;
                }
            return -1.0;
        };
        LocalMapSnapshot chained = base;
        HipOptimizer oc(0), of(0);
        const bool ok1 = oc.marginalize(chained, 5, 4, false);
        const size_t kept1 = oc.prior_landmark_ids().size();
        const std::vector<int64_t> ids1 = oc.prior_landmark_ids();
        drop_last(chained);
        LocalMapSnapshot fresh = chained;
        for (auto& L : fresh.landmarks) L.has_prior = false;      // what a run that lost the first prior would see
        const bool ok2 = oc.marginalize(chained, 4, 3, false);
        const bool ok3 = of.marginalize(fresh, 4, 3, false);
        check(ok1 && ok2 && ok3 && kept1 > 10, "VO: two marginalisations in a row succeed");
        int n_more = 0, n_cmp = 0;
        for (int64_t id : ids1) {
            const double a = info_of(oc, id), b = info_of(of, id);
            if (a < 0 || b < 0) continue;
            n_cmp++; if (a > b * (1.0 + 1e-9)) n_more++;
        }
        std::printf("   landmarks kept by both priors %d, with more information in the chained prior %d (chained kept %zu, fresh kept %zu)\n",
                    n_cmp, n_more, oc.prior_landmark_ids().size(), of.prior_landmark_ids().size());
        check(n_cmp > 5 && n_more == n_cmp, "VO: the second prior carries the first one's information on the landmarks they share");
        check(oc.prior_landmark_ids().size() >= of.prior_landmark_ids().size(), "VO: landmarks of the first prior are resurrected into the second");
    }
    // --- localMapVIOptimization applies IMU::biasDeltaCorrection to the pre-integrations (AOptimizer.cpp:421-434,
    //     IMU.cpp:104-108); singleFrameVIOptimization does not ---
    {
        const int n = 4;
        const double dt = 0.2, gw[3] = {0, 0, -9.81};
        LocalMapSnapshot m = make_map(rng, n, 250);
        std::vector<std::array<double, 3>> P(n), V(n);
        for (int i = 0; i < n; i++) {    // T_f_w = (I, -c): position c
            P[i] = {-m.frames[i].T_f_w.t[0], -m.frames[i].T_f_w.t[1], -m.frames[i].T_f_w.t[2]};
            V[i] = {0.3 / dt, 0, 0};
            m.frames[i].has_imu = true;
            for (int a = 0; a < 3; a++) { m.frames[i].v[a] = V[i][a]; m.frames[i].ba[a] = 0.02 * G(rng); m.frames[i].bg[a] = 0.002 * G(rng); }
        }
        for (int i = n - 1; i > 0; i--) {   // older (i) -> newer (i - 1)
            ImuPair pr{};
            pr.frame_i = i; pr.frame_j = i - 1;
            sadvio_imu_factor& f = pr.f;
            f.dt = dt;
            const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(f.delta_R, I3, sizeof(I3));
            for (int a = 0; a < 3; a++) {
                f.delta_v[a] = V[i - 1][a] - V[i][a] - gw[a] * dt + 0.01 * G(rng);      // a little inconsistent: the biases have to move
                f.delta_p[a] = P[i - 1][a] - P[i][a] - V[i][a] * dt - 0.5 * gw[a] * dt * dt + 0.002 * G(rng);
                f.J_dv_ba[4 * a] = -dt; f.J_dp_ba[4 * a] = -0.5 * dt * dt; f.J_dR_bg[4 * a] = -dt;
                f.J_dv_bg[3 * a + (a + 1) % 3] = 0.3 * dt; f.J_dp_bg[3 * a + (a + 2) % 3] = 0.1 * dt * dt;
            }
            for (int q = 0; q < 9; q++) f.cov[10 * q] = q < 3 ? 1e-6 : (q < 6 ? 1e-4 : 1e-5);
            f.bacc_noise = 3e-3; f.bgyr_noise = 2e-5;
            m.imu_pairs.push_back(pr);
        }
        const LocalMapSnapshot before = m;
        check(opt.localMapVIOptimization(m, 1), "localMapVIOptimization returns true");
        double worst = 0, moved = 0;
        for (size_t k = 0; k < m.imu_pairs.size(); k++) {
            const ImuPair& a = before.imu_pairs[k];
            const ImuPair& b = m.imu_pairs[k];
            double dba[3], dbg[3];
            for (int q = 0; q < 3; q++) { dba[q] = m.frames[a.frame_i].ba[q] - before.frames[a.frame_i].ba[q]; dbg[q] = m.frames[a.frame_i].bg[q] - before.frames[a.frame_i].bg[q]; moved = std::fmax(moved, std::fabs(dba[q]) + std::fabs(dbg[q])); }
            sadvio_imu_factor want = a.f;
            bias_delta_correction(want, dba, dbg);
            for (int q = 0; q < 3; q++) worst = std::fmax(worst, std::fmax(std::fabs(want.delta_p[q] - b.f.delta_p[q]), std::fabs(want.delta_v[q] - b.f.delta_v[q])));
            for (int q = 0; q < 9; q++) worst = std::fmax(worst, std::fabs(want.delta_R[q] - b.f.delta_R[q]));
        }
        std::printf("   bias change %.3e, pre-integration mismatch after correction %.3e (it %d)\n", moved, worst, opt.summary().iterations);
        check(moved > 1e-6 && worst < 1e-12, "localMapVIOptimization: pre-integrations corrected for the solved bias deltas");
        LocalMapSnapshot s = before;
        const std::vector<ImuPair> pairs0 = s.imu_pairs;
        opt.singleFrameVIOptimization(s);
        bool same = true;
        for (size_t k = 0; k < pairs0.size(); k++) same &= std::memcmp(&pairs0[k].f, &s.imu_pairs[k].f, sizeof(sadvio_imu_factor)) == 0;
        check(same, "singleFrameVIOptimization leaves the pre-integrations alone");
    }
    // --- marginalize: the oldest frame goes into a prior that then holds the gauge of the remaining window ---
    // The dense prior is r0 + J dx with dx = the deltas of the NEXT solve: it says "stay where marginalize() found you"
    // (plus r0, whose sign follows the reference as coded: b = +sum J^T r, r0 = -Lambda^-1/2 U^T bk, SURVEY.md quirk B.7,
    // harmless at a converged state where bk = 0). It is therefore built and applied at the converged state.
    // The sparsified prior stores absolute targets and pulls a state perturbed afterwards back.
    for (int sparsif = 0; sparsif < 2; sparsif++) {
        LocalMapSnapshot truth = make_map(rng, 5, 300), m = truth;
        auto perturb = [&](LocalMapSnapshot& s, size_t n_frames) {
            for (size_t i = 0; i < n_frames; i++) { double d[6] = {0.002 * G(rng), 0.002 * G(rng), 0.002 * G(rng), 0.01 * G(rng), 0.01 * G(rng), 0.01 * G(rng)}; apply_pose_delta(s.frames[i].T_f_w, d); }
            for (auto& L : s.landmarks) for (double& x : L.p) x += 0.01 * G(rng);
        };
        const bool ok = opt.marginalize(m, 4, 3, sparsif != 0);
        check(ok && opt.has_prior() && opt.prior_rows() > 3, sparsif ? "marginalize (+ sparsification) builds a prior" : "marginalize builds a dense prior");
        if (sparsif) check(opt.sparse_factor_count() >= 2, "sparsification: landmark prior + chain factors");
        // drop frame0 from the window (it is the last one: newest-first order), keep its landmarks
        auto drop = [](LocalMapSnapshot& s) {
            s.frames.pop_back();
            for (auto& L : s.landmarks) {
                std::vector<Feature> kept;
                for (const Feature& ft : L.features) if (ft.frame != 4) kept.push_back(ft);
                L.features = kept;
            }
        };
        drop(m); drop(truth);
        if (sparsif) perturb(m, 4);
        double before = 0, worst = 0;
        for (size_t i = 0; i < m.frames.size(); i++) before = std::fmax(before, pose_err(m.frames[i].T_f_w, truth.frames[i].T_f_w));
        bool failed = false;
        for (int rep = 0; rep < (sparsif ? 4 : 1); rep++) {
            opt.localMapBA(m, 0);   // no fixed frame: only the prior anchors the window
            if (rep == 0) failed = opt.summary().termination == SADVIO_TERM_FAILURE;
        }
        for (size_t i = 0; i < m.frames.size(); i++) worst = std::fmax(worst, pose_err(m.frames[i].T_f_w, truth.frames[i].T_f_w));
        std::printf("   pose err %.3e -> %.3e (prior rows %d, sparse factors %zu, cost %.3e)\n", before, worst, opt.prior_rows(), opt.sparse_factor_count(), opt.summary().final_cost);
        // (the landmark chain may leave a weakly constrained direction: links whose marginal covariance has an eigenvalue below
        //  the 1e-12 cut drop it, marginalization.cpp:485,505 -- the window is pulled back, not necessarily to 1e-6)
        // Only the FIRST solve is required not to fail: the later repetitions start within 1e-10 of the converged point of a window whose
        // gauge is held by the prior alone (condition number ~1e10 along the landmark chain, see below). There the model cost change
        // of a step is rounding noise around zero, Ceres' `!(model_cost_change > 0)` makes it an invalid step, and five of them in a
        // row end the solve with FAILURE - or not, depending on the summation order of the device's atomics (2 runs in 16). The
        // adapter then leaves the state where it was (5e-11 instead of 3e-15 from the truth).
        check((sparsif ? worst < 0.3 * before : worst < 1e-6) && !failed,
              sparsif ? "window anchored by the sparsified prior is pulled back towards the ground truth" : "gauge-free window + dense prior: usable solve, state stays at the converged point");
    }
    {
        LocalMapSnapshot lonely = make_map(rng, 2, 3);
        lonely.landmarks.clear();
        check(!opt.marginalize(lonely, 1, 0, false) && !opt.has_prior(), "marginalize of a frame without landmarks is refused, prior cleared");
    }
    // --- VIInit: scale, gravity direction and velocities of a visual-only map from pre-integrated IMU factors ---
    // Non-rotating body: Delta_R = I, Delta_v = v_j - v_i - g dt, Delta_p = p_j - p_i - v_i dt - g dt^2 / 2 exactly
    // (IMU.cpp:5-91 with constant orientation), so any (p_k, v_k) sequence gives a consistent factor set.
    {
        const int n = 8;
        const double dt = 0.4, gw[3] = {0, 0, -9.81}, scale_in = 0.5, tilt[3] = {0.04, -0.07, 0.0};
        double Rwi[9];
        exp_so3(tilt, Rwi);
        std::vector<std::array<double, 3>> P(n), V(n);
        for (int k = 0; k < n; k++) {   // k = 0 oldest
            const double t = dt * k;
            P[k] = {1.5 * std::sin(0.7 * t), std::cos(0.5 * t), 0.4 * std::sin(0.9 * t)};
            V[k] = {1.05 * std::cos(0.7 * t), -0.5 * std::sin(0.5 * t), 0.36 * std::cos(0.9 * t)};
        }
        LocalMapSnapshot m;
        for (int i = 0; i < n; i++) {   // newest first
            const int k = n - 1 - i;
            FrameState f;
            f.id = 300 + i; f.has_imu = true;
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) f.T_f_w.R[3 * a + b] = Rwi[3 * b + a];   // R_in = R_true R_w_i^T, R_true = I
            for (int a = 0; a < 3; a++) {
                f.T_f_w.t[a] = 0;
                for (int b = 0; b < 3; b++) f.T_f_w.t[a] -= f.T_f_w.R[3 * a + b] * P[k][b] * scale_in;
                f.v[a] = V[k][a] * scale_in + 0.01 * G(rng);
            }
            if (i == n - 1) { f.has_prior = true; f.T_prior = f.T_f_w; }
            m.frames.push_back(f);
        }
        for (int k = 1; k < n; k++) {
            ImuPair pr{};
            pr.frame_i = n - k; pr.frame_j = n - 1 - k;   // older -> newer
            sadvio_imu_factor& f = pr.f;
            f.dt = dt;
            const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(f.delta_R, I3, sizeof(I3));
            for (int a = 0; a < 3; a++) {
                f.delta_v[a] = V[k][a] - V[k - 1][a] - gw[a] * dt;
                f.delta_p[a] = P[k][a] - P[k - 1][a] - V[k - 1][a] * dt - 0.5 * gw[a] * dt * dt;
            }
            for (int q = 0; q < 9; q++) f.cov[10 * q] = q < 3 ? 1e-6 : (q < 6 ? 1e-4 : 1e-5);
            f.bacc_noise = 3e-3; f.bgyr_noise = 2e-5;
            m.imu_pairs.push_back(pr);
        }
        LandmarkState L; L.p[0] = 1.0; L.p[1] = -2.0; L.p[2] = 3.0;
        m.landmarks.push_back(L);
        double R[9];
        const double s = opt.VIInit(m, R, true);
        double re = 0, ve = 0, fe = 0;
        for (int q = 0; q < 9; q++) re = std::fmax(re, std::fabs(R[q] - Rwi[q]));
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) {
                ve = std::fmax(ve, std::fabs(m.frames[i].v[a] - V[n - 1 - i][a]));
                for (int b = 0; b < 3; b++) fe = std::fmax(fe, std::fabs(m.frames[i].T_f_w.R[3 * a + b] - (a == b ? 1.0 : 0.0)));
            }
        std::printf("   scale %.6f (truth 2) R_w_i err %.2e velocity err %.2e frame rotation err %.2e it %d cost %.3e -> %.3e\n", s, re, ve, fe,
                    opt.summary().iterations, opt.summary().initial_cost, opt.summary().final_cost);
        check(std::fabs(s - 2.0) < 1e-2 && re < 2e-3, "VIInit: scale and gravity direction recovered");
        check(ve < 2e-2 && fe < 2e-3, "VIInit: velocities and frame rotations written back");
        double lp[3] = {0, 0, 0};
        for (int a = 0; a < 3; a++) lp[a] = s * (R[a] * 1.0 + R[3 + a] * -2.0 + R[6 + a] * 3.0);
        check(std::fabs(m.landmarks[0].p[0] - lp[0]) + std::fabs(m.landmarks[0].p[1] - lp[1]) + std::fabs(m.landmarks[0].p[2] - lp[2]) < 1e-12 &&
                  m.frames.back().inf_prior[0] == 100.0 && pose_err(m.frames.back().T_prior, m.frames.back().T_f_w) == 0.0,
              "VIInit: landmarks rescaled, prior re-anchored");
    }
    // --- non-pinhole cameras (double sphere): only the angular backend takes them, as in the reference; bearings come
    //     from the model's getRayCamera, the chi2 gate of landmarkOptimization runs on the host with the model's project ---
    {
        LocalMapSnapshot truth = make_map(rng, 4, 250), m;
        for (auto& f : truth.frames)
            for (auto& c : f.cameras) { c.kind = CameraKind::DoubleSphere; c.fx = 350.0; c.fy = 352.0; c.cx = 376.0; c.cy = 240.0; c.xi = -0.2; c.alpha = 0.58; c.width = 752; c.height = 480; }
        for (auto& L : truth.landmarks)       // re-observe every landmark through the double-sphere model
            for (auto& ft : L.features) {
                const FrameState& f = truth.frames[ft.frame];
                const CameraModel& c = f.cameras[ft.camera];
                double pf[3], pc[3];
                for (int a = 0; a < 3; a++) pf[a] = f.T_f_w.R[3 * a] * L.p[0] + f.T_f_w.R[3 * a + 1] * L.p[1] + f.T_f_w.R[3 * a + 2] * L.p[2] + f.T_f_w.t[a];
                for (int a = 0; a < 3; a++) pc[a] = c.T_s_f.R[3 * a] * pf[0] + c.T_s_f.R[3 * a + 1] * pf[1] + c.T_s_f.R[3 * a + 2] * pf[2] + c.T_s_f.t[a];
                if (!project_camera(c.intrinsics(), pc, ft.u, ft.v)) check(false, "double-sphere projection of a map point is valid");
            }
        m = truth;
        for (size_t i = 0; i + 1 < m.frames.size(); i++) { double d[6] = {0.005 * G(rng), 0.005 * G(rng), 0.005 * G(rng), 0.02 * G(rng), 0.02 * G(rng), 0.02 * G(rng)}; apply_pose_delta(m.frames[i].T_f_w, d); }
        for (auto& L : m.landmarks) for (double& x : L.p) x += 0.03 * G(rng);
        HipOptimizer ang(0, true);
        for (int rep = 0; rep < 3; rep++) ang.localMapBA(m, 1);
        double worst = 0;
        for (size_t i = 0; i < m.frames.size(); i++) worst = std::fmax(worst, pose_err(m.frames[i].T_f_w, truth.frames[i].T_f_w));
        std::printf("   double-sphere rig, angular backend: pose err %.3e it %d cost %.3e -> %.3e '%s'\n", worst, ang.summary().iterations, ang.summary().initial_cost, ang.summary().final_cost, ang.last_error().c_str());
        check(worst < 1e-5, "angular localMapBA with double-sphere cameras recovers the poses");
        LocalMapSnapshot m2 = truth;
        m2.landmarks[7].p[1] += 0.3;
        for (auto& L : m2.landmarks) for (double& x : L.p) x += 0.001 * G(rng);
        ang.landmarkOptimization(m2);
        int n_out = 0;
        for (auto& L : m2.landmarks) n_out += L.outlier;
        check(m2.landmarks[7].outlier && n_out < 10, "host-side chi2 gate with the double-sphere projection flags the displaced landmark only");
        LocalMapSnapshot m3 = truth;
        const Pose before = m3.frames[0].T_f_w;
        opt.localMapBA(m3, 1);   // pixel backend: refused, state untouched
        check(!opt.last_error().empty() && pose_err(m3.frames[0].T_f_w, before) == 0.0, "pixel backend refuses non-pinhole cameras");
    }
    // --- linexd landmarks in localMapBA (angular backend): noise-free end points, perturbed line poses; the line blocks are
    //     in the problem and the lines are written back with T_w_l * (exp(w), t). (The pixel line factor of the reference reads
    //     its 6-vector as a translation while its Jacobian is written for rotation + translation, …Analytic.h:121,156-160: a
    //     map whose only error sits in the lines does not move under it; that behaviour is covered by tests/test_gpu_lines.py.) ---
    {
        LocalMapSnapshot truth = make_map(rng, 5, 300), m;
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        for (int l = 0; l < 6; l++) {
            LineLandmarkState L;
            L.id = 9000 + l;
            const double w[6] = {0.4 * U(rng), 0.4 * U(rng), 0.4 * U(rng), 1.5 * U(rng) + 0.3, 0.8 * U(rng), 5.0 + U(rng)};
            apply_pose_delta(L.T_w_l, w);
            const double len = 0.6;
            L.model[0] = -0.5 * len; L.model[3] = 0.5 * len;
            for (int i = 0; i < 5; i++) {
                double pw[2][3], uv[2][2];
                for (int e = 0; e < 2; e++) {
                    for (int a = 0; a < 3; a++) pw[e][a] = L.T_w_l.R[3 * a] * L.model[3 * e] + L.T_w_l.t[a];
                    project(truth.frames[i], 0, pw[e], uv[e][0], uv[e][1]);
                }
                L.features.push_back({i, 0, uv[0][0], uv[0][1], uv[1][0], uv[1][1]});
            }
            truth.lines.push_back(L);
        }
        m = truth;
        for (auto& L : m.lines) { const double d[6] = {0, 0, 0, 0.02 * G(rng), 0.02 * G(rng), 0.02 * G(rng)}; apply_pose_delta(L.T_w_l, d); }
        m.lines[2].outlier = true;
        const Pose l2 = m.lines[2].T_w_l, l0 = m.lines[0].T_w_l;
        HipOptimizer angl(0, true);
        check(angl.localMapBA(m, 1), "localMapBA with linexd landmarks returns true");
        const double c0 = angl.summary().initial_cost, c1 = angl.summary().final_cost;
        std::printf("   lines: cost %.3e -> %.3e, it %d '%s'\n", c0, c1, angl.summary().iterations, angl.last_error().c_str());
        check(c0 > 0.0 && c1 <= c0, "linexd: the line residuals are in the cost (the as-coded Jacobians are inexact: it only must not grow)");
        check(pose_err(m.lines[0].T_w_l, l0) > 1e-4, "linexd: line poses are written back");
        check(pose_err(m.lines[2].T_w_l, l2) == 0.0, "linexd: outlier line untouched");
        double worst = 0;
        for (size_t i = 0; i < m.frames.size(); i++) worst = std::fmax(worst, pose_err(m.frames[i].T_f_w, truth.frames[i].T_f_w));
        check(worst < 1e-3, "linexd: key-frame poses stay at the ground truth");
    }
    // --- marginalizeRelative: the information of T_0_1 from the shared landmarks is symmetric positive definite, larger with
    //     more landmarks, and refused (zero matrix) when the two frames share nothing ---
    {
        LocalMapSnapshot m = make_map(rng, 3, 200);
        double inf[36], Ak[144];
        check(opt.marginalizeRelative(m, 0, 1, inf, Ak), "marginalizeRelative returns true");
        double asym = 0, mind = 1e300, tr = 0;
        for (int i = 0; i < 6; i++) { mind = std::fmin(mind, inf[7 * i]); tr += inf[7 * i]; for (int j = 0; j < 6; j++) asym = std::fmax(asym, std::fabs(inf[6 * i + j] - inf[6 * j + i])); }
        std::printf("   relative information: trace %.3e min diag %.3e asym %.3e '%s'\n", tr, mind, asym, opt.last_error().c_str());
        check(mind > 0.0 && asym <= 1e-9 * tr, "marginalizeRelative: information symmetric with a positive diagonal");
        LocalMapSnapshot few = m;
        few.landmarks.resize(40);
        double inf2[36];
        check(opt.marginalizeRelative(few, 0, 1, inf2), "marginalizeRelative on fewer landmarks");
        double tr2 = 0;
        for (int i = 0; i < 6; i++) tr2 += inf2[7 * i];
        check(tr2 < tr, "marginalizeRelative: fewer shared landmarks, less information");
        check(!opt.marginalizeRelative(m, 0, 0, inf2) && inf2[0] == 0.0, "marginalizeRelative: bad frame pair refused with a zero matrix");
    }
    // --- the back-end loop over a SEQUENCE (slamBiMonoVIO.cpp:561-614): 20 key-frame steps through this layer —
    //     marginalize(frame0, frame1) [+ sparsification on every other run] -> the frame leaves the window, a new key-frame enters with a
    //     perturbed pose / velocity and fresh, perturbed landmarks -> localMapVIOptimization with the prior the optimizer holds (by id).
    //     Noise-free measurements and pre-integrations: the window must stay at the ground truth, every solve usable.
    for (int sparsif = 0; sparsif < 2; sparsif++) {
        const int n_total = 26, n_win = 6;
        const double dt = 0.2, gw[3] = {0, 0, -9.81}, vx = 0.3 / dt;
        HipOptimizer seq;
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        // frame k (k = 0 oldest) at c = (0.3 k, 0, 0); landmarks spread along the path
        auto frame_at = [&](int k) {
            FrameState f;
            f.id = 1000 + k;
            f.T_f_w.t[0] = -0.3 * k;
            CameraModel c0{458.654, 457.296, 367.215, 248.375, Pose()}, c1 = c0;
            c1.T_s_f.t[0] = -0.11;
            f.cameras = {c0, c1};
            f.has_imu = true;
            f.v[0] = vx;
            return f;
        };
        std::vector<LandmarkState> all;
        for (int l = 0; l < 900; l++) {
            LandmarkState L;
            L.id = 90000 + l;
            L.p[0] = 0.3 * (n_total - 1) * 0.5 * (U(rng) + 1.0) + 1.5 * U(rng); L.p[1] = 1.2 * U(rng); L.p[2] = 4.0 + 2.0 * U(rng);
            all.push_back(L);
        }
        auto imu_pair = [&](int fi, int fj) {     // i older, j newer: exact deltas of the constant-velocity, non-rotating body
            ImuPair pr{};
            pr.frame_i = fi; pr.frame_j = fj;
            sadvio_imu_factor& f = pr.f;
            f.dt = dt;
            const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(f.delta_R, I3, sizeof(I3));
            const double dpos[3] = {0.3, 0, 0}, v0[3] = {vx, 0, 0};
            for (int a = 0; a < 3; a++) {
                f.delta_v[a] = -gw[a] * dt;
                f.delta_p[a] = dpos[a] - v0[a] * dt - 0.5 * gw[a] * dt * dt;
                f.J_dv_ba[4 * a] = -dt; f.J_dp_ba[4 * a] = -0.5 * dt * dt; f.J_dR_bg[4 * a] = -dt;
            }
            for (int q = 0; q < 9; q++) f.cov[10 * q] = q < 3 ? 1e-6 : (q < 6 ? 1e-4 : 1e-5);
            f.bacc_noise = 3e-3; f.bgyr_noise = 2e-5;
            return pr;
        };
        // window over the frames first .. first + n_win - 1 (stored newest first), the landmarks they see at least four times
        auto window = [&](int first, const std::vector<FrameState>& state, const std::vector<LandmarkState>& lm) {
            LocalMapSnapshot m;
            for (int i = 0; i < n_win; i++) m.frames.push_back(state[first + n_win - 1 - i]);
            for (const LandmarkState& L0 : lm) {
                LandmarkState L = L0;
                L.features.clear();
                for (int i = 0; i < n_win; i++)
                    for (int c = 0; c < 2; c++) {
                        double u, v;
                        project(frame_at(first + n_win - 1 - i), c, all[L.id - 90000].p, u, v);   // measurements of the TRUE geometry
                        if (u > 80 && u < 650 && v > 60 && v < 430) L.features.push_back({i, c, u, v});
                    }
                if ((int)L.features.size() >= 4) m.landmarks.push_back(L);
            }
            for (int i = n_win - 1; i > 0; i--) m.imu_pairs.push_back(imu_pair(i, i - 1));
            return m;
        };
        std::vector<FrameState> state;
        for (int k = 0; k < n_total; k++) state.push_back(frame_at(k));
        std::vector<LandmarkState> lm = all;
        state[0].has_prior = true; state[0].T_prior = state[0].T_f_w;
        for (double& x : state[0].inf_prior) x = 100.0;
        double worst = 0.0, worst_rel = 0.0;
        int failures = 0, refused = 0, steps = 0;
        auto rel = [](const Pose& a, const Pose& b) {   // T_a T_b^-1: the relative pose of two frames does not see the window's gauge
            Pose r;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double v = 0; for (int k = 0; k < 3; k++) v += a.R[3 * i + k] * b.R[3 * j + k]; r.R[3 * i + j] = v; }
            for (int i = 0; i < 3; i++) { double v = a.t[i]; for (int k = 0; k < 3; k++) v -= r.R[3 * i + k] * b.t[k]; r.t[i] = v; }
            return r;
        };
        for (int first = 0; first + n_win < n_total; first++, steps++) {
            // the key-frame that entered last and the landmarks only it and its predecessor see start from a perturbed estimate
            {
                FrameState& f = state[first + n_win - 1];
                double d[6] = {0.002 * G(rng), 0.002 * G(rng), 0.002 * G(rng), 0.01 * G(rng), 0.01 * G(rng), 0.01 * G(rng)};
                if (first > 0) { apply_pose_delta(f.T_f_w, d); for (int a = 0; a < 3; a++) f.v[a] += 0.02 * G(rng); }
            }
            LocalMapSnapshot m = window(first, state, lm);
            seq.localMapVIOptimization(m, first == 0 ? 1 : 0);     // after the first step only the prior anchors the window
            if (seq.summary().termination == SADVIO_TERM_FAILURE) failures++;
            for (int i = 0; i < n_win; i++) {
                state[first + n_win - 1 - i] = m.frames[i];
                worst = std::fmax(worst, pose_err(m.frames[i].T_f_w, frame_at(first + n_win - 1 - i).T_f_w));
                if (i + 1 < n_win)
                    worst_rel = std::fmax(worst_rel, pose_err(rel(m.frames[i].T_f_w, m.frames[i + 1].T_f_w),
                                                              rel(frame_at(first + n_win - 1 - i).T_f_w, frame_at(first + n_win - 2 - i).T_f_w)));
            }
            for (const LandmarkState& L : m.landmarks) lm[L.id - 90000] = L;
            if (!seq.marginalize(m, n_win - 1, n_win - 2, sparsif != 0)) refused++;
            for (const LandmarkState& L : m.landmarks) lm[L.id - 90000].has_prior = L.has_prior;
        }
        const auto st = seq.marg_stats();
        std::printf("   %d sliding steps (%s prior): worst pose error %.3e (between consecutive key-frames %.3e), failed solves %d, refused marginalisations %d, routes: %d calls / %d unpivoted / %d fell back\n",
                    steps, sparsif ? "sparsified" : "dense", worst, worst_rel, failures, refused, st[0], st[1], st[2]);
        // The dense prior carries the information exactly: the window stays at the truth (1e-11). The NFR factors of the sparsified prior
        // are an approximation (sparsifyVIO keeps the marginal of the frame and of every kept landmark, not their correlations) that leaves
        // the window weakly conditioned along its landmark links (see the single-step test above: four solves to pull a window back): ONE
        // <= 20-iteration solve per key-frame, as the reference runs it, does not re-converge a frame that entered 1e-2 off — the
        // error stays at the scale of the perturbation instead of accumulating (measured 8e-3 over 20 steps; the same loop against the
        // oracle: tests/test_gpu_sliding_long.py, 1e-8). Held here: no failed solve, no refused marginalisation, no growth.
        check(steps == 20 && failures == 0 && refused == 0 && (sparsif ? worst_rel < 3e-2 : worst < 1e-5),
              sparsif ? "20-step sliding sequence through marginalize + sparsification + localMapVIOptimization stays at the ground truth"
                      : "20-step sliding sequence through marginalize + localMapVIOptimization (dense resident prior) stays at the ground truth");
    }
    std::printf("%s (%d failure%s)\n", fails ? "FAILED" : "PASSED", fails, fails == 1 ? "" : "s");
    return fails ? 1 : 0;
}
