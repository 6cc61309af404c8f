"""Sparse (NFR) prior factors shaped like the sparse branch of addMarginalizationResiduals
(BundleAdjustmentCERESAnalytic.cpp:363-426): VIO = IMUPriordx on the kept frame + one PoseToLandmarkFactor per kept
landmark; VO = Landmark3DPrior on one landmark + a chain of LandmarkToLandmarkFactor."""
import numpy as np

from sadvio_amd import capi
from sadvio_amd.synthetic import T12_to_4


def spd_sqrt(rng, n, scale):
    A = rng.standard_normal((n, n))
    M = A @ A.T / n + np.eye(n)
    return scale * np.linalg.cholesky(M).T  # upper: W^T W = scale^2 M


def vio_sparse_priors(w, kf_keep, lmks, rng, noise=0.02):
    """Factors built at the window's current estimate + a perturbation of the "measured" values."""
    T = T12_to_4(w.kf_T_f_w[kf_keep])
    fs = [{"type": capi.SPARSE_IMU_PRIOR, "kf": kf_keep, "T_prior": w.kf_T_f_w[kf_keep].copy(),
           "v_prior": w.kf_vel[kf_keep] + noise * rng.standard_normal(3), "ba_prior": w.kf_ba[kf_keep].copy(),
           "bg_prior": w.kf_bg[kf_keep].copy(), "sqrt_inf": spd_sqrt(rng, 15, 5.0)}]
    for l in lmks:
        delta = T[:3, :3] @ w.lmk_p[l] + T[:3, 3] + noise * rng.standard_normal(3)
        fs.append({"type": capi.SPARSE_POSE_TO_LMK, "kf": kf_keep, "lmk0": int(l), "delta": delta, "sqrt_inf": spd_sqrt(rng, 3, 8.0)})
    return fs


def vo_sparse_priors(w, lmks, rng, noise=0.02):
    fs = [{"type": capi.SPARSE_LMK_PRIOR, "lmk0": int(lmks[0]), "delta": w.lmk_p[lmks[0]] + noise * rng.standard_normal(3),
           "sqrt_inf": spd_sqrt(rng, 3, 10.0)}]
    for a, b in zip(lmks[:-1], lmks[1:]):
        fs.append({"type": capi.SPARSE_LMK_TO_LMK, "lmk0": int(a), "lmk1": int(b),
                   "delta": w.lmk_p[a] - w.lmk_p[b] + noise * rng.standard_normal(3), "sqrt_inf": spd_sqrt(rng, 3, 8.0)})
    return fs
