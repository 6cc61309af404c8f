"""Generates tests/golden/*.npz: seeded inputs + expected outputs of the BA hot path.

The reference (C++/Eigen/Ceres/OpenCV) cannot be built or imported in this environment, and it stores no
numeric goldens for its visual factors. These vectors are produced by the repo's CPU oracle
AFTER it has been pinned against the reference tests' known answers (tests/test_oracle_*.py); they freeze
those answers so that (a) the oracle cannot drift silently and (b) the HIP path is checked on the GPU box
against committed numbers as well as against the live oracle.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from sadvio_amd import capi, synthetic  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def window_arrays(w):
    return dict(kf_T_f_w=w.kf_T_f_w, kf_const=w.kf_const, cam_K=w.cam_K, cam_T_s_f=w.cam_T_s_f, cam_sigma=w.cam_sigma,
                lmk_p=w.lmk_p, lmk_obs_ptr=w.lmk_obs_ptr, obs_kf=w.obs_kf, obs_cam=w.obs_cam, obs_meas=w.obs_meas,
                kf_id=w.kf_id, lmk_id=w.lmk_id, factor_type=np.int32(w.factor_type),
                prior_kf=np.array([p[0] for p in w.pose_priors], dtype=np.int32),
                prior_T=np.array([p[1] for p in w.pose_priors]).reshape(-1, 12),
                prior_inf=np.array([p[2] for p in w.pose_priors]).reshape(-1, 6))


def make(name, factor, seed):
    w = synthetic.make_window(n_kf=5, n_lmk=120, seed=seed, factor=factor)
    w.to_c()
    # a prior on a free key-frame as well (first frames of a SLAM run carry one, slamBiMono.cpp:17)
    w.pose_priors.append((0, w.kf_T_f_w[0].copy(), 100.0 * np.ones(6)))
    rng = np.random.default_rng(seed + 1)
    pd = 0.01 * rng.standard_normal((w.n_kf, 6)); ld = 0.03 * rng.standard_normal((w.n_lmk, 3))
    r0, Jp0, Jl0, v0 = oracle.linearize(w)
    r1, Jp1, Jl1, v1 = oracle.linearize(w, pd, ld)
    out = window_arrays(w)
    out.update(lin0_r=r0, lin0_Jp=Jp0, lin0_Jl=Jl0, lin_pose_delta=pd, lin_lmk_delta=ld, lin1_r=r1, lin1_Jp=Jp1,
               lin1_Jl=Jl1)
    for tag, opts in (("ref", capi.reference_options()), ("gn5", capi.gn_options(5))):
        res = oracle.solve(w, opts)
        s = res["summary"]
        out.update({f"{tag}_pose": res["pose"], f"{tag}_lmk": res["lmk"],
                    f"{tag}_summary": np.array([s.iterations, s.num_successful_steps, s.num_unsuccessful_steps,
                                                s.termination, s.initial_cost, s.final_cost, s.fixed_cost,
                                                s.final_radius]),
                    f"{tag}_log": res["log"]})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "iterations", int(out["ref_summary"][0]), "cost", out["ref_summary"][4], "->", out["ref_summary"][5])


if __name__ == "__main__":
    make("window_pixel_5kf", capi.FACTOR_PIXEL, 101)
    make("window_angular_5kf", capi.FACTOR_ANGULAR, 202)
