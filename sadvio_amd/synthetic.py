"""Seeded synthetic sliding windows of the BASELINE.json shapes.

EuRoC-shaped stereo rig (intrinsics / extrinsics are the public EuRoC calibration constants the
reference ships in ros/config/dataset/eth.yaml:11-14,18-20,30-33,36-38), key-frames on a gently
curving trajectory, landmarks in the union frustum at 1-15 m depth (the reference caps
triangulation at 20 m, Point3DlandmarkInitializer.cpp:92), each observed by exactly
`obs_per_lmk` (kf, cam) views, 1 px measurement noise, perturbed initial state, oldest key-frame
fixed (`fixed_frame_number: 1`, config.yaml:35) and carrying a 100*I PosePriordx
(slamBiMono.cpp:17). Key-frames are stored newest first (amap.h:28-32).
"""
from __future__ import annotations

import numpy as np

from .capi import FACTOR_ANGULAR, FACTOR_PIXEL, FlatWindow

# EuRoC MAV calibration (T_BS = body <- sensor), public dataset constants.
_T_BS0 = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                   [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                   [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                   [0, 0, 0, 1.0]])
_T_BS1 = np.array([[0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556],
                   [0.999598781151, 0.0130119051815, 0.0251588363115, 0.0453689425024],
                   [-0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038],
                   [0, 0, 0, 1.0]])
_K0 = np.array([458.654, 457.296, 367.215, 248.375])
_K1 = np.array([457.587, 456.134, 379.999, 255.238])


def exp_so3(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    S = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        return np.eye(3) + S
    a = S / th
    return np.eye(3) + (1 - np.cos(th)) * a @ a + np.sin(th) * a


def T_to_12(T4):
    return np.concatenate([T4[:3, :3].reshape(9), T4[:3, 3]])


def T12_to_4(t):
    T = np.eye(4)
    T[:3, :3] = np.asarray(t[:9]).reshape(3, 3)
    T[:3, 3] = t[9:12]
    return T


def inv4(T):
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


def _body_poses(n_kf, length, rng):
    """World<-body poses, oldest first. Body axes (EuRoC): x up, y right, z forward."""
    radius = 25.0
    s = np.linspace(0.0, length, n_kf)
    th = s / radius
    pos = np.stack([radius * np.sin(th), radius * (1 - np.cos(th)), 0.05 * np.sin(0.7 * s)], axis=1)
    Ts = []
    for i in range(n_kf):
        yaw = th[i] + 0.02 * rng.standard_normal()
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        R = np.stack([up, right, fwd], axis=1)  # columns = body axes in world
        R = R @ exp_so3(0.03 * rng.standard_normal(3))
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = pos[i]
        Ts.append(T)
    return Ts


def make_window(n_kf=20, n_lmk=8000, obs_per_lmk=5, seed=20250404, factor=FACTOR_PIXEL, pixel_noise=1.0,
                rot_perturb_deg=0.5, trans_perturb=0.02, lmk_perturb=0.05, length=10.0, fixed=1,
                width=752, height=480, min_depth=1.0, max_depth=15.0, border=5.0, band=None) -> FlatWindow:
    """One VO window (config 2 of BASELINE.json at the defaults: 20 KF x 8 000 landmarks x 40 000 factors).

    band: if set, a landmark seeded in key-frame k is only observed from key-frames within k +- band (the
    co-visibility band of a long trajectory, configs 4 / 5); candidates are then projected in bounded chunks."""
    rng = np.random.default_rng(seed)
    T_w_b = _body_poses(n_kf, length, rng)  # oldest first
    T_s_f = [inv4(_T_BS0), inv4(_T_BS1)]     # frame(body) -> sensor
    Ks = [_K0, _K1]
    n_views = 2 * n_kf
    if obs_per_lmk > n_views:
        raise ValueError(f"obs_per_lmk={obs_per_lmk} exceeds the {n_views} views of {n_kf} stereo key-frames")
    # world -> sensor transforms per (kf, cam), kf index here = oldest-first position
    T_s_w = np.stack([T_s_f[c] @ inv4(T_w_b[k]) for k in range(n_kf) for c in range(2)])  # [n_views,4,4]
    Kv = np.stack([Ks[c] for k in range(n_kf) for c in range(2)])

    lmk = np.zeros((0, 3))
    runs = np.zeros((0,), dtype=np.int64)
    valid_all = np.zeros((0, n_views), dtype=bool)
    chunk = None if band is None else max(512, int(2e6 // n_views))
    while lmk.shape[0] < n_lmk:
        n_try = int(1.6 * (n_lmk - lmk.shape[0])) + 64
        if chunk is not None:
            n_try = min(n_try, chunk)
        k = rng.integers(0, n_kf, n_try)
        u = rng.uniform(20, width - 20, n_try)
        v = rng.uniform(20, height - 20, n_try)
        d = rng.uniform(min_depth, max_depth, n_try)
        pc = np.stack([(u - _K0[2]) / _K0[0] * d, (v - _K0[3]) / _K0[1] * d, d, np.ones(n_try)], axis=1)
        T_w_s = np.stack([T_w_b[i] @ _T_BS0 for i in k])
        pw = np.einsum("nij,nj->ni", T_w_s, pc)[:, :3]
        # project into every view
        ph = np.concatenate([pw, np.ones((n_try, 1))], axis=1)
        pcam = np.einsum("vij,nj->nvi", T_s_w, ph)[:, :, :3]
        z = pcam[:, :, 2]
        uu = Kv[None, :, 0] * pcam[:, :, 0] / z + Kv[None, :, 2]
        vv = Kv[None, :, 1] * pcam[:, :, 1] / z + Kv[None, :, 3]
        # keep measurements inside the reference's validity window [0, 2cx] x [0, 2cy] (Camera.cpp:131-133)
        umax = np.minimum(width, 2 * Kv[None, :, 2]) - border
        vmax = np.minimum(height, 2 * Kv[None, :, 3]) - border
        valid = (z > 0.5) & (uu > border) & (uu < umax) & (vv > border) & (vv < vmax)
        if band is not None:
            view_kf = np.arange(n_views) // 2
            valid &= np.abs(view_kf[None, :] - k[:, None]) <= band
        ok = valid.sum(axis=1) >= obs_per_lmk
        lmk = np.concatenate([lmk, pw[ok]])
        valid_all = np.concatenate([valid_all, valid[ok]])
    lmk = lmk[:n_lmk]
    valid_all = valid_all[:n_lmk]

    # choose obs_per_lmk views: a random contiguous run (in kf-major, cam-minor order) of valid views
    obs_view = np.zeros((n_lmk, obs_per_lmk), dtype=np.int64)
    for i in range(n_lmk):
        idx = np.flatnonzero(valid_all[i])
        s0 = rng.integers(0, len(idx) - obs_per_lmk + 1)
        obs_view[i] = idx[s0:s0 + obs_per_lmk]
    # sort landmarks by their first observing view: consecutive landmarks then share key-frames,
    # like a live SLAM map where landmark ids are created in time order
    order = np.argsort(obs_view[:, 0], kind="stable")
    lmk = lmk[order]
    obs_view = obs_view[order]

    obs_l = np.repeat(np.arange(n_lmk), obs_per_lmk)
    ov = obs_view.reshape(-1)
    kf_old = ov // 2
    cam = (ov % 2).astype(np.int32)
    ph = np.concatenate([lmk[obs_l], np.ones((len(obs_l), 1))], axis=1)
    pcam = np.einsum("nij,nj->ni", T_s_w[ov], ph)[:, :3]
    uv = np.stack([Kv[ov, 0] * pcam[:, 0] / pcam[:, 2] + Kv[ov, 2], Kv[ov, 1] * pcam[:, 1] / pcam[:, 2] + Kv[ov, 3]],
                  axis=1)
    uv_noisy = uv + pixel_noise * rng.standard_normal(uv.shape)

    # newest first ordering for the window (amap.h:28-32)
    kf_new = (n_kf - 1 - kf_old).astype(np.int32)
    T_f_w_true = np.stack([T_to_12(inv4(T_w_b[n_kf - 1 - i])) for i in range(n_kf)])
    kf_const = np.zeros(n_kf, dtype=np.uint8)
    for i in range(n_kf):
        if i > n_kf - fixed - 1:  # BundleAdjustmentCERESAnalytic.cpp:219
            kf_const[i] = 1

    # perturbed initial state
    T_f_w0 = T_f_w_true.copy()
    for i in range(n_kf):
        if kf_const[i]:
            continue
        dw = np.deg2rad(rot_perturb_deg) * rng.standard_normal(3)
        dt = trans_perturb * rng.standard_normal(3)
        T = T12_to_4(T_f_w_true[i])
        D = np.eye(4)
        D[:3, :3] = exp_so3(dw)
        D[:3, 3] = dt
        T_f_w0[i] = T_to_12(T @ D)
    lmk0 = lmk + lmk_perturb * rng.standard_normal(lmk.shape)

    cam_K = np.stack(Ks)
    cam_T = np.stack([T_to_12(T_s_f[0]), T_to_12(T_s_f[1])])
    if factor == FACTOR_PIXEL:
        meas = uv_noisy
        sigma = np.array([1.0, 1.0])  # …Analytic.h:46 default
    else:
        # unit bearing K^-1 [u v 1] normalised (Camera.cpp:15-25); sigma = 1.5 / f (…Angular….cpp:283)
        b = np.stack([(uv_noisy[:, 0] - cam_K[cam, 2]) / cam_K[cam, 0],
                      (uv_noisy[:, 1] - cam_K[cam, 3]) / cam_K[cam, 1], np.ones(len(cam))], axis=1)
        meas = b / np.linalg.norm(b, axis=1, keepdims=True)
        f = 0.5 * (cam_K[:, 0] + cam_K[:, 1])  # Camera.h:46
        sigma = 1.5 / f

    w = FlatWindow(
        kf_T_f_w=T_f_w0, kf_const=kf_const, cam_K=cam_K, cam_T_s_f=cam_T, cam_sigma=sigma, lmk_p=lmk0,
        lmk_obs_ptr=(np.arange(n_lmk + 1) * obs_per_lmk).astype(np.int32), obs_kf=kf_new, obs_cam=cam,
        obs_meas=meas, factor_type=factor, has_imu=0,
        kf_id=(1000 + np.arange(n_kf)[::-1]).astype(np.int64),
        lmk_id=(500000 + 7 * np.arange(n_lmk)).astype(np.int64),
    )
    # PosePriordx 100*I on the oldest key-frame (slamBiMono.cpp:17)
    w.pose_priors = [(n_kf - 1, T_f_w_true[n_kf - 1].copy(), 100.0 * np.ones(6))]
    w.truth = {"T_f_w": T_f_w_true, "lmk": lmk}
    return w


def apply_pose_delta(T12, d6):
    """T_f_w <- T_f_w * (exp(w), t)  (AOptimizer.cpp:329-332; parametersBlock.hpp:34-37)."""
    T = T12_to_4(T12)
    D = np.eye(4)
    D[:3, :3] = exp_so3(d6[:3])
    D[:3, 3] = d6[3:6]
    return T_to_12(T @ D)


def pose_distance(Ta12, Tb12):
    """(rotation angle [rad], translation distance [m]) between two 12-vectors."""
    Ra, Rb = np.asarray(Ta12[:9]).reshape(3, 3), np.asarray(Tb12[:9]).reshape(3, 3)
    c = np.clip(0.5 * (np.trace(Ra @ Rb.T) - 1), -1, 1)
    return float(np.arccos(c)), float(np.linalg.norm(np.asarray(Ta12[9:]) - np.asarray(Tb12[9:])))


def pre_marginalize(w, kf0):
    """Selection rule of Marginalization::preMarginalize (marginalization.cpp:50-88) on a flat window:
    a landmark of frame0 is ignored unless it has exactly 2 features in frame0 (stereo); it is marginalised
    when all its features are in frame0 ("lonely"), kept otherwise."""
    keep, marg = [], []
    for l in range(w.n_lmk):
        o = slice(w.lmk_obs_ptr[l], w.lmk_obs_ptr[l + 1])
        kfs = w.obs_kf[o]
        if not (kfs == kf0).any():
            continue
        if (kfs == kf0).sum() != 2:
            continue
        (marg if (kfs == kf0).all() else keep).append(l)
    return keep, marg
