"""Synthetic visual-inertial initialisation problems (AOptimizer::VIInit, AOptimizer.cpp:448-581): key-frames along a
smooth trajectory with IMU pre-integration between consecutive key-frames (the oracle's restated processIMU), then the
"visual" trajectory is scaled and expressed in a world frame whose z axis is tilted away from gravity — what VIInit has
to recover (scale = exp(lambda), R_w_i)."""
import numpy as np

from imu_helpers import CFG, Chain, arr, factor_dict
from sadvio_amd.synthetic import T12_to_4, T_to_12, exp_so3, inv4

G = np.array([0.0, 0.0, -9.81])


def trajectory(t):
    """Position / velocity / acceleration / body rotation rate of a smooth inertial-frame trajectory."""
    p = np.array([1.5 * np.sin(0.7 * t), 1.0 * np.cos(0.5 * t), 0.4 * np.sin(0.9 * t)])
    v = np.array([1.5 * 0.7 * np.cos(0.7 * t), -1.0 * 0.5 * np.sin(0.5 * t), 0.4 * 0.9 * np.cos(0.9 * t)])
    a = np.array([-1.5 * 0.49 * np.sin(0.7 * t), -1.0 * 0.25 * np.cos(0.5 * t), -0.4 * 0.81 * np.sin(0.9 * t)])
    return p, v, a


def make_viinit(n_kf=10, dt_kf=0.5, rate=200.0, scale=0.5, tilt=(0.05, -0.08), seed=0, vel_noise=0.0):
    """Returns dict(T_f_w [n,12], vel [n,3], factors, truth={scale, R_w_i, vel}) with frames NEWEST FIRST (as
    LocalMap::getLastNFramesIn returns them)."""
    rng = np.random.default_rng(seed)
    omega_b = np.array([0.1, -0.05, 0.2])          # constant body rate
    dt = 1.0 / rate
    n_steps = int(round(dt_kf * rate))
    R = np.eye(3)
    cfg = dict(CFG); cfg["rate_hz"] = rate
    states = []
    # first key-frame
    t = 0.0
    p, v, a = trajectory(t)
    T_w_f = np.eye(4); T_w_f[:3, :3] = R; T_w_f[:3, 3] = p
    acc = R.T @ (a - G)
    ch = Chain(acc, omega_b, 1e9, T_f_w=T_to_12(inv4(T_w_f)), v=v, cfg=cfg)
    states.append(dict(T_w_f=T_w_f.copy(), v=v.copy(), imu=None))
    factors_old_first = []
    for k in range(1, n_kf):
        cur = None
        for _ in range(n_steps):
            t += dt
            R = R @ exp_so3(omega_b * dt)
            p, v, a = trajectory(t)
            acc = R.T @ (a - G)
            cur = ch.step(acc, omega_b, 1e9 + t * 1e9)
        T_w_f = np.eye(4); T_w_f[:3, :3] = R; T_w_f[:3, 3] = p
        cur.T_f_w[:] = list(T_to_12(inv4(T_w_f)))
        cur.v[:] = list(v)
        factors_old_first.append(factor_dict(k - 1, k, cur, dt_kf, cfg))
        states.append(dict(T_w_f=T_w_f.copy(), v=v.copy()))
        ch.set_keyframe(cur)
    # What IMUFactorInit models (residuals.hpp:326-336): R_f_inertial = R_f_w R_w_i, and e^lambda (p_j - p_i) — the
    # frame positions of the INPUT poses — is the inertial displacement. So the input is R_in = R_true R_w_i^T with
    # positions p_in = p_true * scale (t_in = -R_in p_in), velocities v_in = v_true * scale (what the reference's own
    # VIInit test feeds, imu_test.cpp:800-845: translations and velocities multiplied by scale_factor).
    R_w_i = exp_so3(np.array([tilt[0], tilt[1], 0.0]))
    n = len(states)
    T_in = np.zeros((n, 12)); vel = np.zeros((n, 3)); vel_true = np.zeros((n, 3))
    for k, s in enumerate(states):
        R_f_w = s["T_w_f"][:3, :3].T
        Tin = np.eye(4)
        Tin[:3, :3] = R_f_w @ R_w_i.T
        Tin[:3, 3] = -Tin[:3, :3] @ (s["T_w_f"][:3, 3] * scale)
        T_in[k] = T_to_12(Tin)
        vel_true[k] = s["v"]
        vel[k] = s["v"] * scale + vel_noise * rng.standard_normal(3)
    # newest first
    order = np.arange(n)[::-1]
    remap = {int(o): i for i, o in enumerate(order)}
    factors = []
    for f in factors_old_first[::-1]:
        f = dict(f); f["kf_i"] = remap[f["kf_i"]]; f["kf_j"] = remap[f["kf_j"]]
        factors.append(f)
    return dict(T_f_w=T_in[order], vel=vel[order], factors=factors,
                truth=dict(scale=1.0 / scale, R_w_i=R_w_i, tilt=np.array(tilt), vel=vel_true[order]))
