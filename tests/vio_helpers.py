"""Synthetic visual-inertial windows for the parity tests (test infrastructure: uses the oracle's restated
IMU::processIMU to obtain realistic covariance / bias-Jacobian constants)."""
import numpy as np

from imu_helpers import CFG, Chain, arr
from oracle import oracle
from sadvio_amd import capi, synthetic
from sadvio_amd.synthetic import T12_to_4, T_to_12, exp_so3, inv4

G = np.array([0, 0, -9.81])


def log_so3(R):
    return oracle.so3_log(R)


def make_vio_window(n_kf=6, n_lmk=300, seed=3, dt=0.25, factor=capi.FACTOR_PIXEL, fixed=1, noise=True, obs_per_lmk=5, **window_kw):
    """VO window of `synthetic.make_window` + per-key-frame (v, ba, bg) + IMUFactor/IMUBiasFactor between
    consecutive key-frames (AOptimizer.cpp:55-92). Pre-integrated deltas are made consistent with the ground
    truth trajectory (plus noise of the propagated covariance); covariance and bias Jacobians come from the
    restated processIMU run on constant measurements over the same interval at 200 Hz."""
    w = synthetic.make_window(n_kf=n_kf, n_lmk=n_lmk, seed=seed, factor=factor, fixed=fixed, obs_per_lmk=obs_per_lmk, **window_kw)
    rng = np.random.default_rng(seed + 77)
    Tt = [T12_to_4(t) for t in w.truth["T_f_w"]]            # newest first
    pos = [inv4(T)[:3, 3] for T in Tt]
    # time runs from the oldest (index n_kf-1) to the newest (index 0)
    vel = np.zeros((n_kf, 3))
    for i in range(n_kf):
        older = min(i + 1, n_kf - 1); newer = max(i - 1, 0)
        vel[i] = (pos[newer] - pos[older]) / (dt * max(1, older - newer))
    ba_true = 0.02 * rng.standard_normal(3); bg_true = 0.002 * rng.standard_normal(3)
    w.has_imu = 1
    w.kf_vel = vel + (0.05 * rng.standard_normal(vel.shape) if noise else 0.0)
    w.kf_ba = np.tile(ba_true, (n_kf, 1)) + (0.005 * rng.standard_normal((n_kf, 3)) if noise else 0.0)
    w.kf_bg = np.tile(bg_true, (n_kf, 1)) + (0.0005 * rng.standard_normal((n_kf, 3)) if noise else 0.0)
    w.imu_factors = []
    n_steps = int(round(dt * 200))
    for j in range(n_kf - 2, -1, -1):        # kf_j newer, kf_i = j + 1 older
        i = j + 1
        acc = np.array([0.3, -0.2, 9.7]) + 0.1 * rng.standard_normal(3)
        gyr = 0.1 * rng.standard_normal(3)
        ch = Chain(acc, gyr, 1e9, ba=w.kf_ba[i], bg=w.kf_bg[i])
        cur = None
        for s in range(1, n_steps + 1):
            cur = ch.step(acc, gyr, 1e9 + s * 5e6)
        f = oracle.imu_factor_dict(i, j, cur, dt, CFG["bacc_noise"], CFG["bgyr_noise"])
        Ri, Rj = Tt[i][:3, :3], Tt[j][:3, :3]
        dR = Ri @ Rj.T
        dv = Ri @ (vel[j] - vel[i] - G * dt)
        dp = Ri @ (pos[j] - pos[i] - vel[i] * dt - 0.5 * G * dt * dt)
        if noise:
            cov = np.array(f["cov"]).reshape(9, 9)
            n9 = np.linalg.cholesky(cov + 1e-18 * np.eye(9)) @ rng.standard_normal(9)
            dR = dR @ exp_so3(n9[:3]); dv = dv + n9[3:6]; dp = dp + n9[6:9]
        f["delta_R"], f["delta_v"], f["delta_p"] = dR.ravel(), dv, dp
        w.imu_factors.append(f)
    return w
