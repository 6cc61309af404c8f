"""On-disk format of a flattened window ("SADVIOW1", include/sadvio_io.hpp): Python reader / writer.

A live SaDVIO run with the adapter of INTEGRATION.md can dump every problem it hands to the optimizer
(`sadvio::write_window`); `scripts/replay.py` reads those files and solves them with the GPU library and, when
present, the CPU oracle — real-data parity without linking the reference on the GPU box."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .capi import FACTOR_ANGULAR, FlatWindow, ImuFactorC, PosePriorC, fill_imu_factor

MAGIC = b"SADVIOW1"
_IMU_ARRAYS = ("delta_R", "delta_v", "delta_p", "J_dR_bg", "J_dv_ba", "J_dv_bg", "J_dp_ba", "J_dp_bg", "cov")


def _pad(b: bytes) -> bytes:
    return b + b"\0" * ((8 - len(b) % 8) % 8)


def save_window(path: str, w: FlatWindow) -> None:
    md = 3 if w.factor_type == FACTOR_ANGULAR else 2
    has_ids = int(w.kf_id is not None and w.lmk_id is not None)
    hdr = np.array([w.n_kf, w.n_cam, w.n_lmk, w.n_obs, w.factor_type, w.has_imu, len(w.pose_priors), len(w.imu_factors),
                    int(w.lmk_const is not None), int(w.cam_sigma is not None), has_ids, 0], dtype="<i4")
    f64 = lambda a, n: np.ascontiguousarray(a, dtype="<f8").reshape(-1)[:n].tobytes()
    out = [MAGIC, hdr.tobytes()]
    if has_ids:
        out.append(np.ascontiguousarray(w.kf_id, dtype="<i8").tobytes())
    out.append(f64(w.kf_T_f_w, 12 * w.n_kf))
    out.append(_pad(np.ascontiguousarray(w.kf_const, dtype=np.uint8).tobytes()))
    if w.has_imu:
        for a in (w.kf_vel, w.kf_ba, w.kf_bg):
            out.append(f64(np.zeros((w.n_kf, 3)) if a is None else a, 3 * w.n_kf))
    out += [f64(w.cam_K, 4 * w.n_cam), f64(w.cam_T_s_f, 12 * w.n_cam)]
    if w.cam_sigma is not None:
        out.append(f64(w.cam_sigma, w.n_cam))
    if has_ids:
        out.append(np.ascontiguousarray(w.lmk_id, dtype="<i8").tobytes())
    out.append(f64(w.lmk_p, 3 * w.n_lmk))
    if w.lmk_const is not None:
        out.append(_pad(np.ascontiguousarray(w.lmk_const, dtype=np.uint8).tobytes()))
    for a in (w.lmk_obs_ptr, w.obs_kf, w.obs_cam):
        out.append(_pad(np.ascontiguousarray(a, dtype="<i4").tobytes()))
    out.append(f64(w.obs_meas, md * w.n_obs))
    pri, _ = w.priors_c()
    out.append(bytes(pri)[: C.sizeof(PosePriorC) * len(w.pose_priors)])
    imu, _ = w.imus_c()
    out.append(bytes(imu)[: C.sizeof(ImuFactorC) * len(w.imu_factors)])
    with open(path, "wb") as fh:
        fh.write(b"".join(out))


def load_window(path: str) -> FlatWindow:
    buf = open(path, "rb").read()
    if buf[:8] != MAGIC:
        raise ValueError(f"{path}: not a SADVIOW1 file")
    pos = 8
    hdr = np.frombuffer(buf, dtype="<i4", count=12, offset=pos); pos += 48
    n_kf, n_cam, n_lmk, n_obs, ftype, has_imu, n_prior, n_imu, has_lc, has_sig, has_ids, _ = (int(x) for x in hdr)
    md = 3 if ftype == FACTOR_ANGULAR else 2

    def take(dtype, count):
        nonlocal pos
        item = np.dtype(dtype).itemsize
        if pos + item * count > len(buf):
            raise ValueError(f"{path}: truncated")
        a = np.frombuffer(buf, dtype=dtype, count=count, offset=pos).copy()
        pos += item * count
        pos += (8 - pos % 8) % 8
        return a

    kf_id = take("<i8", n_kf) if has_ids else None
    kf_T = take("<f8", 12 * n_kf).reshape(n_kf, 12)
    kf_const = take(np.uint8, n_kf)
    vel = ba = bg = None
    if has_imu:
        vel, ba, bg = (take("<f8", 3 * n_kf).reshape(n_kf, 3) for _ in range(3))
    cam_K = take("<f8", 4 * n_cam).reshape(n_cam, 4)
    cam_T = take("<f8", 12 * n_cam).reshape(n_cam, 12)
    cam_sigma = take("<f8", n_cam) if has_sig else np.ones(n_cam)
    lmk_id = take("<i8", n_lmk) if has_ids else None
    lmk_p = take("<f8", 3 * n_lmk).reshape(n_lmk, 3)
    lmk_const = take(np.uint8, n_lmk) if has_lc else None
    ptr = take("<i4", n_lmk + 1)
    obs_kf = take("<i4", n_obs)
    obs_cam = take("<i4", n_obs)
    meas = take("<f8", md * n_obs).reshape(n_obs, md)
    w = FlatWindow(kf_T_f_w=kf_T, kf_const=kf_const, cam_K=cam_K, cam_T_s_f=cam_T, cam_sigma=cam_sigma, lmk_p=lmk_p,
                   lmk_obs_ptr=ptr, obs_kf=obs_kf, obs_cam=obs_cam, obs_meas=meas, factor_type=ftype, has_imu=has_imu,
                   kf_id=kf_id, lmk_id=lmk_id, lmk_const=lmk_const, kf_vel=vel, kf_ba=ba, kf_bg=bg)
    ps = C.sizeof(PosePriorC)
    for k in range(n_prior):
        p = PosePriorC.from_buffer_copy(buf, pos + ps * k)
        w.pose_priors.append((int(p.kf), np.array(p.T_prior[:]), np.array(p.inf_diag[:])))
    pos += ps * n_prior
    fs = C.sizeof(ImuFactorC)
    for k in range(n_imu):
        f = ImuFactorC.from_buffer_copy(buf, pos + fs * k)
        d = {"kf_i": int(f.kf_i), "kf_j": int(f.kf_j), "dt": float(f.dt), "bacc_noise": float(f.bacc_noise), "bgyr_noise": float(f.bgyr_noise)}
        for name in _IMU_ARRAYS:
            d[name] = np.array(getattr(f, name)[:])
        chk = ImuFactorC(); fill_imu_factor(chk, d)   # the dict must round-trip through the C struct
        w.imu_factors.append(d)
    pos += fs * n_imu
    if pos != len(buf):
        raise ValueError(f"{path}: {len(buf) - pos} trailing bytes")
    return w
