"""Synthetic linexd landmarks for a window of sadvio_amd.synthetic.make_window (test infrastructure).

A line landmark is a pose T_w_l whose x axis is the line direction, with the two model points (-0.5, 0, 0) and
(0.5, 0, 0) scaled by the line length (ModelLine3D, reference landmarks/Line3D.h). A feature holds the two projected
end points (pixel) or their two bearing vectors (angular).
"""
import numpy as np

from sadvio_amd import capi
from sadvio_amd.synthetic import T12_to_4, T_to_12, exp_so3


def add_lines(w, n_line=6, obs_per_line=4, seed=5, noise_px=0.3, pert_rot=0.02, pert_t=0.03, length=0.8, n_const=0):
    rng = np.random.default_rng(seed)
    n_kf = w.n_kf
    T_f_w = [T12_to_4(t) for t in (w.truth.get("T_f_w", w.kf_T_f_w))]
    T_s_f = [T12_to_4(t) for t in w.cam_T_s_f]
    T_true, T_init, model, ptr, okf, ocam, meas = [], [], [], [0], [], [], []
    for l in range(n_line):
        kfs = np.sort(rng.choice(n_kf, size=min(obs_per_line, n_kf), replace=False))
        # centre in front of the first observing key-frame
        Twf = np.linalg.inv(T_f_w[kfs[0]])
        c = (Twf @ np.linalg.inv(T_s_f[0]) @ np.array([rng.uniform(-1.0, 1.0), rng.uniform(-0.7, 0.7), rng.uniform(4.0, 8.0), 1.0]))[:3]
        R = exp_so3(rng.normal(size=3) * 1.2)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = c
        Ti = T.copy()
        Ti[:3, :3] = R @ exp_so3(rng.normal(size=3) * pert_rot)
        Ti[:3, 3] = c + rng.normal(size=3) * pert_t
        m = np.array([-0.5 * length, 0, 0, 0.5 * length, 0, 0])
        T_true.append(T_to_12(T)); T_init.append(T_to_12(Ti)); model.append(m)
        for kf in kfs:
            cam = int(rng.integers(w.n_cam))
            Tsw = T_s_f[cam] @ T_f_w[kf]
            pts = [(Tsw @ T @ np.array([*m[3 * i:3 * i + 3], 1.0]))[:3] for i in range(2)]
            if min(p[2] for p in pts) < 0.5:
                continue
            if w.factor_type == capi.FACTOR_PIXEL:
                fx, fy, cx, cy = w.cam_K[cam]
                z = [np.array([fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy]) + rng.normal(size=2) * noise_px for p in pts]
            else:
                z = []
                for p in pts:
                    b = p / np.linalg.norm(p) + rng.normal(size=3) * noise_px * 1e-3
                    z.append(b / np.linalg.norm(b))
            okf.append(int(kf)); ocam.append(cam); meas.append(np.concatenate(z))
        ptr.append(len(okf))
    const = np.zeros(n_line, dtype=np.uint8)
    const[:n_const] = 1
    w.lines = dict(id=np.arange(n_line, dtype=np.int64) + 7000, T_w_l=np.array(T_init), model=np.array(model),
                   obs_ptr=np.array(ptr, dtype=np.int32), obs_kf=np.array(okf, dtype=np.int32), obs_cam=np.array(ocam, dtype=np.int32),
                   obs_meas=np.array(meas), const=const)
    w.truth["line_T_w_l"] = np.array(T_true)
    return w
