import os

import numpy as np

from sadvio_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_window(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    w = capi.FlatWindow(kf_T_f_w=g["kf_T_f_w"], kf_const=g["kf_const"], cam_K=g["cam_K"], cam_T_s_f=g["cam_T_s_f"],
                        cam_sigma=g["cam_sigma"], lmk_p=g["lmk_p"], lmk_obs_ptr=g["lmk_obs_ptr"], obs_kf=g["obs_kf"],
                        obs_cam=g["obs_cam"], obs_meas=g["obs_meas"], factor_type=int(g["factor_type"]),
                        kf_id=g["kf_id"], lmk_id=g["lmk_id"])
    w.pose_priors = [(int(k), T, i) for k, T, i in zip(g["prior_kf"], g["prior_T"], g["prior_inf"])]
    return w, g
