"""Windows for the marginalisation tests: some landmarks of frame0 are made "lonely" (observed by frame0's stereo pair
only) so that both lists of preMarginalize (marginalization.cpp:50-88) are populated."""
import numpy as np


def with_lonely_landmarks(w, kf0, n_lonely):
    """Drop the observations outside kf0 of the first n_lonely landmarks that kf0 sees with both cameras."""
    keep_obs = np.ones(w.n_obs, dtype=bool)
    done = 0
    for l in range(w.n_lmk):
        if done == n_lonely:
            break
        o = np.arange(w.lmk_obs_ptr[l], w.lmk_obs_ptr[l + 1])
        in0 = w.obs_kf[o] == kf0
        if in0.sum() == 2:
            keep_obs[o[~in0]] = False
            done += 1
    cnt = np.array([keep_obs[w.lmk_obs_ptr[l]:w.lmk_obs_ptr[l + 1]].sum() for l in range(w.n_lmk)])
    w.lmk_obs_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    w.obs_kf = np.ascontiguousarray(w.obs_kf[keep_obs]); w.obs_cam = np.ascontiguousarray(w.obs_cam[keep_obs])
    w.obs_meas = np.ascontiguousarray(w.obs_meas[keep_obs])
    w._keep = []
    return w
