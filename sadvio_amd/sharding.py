"""Landmark partition of one window over several GPUs.

Landmarks (with all their observations) are split into `world` contiguous CSR ranges balanced by observation
count; key-frames, cameras, pose priors and IMU factors are replicated. A sparsified VIO prior follows: the IMUPriordx
factor is replicated, every PoseToLandmarkFactor goes to the owner of its landmark (index re-based to the shard). Contiguous ranges keep the landmark order
and ids of the caller: concatenating the shards' landmark deltas in rank order restores the window's array.
"""
from dataclasses import replace
from typing import List, Tuple

import numpy as np

from .capi import FlatWindow


def landmark_ranges(lmk_obs_ptr: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """`world` contiguous landmark ranges [l0, l1) with (nearly) equal observation counts."""
    ptr = np.asarray(lmk_obs_ptr, dtype=np.int64)
    n_lmk, n_obs = len(ptr) - 1, int(ptr[-1])
    cuts = [0]
    for r in range(1, world):
        target = (n_obs * r) // world
        l = int(np.searchsorted(ptr, target, side="left"))
        cuts.append(min(max(l, cuts[-1]), n_lmk))
    cuts.append(n_lmk)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_window(w: FlatWindow, rank: int, world: int) -> FlatWindow:
    """The part of window `w` owned by `rank`: its landmark range, everything pose-side replicated."""
    l0, l1 = landmark_ranges(w.lmk_obs_ptr, world)[rank]
    o0, o1 = int(w.lmk_obs_ptr[l0]), int(w.lmk_obs_ptr[l1])
    s = replace(
        w,
        lmk_p=np.ascontiguousarray(w.lmk_p[l0:l1]),
        lmk_obs_ptr=np.ascontiguousarray(w.lmk_obs_ptr[l0:l1 + 1] - o0).astype(np.int32),
        obs_kf=np.ascontiguousarray(w.obs_kf[o0:o1]), obs_cam=np.ascontiguousarray(w.obs_cam[o0:o1]),
        obs_meas=np.ascontiguousarray(w.obs_meas[o0:o1]),
        lmk_id=None if w.lmk_id is None else np.ascontiguousarray(w.lmk_id[l0:l1]),
        lmk_const=None if w.lmk_const is None else np.ascontiguousarray(w.lmk_const[l0:l1]),
        _keep=[],
    )
    s.truth = {}
    # NFR factors (capi.SPARSE_*: 0 = IMU prior, 1 = pose-to-landmark): pose-only ones everywhere, landmark ones with their landmark
    sp = []
    for f in getattr(w, "sparse_priors", None) or []:
        if f["type"] == 0:
            sp.append(dict(f))
        elif f["type"] == 1:
            if l0 <= f["lmk0"] < l1:
                g = dict(f); g["lmk0"] = int(f["lmk0"]) - l0
                sp.append(g)
        else:
            raise ValueError("a sharded window carries IMU-prior and pose-to-landmark factors only (sadvio_ba.h)")
    s.sparse_priors = sp
    return s
