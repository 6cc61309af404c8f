"""Landmark partition of one window over several GPUs.

Landmarks (with all their observations) are split into `world` contiguous CSR ranges balanced by observation
count; key-frames, cameras, pose priors and IMU factors are replicated. A sparsified VIO prior follows: the IMUPriordx
factor is replicated, every PoseToLandmarkFactor goes to the owner of its landmark (index re-based to the shard). Contiguous ranges keep the landmark order
and ids of the caller: concatenating the shards' landmark deltas in rank order restores the window's array.

A DENSE prior (MarginalizationFactor) couples its kept landmarks with each other and with the kept frame: they stay in the reduced system,
which every rank solves redundantly — so every rank carries ALL kept landmarks and the whole prior. Their observations go to rank 0 only
(the other ranks hold them as variables without observations), the landmarks the prior does not hold are partitioned as before; rank 0
adds the prior's J^T J / J^T r to the all-reduced system, every rank evaluates its cost (sadvio_ba_set_dense_prior). `shard_window`
returns such a shard with `kept_src` (caller index of every landmark of the shard) and `gather_landmarks` reassembles the window's array.
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


def _shard_with_dense_prior(w: FlatWindow, rank: int, world: int) -> FlatWindow:
    dp = w.dense_prior
    li = np.asarray(dp.get("lmk_index", []), dtype=np.int64); lc = np.asarray(dp.get("lmk_col", []), dtype=np.int64)
    kept = [int(l) for l, c in zip(li, lc) if c >= 0]
    kept_set = set(kept)
    ptr = np.asarray(w.lmk_obs_ptr, dtype=np.int64)
    # the free landmarks, partitioned by observation count
    free = np.array([l for l in range(w.n_lmk) if l not in kept_set], dtype=np.int64)
    cnt = ptr[free + 1] - ptr[free]
    cum = np.concatenate([[0], np.cumsum(cnt)])
    cuts = [int(np.searchsorted(cum, (cum[-1] * r) // world, side="left")) for r in range(world)] + [len(free)]
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    mine = free[cuts[rank]:cuts[rank + 1]]
    src = np.concatenate([mine, np.array(kept, dtype=np.int64)]).astype(np.int64)      # this shard's landmarks: its free ones, then every kept one
    obs_idx, new_ptr = [], [0]
    for j, l in enumerate(src):
        own = j < len(mine) or rank == 0              # kept landmarks: observations on rank 0 only
        if own:
            obs_idx.extend(range(int(ptr[l]), int(ptr[l + 1])))
        new_ptr.append(len(obs_idx))
    obs_idx = np.array(obs_idx, dtype=np.int64)
    pos = {int(l): j for j, l in enumerate(src)}
    s = replace(
        w,
        lmk_p=np.ascontiguousarray(w.lmk_p[src]), lmk_obs_ptr=np.array(new_ptr, dtype=np.int32),
        obs_kf=np.ascontiguousarray(w.obs_kf[obs_idx]), obs_cam=np.ascontiguousarray(w.obs_cam[obs_idx]),
        obs_meas=np.ascontiguousarray(w.obs_meas[obs_idx]),
        lmk_id=None if w.lmk_id is None else np.ascontiguousarray(w.lmk_id[src]),
        lmk_const=None if w.lmk_const is None else np.ascontiguousarray(w.lmk_const[src]),
        _keep=[],
    )
    s.truth = {}
    s.dense_prior = dict(dp, lmk_index=np.array([pos.get(int(l), 0) for l in li], dtype=np.int32), lmk_col=np.asarray(lc, dtype=np.int32))
    if getattr(w, "sparse_priors", None):
        raise ValueError("a sharded window carries either a dense prior or the sparsified one")
    s.sparse_priors = []
    s.kept_src = src
    s.n_own = len(mine)
    return s


def gather_landmarks(w: FlatWindow, shards, per_shard_values):
    """The window's [n_lmk, 3] array from the shards' arrays (sharded with a dense prior: free landmarks from their owner, kept ones from rank 0)."""
    out = np.zeros((w.n_lmk, 3))
    for r, (s, v) in enumerate(zip(shards, per_shard_values)):
        n_own = s.n_own
        out[s.kept_src[:n_own]] = v[:n_own]
        if r == 0:
            out[s.kept_src[n_own:]] = v[n_own:]
    return out


def shard_window(w: FlatWindow, rank: int, world: int) -> FlatWindow:
    """The part of window `w` owned by `rank`: its landmark range, everything pose-side replicated."""
    if getattr(w, "dense_prior", None) is not None and world > 1:
        return _shard_with_dense_prior(w, rank, world)
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
