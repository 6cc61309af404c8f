"""CPU coverage of the N > 1 path: the landmark partition helper, and — with two real processes
over torch.distributed `gloo` — the protocol the sharded solve runs per LM step: each rank Schur-eliminates ITS
landmarks, one SUM all-reduce over [S | g | per-rank cost slots] gives every rank the identical reduced system,
every rank solves it redundantly and back-substitutes its own landmarks. The per-shard linearisation comes from
the CPU oracle (H, g of the un-reduced system); the result must equal the un-sharded first LM step."""
import os
import socket
import sys

import numpy as np
import pytest

from sadvio_amd import capi, sharding, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_landmark_ranges_balanced_contiguous_and_complete():
    w = synthetic.make_window(n_kf=6, n_lmk=333, seed=3)
    for world in (1, 2, 3, 8):
        rg = sharding.landmark_ranges(w.lmk_obs_ptr, world)
        assert len(rg) == world and rg[0][0] == 0 and rg[-1][1] == w.n_lmk
        assert all(rg[i][1] == rg[i + 1][0] for i in range(world - 1))
        obs = [int(w.lmk_obs_ptr[b] - w.lmk_obs_ptr[a]) for a, b in rg]
        assert max(obs) - min(obs) <= 5  # 5 observations per landmark: balanced to one landmark


def test_landmark_ranges_ragged_and_degenerate():
    ptr = np.array([0, 0, 7, 7, 8, 20, 20], dtype=np.int32)  # empty landmarks, one heavy landmark
    for world in (2, 4, 9):
        rg = sharding.landmark_ranges(ptr, world)
        assert rg[0][0] == 0 and rg[-1][1] == 6 and all(a <= b for a, b in rg)
        assert all(rg[i][1] == rg[i + 1][0] for i in range(world - 1))
    assert sharding.landmark_ranges(np.array([0], dtype=np.int32), 3) == [(0, 0)] * 3  # no landmarks at all


def test_sparsified_prior_follows_its_landmarks():
    """shard_window: the IMUPriordx factor of a sparsified VIO prior is replicated, every PoseToLandmarkFactor goes to the owner of
    its landmark with the index re-based to the shard; landmark-holding factor types are refused (sadvio_ba.h)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sparse_helpers import vio_sparse_priors
    from vio_helpers import make_vio_window
    w = make_vio_window(n_kf=5, n_lmk=90, seed=4)
    kept = list(range(0, 90, 3))
    w.sparse_priors = vio_sparse_priors(w, w.n_kf - 2, kept, np.random.default_rng(2), noise=0.01)
    world = 3
    rg = sharding.landmark_ranges(w.lmk_obs_ptr, world)
    seen = []
    for r in range(world):
        sh = sharding.shard_window(w, r, world)
        assert sum(1 for f in sh.sparse_priors if f["type"] == 0) == 1          # replicated on every rank
        for f in sh.sparse_priors:
            if f["type"] == 1:
                assert 0 <= f["lmk0"] < sh.n_lmk
                seen.append(rg[r][0] + f["lmk0"])
    assert sorted(seen) == kept                                                  # each pose-to-landmark factor exactly once
    bad = dict(w.sparse_priors[1]); bad["type"] = 2
    w.sparse_priors = [bad]
    with pytest.raises(ValueError):
        sharding.shard_window(w, 0, 2)


def test_shards_concatenate_back_to_the_window():
    w = synthetic.make_window(n_kf=5, n_lmk=101, seed=9)
    parts = [sharding.shard_window(w, r, 3) for r in range(3)]
    assert sum(p.n_lmk for p in parts) == w.n_lmk and sum(p.n_obs for p in parts) == w.n_obs
    assert np.array_equal(np.concatenate([p.lmk_id for p in parts]), w.lmk_id)
    assert np.array_equal(np.concatenate([p.obs_meas for p in parts]), w.obs_meas)
    for p in parts:
        assert p.lmk_obs_ptr[0] == 0 and p.lmk_obs_ptr[-1] == p.n_obs
        assert np.array_equal(p.kf_T_f_w, w.kf_T_f_w) and len(p.pose_priors) == len(w.pose_priors)


def _lm_diag(Hd, s, radius, lo=1e-6, hi=1e32):
    return np.clip(s * s * Hd, lo, hi) / radius / (s * s)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        w = synthetic.make_window(n_kf=5, n_lmk=90, seed=12)
        opts = capi.reference_options()
        radius = opts.initial_trust_region_radius
        sh = sharding.shard_window(w, rank, world)
        priors = sh.pose_priors
        sh.pose_priors = []                      # pose-only factors are added once, after the reduction
        _, _, H, g = oracle.first_step(sh, opts)  # un-reduced J^T J and J^T r of this rank's factors
        npz = 6 * int((w.kf_const == 0).sum())
        Hpp, Hpl, Hll = H[:npz, :npz], H[:npz, npz:], H[npz:, npz:]
        s_l = 1.0 / (1.0 + np.sqrt(np.diag(Hll)))
        Hll_d = Hll + np.diag(_lm_diag(np.diag(Hll), s_l, radius))  # landmark damping is local to the owner
        Minv = np.linalg.inv(Hll_d)                                  # block diagonal 3x3
        S = Hpp - Hpl @ Minv @ Hpl.T
        gred = g[:npz] - Hpl @ Minv @ g[npz:]
        # one buffer: [S | gred | diag(Hpp) | rank slots (cost-like partials)], SUM all-reduce == gather for the slots
        slots = np.zeros((world, 4)); slots[rank] = [float(g[npz:] @ g[npz:]), rank + 1.0, 0.0, 0.0]
        buf = torch.from_numpy(np.concatenate([S.ravel(), gred, np.diag(Hpp), slots.ravel()]))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf = buf.numpy()
        S = buf[:npz * npz].reshape(npz, npz); gred = buf[npz * npz:npz * npz + npz]
        hd = buf[npz * npz + npz:npz * npz + 2 * npz]; slots = buf[npz * npz + 2 * npz:].reshape(world, 4)
        # pose-only factors once (replicated evaluation), identical on all ranks
        wp = synthetic.make_window(n_kf=5, n_lmk=90, seed=12)
        wp_nol = sharding.shard_window(wp, 0, 1)
        empty = capi.FlatWindow(kf_T_f_w=wp.kf_T_f_w, kf_const=wp.kf_const, cam_K=wp.cam_K, cam_T_s_f=wp.cam_T_s_f,
                                cam_sigma=wp.cam_sigma, lmk_p=np.zeros((0, 3)), lmk_obs_ptr=np.zeros(1, dtype=np.int32),
                                obs_kf=np.zeros(0, dtype=np.int32), obs_cam=np.zeros(0, dtype=np.int32),
                                obs_meas=np.zeros((0, 2)))
        empty.pose_priors = priors
        _, _, Hp, gp = oracle.first_step(empty, opts)
        S = S + Hp; gred = gred + gp; hd = hd + np.diag(Hp)
        s_p = 1.0 / (1.0 + np.sqrt(hd))
        dpose = -np.linalg.solve(S + np.diag(_lm_diag(hd, s_p, radius)), gred)
        dl = -Minv @ (g[npz:] + Hpl.T @ dpose)   # back-substitution of this rank's landmarks
        q.put((rank, dpose, dl, slots.copy()))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_sharded_first_step_equals_unsharded(oracle_lib):
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = synthetic.make_window(n_kf=5, n_lmk=90, seed=12)
    dp, dl, _, _ = oracle_lib.first_step(w, capi.reference_options())
    free = w.kf_const == 0
    for rank, dpose, _, slots in res:
        assert np.allclose(dpose, dp[free].ravel(), rtol=1e-8, atol=1e-11)      # every rank: the un-sharded pose step
        assert np.array_equal(slots[:, 1], np.arange(1, world + 1))               # SUM over disjoint slots == gather
    assert np.array_equal(res[0][1], res[1][1])                                   # bit-identical on both ranks
    assert np.allclose(np.concatenate([r[2] for r in res]), dl.ravel(), rtol=1e-7, atol=1e-10)


def test_shards_of_a_window_with_a_dense_prior():
    """Dense prior on a sharded window: every kept landmark is on every rank (observations on rank 0 only), every free landmark on
    exactly one rank with all its observations; gather_landmarks restores the caller's array."""
    import numpy as np
    from sadvio_amd import sharding, synthetic
    w = synthetic.make_window(n_kf=6, n_lmk=500, seed=9)
    rng = np.random.default_rng(3)
    li = np.sort(rng.permutation(w.n_lmk)[:41]).astype(np.int32)
    lc = np.full(41, -1, dtype=np.int32)
    col = 0
    for i in range(41):
        if i == 1:
            continue                                  # a skipped landmark (lmk_col = -1) is an ordinary free one
        lc[i] = col; col += 3
    w.dense_prior = {"J": np.zeros((col, col)), "r0": np.zeros(col), "kf_keep": -1, "kf_col": 0, "lmk_index": li, "lmk_col": lc}
    kept = [int(l) for l, c in zip(li, lc) if c >= 0]
    for world in (2, 3, 5):
        shards = [sharding.shard_window(w, r, world) for r in range(world)]
        seen = np.zeros(w.n_lmk, dtype=int)
        n_obs = 0
        for r, s in enumerate(shards):
            assert s.n_lmk == s.n_own + len(kept)
            assert list(s.kept_src[s.n_own:]) == kept
            seen[s.kept_src[:s.n_own]] += 1
            n_obs += s.n_obs
            cnt = np.diff(s.lmk_obs_ptr)
            assert (cnt[s.n_own:] > 0).all() if r == 0 else (cnt[s.n_own:] == 0).all()
            # the prior's landmark list points at the shard's copies, in the prior's order
            for l, c, ls in zip(li, lc, s.dense_prior["lmk_index"]):
                if c >= 0:
                    assert s.kept_src[ls] == l and np.array_equal(s.lmk_p[ls], w.lmk_p[l])
            for j, l in enumerate(s.kept_src[:s.n_own]):
                a, b = s.lmk_obs_ptr[j], s.lmk_obs_ptr[j + 1]
                assert np.array_equal(s.obs_meas[a:b], w.obs_meas[w.lmk_obs_ptr[l]:w.lmk_obs_ptr[l + 1]])
        free = np.ones(w.n_lmk, dtype=bool); free[kept] = False
        assert (seen[free] == 1).all() and (seen[~free] == 0).all() and n_obs == w.n_obs
        vals = [np.tile(np.arange(s.n_lmk)[:, None] + 1000.0 * r, (1, 3)) for r, s in enumerate(shards)]
        out = sharding.gather_landmarks(w, shards, vals)
        for r, s in enumerate(shards):
            assert np.array_equal(out[s.kept_src[:s.n_own], 0], vals[r][:s.n_own, 0])
        assert np.array_equal(out[kept, 0], vals[0][shards[0].n_own:, 0])
