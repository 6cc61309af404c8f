"""A window spanning several "devices": the landmark-sharded solve with one all-reduce of the reduced system per
LM step. The GPU box has ONE MI355X, so two handles (two HIP streams on device 0, two host
threads) stand in for two ranks and the collective is a host-mediated sum installed through
sadvio_ba_set_collective; the RCCL path itself (sadvio_ba_comm_init_rccl) is exercised with world = 1."""
import ctypes as C
import threading

import numpy as np
import pytest

from sadvio_amd import capi, sharding, synthetic

from golden_util import lmk_err

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-6
LMK_TOL = 1e-6   # = the pose bar; relative for landmarks that move by more than a metre (golden_util.lmk_err)


class HostAllReduce:
    """In-place sum over `world` device buffers through host memory (test stand-in for ncclAllReduce)."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.bufs = [None] * world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.calls = 0
        self.max_count = 0

    def fn(self, rank):
        def allreduce(ctx, dev, count, stream):
            try:
                if self.hip.hipStreamSynchronize(stream) != 0:
                    return 1
                a = np.empty(count, dtype=np.float64)
                if self.hip.hipMemcpy(a.ctypes.data, dev, 8 * count, 2) != 0:  # device -> host
                    return 2
                self.bufs[rank] = a
                self.barrier.wait(timeout=60)
                total = self.bufs[0].copy()
                for r in range(1, self.world):
                    total += self.bufs[r]
                self.barrier.wait(timeout=60)
                if self.hip.hipMemcpy(dev, total.ctypes.data, 8 * count, 1) != 0:  # host -> device
                    return 3
                if rank == 0:
                    self.calls += 1
                    self.max_count = max(self.max_count, int(count))
                return 0
            except Exception:
                return 4
        return allreduce


def solve_sharded(backend_cls, w, opts, world):
    coll = HostAllReduce(world)
    out = [None] * world

    def run(rank):
        be = backend_cls(device=0)
        try:
            be.set_collective(rank, world, coll.fn(rank))
            be.set_windows([sharding.shard_window(w, rank, world)])
            s = be.solve(opts)[0]
            out[rank] = (s, be.get_deltas(0), be.get_ids(0))
        except Exception as e:  # surface in the main thread
            out[rank] = e
            coll.barrier.abort()
        finally:
            be.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for o in out:
        if isinstance(o, Exception):
            raise o
    return out, coll


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_window_matches_single_device_and_oracle(backend_cls, oracle_lib, world):
    w = synthetic.make_window(n_kf=8, n_lmk=1500, seed=41)
    opts = capi.reference_options()
    out, coll = solve_sharded(backend_cls, w, opts, world)
    ref = oracle_lib.solve(w, opts)
    rs = ref["summary"]
    lmk = np.concatenate([o[1]["lmk"] for o in out])
    ids = np.concatenate([o[2][1] for o in out])
    assert np.array_equal(ids, w.lmk_id)  # concatenation in rank order restores the caller's landmark order
    for s, d, _ in out:
        assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
        assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-9) and np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-10)
        assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL
    # every rank holds bit-identical pose deltas (they solved the same all-reduced system)
    for r in range(1, world):
        assert np.array_equal(out[r][1]["pose"], out[0][1]["pose"])
    assert lmk_err(lmk, ref["lmk"]) <= LMK_TOL
    assert coll.calls == 2 * rs.iterations or coll.calls >= 2  # two collectives per executed LM step


def test_sharded_config4_window_out_of_lds(backend_cls):
    """Config 4 (100 KF x 50 k landmarks) sharded 4-way, HBM-resident reduced system (N_p = 594): equals the
    single-device solve of the same window."""
    w = synthetic.make_window(n_kf=100, n_lmk=50000, length=50.0, band=6, seed=4)
    opts = capi.gn_options(4)
    be = backend_cls(device=0)
    be.set_windows([w])
    s1 = be.solve(opts)[0]
    d1 = be.get_deltas(0)
    be.close()
    out, coll = solve_sharded(backend_cls, w, opts, 4)
    # only the band of the 594 x 594 reduced system travels ("reduce only the non-zero blocks")
    assert 594 * 30 < coll.max_count < 594 * 594 // 4
    lmk = np.concatenate([o[1]["lmk"] for o in out])
    for s, d, _ in out:
        assert np.isclose(s.final_cost, s1.final_cost, rtol=1e-9)
        assert np.abs(d["pose"] - d1["pose"]).max() <= POSE_TOL
    assert lmk_err(lmk, d1["lmk"]) <= LMK_TOL


def test_rccl_collective_world_1(backend_cls, oracle_lib):
    """The built-in RCCL hook on a one-rank communicator: ncclCommInitRank + the library's own all-reduce path."""
    w = synthetic.make_window(n_kf=5, n_lmk=300, seed=42)
    opts = capi.reference_options()
    be = backend_cls(device=0)
    uid = be.rccl_unique_id()
    be.comm_init_rccl(0, 1, uid)
    info = be.comm_info()            # ncclCommCount / ncclCommUserRank of the communicator itself
    assert info["is_rccl"] and (info["nranks"], info["rank"], info["device"]) == (1, 0, 0)
    be.set_windows([w])
    s = be.solve(opts)[0]
    d = be.get_deltas(0)
    be.close()
    ref = oracle_lib.solve(w, opts)
    assert np.isclose(s.final_cost, ref["summary"].final_cost, rtol=1e-9)
    assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL


def test_collective_must_precede_set_windows(backend_cls):
    be = backend_cls(device=0)
    be.set_windows([synthetic.make_window(n_kf=4, n_lmk=100, seed=1)])
    with pytest.raises(capi.SadvioError):
        be.set_collective(0, 2, lambda *a: 0)
    be.close()


def test_marginalize_relative_refused_on_a_sharded_window(backend_cls):
    """Each rank of a sharded window only holds its landmark partition: the relative-pose information would be a partial sum."""
    w = synthetic.make_window(n_kf=5, n_lmk=300, seed=42)
    be = backend_cls(device=0)
    be.set_collective(0, 2, lambda *a: 0)
    assert be.comm_info() == {"nranks": 2, "rank": 0, "device": 0, "is_rccl": False}
    be.set_windows([sharding.shard_window(w, 0, 2)])
    with pytest.raises(capi.SadvioError):
        be.marginalize_relative(0, 0, 1)
    be.close()


def test_sharded_vio_window_with_the_sparsified_prior(backend_cls, oracle_lib):
    """A VIO window that carries a sparsified marginalisation prior (IMUPriordx on the kept frame + pose-to-landmark factors, the
    sparse branch of addMarginalizationResiduals, BundleAdjustmentCERESAnalytic.cpp:363-426) split over two and three ranks:
    the IMU prior is replicated, each pose-to-landmark factor rides its owner's elimination; equals the oracle's un-sharded
    solve, iteration by iteration in the summary, and every rank ends on identical poses."""
    from sparse_helpers import vio_sparse_priors
    from vio_helpers import make_vio_window
    w = make_vio_window(n_kf=6, n_lmk=900, seed=47)
    w.sparse_priors = vio_sparse_priors(w, w.n_kf - 2, list(range(5, 800, 13)), np.random.default_rng(5), noise=0.03)
    opts = capi.reference_options()
    ref = oracle_lib.solve(w, opts)
    rs = ref["summary"]
    for world in (2, 3):
        out, _ = solve_sharded(backend_cls, w, opts, world)
        lmk = np.concatenate([o[1]["lmk"] for o in out])
        for s, d, _ in out:
            assert (s.iterations, s.termination) == (rs.iterations, rs.termination)
            assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-9) and np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-10)
            assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL
            for q in ("dv", "dba", "dbg"):
                assert np.abs(d[q] - ref[q]).max() <= POSE_TOL
        # bit-identical on every rank: on a sharded window the pose-only factor families (priors, IMU + bias, NFR) are added to
        # the reduced system one factor at a time with plain adds instead of order-dependent LDS atomics (k_solve, `det`)
        for r in range(1, world):
            assert np.array_equal(out[r][1]["pose"], out[0][1]["pose"])
            for q in ("dv", "dba", "dbg"):
                assert np.array_equal(out[r][1][q], out[0][1][q])
        assert lmk_err(lmk, ref["lmk"]) <= LMK_TOL


def test_landmark_holding_factors_refused_on_a_sharded_window(backend_cls):
    from sparse_helpers import vo_sparse_priors
    w = synthetic.make_window(n_kf=5, n_lmk=300, seed=42)
    sh = sharding.shard_window(w, 0, 2)
    sh.sparse_priors = vo_sparse_priors(sh, [3, 4, 5], np.random.default_rng(1))
    be = backend_cls(device=0)
    be.set_collective(0, 2, lambda *a: 0)
    with pytest.raises(capi.SadvioError):
        be.set_windows([sh])
    be.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_vio_window_with_the_dense_prior(backend_cls, oracle_lib, world):
    """Round 5: the dense MarginalizationFactor on a window that spans devices. Every rank carries the prior and its
    kept landmarks (observations on rank 0 only, sharding.py), rank 0 adds J^T J / J^T r to the all-reduced system, every rank evaluates
    the prior's cost from row-block partials summed in index order — the ranks must leave every step with the same bits."""
    from test_gpu_prior import random_prior
    from vio_helpers import make_vio_window
    w = make_vio_window(n_kf=6, n_lmk=900, seed=173)
    w.dense_prior = random_prior(w, 60, w.n_kf - 2, np.random.default_rng(61), rank_deficit=2)      # N_p = 75 + 180: a dense system out of LDS
    opts = capi.reference_options()
    coll = HostAllReduce(world)
    out, shards = [None] * world, [sharding.shard_window(w, r, world) for r in range(world)]

    def run(rank):
        be = backend_cls(device=0)
        try:
            be.set_collective(rank, world, coll.fn(rank))
            be.set_windows([shards[rank]])
            s = be.solve(opts)[0]
            out[rank] = (s, be.get_deltas(0))
        except Exception as e:
            out[rank] = e
            coll.barrier.abort()
        finally:
            be.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for o in out:
        if isinstance(o, Exception):
            raise o
    ref = oracle_lib.solve(w, opts, dense_prior=w.dense_prior)
    rs = ref["summary"]
    assert rs.num_successful_steps > 0
    for s, d in out:
        assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
        assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-9)
        for k in ("pose", "dv", "dba", "dbg"):
            assert np.abs(d[k] - ref[k]).max() <= POSE_TOL, k
    for r in range(1, world):
        for k in ("pose", "dv", "dba", "dbg"):
            assert np.array_equal(out[r][1][k], out[0][1][k]), k                  # same bits on every rank
        n_own = shards[r].n_own
        assert np.array_equal(out[r][1]["lmk"][n_own:], out[0][1]["lmk"][shards[0].n_own:])   # the kept landmarks too
    lmk = sharding.gather_landmarks(w, shards, [o[1]["lmk"] for o in out])
    assert lmk_err(lmk, ref["lmk"]) <= LMK_TOL
