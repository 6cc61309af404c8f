"""On-disk window format (include/sadvio_io.hpp, sadvio_amd/io.py): Python round trip, C++ writer -> Python reader,
Python writer -> oracle solve identical to the in-memory window."""
import os
import subprocess

import numpy as np

from sadvio_amd import capi, io, synthetic
from vio_helpers import make_vio_window

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a: capi.FlatWindow, b: capi.FlatWindow):
    for k in ("kf_T_f_w", "kf_const", "cam_K", "cam_T_s_f", "cam_sigma", "lmk_p", "lmk_obs_ptr", "obs_kf", "obs_cam", "obs_meas",
              "kf_vel", "kf_ba", "kf_bg", "lmk_const", "kf_id", "lmk_id"):
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None), k
        if x is not None:
            assert np.array_equal(np.asarray(x), np.asarray(y)), k
    assert (a.factor_type, a.has_imu, len(a.pose_priors), len(a.imu_factors)) == (b.factor_type, b.has_imu, len(b.pose_priors), len(b.imu_factors))
    for (k0, T0, i0), (k1, T1, i1) in zip(a.pose_priors, b.pose_priors):
        assert k0 == k1 and np.array_equal(np.ravel(T0), np.ravel(T1)) and np.array_equal(np.ravel(i0), np.ravel(i1))
    for f0, f1 in zip(a.imu_factors, b.imu_factors):
        for k in f1:
            assert np.array_equal(np.ravel(f0[k]), np.ravel(f1[k])), k


def test_python_round_trip(tmp_path, oracle_lib):
    for i, w in enumerate([synthetic.make_window(n_kf=4, n_lmk=60, seed=70),
                           synthetic.make_window(n_kf=3, n_lmk=40, seed=71, factor=capi.FACTOR_ANGULAR),
                           make_vio_window(n_kf=4, n_lmk=50, seed=72)]):
        w.kf_id = np.arange(w.n_kf, dtype=np.int64) + 10; w.lmk_id = np.arange(w.n_lmk, dtype=np.int64) * 3
        if i == 0:
            w.lmk_const = (np.arange(w.n_lmk) % 7 == 0).astype(np.uint8)
        p = str(tmp_path / f"w{i}.sadvio")
        io.save_window(p, w)
        r = io.load_window(p)
        _same(w, r)
        a, b = oracle_lib.solve(w, capi.reference_options()), oracle_lib.solve(r, capi.reference_options())
        assert a["summary"].final_cost == b["summary"].final_cost and np.array_equal(a["pose"], b["pose"])


def test_cpp_writer_python_reader(tmp_path):
    exe = str(tmp_path / "test_io")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "test_io.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    path = str(tmp_path / "cpp.sadvio")
    r = subprocess.run([exe, path], capture_output=True, text=True)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout + r.stderr
    w = io.load_window(path)
    assert (w.n_kf, w.n_cam, w.n_lmk, w.n_obs, w.has_imu) == (3, 2, 4, 7, 1)
    assert np.array_equal(w.kf_id, [7, 8, 9]) and np.array_equal(w.lmk_obs_ptr, [0, 2, 2, 5, 7])
    assert np.allclose(w.obs_meas.ravel(), 10.0 * np.arange(14) + 0.25) and np.array_equal(w.lmk_const, [0, 1, 0, 0])
    assert w.pose_priors[0][0] == 2 and np.array_equal(w.pose_priors[0][2], 100.0 * np.ones(6))
    f = w.imu_factors[0]
    assert (f["kf_i"], f["kf_j"], f["dt"]) == (1, 0, 0.25) and f["delta_v"][2] == 2.45 and f["cov"][10] == 1e-4 * 11
    # Python writes the same bytes the C++ writer produced
    p2 = str(tmp_path / "py.sadvio")
    io.save_window(p2, w)
    assert open(p2, "rb").read() == open(path, "rb").read()


def test_reader_rejects_garbage(tmp_path):
    p = str(tmp_path / "bad.sadvio")
    open(p, "wb").write(b"NOTAWINDOW" * 4)
    try:
        io.load_window(p)
        assert False
    except ValueError:
        pass
