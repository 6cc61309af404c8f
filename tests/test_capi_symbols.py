"""The C-ABI library loads on a CPU-only box and exports every symbol include/sadvio_ba.h declares.
No compute call is made here (there is no CPU fallback: compute needs a gfx950 device)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sadvio_ba.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sadvio_ba_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for needed in ("sadvio_ba_create", "sadvio_ba_destroy", "sadvio_ba_set_windows", "sadvio_ba_set_pose_priors",
                   "sadvio_ba_set_imu_factors", "sadvio_ba_set_dense_prior", "sadvio_ba_solve", "sadvio_ba_get_deltas",
                   "sadvio_ba_last_error", "sadvio_ba_device_count"):
        assert needed in syms


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build_hip()
    from sadvio_amd import capi
    lib = capi.load_library()
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/sadvio_ba.h but not exported"
    assert b"gfx950" in lib.sadvio_ba_version()


def test_no_cpu_fallback_without_device():
    """On a box without a gfx950 device, creating a backend must fail loudly (never fall back)."""
    from sadvio_amd import capi
    lib = capi.load_library()
    if lib.sadvio_ba_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    with pytest.raises(capi.SadvioError):
        capi.Backend(device=0)


def test_product_path_never_imports_the_oracle():
    """sadvio_amd/ (the product) must not import, include, link or dlopen anything under oracle/."""
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#\s*include[^\n]*oracle)|libsadvio_oracle|oracle/_build|oracle\.oracle",
                     re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sadvio_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(text), f"{f} reaches into oracle/"
