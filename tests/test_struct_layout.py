"""Both Python mirrors of include/sadvio_ba.h — the product binding (sadvio_amd/capi.py) and the oracle's own copy
(oracle/structs.py), written independently — must have the C compiler's sizeof / offsetof for every field: a field-order
slip in either mirror fails here instead of being shared silently by checker and checked."""
import ctypes as C
import os
import re
import subprocess

import pytest

from oracle import structs as O
from sadvio_amd import capi as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# C struct -> (product mirror, oracle mirror); Python-side names that differ from C ("lambda" is a keyword)
PAIRS = {
    "sadvio_flat_window": (P.FlatWindowC, O.flat_window),
    "sadvio_imu_factor": (P.ImuFactorC, O.imu_factor),
    "sadvio_pose_prior": (P.PosePriorC, O.pose_prior),
    "sadvio_line_set": (P.LineSetC, O.line_set),
    "sadvio_sparse_prior": (P.SparsePriorC, O.sparse_prior),
    "sadvio_solve_options": (P.SolveOptions, O.solve_options),
    "sadvio_solve_summary": (P.SolveSummary, O.solve_summary),
    "sadvio_viinit_problem": (P.ViInitProblemC, O.viinit_problem),
    "sadvio_viinit_result": (P.ViInitResultC, O.viinit_result),
    "sadvio_marg_request": (P.MargRequestC, None),
    "sadvio_marg_result": (P.MargResultC, None),
    "sadvio_prior_info": (P.PriorInfoC, None),
    "sadvio_ba_config": (P.Config, None),
}
RENAME = {"lambda": "lambda_"}


def c_layout(tmp_path):
    hdr = open(os.path.join(ROOT, "include", "sadvio_ba.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sadvio_ba.h"', "int main(void) {"]
    for name in PAIRS:
        m = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + ";", hdr, flags=re.S)
        assert m, name
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            decl = re.sub(r"\[[^\]]*\]", "", decl)
            parts = decl.split(",")
            first = parts[0].split()[-1].lstrip("*")
            fields.append(first)
            for extra in parts[1:]:
                fields.append(extra.strip().lstrip("*"))
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in fields:
            lines.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    lay = {}
    for ln in out.splitlines():
        k, v = ln.split()
        lay[k] = int(v)
    return lay


def test_python_mirrors_match_the_c_header(tmp_path):
    lay = c_layout(tmp_path)
    checked = 0
    for name, mirrors in PAIRS.items():
        for mirror in mirrors:
            if mirror is None:
                continue
            assert C.sizeof(mirror) == lay[name], (name, mirror.__module__)
            c_fields = [k.split(".")[1] for k in lay if k.startswith(name + ".")]
            py_fields = [f[0] for f in mirror._fields_]
            assert [RENAME.get(f, f) for f in c_fields] == py_fields, (name, mirror.__module__)
            for f in c_fields:
                assert getattr(mirror, RENAME.get(f, f)).offset == lay[f"{name}.{f}"], (name, f, mirror.__module__)
                checked += 1
    assert checked > 200


def test_oracle_binding_does_not_import_the_product_binding():
    for fn in ("oracle.py", "structs.py", "twin.py"):
        src = open(os.path.join(ROOT, "oracle", fn)).read()
        assert not re.search(r"^\s*(from|import)\s+sadvio_amd", src, flags=re.M), fn
