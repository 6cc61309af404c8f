"""The sharded-window protocol across REAL processes: two / three processes launched by
torch.distributed.run, each owning a handle on the box's one GPU, the per-LM-step all-reduce of the reduced system going
through sadvio_ba_set_collective into torch.distributed (gloo). This drives the library's rank logic exactly as an
8-GPU node would (there the hook is the built-in RCCL one, sadvio_ba_comm_init_rccl): different address spaces, no
shared Python state, a collective that really crosses process boundaries."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_window_across_processes(tmp_path, world):
    out = tmp_path / "record.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "mp_sharded_worker.py"), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rec = json.load(open(out))
    a = rec["lds"]
    assert a["iterations"] == [a["single_iterations"]] * world == [a["oracle_iterations"]] * world
    assert a["termination"] == [a["single_termination"]] * world
    assert a["pose_identical_across_ranks"]
    assert a["dpose_vs_oracle"] <= 1e-6 and a["dlmk_vs_oracle"] <= 1e-5 and a["dpose_vs_single"] <= 1e-9
    assert a["allreduce_calls"] >= 2 * a["single_iterations"] - 1       # two collectives per executed LM step
    b = rec["banded_hbm"]
    assert b["pose_identical_across_ranks"] and b["dpose_vs_single"] <= 1e-6 and b["dlmk_vs_single"] <= 1e-5 and b["dcost_rel"] <= 1e-9
    assert b["max_count"] < b["n_p"] * b["n_p"] // 2                    # only the band of the reduced system travels
