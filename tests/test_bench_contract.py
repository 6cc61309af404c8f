"""The driver reads ONE JSON line from bench.py's stdout. Libraries write to stdout as well (RCCL prints a banner through C stdio when a
communicator is created; buffered, it used to land after the record): bench.py claims fd 1 for the record alone."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stdout_carries_the_record_alone():
    code = (
        "import bench, ctypes\n"
        "emit = bench._claim_stdout()\n"
        "print('python noise')\n"
        "ctypes.CDLL(None).printf(b'buffered C noise before the record\\n')\n"
        "emit({'metric': 'x', 'value': 1.5})\n"
        "ctypes.CDLL(None).printf(b'C noise after the record\\n')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "x", "value": 1.5}
    for noise in ("python noise", "buffered C noise before the record", "C noise after the record"):
        assert noise in r.stderr


def test_bench_line_fields_of_the_committed_record():
    """profiles/r06_bench.json is a full line of the profiling box: the contract's fields, the roofline and the CPU baseline objects."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["dtype"] == "f64" and rec["vs_baseline"] is None and "workload" in rec["config"]
    assert rec["roofline"]["bound"] == "latency" and rec["batched"]["fp64"]["flops_per_lm_step_survey_8d"] == 470 * 40000 + 4100 * 8000 + 114 ** 3 // 3
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-4
    assert abs(rec["value"] - rec["config"]["solves_per_step"] * rec["config"]["iterations_per_solve"] / (rec["ms_per_step"] * 1e-3)) / rec["value"] < 1e-3


def test_gpus_n_without_n_gpus_refuses():
    """`bench.py --gpus 2` on a node without two GPUs fails with "needs 2 GPUs" - it never prints a one-rank record."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return   # a multi-GPU box: the launch itself is covered by tests/test_gpu_multiprocess.py
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr and r.stdout.strip() == ""


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE = 2" in r.stderr and r.stdout.strip() == ""
