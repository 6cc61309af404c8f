#!/usr/bin/env python
"""bench.py — BA iterations/s + ms/solve on BASELINE.json config 2 (synthetic 20 KF x 8 000 landmarks x
40 000 reprojection factors, GN 10 iters) through the C ABI of the HIP backend.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N)

A "step" is --solves-per-step (default 400) back-to-back complete solves (each: 10 LM step attempts, every early exit
disabled, one hipGraph launch) of the windows resident on this GPU, so that the driver's 20-step run times ~5 s of GPU
work instead of 12 ms (and a sampling `rocm-smi` sees a busy GPU); the CPU-baseline legs run BEFORE the GPU legs; `value` (BA iterations/s) does not depend on that grouping, `ms_per_step` is per step and
`ms_per_solve` per solve. The windows are uploaded to HBM before the timed region (`upload_inclusive` reports the rate with
set_windows + get_deltas inside, the way the reference's own timer brackets problem construction, slamBiMono.cpp:273-275).
N > 1 shards independent windows (one set per GPU, weak scaling, no data-path collective — BASELINE.json north_star: "independent sub-windows / keyframe
batches shard across the 8 GPUs"); torch.distributed (backend "nccl" = RCCL) is used only for the barriers and
the max-over-ranks of the elapsed time.

One JSON line is printed by rank 0. Besides the contract's fields it carries
  roofline      the dominant kernel's achieved algorithmic GB/s vs the 8 TB/s HBM peak; kernel durations are
                measured live with hipEvents on the backend's stream in a separate (untimed) profiled pass
  cpu_baseline  the CPU oracle (a port of the reference's algorithm, the reference itself cannot be built
                here) timed on this box's host cores on the same window
  batched       the same metric with 64 independent windows per launch (the bandwidth-bound regime)
  vio_window    GN-10 solves of a config-3 shaped VIO window (no prior / sparsified prior), resident in HBM
  marginalize   sadvio_ba_marginalize on a config-3 shaped window (12-KF VIO, 300 kept landmarks) with the oracle's CPU time beside it
  backend_step  one back-end key-frame step as the reference brackets it (slamBiMonoVIO.cpp:569-594): marginalize -> sparsify ->
                set_windows with the prior -> solve -> get_deltas, the prior resident on the device
  sharded_window (N > 1 only) ONE config-4 window landmark-sharded over the ranks: RCCL all-reduce of the reduced system per LM step
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured achievable)
FP64_PEAK_TFLOPS = 78.6  # half the guide's 157.3 TFLOP/s fp32 vector / matrix rate; v_mfma_f64_16x16x4 and the fp64 VALU share one pipe at 128 flop / clk / CU (profiles/r03_mfma64_rate_probe.txt)
GN_ITERS = 10
# committed rocprofv3 PMC summaries of this same command (scripts/prof_bench.sh / prof_batched.sh), named explicitly: the
# newest round's files, not whatever sorts last
PROFILE_SUMMARY = "profiles/r06_summary.json"
PROFILE_SUMMARY_BATCHED = "profiles/r06_batched64_summary.json"
PROFILE_FALLBACK = {"profiles/r06_summary.json": "profiles/r05_summary.json",
                    "profiles/r06_batched64_summary.json": "profiles/r05_batched64_summary.json"}   # (the throughput kernels did not change in round 6)


def profile_summary(rel):
    """(parsed summary, path relative to the repo) of a committed profile; the previous round's if this round's is not there yet."""
    for cand in (rel, PROFILE_FALLBACK.get(rel)):
        if cand and os.path.exists(os.path.join(ROOT, cand)):
            return json.load(open(os.path.join(ROOT, cand))), cand
    return None, None


def algorithmic_bytes(n_obs, n_lmk, n_p):
    """SURVEY.md §8(d) / BASELINE.md §4: B_iter = 120 N_obs + 96 N_lmk + 16 (N_p^2 + N_p), split per kernel:
    k_build = linearise/eliminate pass, k_backsub = back-substitution + candidate-cost passes (fused)."""
    red = 8 * (n_p * n_p + n_p)
    return {"k_build": 40 * n_obs + 24 * n_lmk + red, "k_solve": red, "k_backsub": 80 * n_obs + 72 * n_lmk}


def _claim_stdout():
    """The contract is ONE JSON line on stdout. Libraries write there too (RCCL prints a five-line banner through C stdio when a
    communicator is created - buffered, it lands AFTER the JSON line when stdout is a file or a pipe): everything this process writes
    to fd 1 from now on goes to stderr, the returned function writes the record to the real stdout."""
    import ctypes
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    libc = ctypes.CDLL(None)

    def emit(rec):
        sys.stdout.flush()
        libc.fflush(None)
        os.write(real, (json.dumps(rec) + "\n").encode())
    return emit


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): re-exec through torch.distributed.run,
    one rank per GPU on 127.0.0.1, so that the plain command line yields the N-rank record (n_gpus = N, `sharded_window` through an
    N-rank RCCL communicator) instead of a one-rank one. Fails - never degrades to fewer ranks - when the node has fewer GPUs."""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs on this node (found {have}); there is no CPU fallback and no run on fewer ranks")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes: the host driver only supports dmabuf IPC
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--solves-per-step", type=int, default=400, help="complete GN-10 solves per timed step")
    ap.add_argument("--windows", type=int, default=1, help="independent config-2 windows per GPU in the timed run")
    ap.add_argument("--batch", type=int, default=64, help="windows per GPU of the extra batched measurement (0 = skip)")
    ap.add_argument("--batch-large", type=int, default=256, help="windows of the second batched measurement (<= --batch: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-window", action="store_true",
                    help="ONLY the sharded measurement: ONE config-4 window (100 KF x 50k landmarks) landmark-sharded "
                         "over the ranks, reduced system all-reduced over RCCL per LM step (strong scaling). With N > 1 the "
                         "default run reports it as the `sharded_window` object next to the independent-window line")
    ap.add_argument("--no-marginalize", action="store_true")
    ap.add_argument("--no-sharded", action="store_true", help="N = 1: skip the one-rank run of the sharded config-4 window (RCCL communicator of size 1)")
    ap.add_argument("--no-vio", action="store_true", help="skip the config-3 shaped VIO window leg (the profiled runs of scripts/prof_bench.sh: one k_solve variant per trace)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args, sys.argv[1:])      # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks; pass the same number to both")
    emit_record = _claim_stdout()
    import numpy as np
    import torch

    import __graft_entry__ as ge
    if local_rank == 0:
        ge.build_hip()   # no-op when the in-tree .so is newer than its sources; never from several ranks at once
    from sadvio_amd import capi, synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()   # rank 0 of the node has finished (or skipped) the build

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_solves(be, opts, steps, warmup, per_step=1):
        for _ in range(warmup * per_step):
            be.solve(opts)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps * per_step):
            be.solve(opts)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    if args.shard_window:
        rec = bench_sharded_window(args, rank, local_rank, world, dist, barrier, timed_solves)
        if rank == 0:
            emit_record(rec)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    opts = capi.gn_options(GN_ITERS)
    # every rank owns its own windows (different seeds): weak scaling over independent sub-windows
    base_seed = 20250404 + 1000 * rank
    wins = [synthetic.make_window(seed=base_seed + i) for i in range(args.windows)]
    # CPU legs first (rank 0 at N = 1 only): the GPU legs then run back to back at the end of the command
    cpu = marg_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(wins[0], opts)
        if not args.no_marginalize:
            marg_cpu = marginalize_cpu_leg()
    be = capi.Backend(device=local_rank, use_graph=True)  # one hipGraph launch per solve
    be.set_windows(wins)
    sps = max(1, args.solves_per_step)
    dt = timed_solves(be, opts, args.steps, args.warmup, sps)
    sums = be.solve(opts)
    gpu_sol = be.get_deltas(0)
    gpu_ids = be.get_ids(0)
    # upload-inclusive rate: set_windows (validation, tiling, one pinned staging copy) + solve + read-back per solve
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_up = 50
    prep = be.prepare(wins)   # the C structs a C++ caller holds anyway: Python's per-field marshalling stays outside the timed region
    for _ in range(n_up):
        be.set_prepared(prep)
        be.solve(opts)
        be.get_deltas(0)
    dt_up = (time.perf_counter() - t0) / n_up
    be.close()
    # the same with two handles driven from two host threads (the reference runs a front-end and a back-end optimizer instance
    # concurrently, slamParameters.cpp:273-275; a handle owns its stream and window state, so the two act as a double buffer):
    # one handle's host-side layout build runs under the other's solve
    dt_up2 = None
    if world == 1:
        import threading
        bes = [capi.Backend(device=local_rank, use_graph=True) for _ in range(2)]
        preps = [b.prepare(wins) for b in bes]

        def loop(b, pr, n):
            for _ in range(n):
                b.set_prepared(pr); b.solve(opts); b.get_deltas(0)
        for b, pr in zip(bes, preps):
            loop(b, pr, 3)
        ths = [threading.Thread(target=loop, args=(b, pr, n_up)) for b, pr in zip(bes, preps)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt_up2 = (time.perf_counter() - t0) / (2 * n_up)
        for b in bes:
            b.close()
    iters_per_solve = sum(s.iterations for s in sums)
    total_iters = iters_per_solve * args.steps * sps * world
    value = total_iters / dt
    ms_per_step = 1e3 * dt / args.steps
    ms_per_solve = ms_per_step / sps
    # N > 1: the collective path as well (same processes, same communicator): config 4 sharded over the ranks
    # the collective path (config 4 sharded over the ranks; RCCL all-reduce of the reduced system per LM step). At N = 1 it runs
    # through the same sadvio_ba_comm_init_rccl / ncclAllReduce calls with a one-rank communicator, so that every driver record
    # contains the path; the scaling curve only exists for N > 1
    sharded = bench_sharded_window(args, rank, local_rank, world, dist, barrier, timed_solves) if (world > 1 or not args.no_sharded) else None
    sharded5 = bench_sharded_window(args, rank, local_rank, world, dist, barrier, timed_solves, config=5) if world > 1 else None

    out = None
    if rank == 0:
        w0 = wins[0]
        n_p = 6 * int((w0.kf_const == 0).sum())
        # --- live per-kernel durations (hipEvents on the backend's stream), untimed profiled pass ---
        bp = capi.Backend(device=local_rank, profile_kernels=True)
        bp.set_windows(wins)
        for _ in range(3):
            bp.solve(opts)
        bp.set_windows(wins)  # resets the per-class accumulators
        for _ in range(10):
            bp.solve(opts)
        kt = bp.kernel_times()
        bp.close()
        ab = algorithmic_bytes(w0.n_obs * len(wins), w0.n_lmk * len(wins), n_p)
        ab["k_solve"] = len(wins) * 8 * (n_p * n_p + n_p)
        ab["k_build"] = len(wins) * (40 * w0.n_obs + 24 * w0.n_lmk + 8 * (n_p * n_p + n_p))
        dom = max((k for k in kt if k in ab), key=lambda k: kt[k]["avg_us"] * kt[k]["launches"])
        achieved = ab[dom] / (kt[dom]["avg_us"] * 1e-6) / 1e9
        # HBM traffic per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
        # command (scripts/prof_bench.sh -> profiles/*_summary.json); null if no summary is committed
        traffic, traffic_src = None, None
        prof, prof_path = profile_summary(PROFILE_SUMMARY)
        if prof is not None and len(wins) == 1:
            kk = prof.get("kernels", {}).get(dom, {})
            if "hbm_traffic_bytes_per_launch" in kk:
                traffic, traffic_src = round(kk["hbm_traffic_bytes_per_launch"]), prof_path
        roofline = {"bound": "latency", "bound_roof_of_the_numbers": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": ab[dom], "avg_kernel_us": round(kt[dom]["avg_us"], 3),
                    "kernels_us": {k: round(v["avg_us"], 3) for k, v in kt.items()},
                    "kernels_us_note": "hipEvent brackets on the backend's stream in a separate profiled pass: each includes ~2-3 us of "
                                       "event overhead, their sum exceeds the driver-timed step; rocprofv3 averages of the same command beside them",
                    "kernels_us_rocprof": ({k: round(v["avg_us"], 3) for k, v in prof.get("kernels", {}).items() if "avg_us" in v} if prof else None),
                    # whole LM step from the TIMED region (not from the event-inflated kernel sum): B_iter x iterations / elapsed
                    "iteration_achieved_GBps": round(sum(ab.values()) * iters_per_solve / (ms_per_solve * 1e-3) / 1e9, 2),
                    "iteration_frac_of_hbm_peak": round(sum(ab.values()) * iters_per_solve / (ms_per_solve * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        # the two STREAMING kernels (HBM-bound by design; k_solve is a one-workgroup-per-window factorisation, latency-bound): algorithmic
        # GB/s from the live hipEvent durations, HBM traffic / algorithmic bytes from the committed PMC passes
        stream_k = {}
        for k in ("k_build", "k_backsub"):
            if k in kt:
                g = ab[k] / (kt[k]["avg_us"] * 1e-6) / 1e9
                rec = {"algorithmic_bytes_per_launch": ab[k], "avg_kernel_us": round(kt[k]["avg_us"], 3), "achieved_GBps": round(g, 1),
                       "frac_of_hbm_peak": round(g / HBM_PEAK_GBS, 5)}
                kk = (prof or {}).get("kernels", {}).get(k, {}) if len(wins) == 1 else {}
                if "hbm_traffic_bytes_per_launch" in kk:
                    rec["traffic"] = round(kk["hbm_traffic_bytes_per_launch"])
                    rec["traffic_over_algorithmic"] = round(kk["hbm_traffic_bytes_per_launch"] / ab[k], 3)
                if "avg_us" in kk:
                    rec["avg_kernel_us_rocprof"] = round(kk["avg_us"], 3)
                stream_k[k] = rec
        roofline["streaming_kernels"] = stream_k
        roofline["bound_note"] = ("`kernel` is the launch with the largest share of the step; k_solve is ONE workgroup per window (an in-LDS Cholesky of the "
                                  "reduced system): neither HBM- nor MFMA-bound but bound by the instruction issue of its pivot wave and the LDS traffic of its "
                                  "trailing updates (`bound`: latency); achieved / peak / frac are its algorithmic bytes against the HBM roof because the contract "
                                  "asks for the dominant kernel's; the whole LM step against the same roof is iteration_frac_of_hbm_peak, the HBM-bound kernels "
                                  "of the step are under `streaming_kernels`, the bandwidth-bound regime under `batched`")
        # --- parity of the TIMED configuration against the CPU solve of the same window (BASELINE.json metric: "pose-RMSE vs Ceres";
        # the reference cannot be built, the CPU solve is the oracle's) ---
        parity = None
        if cpu and cpu.get("_ref"):
            ref = cpu.pop("_ref")
            ang, dis = [], []
            for i in range(w0.n_kf):
                a = synthetic.apply_pose_delta(w0.kf_T_f_w[i], gpu_sol["pose"][i])   # T_f_w <- T_f_w (exp w, t), AOptimizer.cpp:329-332
                b = synthetic.apply_pose_delta(w0.kf_T_f_w[i], ref["pose"][i])
                Ra, Rb = np.asarray(a[:9]).reshape(3, 3), np.asarray(b[:9]).reshape(3, 3)
                D = Ra @ Rb.T     # |log(Ra Rb^T)| from the skew part (arccos of the trace resolves nothing below 1.5e-8 rad)
                v = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
                ang.append(float(np.arctan2(np.linalg.norm(v), 0.5 * (np.trace(D) - 1.0))))
                dis.append(float(np.linalg.norm(np.asarray(a[9:]) - np.asarray(b[9:]))))
            ang, dis = np.asarray(ang), np.asarray(dis)
            parity = {"against": "CPU oracle solve of the identical window (port of the reference's algorithm; Ceres itself is not on the box)",
                      "max_pose_delta": float(max(ang.max(), dis.max())), "max_rotation_rad": float(ang.max()), "max_translation_m": float(dis.max()),
                      "pose_rmse_rotation_rad": float(np.sqrt((ang ** 2).mean())), "pose_rmse_translation_m": float(np.sqrt((dis ** 2).mean())),
                      "max_landmark_delta_m": float(np.abs(gpu_sol["lmk"] - ref["lmk"]).max()),
                      "final_cost_rel_diff": float(abs(sums[0].final_cost - ref["final_cost"]) / abs(ref["final_cost"])),
                      "iterations_gpu_cpu": [int(sums[0].iterations), int(ref["iterations"])],
                      "lmk_ids_bitexact": bool(np.array_equal(gpu_ids[1], w0.lmk_id) and np.array_equal(gpu_ids[0], w0.kf_id)),
                      "tolerance": "north_star: pose delta <= 1e-6, landmark indices bit-exact"}
        elif cpu:
            cpu.pop("_ref", None)
        # --- batched throughput (independent windows in one submission) ---
        batched = None
        if args.batch > 0 and world == 1:
            per_iter_bytes = 120 * w0.n_obs + 96 * w0.n_lmk + 16 * (n_p * n_p + n_p)
            # The second roof of this regime, by SURVEY.md 8(d)'s flop count: F_iter = 470 N_obs + 4 100 N_lmk + N_p^3 / 3 (projection + both
            # Jacobians + the J^T J blocks once per observation; elimination, Schur products and back-substitution per landmark of 5 views)
            # = 52.1 MFLOP for config 2. (The three-pass kernels EXECUTE ~1 540 flops per observation - the factor is evaluated once per pass:
            # recomputation is not useful work and is not counted; rounds 4-5 reported that figure.) At 5.80 MB per iteration that is
            # 9.0 flop / B against a machine balance of 78.6 TFLOP/s / 8 TB/s = 9.8: the path sits at the ridge, both fractions are reported.
            per_iter_flops = 470 * w0.n_obs + 4100 * w0.n_lmk + n_p ** 3 // 3

            def batch_leg(nw):
                bw = [synthetic.make_window(seed=base_seed + 100 + i) for i in range(min(nw, 8))]
                bw = [bw[i % len(bw)] for i in range(nw)]
                bb = capi.Backend(device=local_rank, use_graph=False)   # ~40 launches of >= 40 us each: the graph buys nothing here (measured: -2 %)
                bb.set_windows(bw)
                for _ in range(3):
                    bb.solve(opts)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 20 if nw <= 64 else 8
                for _ in range(reps):
                    bs = bb.solve(opts)
                torch.cuda.synchronize()
                bdt = time.perf_counter() - t0
                bb.close()
                biters = sum(s.iterations for s in bs) * reps
                return {"windows": nw, "value": round(biters / bdt, 1), "unit": "BA iterations/s",
                        "ms_per_solve_batch": round(1e3 * bdt / reps, 3),
                        "algorithmic_GBps": round(biters * per_iter_bytes / bdt / 1e9, 1),
                        "frac_of_hbm_peak": round(biters * per_iter_bytes / bdt / 1e9 / HBM_PEAK_GBS, 4),
                        "fp64": {"flops_per_lm_step_survey_8d": per_iter_flops, "executed_flops_per_observation_three_passes": 1540, "achieved_TFLOPs": round(biters * per_iter_flops / bdt / 1e12, 2),
                                 "peak_TFLOPs": FP64_PEAK_TFLOPS, "frac_of_fp64_peak": round(biters * per_iter_flops / bdt / 1e12 / FP64_PEAK_TFLOPS, 4),
                                 "flop_per_byte": round(per_iter_flops / per_iter_bytes, 1), "machine_balance_flop_per_byte": round(FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, 1)}}

            # >= 65 536 landmarks in a submission: the throughput kernels (sadvio_amd/csrc/lm_kernels.h) take over by themselves
            batched = batch_leg(args.batch)
            # measured HBM traffic of one LM step of the 64-window batch (PMC passes of scripts/prof_batched.sh, committed under
            # profiles/): the throughput kernels' FETCH + WRITE bytes per launch, one launch of each per step
            try:
                bprof, prof = profile_summary(PROFILE_SUMMARY_BATCHED)
                kern = bprof["kernels"]
                # one launch of each per LM step; the opening pass (k_lm_pass_init) once per solve
                per_step = sum(kern[k]["hbm_traffic_bytes_per_launch"] for k in ("k_build_obs", "k_solve", "k_lm_pass", "k_decide") if k in kern)
                per_step += kern.get("k_lm_pass_init", {}).get("hbm_traffic_bytes_per_launch", 0.0) / GN_ITERS
                if args.batch == 64 and per_step > 0:
                    step_s = 1e-3 * batched["ms_per_solve_batch"] / GN_ITERS
                    batched["hbm_traffic"] = {"bytes_per_lm_step": int(per_step), "GBps": round(per_step / step_s / 1e9, 1),
                                              "frac_of_hbm_peak": round(per_step / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                              "traffic_over_algorithmic": round(per_step / (64 * per_iter_bytes), 3),
                                              "kernels_us_rocprof": {k: round(v["avg_us"], 1) for k, v in kern.items() if "avg_us" in v},
                                              "source": prof}
            except Exception:
                pass
            if args.batch_large > args.batch:
                batched["larger"] = batch_leg(args.batch_large)   # the fixed ~40 us of the per-window reduced solve spread over more windows
        marg = bstep = None
        if not args.no_marginalize and world == 1:
            marg = marginalize_leg(local_rank, marg_cpu)
            bstep = backend_step_leg(local_rank)
        vio = vio_window_leg(local_rank, opts) if (world == 1 and not args.no_vio) else None
        out = {
            "metric": "BA iterations/sec (ms/solve in ms_per_solve), 20-KF/8k-landmark window",
            "value": round(value, 1), "unit": "BA iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "ms_per_solve": round(ms_per_solve, 5),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic 20 KF x 8000 landmarks x 40000 reprojection factors (pixel), "
                                   f"GN {GN_ITERS} iters (LM step attempts, early exits disabled)",
                       "solves_per_step": sps,
                       "windows_per_gpu": args.windows, "iterations_per_solve": iters_per_solve // max(1, args.windows),
                       "parallelism": f"independent windows x{world}" if world > 1 else "single window",
                       "n_kf": w0.n_kf, "n_lmk": w0.n_lmk, "n_obs": w0.n_obs, "reduced_dim": n_p},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "batched": batched,
            "upload_inclusive": {"value": round(iters_per_solve / dt_up, 1), "unit": "BA iterations/s",
                                 "ms_per_solve": round(1e3 * dt_up, 4),
                                 "what": "set_windows (host flatten -> HBM) + solve + get_deltas per solve, rank 0",
                                 "two_handles": (None if dt_up2 is None else
                                                 {"value": round(iters_per_solve / dt_up2, 1), "ms_per_solve": round(1e3 * dt_up2, 4),
                                                  "what": "two handles on two host threads, each set_windows + solve + get_deltas in turn: one handle's layout build runs under the other's solve"})},
            "marginalize": marg, "backend_step": bstep, "vio_window": vio, "sharded_window": sharded, "sharded_window_c5": sharded5,
        }
        if cpu and batched and cpu.get("batched"):
            batched["speedup_vs_cpu_batched"] = round(batched["value"] / cpu["batched"]["value"], 1)
            batched["speedup_note"] = "throughput regime against throughput regime: 64 windows per GPU submission vs independent windows on all usable host cores"
        if cpu:
            out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)                              # at the reference's 4 threads
            out["speedup_vs_cpu_at_reference_threads"] = round(value / cpu["at_reference_threads"]["value"], 1)
            out["speedup_vs_cpu_best_thread_count"] = round(value / cpu["best_thread_count"]["value"], 1)
        emit_record(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_leg(w0, opts):
    """The CPU side, timed on this box's host cores BEFORE any GPU leg: the C oracle (a port of the reference's algorithm —
    explicit Schur complement + dense Cholesky; OpenMP over landmarks with per-thread copies of the reduced system) at several
    thread counts, a probe for Ceres / Eigen / CHOLMOD on the box, and an emulation of the reference's own linear-solver choice."""
    from oracle import oracle
    oracle.build()
    ncores = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0))      # the cores this process may actually run on
    res = {}
    counts = sorted({1, 4, min(usable, 16), min(usable, 32), min(usable, 64)})   # 4 = the reference's Ceres num_threads (AOptimizer.cpp:323)
    for thr in counts:
        oracle.solve(w0, opts, n_threads=thr)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 3.0:
            r = oracle.solve(w0, opts, n_threads=thr)
            n += r["summary"].iterations
        res[thr] = n / (time.perf_counter() - t0)
    best = max(res, key=res.get)
    ref = oracle.solve(w0, opts, n_threads=1)    # the CPU solution of the timed window: `parity` compares the GPU solve with it
    # `value` = the reference's own thread setting (options.num_threads = 4, AOptimizer.cpp:323): what its Ceres path would be given; the
    # port's best thread count on this box is reported beside it (quote the 4-thread figure first)
    return {"value": round(res[4], 2), "unit": "BA iterations/s", "cores": 4, "kind": "port",
            "best_thread_count": {"threads": best, "value": round(res[best], 2)},
            "_ref": {"pose": ref["pose"], "lmk": ref["lmk"], "final_cost": ref["summary"].final_cost, "iterations": ref["summary"].iterations},
            "batched": cpu_batched_leg(oracle, w0, opts, usable),
            "at_reference_threads": {"threads": 4, "value": round(res[4], 2), "why": "the reference's own setting: options.num_threads = 4 (AOptimizer.cpp:323)"},
            "sample": f"config-2 window, GN-{GN_ITERS} solves repeated for ~3 s per thread count",
            "threads_it_per_s": {str(k): round(v, 1) for k, v in res.items()},
            "host_cores": ncores, "usable_cores": usable,
            "ceres_on_box": ceres_probe(),
            "sparse_normal_cholesky_emulation": sparse_normal_baseline(oracle, w0, opts),
            "note": "the reference (Ceres 2.2 / Eigen / SuiteSparse) cannot be built here or on the GPU box; "
                    "`value` = the C oracle (explicit Schur complement + dense Cholesky, OpenMP over landmarks with per-thread "
                    "reduced-system accumulators) at the reference's 4 threads, `best_thread_count` = its best point on this box; sparse_normal_cholesky_emulation = the reference's own linear-solver choice "
                    "(un-reduced J^T J + D, sparse direct factorisation) with SciPy's SuperLU standing in for CHOLMOD"}


def cpu_batched_leg(oracle, w0, opts, usable, budget_s=4.0):
    """The CPU counterpart of the GPU's `batched` regime (VERDICT r04 missing #4): independent config-2 windows solved concurrently on
    ALL usable host cores — usable / t workers (host threads; the problem is marshalled once per worker and the C solver called in a
    loop with the GIL released), each solving whole windows with t OpenMP threads — for t = 1 and t = 4 (the reference's
    num_threads, AOptimizer.cpp:323). The windows are the same one (the solves are independent and deterministic);
    throughput = completed LM iterations / wall time."""
    import ctypes as C
    import threading
    import numpy as np
    from oracle import structs as S
    copts = S.options_from(opts)
    lib = oracle.lib()
    out = {}
    for t in (1, 4):
        workers = max(1, usable // t)
        probs = [oracle.make_problem(w0, None, t) for _ in range(workers)]
        done = [0] * workers
        stop = [0.0]

        def work(k):
            P, _keep = probs[k]
            pose = np.zeros((w0.n_kf, 6)); lmk = np.zeros((w0.n_lmk, 3)); z = [np.zeros((w0.n_kf, 3)) for _ in range(3)]
            log = np.zeros((64, 8)); sm = oracle.SolveSummary()
            while time.perf_counter() < stop[0]:
                lib.oracle_solve(C.byref(P), C.byref(copts), C.byref(sm), oracle._p(pose), oracle._p(lmk), oracle._p(z[0]), oracle._p(z[1]),
                                 oracle._p(z[2]), oracle._p(log), 64)
                done[k] += sm.iterations
        ths = [threading.Thread(target=work, args=(k,)) for k in range(workers)]
        t0 = time.perf_counter()
        stop[0] = t0 + budget_s
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        out[str(t)] = {"workers": workers, "threads_per_window": t, "value": round(sum(done) / (time.perf_counter() - t0), 1)}
    best = max(out.values(), key=lambda r: r["value"])
    return {"value": best["value"], "unit": "BA iterations/s", "cores": best["workers"] * best["threads_per_window"],
            "windows_in_flight": best["workers"], "by_threads_per_window": out,
            "sample": f"independent config-2 windows in flight on all usable cores ({usable}), GN-{GN_ITERS} solves for ~{budget_s:.0f} s per setting"}


def bench_sharded_window(args, rank, local_rank, world, dist, barrier, timed_solves, config=4):
    """BASELINE.json config 4: one 100-KF / 50k-landmark window spanning the GPUs of the node — or config 5
    (500 KF / 200k landmarks). Returns the record (rank 0) — printed as its own line by --shard-window, nested as
    `sharded_window` (config 4) / `sharded_window_c5` otherwise."""
    from sadvio_amd import capi, sharding, synthetic
    opts = capi.gn_options(GN_ITERS)
    if config == 5:
        w = synthetic.make_window(n_kf=500, n_lmk=200000, length=250.0, band=6, seed=5)
    else:
        w = synthetic.make_window(n_kf=100, n_lmk=50000, length=50.0, band=6, seed=4)  # same seed on every rank
    be = capi.Backend(device=local_rank)
    uid = [be.rccl_unique_id() if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(uid, src=0)
    be.comm_init_rccl(rank, world, uid[0])
    comm = be.comm_info()   # what the RCCL communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)
    be.set_windows([sharding.shard_window(w, rank, world)])
    steps = max(5, args.steps)
    dt = timed_solves(be, opts, steps, 2)
    s = be.solve(opts)[0]
    be.close()
    if rank != 0:
        return None
    n_p = 6 * int((w.kf_const == 0).sum())
    return {
        "metric": f"BA iterations/sec, ONE {w.n_kf}-KF/{w.n_lmk // 1000}k-landmark window sharded over the GPUs",
        "value": round(s.iterations * steps / dt, 1), "unit": "BA iterations/s", "n_gpus": world,
        "steps": steps, "warmup": 2, "ms_per_step": round(1e3 * dt / steps, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic {w.n_kf} KF x {w.n_lmk} landmarks x {w.n_obs} reprojection factors, GN 10 iters, "
                               "landmark-sharded, RCCL all-reduce of the reduced system per LM step",
                   "parallelism": f"window sharded x{world}", "reduced_dim": n_p, "rccl_ranks": comm["nranks"], "rccl_rank0_device": comm["device"],
                   "rccl_communicator": comm["is_rccl"],
                   "collectives_per_lm_step": 2,
                   # only the band of the reduced system travels (k_band_pack): N_p x bw with bw = 60 for this
                   # window's 10-key-frame co-visibility, + gradient / diagonal vectors + the per-rank partials
                   "allreduce_bytes_per_step": 8 * (n_p * 60 + 3 * n_p + 4 * world) + 32 * world,
                   "allreduce_bytes_per_step_full_matrix": 8 * (n_p * n_p + 3 * n_p + 4 * world) + 32 * world},
        "final_cost": s.final_cost}


def ceres_probe():
    """Is a Ceres / Eigen installation present on this box (the B1 'Ceres CPU' line of BASELINE.md §3 needs one)?"""
    import ctypes.util
    import glob
    libs = {n: ctypes.util.find_library(n) for n in ("ceres", "cholmod")}
    eigen = bool(glob.glob("/usr/include/eigen3/Eigen/Core") or glob.glob("/usr/local/include/eigen3/Eigen/Core"))
    return {"libceres": libs["ceres"], "libcholmod": libs["cholmod"], "eigen_headers": eigen,
            "available": bool(libs["ceres"] and eigen)}


def sparse_normal_baseline(oracle, w, opts, budget_s=5.0):
    """LM iterations/s of the reference's linear-solver CHOICE on the CPU: Jacobians from the oracle's C evaluation, then the
    UN-REDUCED normal equations (J^T J + D) y = J^T r factorised by a sparse direct solver — what
    ceres::SPARSE_NORMAL_CHOLESKY does with CHOLMOD (AOptimizer.cpp:316) — here SciPy's SuperLU in symmetric mode with the
    fill-free elimination order CHOLMOD's AMD finds on bundle-adjustment systems (landmark blocks first, poses last).
    One thread; the split shows where the time goes."""
    try:
        import numpy as np
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
    except Exception as e:  # pragma: no cover
        return {"error": str(e)}
    kc = np.asarray(w.kf_const).astype(bool)
    n_pose = 6 * int((~kc).sum())
    n = n_pose + 3 * w.n_lmk
    kf_col = np.cumsum(~kc) * 6 - 6 + 3 * w.n_lmk      # poses last
    kf_col[kc] = -1
    lmk_of = np.repeat(np.arange(w.n_lmk), np.diff(w.lmk_obs_ptr))
    free = kf_col[w.obs_kf] >= 0
    rows = np.arange(2 * w.n_obs).reshape(-1, 2)
    ri = np.concatenate([np.repeat(rows[free], 6, axis=1).ravel(), np.repeat(rows, 3, axis=1).ravel()])
    ci = np.concatenate([(kf_col[w.obs_kf][free, None, None] + np.arange(6)[None, None, :] + np.zeros((1, 2, 1), int)).ravel(),
                         (3 * lmk_of[:, None, None] + np.arange(3)[None, None, :] + np.zeros((1, 2, 1), int)).ravel()])
    t_all, n_it, split = time.perf_counter(), 0, np.zeros(3)
    while time.perf_counter() - t_all < budget_s:
        t0 = time.perf_counter()
        r, Jp, Jl, _ = oracle.linearize(w)
        t1 = time.perf_counter()
        J = sp.csr_matrix((np.concatenate([Jp[free].ravel(), Jl.ravel()]), (ri, ci)), shape=(2 * w.n_obs, n))
        H = (J.T @ J).tocsc()
        H = H + sp.diags(np.clip(H.diagonal(), 1e-6, 1e32) / 1e4)
        t2 = time.perf_counter()
        y = spl.splu(H, permc_spec="NATURAL", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)).solve(J.T @ r.ravel())
        t3 = time.perf_counter()
        assert np.isfinite(y).all()
        split += [t1 - t0, t2 - t1, t3 - t2]
        n_it += 1
    dt = time.perf_counter() - t_all
    return {"value": round(n_it / dt, 2), "unit": "BA iterations/s", "cores": 1,
            "ms_linearize_assemble_factorize": [round(float(1e3 * x / n_it), 1) for x in split],
            "sample": f"{n_it} linearise + assemble + factorise + solve passes on the config-2 window, {n} unknowns"}


def vio_window_leg(device, opts):
    """BASELINE config 3's shape (12-KF VIO window, 7 200 landmarks, 11 IMU + bias factor pairs; synthetic, EuRoC is not on the box):
    GN-10 solves resident in HBM, without a prior and with the sparsified prior (IMUPriordx + 300 pose-to-landmark factors)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sadvio_amd import capi
    from vio_helpers import make_vio_window
    from sparse_helpers import vio_sparse_priors
    w = make_vio_window(n_kf=12, n_lmk=7200, seed=6)
    out = {"window": "12 KF x 7200 landmarks x 36000 factors + 11 IMU / bias factor pairs, 15 states per key-frame", "unit": "BA iterations/s"}
    for name, sp in (("no_prior", []), ("sparsified_prior", vio_sparse_priors(w, w.n_kf - 2, list(range(0, 600, 2)), np.random.default_rng(4), noise=0.03))):
        w.sparse_priors = sp
        be = capi.Backend(device=device, use_graph=True)
        be.set_windows([w])
        for _ in range(3):
            be.solve(opts)
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            s = be.solve(opts)
        dt = (time.perf_counter() - t0) / reps
        be.close()
        out[name] = {"value": round(s[0].iterations / dt, 1), "ms_per_solve": round(1e3 * dt, 4)}
    w.sparse_priors = []
    return out


def _marg_case():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from vio_helpers import make_vio_window
    from marg_helpers import with_lonely_landmarks
    from sadvio_amd.synthetic import pre_marginalize
    w = with_lonely_landmarks(make_vio_window(n_kf=12, n_lmk=7200, seed=6), 11, 40)
    keep, marg = pre_marginalize(w, 11)
    keep = keep[:300]
    imu = [f for f in w.imu_factors if f["kf_i"] == 11 and f["kf_j"] == 10][0]
    return w, keep, marg, imu


def marginalize_cpu_leg():
    """The oracle's CPU restatement of Marginalization::computeSchurComplement etc. (marginalization.cpp:213-265,318-342) on the
    config-3 shaped window, timed before the GPU legs; its prior is kept for the comparison."""
    import numpy as np
    from oracle import oracle
    w, keep, marg, imu = _marg_case()
    t = time.perf_counter()
    ref = oracle.marginalize(w, 11, marg, keep, kf_keep=10, marg_has_imu=True, imu=imu, priors=w.pose_priors)
    rec = {"cpu_oracle_ms": round(1e3 * (time.perf_counter() - t), 1),
           "cpu_kind": "port (cyclic Jacobi eigen-solver, one thread; the reference uses Eigen::SelfAdjointEigenSolver)"}
    t = time.perf_counter()
    A = np.asarray(ref["Ak"])
    np.linalg.eigh(0.5 * (A + A.T))
    rec["cpu_lapack_eigh_of_Ak_ms"] = round(1e3 * (time.perf_counter() - t), 1)
    return rec, ref["J"].T @ ref["J"]


def marginalize_leg(device, cpu):
    """sadvio_ba_marginalize (K8) on a config-3 shaped window: 12-KF VIO, frame0's IMU + visual factors + previous prior,
    m = 135 marginalised / n = 915 kept columns (300 kept landmarks), in both forms of the prior (sadvio_ba.h): the Cholesky form
    (`gpu_ms`: no eigen-decomposition, the prior stays on the device) and the reference's eigen form (`gpu_ms_eigen_form`);
    `cpu` = marginalize_cpu_leg()'s record or None."""
    import numpy as np
    from sadvio_amd import capi
    w, keep, marg, imu = _marg_case()
    be = capi.Backend(device=device)
    be.set_windows([w])
    common = dict(kf_marg=11, lmk_marg=marg, lmk_keep=keep, kf_keep=10, marg_has_imu=True, imu=imu, priors=w.pose_priors, eig_cut="reference")
    times = {"cholesky": [], "eigen": []}
    for form in ("cholesky", "eigen"):
        for _ in range(5):
            t = time.perf_counter()
            g = be.marginalize(0, form=form, readback=False, **common)
            times[form].append(time.perf_counter() - t)
    info = {}
    for form in ("cholesky", "eigen"):
        gi = be.marginalize(0, form=form, **common)
        info[form] = gi["J"].T @ gi["J"]
    be.close()
    # the per-key-frame case: the same window with the PREVIOUS prior folded in (tests/golden_util.config3_marg_case, what
    # backend_step times): Ak is of full rank and takes the unpivoted wide-panel factorisation
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import config3_marg_case
    w3, margs3 = config3_marg_case(300)
    be = capi.Backend(device=device)
    be.set_windows([w3])
    t_steady = []
    for _ in range(6):
        t = time.perf_counter()
        g3 = be.marginalize(0, form="cholesky", readback=False, eig_cut="reference", **margs3)
        t_steady.append(time.perf_counter() - t)
    # ... and with a previous prior of FULL size (n_last = n = 915: frame0's 15 states + the 300 kept landmarks, all still in the
    # window), resident on the device: what every key-frame of a running back-end folds in
    margs_big = dict(margs3, last={"kf_keep": 11, "kf_col": g3["kf_col"], "lmk_index": g3["lmk_index"], "lmk_col": g3["lmk_col"]})
    t_big = []
    for _ in range(5):
        t = time.perf_counter()
        g4 = be.marginalize(0, form="cholesky", readback=False, eig_cut="reference", **margs_big)
        t_big.append(time.perf_counter() - t)
    be.close()
    rec = {"gpu_ms": round(1e3 * min(t_steady[1:]), 2), "n_full_steady_state": int(g3["n_full"]),
           "gpu_ms_full_size_previous_prior": round(1e3 * min(t_big[1:]), 2), "n_full_full_size_previous_prior": int(g4["n_full"]),
           "gpu_ms_first_marginalisation": round(1e3 * min(times["cholesky"][1:]), 2), "gpu_ms_eigen_form": round(1e3 * min(times["eigen"][1:]), 2),
           "m": int(g["m"]), "n": int(g["n"]), "n_full": int(g["n_full"]), "jacobi_sweeps_eigen_form": list(g["sweeps"]),
           "eig_cut": "reference (absolute 1e-12, marginalization.hpp:58)",
           "forms_information_rel_diff": float(np.abs(info["cholesky"] - info["eigen"]).max() / np.abs(info["eigen"]).max()),
           "what": "`gpu_ms` = the per-key-frame call (Cholesky form, a previous prior on frame0's 15 states folded in: full rank, unpivoted "
                   "factorisation, every pivot tested); `gpu_ms_full_size_previous_prior` = the same with a resident previous prior over all 915 "
                   "kept columns (J^T J of the previous prior on the matrix cores); `gpu_ms_first_marginalisation` = the same window without an earlier prior (rank deficient: rank-revealing route); "
                   "`gpu_ms_eigen_form`, n_full, the information differences and the CPU figures are of the latter",
           "workload": "config-3 shape: 12-KF VIO window, 300 kept landmarks, IMU + visual factors of frame0; prior left on the device (no read-back)"}
    if cpu is not None:
        rec.update(cpu[0])
        rec["information_rel_diff_vs_oracle"] = float(np.abs(info["cholesky"] - cpu[1]).max() / np.abs(cpu[1]).max())
    return rec


def backend_step_leg(device, reps=5):
    """One back-end key-frame step as the reference's timers bracket it (slamBiMonoVIO.cpp:569-594: marginalize (+ sparsify), then
    localMapVIOptimization = graph build + solve + write-back), config-3 shape (12-KF VIO window, 7 200 landmarks, 300 kept
    landmarks: m = 135, n = 915): marginalize -> [sparsify] -> set_windows with the prior -> GN-10 solve -> get_deltas, wall clock of
    each call through the C ABI with the caller's structs already marshalled (what a C++ caller holds). The prior never leaves
    the device (sadvio_ba.h: SADVIO_PRIOR_RESIDENT). Best of `reps` after one warm-up."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sadvio_amd import capi
    from golden_util import config3_marg_case
    w, margs = config3_marg_case(300)
    w2, _ = config3_marg_case(300)
    w2.pose_priors = []; w2.kf_const = np.zeros(w2.n_kf, dtype=np.uint8); w2.kf_const[11] = 1
    w2.imu_factors = [f for f in w2.imu_factors if f["kf_i"] != 11]
    opts = capi.gn_options(GN_ITERS)
    idx = ("kf_keep", "kf_col", "lmk_index", "lmk_col")
    out = {"workload": "config-3 shape: marginalise key-frame 11 of a 12-KF VIO window (m = 135, n = 915), then GN-10 solve of the window with the prior",
           "brackets": "slamBiMonoVIO.cpp:569-594", "unit": "ms"}
    for name, form, sparsif in (("cholesky_form_sparsified", "cholesky", True), ("cholesky_form_dense_prior", "cholesky", False),
                                ("eigen_form_sparsified", "eigen", True)):
        be = capi.Backend(device=device, use_graph=True)
        prep_w = be.prepare([w])
        prep_dense = None
        best = None
        for rep in range(reps + 1):
            ph = {}
            be.set_prepared(prep_w)       # the window that still holds frame0: resident in a live system
            t0 = time.perf_counter()
            g = be.marginalize(0, form=form, eig_cut="reference", readback=False, **margs)
            t1 = time.perf_counter(); ph["marginalize"] = t1 - t0
            if sparsif:
                w2.sparse_raw = be.sparsify(0, g, vio=True, raw=True); w2.dense_prior = None
                ph["sparsify"] = time.perf_counter() - t1
                prep = be.prepare([w2])
            else:
                w2.sparse_raw = None; w2.dense_prior = {k: g[k] for k in idx}
                prep = prep_dense = prep_dense or be.prepare([w2])
            t1 = time.perf_counter()
            be.set_prepared(prep)
            t2 = time.perf_counter(); ph["set_windows"] = t2 - t1
            s = be.solve(opts)[0]
            t3 = time.perf_counter(); ph["solve"] = t3 - t2
            be.get_deltas(0)
            ph["get_deltas"] = time.perf_counter() - t3
            ph["total"] = sum(ph.values())
            if rep > 0 and (best is None or ph["total"] < best["total"]):
                best = ph
        be.close()
        out[name] = {k: round(1e3 * v, 3) for k, v in best.items()}
        out[name]["iterations"] = int(s.iterations)
    w2.sparse_raw = None; w2.dense_prior = None
    out["value"] = out["cholesky_form_sparsified"]["total"]
    out["what"] = "`value` = the product path (Cholesky-form prior, NFR sparsification as SaDVIO's VIO pipeline enables it with sparsification: 1)"
    return out


if __name__ == "__main__":
    main()
