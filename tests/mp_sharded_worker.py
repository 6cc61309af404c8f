"""Worker of tests/test_gpu_multiprocess.py: one PROCESS per rank, every rank with its own handle on GPU 0, the
library's in-place all-reduce hook (sadvio_ba_set_collective) backed by torch.distributed (gloo, host memory). Launched by
torch.distributed.run; rank 0 writes the comparison record to argv[1]. With argv[2] == "rccl" every rank takes ITS OWN GPU
(LOCAL_RANK) and the library's built-in RCCL all-reduce (sadvio_ba_comm_init_rccl over xGMI) instead: the N > 1 path as an 8-GPU
node runs it (tests/test_gpu_multiprocess.py skips that case on a one-GPU box)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")


def main():
    import torch
    import torch.distributed as dist
    from sadvio_amd import capi, sharding, synthetic
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rccl = len(sys.argv) > 2 and sys.argv[2] == "rccl"
    device = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
    dist.init_process_group(backend="gloo")
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    stats = {"calls": 0, "max_count": 0}

    def allreduce(ctx, dev, count, stream):       # in-place SUM over the ranks' device buffers, through host memory
        try:
            if hip.hipStreamSynchronize(stream) != 0:
                return 1
            a = np.empty(count, dtype=np.float64)
            if hip.hipMemcpy(a.ctypes.data, dev, 8 * count, 2) != 0:
                return 2
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if hip.hipMemcpy(dev, a.ctypes.data, 8 * count, 1) != 0:
                return 3
            stats["calls"] += 1
            stats["max_count"] = max(stats["max_count"], int(count))
            return 0
        except Exception:
            return 4

    record = {}
    cases = [("lds", dict(n_kf=8, n_lmk=1500, seed=41), capi.reference_options()),
             ("banded_hbm", dict(n_kf=60, n_lmk=9000, length=30.0, band=6, seed=43), capi.gn_options(4))]
    for name, kw, opts in cases:
        w = synthetic.make_window(**kw)            # same seed on every rank: replicated pose side
        be = capi.Backend(device=device)
        if rccl:
            uid = [be.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            be.comm_init_rccl(rank, world, uid[0])
            info = be.comm_info()
            assert info["is_rccl"] and (info["nranks"], info["rank"], info["device"]) == (world, rank, device), info
        else:
            be.set_collective(rank, world, allreduce)
        be.set_windows([sharding.shard_window(w, rank, world)])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        be.close()
        # gather the landmark shards and every rank's pose deltas on rank 0
        parts = [None] * world
        dist.all_gather_object(parts, (d["pose"], d["lmk"], s.iterations, s.termination, s.final_cost))
        if rank == 0:
            single = capi.Backend(device=device)
            single.set_windows([w])
            s1 = single.solve(opts)[0]
            d1 = single.get_deltas(0)
            single.close()
            lmk = np.concatenate([p[1] for p in parts])
            record[name] = {
                "iterations": [p[2] for p in parts], "termination": [p[3] for p in parts], "single_iterations": s1.iterations,
                "single_termination": s1.termination,
                "pose_identical_across_ranks": all(np.array_equal(p[0], parts[0][0]) for p in parts),
                "dpose_vs_single": float(np.abs(parts[0][0] - d1["pose"]).max()), "dlmk_vs_single": float(np.abs(lmk - d1["lmk"]).max()),
                "dcost_rel": abs(parts[0][4] - s1.final_cost) / abs(s1.final_cost),
                "rccl": rccl, "allreduce_calls": stats["calls"], "max_count": stats["max_count"], "n_p": 6 * int((w.kf_const == 0).sum())}
            if name == "lds":
                from oracle import oracle
                ref = oracle.solve(w, opts)
                record[name]["dpose_vs_oracle"] = float(np.abs(parts[0][0] - ref["pose"]).max())
                record[name]["dlmk_vs_oracle"] = float(np.abs(lmk - ref["lmk"]).max())
                record[name]["oracle_iterations"] = ref["summary"].iterations
        stats["calls"] = 0; stats["max_count"] = 0
        dist.barrier()
    if rank == 0:
        json.dump(record, open(sys.argv[1], "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
