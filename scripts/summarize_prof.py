#!/usr/bin/env python
"""Summarise a scripts/prof_bench.sh output directory: per-kernel average duration from the rocprofv3 kernel
trace and per-launch HBM traffic from the FETCH_SIZE / WRITE_SIZE PMC passes.

Corrections (MI355X_MICROARCH.md, "HBM"): both counters are reported in KiB-like units of 1024 B by rocprofv3;
on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so it is doubled. WRITE_SIZE is uncalibrated and is
reported as is."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    if "k_lm_pass" in name:   # the opening pass of a solve (INIT = true: one linearisation) and the per-step pass are different kernels
        return "k_lm_pass_init" if ", true>" in name.split("(")[0] else "k_lm_pass"
    for k in ("k_build_obs", "k_build", "k_solve", "k_backsub", "k_init_tables", "k_reset", "k_final", "k_decide", "k_linearize_probe"):
        if k in name:
            return k
    return None


def main(out):
    res = {"kernels": {}}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Name"])
            if k:
                # template variants of one kernel (k_solve<0, false> / <0, true> ...): calls-weighted average; prof_bench.sh profiles
                # the config-2 legs only, so that normally ONE variant is present
                d = res["kernels"].setdefault(k, {"calls": 0, "avg_us": 0.0, "min_us": 1e30, "max_us": 0.0})
                c = int(row["Calls"])
                d["avg_us"] = (d["avg_us"] * d["calls"] + float(row["AverageNs"]) / 1e3 * c) / max(d["calls"] + c, 1)
                d["calls"] += c
                d["min_us"] = min(d["min_us"], float(row["MinNs"]) / 1e3); d["max_us"] = max(d["max_us"], float(row["MaxNs"]) / 1e3)
                d.setdefault("variants", []).append(row["Name"].split("(")[0].replace("void ", "").replace("sadvio::", ""))
    for tag, scale in (("fetch", 2.0), ("write", 1.0)):
        acc = defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(out, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if k:
                    acc[k][0] += float(row["Counter_Value"])
                    acc[k][1] += 1
        for k, (v, n) in acc.items():
            res["kernels"].setdefault(k, {})[tag + "_bytes_per_launch"] = scale * 1024.0 * v / max(n, 1)
        if acc:   # per-kernel table of the raw counter (the per-launch rows are not kept)
            with open(os.path.join(out, "pmc_%s_per_kernel.csv" % tag), "w") as fo:
                fo.write("kernel,launches,%s_SIZE_sum,%s_SIZE_avg_per_launch\n" % (tag.upper(), tag.upper()))
                for k, (v, n) in sorted(acc.items()):
                    fo.write("%s,%d,%.1f,%.3f\n" % (k, n, v, v / max(n, 1)))
    for k, d in res["kernels"].items():
        if "fetch_bytes_per_launch" in d and "write_bytes_per_launch" in d:
            d["hbm_traffic_bytes_per_launch"] = d["fetch_bytes_per_launch"] + d["write_bytes_per_launch"]
    res["corrections"] = "FETCH_SIZE x 1024 B x 2 (gfx950 half-count); WRITE_SIZE x 1024 B (uncalibrated)"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
