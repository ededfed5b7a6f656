#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory into a small text summary (per-kernel
time from --kernel-trace --stats; per-kernel mean PMC values from the --pmc passes)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("nn_sweep", "nn_pack", "nn_rowmerge", "nn_colresolve", "nn_resolve", "nn_grad", "pack_kernel",
                "am_rowk_kernelILb1ELb1", "am_rowk_kernelILb0ELb1", "am_rowl", "am_match", "am_init", "mcg_kernel",
                "mc_partial", "fps_reg", "query_ball", "three_nn"):
        if key in name:
            return key
    return name[:60]


def main(out):
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats:", os.path.relpath(f, out))
        for row in csv.DictReader(open(f)):
            print("  {:40s} calls={:>6s} total_ns={:>12s} avg_ns={:>10s} pct={}".format(
                short(row.get("Name", "")), row.get("Calls", ""), row.get("TotalDurationNs", ""),
                row.get("AverageNs", ""), row.get("Percentage", "")))
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                acc[short(row.get("Kernel_Name", ""))][row.get("Counter_Name", "")].append(
                    float(row.get("Counter_Value", 0)))
            print("== counters:", os.path.relpath(f, out))
            for k, cs in sorted(acc.items()):
                for c, vals in sorted(cs.items()):
                    print(f"  {k:30s} {c:24s} mean/dispatch={sum(vals) / len(vals):.6g} n={len(vals)}")


if __name__ == "__main__":
    main(sys.argv[1])
