#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory into a small text summary (per-kernel
time from --kernel-trace --stats; per-kernel mean PMC values from the --pmc passes)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    if "am_rowk_kernel<" in name:  # <P3, P1 mode, rows per lane>: one name per fused phase
        return "am_rowk<" + name.split("am_rowk_kernel<")[1].split(">")[0].replace(" ", "") + ">"
    for key in ("nnp_grad_sorted", "nnp_sweep", "nnp_sort_reg", "nnp_sort", "nn_sweep", "nn_pack", "nn_rowmerge", "nn_colresolve", "nn_resolve", "nn_grad", "pack_kernel",
                "am_rowk_kernelILb1ELb1", "am_rowk_kernelILb0ELb1", "am_rowl", "am_p2_live", "am_compact", "am_match", "am_init", "mcg_rows", "mcg_kernel",
                "mc_partial", "mc_final", "emd_fused", "emd_pack_cols", "am_cull", "fps_sorted", "fps_reg", "rows_csr_build", "rows_csr_gather", "query_ball_boxes", "qx_flags", "query_ball_lanes", "query_ball", "three_nn_boxes", "three_nn", "three_interpolate_rows", "three_interpolate_grad_tile",
                "group_point_grad", "group_point", "gather_kernel"):
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
    # per-dispatch trace: split each kernel by launch shape (bench.py also runs the kernel at the
    # north-star 16384^2 shape as an extra, so the per-name average of --stats mixes two workloads)
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))
            acc[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        print("== kernel trace by launch shape:", os.path.relpath(f, out))
        for (k, grid, wg), v in sorted(acc.items()):
            if "at::native" in k or "rocclr" in k:
                continue
            print(f"  {k:30s} grid_x={grid:>9d} wg={wg:>5d} calls={len(v):>5d} avg_ns={sum(v) / len(v):12.1f}")
    for sub in ("pmc_fetch", "pmc_fetchsize", "pmc_write", "pmc_sq", "pmc_sq2"):
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                nm = short(row.get("Kernel_Name", ""))
                # per launch SHAPE: one kernel name serves several workloads in a run (C4, training size, evaluation size)
                nm = "%s grid=%s wg=%s" % (nm[:34], row.get("Grid_Size", row.get("Grid_Size_X", "?")), row.get("Workgroup_Size", row.get("Workgroup_Size_X", "?")))
                acc[nm][row.get("Counter_Name", "")].append(float(row.get("Counter_Value", 0)))
            print("== counters:", os.path.relpath(f, out))
            for k, cs in sorted(acc.items()):
                for c, vals in sorted(cs.items()):
                    print(f"  {k:58s} {c:24s} mean/dispatch={sum(vals) / len(vals):.6g} n={len(vals)}")


if __name__ == "__main__":
    main(sys.argv[1])
