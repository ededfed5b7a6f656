#!/usr/bin/env bash
# round 4, first GPU call: issue-cost ubenches for the sub-wave tile decision + PMC evidence for the EMD / C3 kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out/r04a"
"$R/tools/ubench/valu_rate" > "$R/gpurun_out/r04a/valu_rate.txt" 2>&1
"$R/tools/ubench/gather_scan" > "$R/gpurun_out/r04a/gather_scan.txt" 2>&1
bash "$R/tools/profile_op.sh" emd r04a/emd
bash "$R/tools/profile_op.sh" fps r04a/c3
tail -30 "$R/gpurun_out/r04a/valu_rate.txt"; cat "$R/gpurun_out/r04a/gather_scan.txt"
