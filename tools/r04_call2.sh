#!/usr/bin/env bash
# round 4, second GPU call: clean C4-only PMC passes, sort phase stamps (randn / collapsed), gpu test suite on the de-knobbed build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b; mkdir -p "$O"
cd "$R"
python3 tools/experiments/sort_stamps.py > "$O/sort_stamps_randn.txt" 2>&1
python3 tools/experiments/sort_stamps_collapsed.py > "$O/sort_stamps_collapsed.txt" 2>&1
bash tools/profile_op.sh c4 r04b/c4
( time python3 -m pytest tests -m gpu -x -q ) > "$O/pytest_gpu.txt" 2>&1
tail -5 "$O/pytest_gpu.txt"; cat "$O/sort_stamps_randn.txt" "$O/sort_stamps_collapsed.txt"
