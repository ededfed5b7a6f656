#!/usr/bin/env bash
# round 4, GPU call 13: product state -- full gpu suite, soaks, the profile set of tools/profile_bench.sh, A/B against the shared-group build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04m; mkdir -p "$O"
cd "$R"
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > "$O/pytest_gpu.txt" 2>&1
tail -5 "$O/pytest_gpu.txt"
timeout 600 python3 tools/ab_step.py base shared4 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
timeout 400 python3 tools/soak_culled.py 300 > "$O/soak_culled.txt" 2>&1; tail -2 "$O/soak_culled.txt"
timeout 400 python3 tools/soak_step.py 300 > "$O/soak_step.txt" 2>&1; tail -2 "$O/soak_step.txt"
bash tools/profile_bench.sh r04m/prof
