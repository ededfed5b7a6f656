#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04v; mkdir -p "$O"
cd "$R"
timeout 900 python3 tools/ab_step.py base big48 big32 big96 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
