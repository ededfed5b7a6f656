#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
timeout 700 python3 tools/soak_culled.py 600 > $O/soak_culled_long.txt 2>&1; tail -1 $O/soak_culled_long.txt
timeout 700 python3 tools/soak_step.py 600 > $O/soak_step_long.txt 2>&1; tail -1 $O/soak_step_long.txt
