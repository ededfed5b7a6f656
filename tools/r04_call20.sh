#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r04t
python3 tools/experiments/crowded_flag_probe.py 2>&1 | tail -5
