#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
timeout 900 python3 tools/ab_step.py base cap24 cap24u3 cap20 > $O/ab_step_cap.txt 2>&1; cat $O/ab_step_cap.txt
