#!/usr/bin/env bash
# round 4, GPU call 12: where the fused step's time goes (timing-only ablations)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04l; mkdir -p "$O"
cd "$R"
timeout 900 python3 tools/ab_step.py base nofuse fa1 fa2 fa4 fa7 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
