#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
for rnd in 0 1; do
timeout 300 python3 tools/experiments/match_order.py 2>&1 | grep -v amdgpu | sed 's/^/base  /'
RFOPS_LIB=rfnet_amd/variants/librfops_mcfwd.so timeout 300 python3 tools/experiments/match_order.py 2>&1 | grep -v amdgpu | sed 's/^/mcfwd /'
done > $O/match_order.txt; cat $O/match_order.txt
