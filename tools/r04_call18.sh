#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04r; mkdir -p "$O"
cd "$R"
timeout 300 python3 tools/culled_stats.py 2>&1 | grep -E "sweep|dir" > "$O/base.txt"; cat "$O/base.txt"
RFOPS_LIB=rfnet_amd/variants/librfops_pack4.so timeout 300 python3 tools/culled_stats.py 2>&1 | grep -E "sweep|dir" > "$O/pack4.txt"; cat "$O/pack4.txt"
