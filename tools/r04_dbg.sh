#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out/r04z
for lib in "" nozs div2 nohagg l64; do echo "=== $lib"; if [ -n "$lib" ]; then export RFOPS_LIB=rfnet_amd/variants/librfops_$lib.so; fi; timeout 200 python3 tools/experiments/soak_step_case.py 176 2>&1 | grep -v amdgpu | head -24; done > gpurun_out/r04z/case176.txt 2>&1
cat gpurun_out/r04z/case176.txt
