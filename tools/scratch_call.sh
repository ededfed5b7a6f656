#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
timeout 600 python3 tools/ab_step.py base prev > $O/ab_step_sort3.txt 2>&1; cat $O/ab_step_sort3.txt
timeout 200 python3 tools/experiments/sort_stamps.py 2>&1 | grep -v amdgpu
timeout 200 python3 tools/soak_culled.py 60 | tail -1
timeout 200 python3 tools/soak_step.py 60 | tail -1
