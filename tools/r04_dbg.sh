#!/usr/bin/env bash
# scratch GPU call of round 4 (edited per experiment)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py -x -q 2>&1 | tail -3 > $O/pytest_ch.txt; cat $O/pytest_ch.txt
timeout 600 python3 tools/ab_step.py base narrow > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
timeout 200 python3 tools/experiments/sort_stamps.py > $O/sort_stamps.txt 2>&1; cat $O/sort_stamps.txt
RFOPS_LIB=rfnet_amd/variants/librfops_narrow.so timeout 200 python3 tools/experiments/sort_stamps.py > $O/sort_stamps_narrow.txt 2>&1; cat $O/sort_stamps_narrow.txt
timeout 300 python3 tools/soak_culled.py 60 > $O/soak_culled.txt 2>&1; tail -2 $O/soak_culled.txt
