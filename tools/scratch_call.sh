#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_all.txt; cat $O/pytest_all.txt
timeout 600 python3 tools/ab_step.py base prev > $O/ab_step_sortfix.txt 2>&1; cat $O/ab_step_sortfix.txt
timeout 200 python3 tools/experiments/sort_stamps.py > $O/sort_stamps.txt 2>&1; cat $O/sort_stamps.txt
timeout 300 python3 tools/opbench.py fps --iters 10 > $O/opbench_fps.txt 2>&1; grep -i "fps\|farthest" $O/opbench_fps.txt | head -5
RFOPS_LIB=rfnet_amd/variants/librfops_prev.so timeout 300 python3 tools/opbench.py fps --iters 10 > $O/opbench_fps_prev.txt 2>&1; grep -i "fps\|farthest" $O/opbench_fps_prev.txt | head -5
