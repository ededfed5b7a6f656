#!/usr/bin/env bash
# round 4, GPU call 6: quad tiles with index-carrying scans; dense-bin sort loops (separate copies); match_cost_grad tile knobs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04g; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py -x -q ) > "$O/pytest_chamfer.txt" 2>&1
tail -5 "$O/pytest_chamfer.txt"
timeout 300 python3 tools/culled_stats.py > "$O/culled_stats.txt" 2>&1; cat "$O/culled_stats.txt"
RFOPS_LIB=rfnet_amd/variants/librfops_t16stamps.so timeout 300 python3 tools/culled_stats.py 2>&1 | head -4 > "$O/culled_stats_stamps.txt"; cat "$O/culled_stats_stamps.txt"
timeout 600 python3 tools/ab_step.py base shared4 nohagg > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
python3 tools/experiments/sort_stamps.py > "$O/sort_stamps_randn.txt" 2>&1; cat "$O/sort_stamps_randn.txt"
python3 tools/experiments/sort_stamps_collapsed.py > "$O/sort_stamps_collapsed.txt" 2>&1; cat "$O/sort_stamps_collapsed.txt"

timeout 100 python3 tools/soak_culled.py 60 > "$O/soak_culled.txt" 2>&1; tail -2 "$O/soak_culled.txt"
timeout 100 python3 tools/soak_step.py 60 > "$O/soak_step.txt" 2>&1; tail -2 "$O/soak_step.txt"
