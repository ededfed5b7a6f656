#!/usr/bin/env bash
# round 4, GPU call 15: STR leaf of small clouds (32 in the product now), grid order variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04o; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py -x -q ) > "$O/pytest.txt" 2>&1
tail -3 "$O/pytest.txt"
timeout 900 python3 tools/ab_step.py base leaf64 leaf16 mix2 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
timeout 300 python3 tools/culled_stats.py > "$O/culled_stats.txt" 2>&1; head -3 "$O/culled_stats.txt"
timeout 600 python3 tools/ab_modes.py > "$O/ab_modes.txt" 2>&1; tail -30 "$O/ab_modes.txt"
