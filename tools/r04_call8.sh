#!/usr/bin/env bash
# round 4, GPU call 8: quad tiles (block re-scan form, packed list entries), dwordx3 sort loads, match_cost_grad prefetch, EMD tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py tests/test_gpu_emd.py -x -q ) > "$O/pytest.txt" 2>&1
tail -5 "$O/pytest.txt"
timeout 300 python3 tools/culled_stats.py > "$O/culled_stats.txt" 2>&1; cat "$O/culled_stats.txt"
RFOPS_LIB=rfnet_amd/variants/librfops_t16stamps.so timeout 300 python3 tools/culled_stats.py 2>&1 | head -4 > "$O/culled_stats_stamps.txt"; cat "$O/culled_stats_stamps.txt"
timeout 600 python3 tools/ab_step.py base shared4 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
python3 tools/experiments/sort_stamps.py > "$O/sort_stamps_randn.txt" 2>&1; cat "$O/sort_stamps_randn.txt"
timeout 600 python3 tools/ab_mcg.py base mgold > "$O/ab_mcg.txt" 2>&1; cat "$O/ab_mcg.txt"
timeout 100 python3 tools/soak_culled.py 60 > "$O/soak_culled.txt" 2>&1; tail -2 "$O/soak_culled.txt"
timeout 100 python3 tools/soak_step.py 60 > "$O/soak_step.txt" 2>&1; tail -2 "$O/soak_step.txt"
