#!/usr/bin/env bash
# round 4, GPU call 11: fused step (sweep + backward in one launch): parity, A/B, soak; match_cost_grad atomics ablation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04k; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py tests/test_gpu_cabi.py tests/test_gpu_glue.py tests/test_gpu_chamfer_ext.py -x -q ) > "$O/pytest.txt" 2>&1
tail -4 "$O/pytest.txt"
timeout 600 python3 tools/ab_step.py base nofuse shared4 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
timeout 200 python3 tools/soak_step.py 120 > "$O/soak_step.txt" 2>&1; tail -2 "$O/soak_step.txt"
timeout 600 python3 tools/ab_mcg.py base mgabl1 > "$O/ab_mcg.txt" 2>&1; cat "$O/ab_mcg.txt"
