#!/usr/bin/env bash
# round 4, GPU call 14: match_cost_grad with lane-held columns; interleaved sweep grid
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04n; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_emd.py tests/test_gpu_fuzz.py -x -q ) > "$O/pytest.txt" 2>&1
tail -3 "$O/pytest.txt"
timeout 600 python3 tools/ab_mcg.py base mgscal mgold > "$O/ab_mcg.txt" 2>&1; cat "$O/ab_mcg.txt"
timeout 600 python3 tools/ab_step.py base mix > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
