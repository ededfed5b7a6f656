#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04u; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py tests/test_gpu_emd.py -x -q ) > "$O/pytest.txt" 2>&1
tail -4 "$O/pytest.txt"
timeout 300 python3 tools/opbench.py emd --iters 8 2>&1 | grep -A7 "50-level" > "$O/opbench_emd50.txt"; cat "$O/opbench_emd50.txt"
