#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04w; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py -x -q ) > "$O/pytest.txt" 2>&1
tail -3 "$O/pytest.txt"
for lib in "" rfnet_amd/variants/librfops_big64.so; do echo "== RFOPS_LIB=$lib"; RFOPS_LIB=$lib timeout 300 python3 tools/opbench.py chamfer --iters 20 2>&1 | grep -E "nn_distance fwd|C2 nn" ; RFOPS_LIB=$lib timeout 300 python3 tools/opbench.py ns --iters 20 2>&1 | grep -E "north"; RFOPS_LIB=$lib timeout 300 python3 tools/opbench.py model --iters 20 2>&1 | grep -E "nn_distance fwd"; done > "$O/opbench_leaf.txt" 2>&1; cat "$O/opbench_leaf.txt"
timeout 600 python3 tools/ab_step.py base big64 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
