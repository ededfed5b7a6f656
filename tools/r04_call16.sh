#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04p; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_emd.py tests/test_gpu_fuzz.py -x -q ) > "$O/pytest.txt" 2>&1
tail -3 "$O/pytest.txt"
timeout 600 python3 tools/ab_mcg.py base mgscal mga2 mga3 > "$O/ab_mcg.txt" 2>&1; cat "$O/ab_mcg.txt"
