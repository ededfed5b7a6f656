#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04s; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py -x -q ) > "$O/pytest.txt" 2>&1
tail -4 "$O/pytest.txt"
python3 tools/experiments/sort_stamps.py > "$O/sort_stamps_randn.txt" 2>&1; cat "$O/sort_stamps_randn.txt"
python3 tools/experiments/sort_stamps_collapsed.py > "$O/sort_stamps_collapsed.txt" 2>&1; cat "$O/sort_stamps_collapsed.txt"
timeout 900 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err"; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step']); print({k:(round(v['ms_per_step'],4), {a:round(b*1e3,1) for a,b in v['auto_kernels_ms'].items()}, v['identical_to_dense_sweep']) for k,v in d['by_distribution'].items()})"
