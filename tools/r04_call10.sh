#!/usr/bin/env bash
# round 4, GPU call 10: crowded-cloud fallback, LDS budget of the tiles, match_cost_grad variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04j; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py tests/test_gpu_emd.py tests/test_gpu_glue.py tests/test_gpu_chamfer_ext.py -x -q ) > "$O/pytest.txt" 2>&1
tail -4 "$O/pytest.txt"
timeout 600 python3 tools/ab_step.py base shared4 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
timeout 600 python3 tools/ab_mcg.py base mgold mgl mgl2 > "$O/ab_mcg.txt" 2>&1; cat "$O/ab_mcg.txt"
timeout 300 python3 tools/ab_dist.py base shared4 > "$O/ab_dist.txt" 2>&1; tail -20 "$O/ab_dist.txt"
timeout 100 python3 tools/soak_culled.py 60 > "$O/soak_culled.txt" 2>&1; tail -2 "$O/soak_culled.txt"
timeout 900 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$O/bench.json" 2> "$O/bench.err"; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step']); print({k:(round(v['ms_per_step'],4), {a:round(b*1e3,1) for a,b in v['auto_kernels_ms'].items()}, v['identical_to_dense_sweep']) for k,v in d['by_distribution'].items()}); print(d['per_op_roofline']['match_cost_grad'])"
