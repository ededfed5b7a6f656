#!/usr/bin/env bash
# round 4, GPU call 9: full gpu suite, match_cost_grad (LDS columns, two barriers), cloud finishing times, bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04i; mkdir -p "$O"
cd "$R"
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > "$O/pytest_gpu.txt" 2>&1
tail -6 "$O/pytest_gpu.txt"
timeout 600 python3 tools/ab_mcg.py base mgold > "$O/ab_mcg.txt" 2>&1; cat "$O/ab_mcg.txt"
RFOPS_LIB=rfnet_amd/variants/librfops_cloudend.so timeout 200 python3 tools/experiments/cloud_end_times.py > "$O/cloud_end_times.txt" 2>&1; cat "$O/cloud_end_times.txt"
timeout 900 python3 bench.py --steps 100 --warmup 10 > "$O/bench.json" 2> "$O/bench.err"; tail -3 "$O/bench.err"; python3 -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','roofline')}); print(d.get('emd',{}).get('roofline')); print(d.get('emd',{}).get('extended_50')); print(d.get('per_op_roofline')); print({k:(v.get('ms_per_step'), v.get('kernels_us')) for k,v in d.get('by_distribution',{}).items()})"
