#!/usr/bin/env bash
# scratch GPU call of round 4 (edited per experiment)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_emd.py -x -q 2>&1 | tail -3 > $O/pytest_ch.txt; cat $O/pytest_ch.txt
timeout 600 python3 tools/ab_step.py base prev > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
timeout 600 python3 tools/ab_step.py --shape 32,16384,16384 base prev > $O/ab_step_ns.txt 2>&1; cat $O/ab_step_ns.txt
RFOPS_LIB=rfnet_amd/variants/librfops_tl.so timeout 300 python3 tools/experiments/wave_timeline.py > $O/timeline.txt 2>&1; cat $O/timeline.txt
timeout 300 python3 tools/soak_culled.py 60 > $O/soak_culled.txt 2>&1; tail -2 $O/soak_culled.txt
timeout 300 python3 tools/soak_step.py 60 > $O/soak_step.txt 2>&1; tail -2 $O/soak_step.txt
timeout 300 python3 tools/ab_mcg.py base prev > $O/ab_mcg.txt 2>&1; cat $O/ab_mcg.txt
