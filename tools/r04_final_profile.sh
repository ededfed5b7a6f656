#!/usr/bin/env bash
# round 4: the evidence set at the round's final build -- full gpu suite, smoke, same-device A/B against the round-3 sweep form, tools/profile_bench.sh (bench + rocprofv3 stats + PMC), tools/profile_op.sh c4 (EMD counters), soaks.  usage (GPU box): bash tools/r04_final_profile.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h; mkdir -p "$O"
cd "$R"
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > "$O/pytest_gpu.txt" 2>&1
tail -4 "$O/pytest_gpu.txt"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.txt" 2>&1; tail -1 "$O/smoke.txt"
timeout 600 python3 tools/ab_step.py base shared4 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
bash tools/profile_bench.sh r04h/prof
bash tools/profile_op.sh c4 r04h/c4
timeout 300 python3 tools/soak_culled.py 200 > "$O/soak_culled.txt" 2>&1; tail -1 "$O/soak_culled.txt"
timeout 300 python3 tools/soak_step.py 200 > "$O/soak_step.txt" 2>&1; tail -1 "$O/soak_step.txt"
