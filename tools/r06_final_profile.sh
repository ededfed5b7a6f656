#!/usr/bin/env bash
# round 6: the evidence set at the round's final build -- full gpu suite, smoke, tools/profile_bench.sh (bench + rocprofv3 stats + PMC),
# tools/profile_op.sh c4 (EMD counters), the EMD launch timeline, same-device A/Bs of this round's changes, soaks.
# usage (GPU box): bash tools/r06_final_profile.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06h; mkdir -p "$O"
cd "$R"
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > "$O/pytest_gpu.txt" 2>&1
tail -4 "$O/pytest_gpu.txt"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.txt" 2>&1; tail -1 "$O/smoke.txt"
bash tools/profile_bench.sh r06h/prof
bash tools/profile_op.sh c4 r06h/c4
bash tools/experiments/trace_emd.sh "" 34 > "$O/emd_launches.txt" 2>&1
bash tools/experiments/trace_emd.sh 50 106 > "$O/emd50_launches.txt" 2>&1
bash tools/experiments/trace_emd.sh big 26 > "$O/emd_big_launches.txt" 2>&1
timeout 200 python3 tools/ab_emd_modes.py > "$O/ab_emd_modes.txt" 2>&1; cat "$O/ab_emd_modes.txt"
timeout 300 python3 tools/ab_group_grad.py > "$O/ab_group_grad.txt" 2>&1; cut -c1-260 "$O/ab_group_grad.txt"
timeout 200 python3 tools/ab_c3.py > "$O/ab_c3.txt" 2>&1; cat "$O/ab_c3.txt"
timeout 300 python3 tools/soak_emd_live.py 120 6 > "$O/soak_emd_live.txt" 2>&1; tail -2 "$O/soak_emd_live.txt"
timeout 300 python3 tools/soak_emd_live.py 60 3 large > "$O/soak_emd_large.txt" 2>&1; tail -2 "$O/soak_emd_large.txt"
timeout 300 python3 tools/soak_emd_live.py 60 5 batch > "$O/soak_emd_batch.txt" 2>&1; tail -2 "$O/soak_emd_batch.txt"
timeout 200 python3 tools/ab_emd_cull.py base > "$O/emd_sizes.txt" 2>&1; cut -c1-200 "$O/emd_sizes.txt"
timeout 300 python3 tools/soak_step.py 100 > "$O/soak_step.txt" 2>&1; tail -2 "$O/soak_step.txt"
timeout 300 python3 tools/soak_culled.py 60 > "$O/soak_culled.txt" 2>&1; tail -2 "$O/soak_culled.txt"
timeout 300 python3 tools/soak_three_nn.py 60 3 > "$O/soak_three_nn.txt" 2>&1; tail -2 "$O/soak_three_nn.txt"
( RF_FUZZ_SCALE=20 timeout 900 python3 -m pytest tests/test_gpu_fuzz.py -q 2>&1 | grep -E "^E  .*Assertion|passed|failed" ) > "$O/fuzz20.txt" 2>&1; tail -6 "$O/fuzz20.txt"
