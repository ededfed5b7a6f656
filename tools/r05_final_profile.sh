#!/usr/bin/env bash
# round 5: the evidence set at the round's final build -- full gpu suite, smoke, tools/profile_bench.sh (bench + rocprofv3 stats + PMC),
# tools/profile_op.sh c4 (EMD counters), the C3 ball query / one-call profile, same-device A/Bs.  usage (GPU box): bash tools/r05_final_profile.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05h; mkdir -p "$O"
cd "$R"
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > "$O/pytest_gpu.txt" 2>&1
tail -4 "$O/pytest_gpu.txt"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.txt" 2>&1; tail -1 "$O/smoke.txt"
bash tools/profile_bench.sh r05h/prof
bash tools/profile_op.sh c4 r05h/c4
bash tools/profile_py.sh r05h/ball tools/run_ball_once.py 0.1
bash tools/profile_py.sh r05h/interp tools/run_interp_once.py
timeout 200 python3 tools/ab_ball.py > "$O/ab_ball.txt" 2>&1; cat "$O/ab_ball.txt"
timeout 200 python3 tools/ab_c3.py > "$O/ab_c3.txt" 2>&1; cat "$O/ab_c3.txt"
# same-device A/Bs against builds with this round's EMD changes switched off (python tools/build_variant.py, see profiles/README.md):
#   r4emd  = -DRFA_FGT_MIN_PAIRS=1e30 -DRFA_ROWSORT_MIN_PAIRS=1e30 -DRFA_MATCH_NT=0 -DRFA_MC_NT=0 -DRFA_MCG_ROWS=0 -DRFA_MG_NT=0 -DRFA_PK=0
#            -DRFA_PP_DENSE=0 -DRFA_PK_FUSED=0 -DRFA_SKIP_MAXT=0.2f
#   nofgt = -DRFA_FGT_MIN_PAIRS=1e30   nopk = -DRFA_PK=0 -DRFA_PP_DENSE=0 -DRFA_SKIP_MAXT=0.2f -DRFA_PK_FUSED=0   mcgold = -DRFA_MCG_ROWS=0 -DRFA_MG_NT=0
#   nolist = -DRFA_SKIP_MASK=0 (level 0's sweeps do not list the columns for level 1's)   gsf32 = -DRFP_GS_F64=0 -DRFP_GS_SEG=1 (the backward's fp32 LDS sums)
timeout 400 python3 tools/ab_emd_kernels.py r4emd nofgt nopk nolist base > "$O/ab_emd.txt" 2>&1; cut -c1-200 "$O/ab_emd.txt"
timeout 300 python3 tools/ab_emd_sizes.py nofgt base > "$O/ab_emd_sizes.txt" 2>&1; cut -c1-300 "$O/ab_emd_sizes.txt"
AB_MCG_SHAPES=1 timeout 300 python3 tools/ab_mcg.py mcgold base > "$O/ab_mcg.txt" 2>&1; cut -c1-300 "$O/ab_mcg.txt"
bash tools/experiments/trace_emd.sh > "$O/emd_launches.txt" 2>&1
timeout 300 python3 tools/ab_fps_sorted.py base > "$O/ab_fps_sorted.txt" 2>&1; cut -c1-300 "$O/ab_fps_sorted.txt"
timeout 400 python3 tools/ab_fps_sizes.py > "$O/ab_fps_sizes.txt" 2>&1; tail -9 "$O/ab_fps_sizes.txt"
./tools/ubench/stream_rate > "$O/stream_rate.txt" 2>&1; cat "$O/stream_rate.txt"
./tools/ubench/lds_atomic_rate > "$O/lds_atomic_rate.txt" 2>&1; cat "$O/lds_atomic_rate.txt"
timeout 600 python3 tools/ab_step.py gsf32 base > "$O/ab_step_f64.txt" 2>&1; cut -c1-400 "$O/ab_step_f64.txt"
timeout 300 python3 tools/ab_three_nn.py > "$O/ab_three_nn.txt" 2>&1; cat "$O/ab_three_nn.txt"
timeout 300 python3 tools/experiments/three_interpolate_rate.py > "$O/three_interpolate_rate.txt" 2>&1; cat "$O/three_interpolate_rate.txt"
timeout 300 python3 tools/experiments/group_point_rate.py > "$O/group_point_rate.txt" 2>&1; cat "$O/group_point_rate.txt"
