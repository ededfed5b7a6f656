#!/usr/bin/env bash
# round 4, GPU call 4: quad tiles with dynamic LDS + unrolled list rounds: parity, phase stamps, A/B; EMD waves-per-SIMD A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py -x -q ) > "$O/pytest_chamfer.txt" 2>&1
tail -5 "$O/pytest_chamfer.txt"
timeout 300 python3 tools/culled_stats.py > "$O/culled_stats.txt" 2>&1; cat "$O/culled_stats.txt"
RFOPS_LIB=rfnet_amd/variants/librfops_t16stamps.so timeout 300 python3 tools/culled_stats.py 2>&1 | head -4 > "$O/culled_stats_stamps.txt"; cat "$O/culled_stats_stamps.txt"
timeout 600 python3 tools/ab_step.py base shared4 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
AB_ENV=RFOPS_LIB timeout 600 python3 tools/ab_emd.py rfnet_amd/librfops.so rfnet_amd/variants/librfops_amw8k.so > "$O/ab_emd_waves.txt" 2>&1; cat "$O/ab_emd_waves.txt"
