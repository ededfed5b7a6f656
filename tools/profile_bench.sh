#!/usr/bin/env bash
# Profiles bench.py on the GPU box: kernel-trace stats, then HBM traffic and SQ counters in
# SEPARATE --pmc passes (gpurun refuses pmc combined with trace domains other than kernel-trace).
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_sq.err"
find "$OUT" -name "*.csv" | head -50
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq2" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_sq2.err"
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
