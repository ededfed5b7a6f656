#!/usr/bin/env bash
# Profiles bench.py on the GPU box: kernel-trace stats of the default run, then HBM traffic and SQ
# counters in SEPARATE --pmc passes (gpurun refuses pmc combined with trace domains other than
# kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit one pass) on the headline step alone
# (--no-extras), and a kernel-trace of the C5 forward.
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag>
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 100 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"
# (a) the headline step alone: every nnp_sweep launch of the C2 grid is a randn launch, so the trace's
#     per-kernel average is directly comparable with bench.py's own hipEvent figure in the same run
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
# (b) the default command (by_distribution, EMD, north star, C5 extras included: the same grid then
#     also carries the other distributions' launches)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/bench_full_under_rocprof.json" 2> "$OUT/stats_full.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetchsize" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$OUT/pmc_write.err"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$OUT/pmc_sq.err"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES --output-format csv -d "$OUT/pmc_sq2" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$OUT/pmc_sq2.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -- python3 "$R/tools/c5_forward.py" 6 > "$OUT/c5.log" 2>&1
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
python3 "$R/tools/opbench.py" all --iters 20 > "$OUT/opbench.txt" 2>&1
python3 "$R/tools/opbench.py" c5 --iters 12 >> "$OUT/opbench.txt" 2>&1
tail -5 "$OUT/summary.txt"
