#!/usr/bin/env bash
# Memory-pipe counters of the headline step's kernels: average latency of vector / scalar / LDS instructions
# (SQ_INST_LEVEL_* / SQ_INSTS_*), texture-addresser and L1 stalls, scalar-cache hit rate.  Separate --pmc passes, kernel-trace off.
# usage (GPU box, repo root): bash tools/pmc_latency.sh <tag>
set -u
TAG=${1:-pmc_lat}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_a" -- python3 $B > /dev/null 2> "$OUT/a.err"
# (all eight texture-pipe counters in one pass: 'Request exceeds the capabilities of the hardware', and rocprofv3 then hung until the call's limit)
timeout 600 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d "$OUT/pmc_b" -- python3 $B > /dev/null 2> "$OUT/b.err"
timeout 600 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pmc_c" -- python3 $B > /dev/null 2> "$OUT/c.err"
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "nnp_" "$OUT/summary.txt"
tail -3 "$OUT/a.err" "$OUT/b.err" "$OUT/c.err" | grep -i "error\|fail\|invalid" | head
