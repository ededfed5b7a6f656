#!/usr/bin/env bash
# rocprofv3 kernel trace + two SQ counter passes (+ FETCH_SIZE / WRITE_SIZE, each in a pass of its own) over one python tool.
# usage: bash tools/profile_py.sh <tag> <script.py> [args...]   -> gpurun_out/<tag>/summary.txt
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
SCRIPT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$SCRIPT" "$@" > "$OUT/run.txt" 2> "$OUT/stats.err"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$SCRIPT" "$@" > /dev/null 2> "$OUT/pmc_sq.err"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- python3 "$SCRIPT" "$@" > /dev/null 2> "$OUT/pmc_sq2.err"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetchsize" -- python3 "$SCRIPT" "$@" > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$SCRIPT" "$@" > /dev/null 2> "$OUT/pmc_write.err"
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
