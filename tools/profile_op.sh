#!/usr/bin/env bash
# rocprofv3 kernel-trace + PMC passes (SQ counters, then FETCH_SIZE and WRITE_SIZE each in a pass of its own) over one opbench workload.  usage: bash tools/profile_op.sh <what> <tag>
set -u
WHAT=${1:-emd}; TAG=${2:-op}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/opbench.py" "$WHAT" --iters 4 > "$OUT/opbench.txt" 2> "$OUT/stats.err"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/tools/opbench.py" "$WHAT" --iters 2 > /dev/null 2> "$OUT/pmc_sq.err"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- python3 "$R/tools/opbench.py" "$WHAT" --iters 2 > /dev/null 2> "$OUT/pmc_sq2.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetchsize" -- python3 "$R/tools/opbench.py" "$WHAT" --iters 2 > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/tools/opbench.py" "$WHAT" --iters 2 > /dev/null 2> "$OUT/pmc_write.err"
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
