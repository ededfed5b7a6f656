#!/usr/bin/env bash
# rocprofv3 passes over one Chamfer shape.  usage: bash tools/profile_nn.sh <tag> <b> <n> <m> <mode>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/run_nn_once.py" "$@" 8 > "$OUT/run.txt" 2> "$OUT/stats.err"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/tools/run_nn_once.py" "$@" 3 > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- python3 "$R/tools/run_nn_once.py" "$@" 3 > /dev/null 2> "$OUT/pmc_sq2.err"
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQC_ICACHE_MISSES SQC_ICACHE_REQ --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/tools/run_nn_once.py" "$@" 3 > /dev/null 2> "$OUT/pmc_fetch.err"
# FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md: TCC counter budget) -- asked for
# together the run hung until the box's time limit
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetchsize" -- python3 "$R/tools/run_nn_once.py" "$@" 3 > /dev/null 2> "$OUT/pmc_fetchsize.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/tools/run_nn_once.py" "$@" 3 > /dev/null 2> "$OUT/pmc_write.err"
python3 "$R/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
tail -3 "$OUT"/pmc_fetch.err
