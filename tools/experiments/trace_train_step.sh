R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_train; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/c5_train_step.py" 6 > "$OUT/log.txt" 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python3 $R/tools/kstats.py $f 6 24 | cut -c1-170
tail -3 $OUT/log.txt
