#!/usr/bin/env bash
# gpurun_out/r06h/ (tools/r06_final_profile.sh, merged back from the GPU box) -> profiles/r06_*: the judged copies.
# The newest file by mtime wins where earlier calls left same-named artefacts in the merged scratch tree.
set -eu
cd "$(dirname "$0")/.."
O=gpurun_out/r06h; P=profiles
newest() { ls -t $1 2>/dev/null | head -1; }
cp $O/prof/bench.json $P/r06_bench.json
cp $O/prof/bench_under_rocprof.json $P/r06_bench_headline_under_rocprof.json
cp $O/prof/bench_full_under_rocprof.json $P/r06_bench_full_under_rocprof.json
cp "$(newest "$O/prof/stats/*/*kernel_stats.csv")" $P/r06_kernel_stats_headline.csv
cp "$(newest "$O/prof/stats_full/*/*kernel_stats.csv")" $P/r06_kernel_stats_full.csv
cp "$(newest "$O/prof/c5/*/*kernel_stats.csv")" $P/r06_c5_forward_kernel_stats.csv
{ cat $O/prof/summary.txt; echo; echo "==== tools/profile_op.sh c4: approx_match + match_cost + match_cost_grad at C4 ===="; cat $O/c4/summary.txt; } > $P/r06_rocprofv3_summary.txt
grep -v amdgpu.ids $O/prof/opbench.txt > $P/r06_opbench.txt
for f in emd_launches emd50_launches emd_big_launches ab_emd_modes ab_group_grad ab_c3 emd_sizes; do grep -v amdgpu.ids $O/$f.txt > $P/r06_$f.txt; done
echo "copied; soaks and the fuzz run: $O/soak_*.txt, $O/fuzz20.txt, $O/pytest_gpu.txt -> profiles/r06_soak.txt by hand"
