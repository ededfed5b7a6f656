R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_emd_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/stats" -- python3 "$R/tools/run_emd_once.py" $1 > /dev/null 2> "$OUT/err.txt"
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/stats/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last call's sequence of kernels
seq=[(r["Kernel_Name"],int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in rows]
import re
def short(n):
    m=re.search(r"(am_row[kl]_kernel<[^>]*>|am_match_kernel<[^>]*>|fgt_\w+|nnp_sort\w*|mc_\w+|am_init\w*|zero_kernel|am_\w+)",n)
    return m.group(1) if m else n[:40]
# take the last 40 kernels
import sys
for n,d in seq[-(int('${2:-34}')):]:
    print(f"{short(n):50s} {d/1e3:8.1f} us")
PY
