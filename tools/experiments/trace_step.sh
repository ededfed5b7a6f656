# kernel-trace timeline of the headline step (sort -> sweep -> backward), gaps included.  usage (GPU box): bash tools/experiments/trace_step.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_step_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> "$OUT/err.txt"
python3 - <<PY
import csv,glob,re,statistics
f=sorted(glob.glob("$OUT/stats/**/*kernel_trace.csv",recursive=True))[-1]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    for k in ("nnp_sort","nnp_sweep","nnp_grad_sorted"):
        if k in n: return k
    return None
seq=[(short(r["Kernel_Name"]),int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows if short(r["Kernel_Name"])]
# find steady-state triples
steps=[]
for i in range(len(seq)-3):
    if [s[0] for s in seq[i:i+4]]==["nnp_sort","nnp_sweep","nnp_grad_sorted","nnp_sort"]:
        a,b,c,d=seq[i:i+4]
        steps.append((a[2]-a[1], b[1]-a[2], b[2]-b[1], c[1]-b[2], c[2]-c[1], d[1]-c[2], d[1]-a[1]))
steps=steps[10:60]
names=["sort","gap sort->sweep","sweep","gap sweep->grad","grad","gap grad->next sort","step (sort start to next sort start)"]
for i,n in enumerate(names):
    v=[s[i]/1e3 for s in steps]
    print(f"{n:40s} median {statistics.median(v):7.2f} us  min {min(v):7.2f}  max {max(v):7.2f}   ({len(v)} steps)")
PY
