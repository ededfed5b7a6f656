#!/usr/bin/env bash
# PMC passes over rf_chamfer_step at C2 (separate passes: FETCH_SIZE / WRITE_SIZE do not fit one, and gpurun
# refuses pmc combined with other trace domains).  usage (GPU box, repo root): bash tools/pmc_step.sh <tag>
set -u
TAG=${1:-pmc_step}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetchsize" -- python3 "$R/tools/run_step_once.py" > /dev/null 2> "$OUT/f.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/tools/run_step_once.py" > /dev/null 2> "$OUT/w.err"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_sq" -- python3 "$R/tools/run_step_once.py" > /dev/null 2> "$OUT/h.err"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        nm = row.get("Kernel_Name", "")
        nm = nm[nm.find("nn"):][:28] if "nn" in nm else nm[:28]
        acc[nm][row.get("Counter_Name", "")].append(float(row.get("Counter_Value", 0)))
    for k, cs in sorted(acc.items()):
        for c, vals in sorted(cs.items()):
            print(f"{k:30s} {c:16s} mean/dispatch={sum(vals) / len(vals):.6g} n={len(vals)}")
PY
