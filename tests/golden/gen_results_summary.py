#!/usr/bin/env python3
"""Generates tests/golden/results_summary.json from the reference's own evaluation artefact
(/root/reference/results/recon/results.csv, written by recon_test.py:42-44,68): row count, overall
and per-category means and the first rows.  Data only; run in the build container."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from rfnet_amd import evalio  # noqa: E402

rows = evalio.read_results_csv("/root/reference/results/recon/results.csv")
cat = evalio.per_category_means(rows)
out = {
    "rows": len(rows),
    "mean_cd": float(sum(r[1] for r in rows) / len(rows)),
    "mean_emd": float(sum(r[2] for r in rows) / len(rows)),
    "per_category": {k: [float(v[0]), float(v[1])] for k, v in sorted(cat.items())},
    "per_category_count": {k: sum(1 for r in rows if r[0].startswith(k + "/")) for k in sorted(cat)},
    "head": [list(r) for r in rows[:4]],
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "results_summary.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("rows", "mean_cd", "mean_emd")}))
