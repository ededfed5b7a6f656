#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; O=gpurun_out/r04z; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_all.txt; cat $O/pytest_all.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python3 bench.py --steps 100 --warmup 10 ) > $O/bench_last.json 2> $O/bench_last.err; tail -3 $O/bench_last.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04z/bench_last.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['emd']['value'], d['emd']['ms_per_call'], d['per_op_roofline']['match_cost_grad']['frac'])
PY
