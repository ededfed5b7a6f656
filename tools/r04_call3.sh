#!/usr/bin/env bash
# round 4, third GPU call: first run of the quad-per-query tiles (RFP_TILE16): parity tests, stats, A/B against the shared-group build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c; mkdir -p "$O"
cd "$R"
( timeout 900 python3 -m pytest tests/test_gpu_chamfer_culled.py tests/test_gpu_chamfer_step_sorted.py tests/test_gpu_chamfer.py -x -q ) > "$O/pytest_chamfer.txt" 2>&1
tail -15 "$O/pytest_chamfer.txt"
timeout 300 python3 tools/culled_stats.py > "$O/culled_stats.txt" 2>&1; cat "$O/culled_stats.txt"
RFOPS_LIB=rfnet_amd/variants/librfops_shared4.so timeout 300 python3 tools/culled_stats.py > "$O/culled_stats_shared4.txt" 2>&1; cat "$O/culled_stats_shared4.txt"
timeout 600 python3 tools/ab_step.py base shared4 > "$O/ab_step.txt" 2>&1; cat "$O/ab_step.txt"
python3 - <<'PY' > "$O/collapsed_dump.txt" 2>&1
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from rfnet_amd.rfnet import RFNet
rng = np.random.RandomState(300); torch.manual_seed(0)
net = RFNet().cuda()
partial = torch.from_numpy((rng.rand(32, 3000, 3) - 0.5).astype(np.float32)).cuda()
with torch.no_grad():
    out = net(partial)[3].contiguous()
np.save("gpurun_out/r04c/collapsed_cloud0.npy", out[0].cpu().numpy())
print("saved", out.shape)
PY
