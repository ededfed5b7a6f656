"""How many 64-source groups does a destination tile of the sorted-space backward visit?  Reads the sweep's
group masks out of a ChamferStep workspace (layout: nn_pruned.hip emit_layout) at C2 and prints, per direction,
the buckets per group and the listed groups per tile."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R  # noqa: E402
from rfnet_amd._lib import lib  # noqa: E402

B, N, M = 32, 2048, 16384
kind = sys.argv[1] if len(sys.argv) > 1 else "randn"
rng = np.random.RandomState(100)
if kind == "randn":
    a, c = rng.randn(B, N, 3), rng.randn(B, M, 3)
else:
    a, c = rng.rand(B, N, 3) - 0.5, rng.rand(B, M, 3) - 0.5
a = torch.from_numpy(a.astype(np.float32)).cuda()
c = torch.from_numpy(c.astype(np.float32)).cuda()
plan = R.ChamferStep(B, N, M, "cuda")
plan(a, c, torch.ones(B, N, device="cuda"), torch.ones(B, M, device="cuda"))
torch.cuda.synchronize()
ws = plan.ws.cpu().numpy()
al = lambda v: (v + 255) // 256 * 256
npad = [2048, 16384 + 64]
off = lib.rf_nn_sort_bytes(B, N) + lib.rf_nn_sort_bytes(B, M)
masks = []
for k in range(2):
    off_rec = off
    off_own = off_rec + al(B * npad[k] * 4)
    off_mask = off_own + al(B * npad[k] * 12)
    nm = B * (npad[k] // 64)
    masks.append(ws[off_mask:off_mask + nm * 8].view(np.uint64).reshape(B, npad[k] // 64))
    rec = ws[off_rec:off_rec + B * npad[k] * 4].view(np.int32).reshape(B, npad[k])
    print(f"set {k}: valid records {int((rec >= 0).sum())} of {B * npad[k]}")
    off = off_mask + al(nm * 8)
for S in range(2):
    D = 1 - S
    mk = masks[S]
    pc = np.array([bin(int(x)).count("1") for x in mk.ravel()])
    print(f"groups of set {S} ({mk.shape[1]} per cloud): buckets of set {D} per group mean {pc.mean():.2f} max {pc.max()}")
    tiles = 32
    cnt = np.zeros((B, tiles))
    for t in range(tiles):
        tm = np.uint64(3 << (2 * t))
        cnt[:, t] = ((mk & tm) != 0).sum(1)
    print(f"  tiles of set {D} (2 buckets each): listed groups per tile mean {cnt.mean():.1f} max {cnt.max():.0f} "
          f"of {mk.shape[1]}; visits per cloud {cnt.sum(1).mean():.0f}")
