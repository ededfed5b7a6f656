"""Phase stamps (s_memtime / clock64) of the register-resident sort's last workgroup at C2."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
a = torch.from_numpy(rng.randn(32, 2048, 3).astype(np.float32)).cuda()
c = torch.from_numpy(rng.randn(32, 16384, 3).astype(np.float32)).cuda()
names = ["start", "loads issued+zeroed", "bbox+tables", "quantiles", "keys+hist", "scan", "positions", "staging: scatter to LDS", "staging: boxes", "staging: write-out", "round 2: scatter", "round 2: boxes", "round 2: write-out"]
acc = []
for _ in range(5):
    st = []
    R.nn_distance(a, c, mode="culled", stats=st)
    s = [x for x in st[16:32] if x]
    acc.append(s)
k = min(len(x) for x in acc)
s = np.median(np.array([x[:k] for x in acc], dtype=np.float64), 0)
for i in range(1, len(s)):
    print(f"{names[i]:26s} {int(s[i] - s[i - 1]):8d} ticks")
print("total", int(s[-1] - s[0]), "shader-clock ticks (the last workgroup of the launch: second half of the last 16384-point cloud)")
