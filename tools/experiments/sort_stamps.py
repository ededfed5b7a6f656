"""Phase stamps (s_memtime / clock64) of the register-resident sort's last workgroup at C2."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
a = torch.from_numpy(rng.randn(32, 2048, 3).astype(np.float32)).cuda()
c = torch.from_numpy(rng.randn(32, 16384, 3).astype(np.float32)).cuda()
names = ["start", "loaded+zeroed", "bbox+tables", "quantiles", "keys+hist", "scan", "positions", "half0 done", "half1 done"]
acc = []
for _ in range(5):
    st = []
    R.nn_distance(a, c, mode="culled", stats=st)
    s = [x for x in st[16:32] if x]
    acc.append(s)
s = np.median(np.array([x[:9] for x in acc if len(x) >= 9], dtype=np.float64), 0)
for i in range(1, len(s)):
    print(f"{names[i]:16s} {int(s[i] - s[i - 1]):8d} ticks")
print("total", int(s[-1] - s[0]), "ticks (100 MHz clock64? or shader clock -- compare with 29 us)")
