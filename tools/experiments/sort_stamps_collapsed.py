"""Phase stamps of the sort (last workgroup) and sweep counters on the UNTRAINED RFNet's own output (16384 points collapsed
onto ~120 spots per cloud: bench.py's `rfnet_untrained_output`)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R
from rfnet_amd.rfnet import RFNet
rng = np.random.RandomState(300)
torch.manual_seed(0)
net = RFNet().cuda()
partial = torch.from_numpy((rng.rand(32, 3000, 3) - 0.5).astype(np.float32)).cuda()
with torch.no_grad():
    out = net(partial)[3].contiguous()
a = partial[:, :2048].contiguous()
x = out[0].cpu().numpy()
print("distinct points in cloud 0 (exact):", len(np.unique(x, axis=0)), " at 1e-4:", len(np.unique(x.round(4), axis=0)))
names = ["start", "loads issued+zeroed", "bbox+tables", "quantiles", "keys+hist", "scan", "positions", "staging round 1", "staging round 2"]
acc = []
for _ in range(5):
    st = []
    R.nn_distance(a, out, mode="culled", stats=st)
    acc.append(st)
s = np.median(np.array([[v for v in x[16:32]] for x in acc], dtype=np.float64), 0)
s = [v for v in s if v]
for i in range(1, len(s)):
    print(f"{names[i]:22s} {int(s[i] - s[i - 1]):8d} ticks")
print("total", int(s[-1] - s[0]))
st = acc[-1]
print("dir0: waves %d steps %d (max %d) scans %d (max %d) | dir1: waves %d steps %d (max %d) scans %d (max %d)" % (st[0], st[1], st[2], st[3], st[8], st[4], st[5], st[6], st[7], st[9]))
