import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from rfnet_amd import _raw as R
from oracle.oracle import Oracle
orc = Oracle()
ns = 8
rng = np.random.RandomState(ns)
pts = rng.rand(4, 3000, 3).astype(np.float32)
q = pts[:, :90].copy()
pts[0, 3, 1] = np.nan
pts[1, 2650] = np.nan
pts[2, 100, 0] = np.inf
pts[2, 200, 2] = -np.inf
q[0, 7, 0] = np.nan
q[2, 5, 0] = np.inf
q[3, 11, 2] = np.nan
q[3, 12, 1] = -np.inf
oi, oc = orc.query_ball_point(np.float32(0.15), ns, pts, q)
cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
gi, gc = R.query_ball_point(0.15, ns, cu(pts), cu(q), form="boxes")
gi, gc = gi.cpu().numpy(), gc.cpu().numpy()
bad = np.argwhere((gc != oc) | (gi != oi).any(-1))
print("mismatching (cloud, query):", bad[:20].tolist(), len(bad))
for b_, q_ in bad[:5]:
    print(b_, q_, "oracle", oc[b_, q_], oi[b_, q_], "got", gc[b_, q_], gi[b_, q_])
h = R.nn_sort(cu(pts))
buf = h.buf
print(type(buf), buf.dtype, buf.numel())
from rfnet_amd._lib import lib
nbytes = lib.rf_nn_sort_bytes(4, 3000)
raw = buf.view(torch.uint8).cpu().numpy()[:nbytes]
tail = np.frombuffer(raw[-512:].tobytes(), np.int32)
print("tail ints:", tail[:64])
tail = np.frombuffer(raw[-256:].tobytes(), np.int32)
print("pos0|crowded|nonfinite:", tail[:12])
