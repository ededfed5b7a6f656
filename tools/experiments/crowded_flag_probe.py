"""Reads the sort's per-cloud `crowded` flag (nn_pruned.hip, behind pos0 in a sorted set) for a few cloud kinds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R

def a256(v): return (v + 255) // 256 * 256
def flags(t):
    b, n = t.shape[0], t.shape[1]
    h = R.nn_sort(t)
    split = 2 if 8192 < n <= 16384 else 1
    npad = (n + 63) // 64 * 64 + (split - 1) * 64
    off = a256(b * npad * 12 + 256) + a256(b * npad * 4) + a256(b * (npad // 64) * 96) + a256(b * (npad // 64) * 32)
    torch.cuda.synchronize()
    w = h.buf.view(torch.uint8)[off: off + 8 * b].view(torch.int32).cpu().numpy()
    return w[:b], w[b:2 * b]

rng = np.random.RandomState(1)
b, k = 4, 16384
spots = rng.random_sample((b, 120, 3)) - 0.5
sid = np.where(rng.random_sample((b, k)) < 0.7, (np.arange(k) * 120 // k)[None], rng.randint(0, 120, (b, k)))
coll = (spots[np.arange(b)[:, None], sid] + 6e-6 * rng.randn(b, k, 3)).astype(np.float32)
for name, x in (("randn", rng.randn(b, k, 3).astype(np.float32)), ("collapsed (synthetic, coherent runs)", coll)):
    p0, fl = flags(torch.from_numpy(x).cuda())
    print(name, "pos0", p0, "crowded", fl)
