"""Re-runs ONE case of tools/soak_step.py (same seeds) and says which output differs where.  usage: soak_step_case.py CASE"""
import sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 3)[0])
from rfnet_amd import _raw
case = int(sys.argv[1])
rng = np.random.RandomState(987654 + case)
n, m = (int(v) for v in np.exp(rng.uniform(np.log(1024), np.log(18000), size=2)))
b = int(rng.choice([1, 2, 3, 5, 8, 17, 33]))
while b * n * m < (1 << 27) and b < 64:
    b += int(rng.randint(1, 8))
if b * (n + m) > 700000:
    b = max(1, 700000 // (n + m))
kind = case % 6
def cloud(k):
    if kind == 0: return rng.randn(b, k, 3)
    if kind == 1: return rng.random_sample((b, k, 3))
    if kind == 2: return rng.randint(0, 6, size=(b, k, 3)) * 0.5
    if kind == 3:
        base = rng.randn(b, max(k // 7, 1), 3)
        return np.take_along_axis(base, rng.randint(0, base.shape[1], (b, k))[..., None], 1)
    if kind == 4:
        x = rng.randn(b, k, 3); return x / np.linalg.norm(x, axis=-1, keepdims=True)
    c = rng.randn(b, 5, 3)
    return c[np.arange(b)[:, None], rng.randint(0, 5, (b, k))] + 0.02 * rng.randn(b, k, 3)
a = torch.from_numpy(cloud(n).astype(np.float32)).cuda()
c = torch.from_numpy(cloud(m).astype(np.float32)).cuda()
g1 = torch.from_numpy((rng.rand(b, n) + 0.1).astype(np.float32) * rng.choice([-1.0, 1.0], (b, n)).astype(np.float32)).cuda()
g2 = torch.from_numpy((rng.rand(b, m) + 0.1).astype(np.float32)).cuda()
print("case", case, "b n m", b, n, m, "kind", kind)
for rep in range(3):
    out = _raw.ChamferStep(b, n, m, "cuda")(a, c, g1, g2)
    ref = _raw.nn_distance(a, c, mode="dense")
    cul = _raw.nn_distance(a, c, mode="culled")
    for nm, x, y, z in zip(("dist1", "idx1", "dist2", "idx2"), ref, out[:4], cul):
        bad = (x != y).nonzero()
        bad2 = (x != z).nonzero()
        print(rep, nm, "step vs dense: mismatches", len(bad), bad[:3].tolist(), "| culled fwd vs dense:", len(bad2), bad2[:3].tolist())
        if len(bad):
            i = tuple(bad[0].tolist()); print("   dense", x[i].item(), "step", y[i].item())
    r1, r2 = _raw.nn_distance_grad(a, c, g1, ref[1], g2, ref[3])
    for nm, x, y in (("grad1", r1, out[4]), ("grad2", r2, out[5])):
        tol = 1e-5 * x.abs() + 2e-5 * float(x.abs().max())
        bad = ((x - y).abs() > tol).nonzero()
        print(rep, nm, "outside tolerance:", len(bad), bad[:3].tolist(), "max abs diff", float((x - y).abs().max()), "max |ref|", float(x.abs().max()))
