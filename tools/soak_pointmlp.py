"""Random-shape soak of the model-graph helper kernels (pointmlp.hip) against tensor ops:
rf_point_affine, rf_maxpool_points(+_idx), rf_act_grad_colsum.  argv: seconds"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _raw

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(7)
g = torch.Generator(device="cuda").manual_seed(7)
t0, cases, bad = time.time(), 0, 0
while time.time() - t0 < secs:
    b = int(rng.randint(1, 6))
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 3000, 4024, 16384, int(rng.randint(1, 20000))]))
    c = 4 * int(rng.choice([1, 2, 16, 27, 32, 64, 96, 128, 256, int(rng.randint(1, 257))]))
    x = torch.randn(b, n, c, device="cuda", generator=g)
    ok = True
    # pooling
    ok &= bool(torch.equal(_raw.maxpool_points(x), x.amax(1, keepdim=True)))
    v, i = _raw.maxpool_points_idx(x)
    ok &= bool(torch.equal(v, x.amax(1, keepdim=True)))
    ok &= bool(torch.equal(torch.gather(x, 1, i.long().unsqueeze(1)), v))
    ok &= bool(torch.equal(i.long(), (x == v).float().argmax(1)))
    # activation gradient + column sums
    out = torch.randn(b, n, c, device="cuda", generator=g)
    for act, ref in (("relu", torch.where(out > 0, x, torch.zeros_like(x))), ("leaky_relu", torch.where(out > 0, x, x * 0.2)),
                     ("tanh", x * (1.0 - out * out)), (None, x)):
        gg, sums = _raw.act_grad_colsum(x, out, act)
        ok &= bool(torch.allclose(gg, ref, rtol=1e-6, atol=1e-7))
        exp = ref.double().sum(1)
        ok &= bool(torch.allclose(sums.double(), exp, rtol=1e-5, atol=1e-5 * float(exp.abs().max()) + 1e-6))
    # fused layer tail
    kp = int(rng.choice([0, 3, 16]))
    p = torch.randn(b, n, kp, device="cuda", generator=g) if kp else None
    w = torch.randn(kp, c, device="cuda", generator=g) if kp else None
    r = torch.randn(b, 1, c, device="cuda", generator=g)
    for act in ("relu", "tanh", None):
        got = _raw.point_affine(x, p, w, r, act)
        ref = x + r + (p @ w if kp else 0)
        ref = torch.relu(ref) if act == "relu" else torch.tanh(ref) if act == "tanh" else ref
        ok &= bool(torch.allclose(got, ref, rtol=1e-5, atol=1e-5))
    cases += 1
    if not ok:
        bad += 1
        print("MISMATCH", b, n, c, kp, flush=True)
print(f"{cases} shapes, {bad} mismatches, {time.time() - t0:.0f} s")
