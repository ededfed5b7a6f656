"""The cases of tools/experiments/emd_cull_live_cost.py (the large-shape stream of tools/soak_emd_live.py, seed 3) whose culled cost-only
earth_mover leaves rel 1e-5 of the swept route's cost: which of the routes is nearer the ORACLE's cost there?  (One oracle chain of
4.4e7 pairs is ~10 s on a host core.)  usage: python tools/experiments/emd_cull_live_cost_oracle.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from rfnet_amd import _raw as R
from oracle.oracle import Oracle
orc = Oracle()
rng = np.random.RandomState(3)
def cloud(b, n, kind):
    if kind == 0: return rng.random_sample((b, n, 3)) - 0.5
    if kind == 1:
        s = 1.0 if rng.rand() < 0.5 else -1.0
        return np.clip(s * 0.45 + 0.03 * rng.randn(b, n, 3), -0.5, 0.5)
    if kind == 2: return np.clip(0.3 * rng.randn(b, n, 3) * rng.rand(1, 1, 3), -0.5, 0.5)
    if kind == 3: return (rng.random_sample((b, n, 3)) - 0.5) * float(np.exp(rng.uniform(np.log(0.3), np.log(3.0))))
    x = rng.random_sample((b, n, 3)) - 0.5
    x[:, n // 2:] = x[:, : n - n // 2]
    return x
want = {(8604, 6075), (8575, 4920), (7381, 5027), (7930, 5575)}
found = 0
while found < 4:
    b = rng.randint(1, 3)
    n = int(round(np.exp(rng.uniform(np.log(4096), np.log(9000)))))
    m = n if rng.rand() < 0.4 else int(round(np.exp(rng.uniform(np.log(4096), np.log(9000)))))
    ka, kc = rng.randint(0, 5), rng.randint(0, 5)
    a, c = cloud(b, n, ka).astype(np.float32), cloud(b, m, kc).astype(np.float32)
    if (n, m) not in want:
        continue
    found += 1
    ta, tc = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
    fs = R.earth_mover(ta, tc, mode="swept").cpu().numpy().astype(np.float64)
    fc = R.earth_mover(ta, tc).cpu().numpy().astype(np.float64)
    fl = R.match_cost(ta, tc, R.approx_match(ta, tc)).cpu().numpy().astype(np.float64)
    t0 = time.time()
    om = orc.approx_match(a[:1], c[:1])
    oc = float(orc.match_cost(a[:1], c[:1], om)[0])
    print(f"b={b} n={n} m={m} kinds={ka},{kc} sample 0: oracle {oc:.6f} ({time.time() - t0:.0f} s) | swept {fs[0]:.6f} ({abs(fs[0] / oc - 1):.2e}) | "
          f"culled + live (cost-only earth_mover) {fc[0]:.6f} ({abs(fc[0] / oc - 1):.2e}) | approx_match (live) + match_cost {fl[0]:.6f} ({abs(fl[0] / oc - 1):.2e})", flush=True)
