# replays tools/soak_emd_live.py's large cases (same rng stream) and prints those whose culled cost-only earth_mover leaves rel 1e-5 of the swept route's
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from rfnet_amd import _raw as R
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
t0 = time.time(); cases = 0; worst = 0.0; bad = 0
while time.time() - t0 < float(sys.argv[1]):
    b = rng.randint(1, 3)
    n = int(round(np.exp(rng.uniform(np.log(4096), np.log(9000)))))
    m = n if rng.rand() < 0.4 else int(round(np.exp(rng.uniform(np.log(4096), np.log(9000)))))
    ka, kc = rng.randint(0, 5), rng.randint(0, 5)
    a = torch.from_numpy(cloud(b, n, ka).astype(np.float32)).cuda(); c = torch.from_numpy(cloud(b, m, kc).astype(np.float32)).cuda()
    fs = R.earth_mover(a, c, mode="swept"); fc = R.earth_mover(a, c)
    r = float(((fc - fs).abs() / fs.abs().clamp_min(1e-30)).max())
    cases += 1; worst = max(worst, r)
    if r > 1e-5:
        bad += 1; print(f"b={b} n={n} m={m} kinds={ka},{kc}: culled cost-only vs swept {r:.2e}")
print(cases, "cases,", bad, "beyond 1e-5, worst", worst)
