"""am_match (the materialisation of `match`, 512 MiB at C4) over workgroup shapes: variants built by tools/build_variant.py with
-DRFA_MATCH_TPB / -DRFA_MATCH_LSEG (and -DRFA_MATCH_NOCOMPUTE: the same stores of a value that costs nothing -- the kernel's store
floor).  Each variant in its own process (RFOPS_LIB); per-kernel times by the library's hipEvents.
usage: python tools/ab_am_match.py [variant.so ...]      (no arguments: the product library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from rfnet_amd import _raw as R, _lib
rng = np.random.RandomState(100)
a = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
c = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: R.match_cost(a, c, R.approx_match(a, c)))
m = R.approx_match(a, c); cost = R.match_cost(a, c, m)
_lib.profile_collect(); _lib.profile_enable(True)
for _ in range(20): R.match_cost(a, c, R.approx_match(a, c))
torch.cuda.synchronize(); _lib.profile_enable(False)
pr = _lib.profile_collect()
print("%%-44s approx_match + match_cost %%.4f ms   am_match %%6.1f us   mc_partial %%6.1f us   cost sum %%.6f" %% (
    os.path.basename(os.environ.get("RFOPS_LIB", "product")), ms, pr["am_match"][0] / 20 * 1e3, pr["mc_partial"][0] / 20 * 1e3 if "mc_partial" in pr else -1, float(cost.double().sum())))
""" % ROOT
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib: env["RFOPS_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-800:])
