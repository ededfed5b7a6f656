"""approx_match + match_cost at C4 on the two routes of rf_approxmatch_mode (auto / swept), hipEvent-timed in one process,
with the per-kernel split of the default route.  usage: python tools/ab_emd_modes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rfnet_amd import _raw as R, _lib

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

rng = np.random.RandomState(100)
a = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
c = torch.from_numpy((rng.random_sample((32, 2048, 3)) - 0.5).astype(np.float32)).cuda()
for mode in ("auto", "swept"):
    ms = timed(lambda: R.match_cost(a, c, R.approx_match(a, c, mode=mode)))
    mf = timed(lambda: R.earth_mover(a, c, mode=mode))
    mg = timed(lambda: R.earth_mover(a, c, with_grad=True, mode=mode))
    print(f"C4 approx_match+match_cost {mode:9s} {ms:.4f} ms/call   fused earth_mover {mf:.4f} ms/call   with gradients {mg:.4f} ms/call")
_lib.profile_collect(); _lib.profile_enable(True)
for _ in range(10): R.approx_match(a, c)
torch.cuda.synchronize(); _lib.profile_enable(False)
for k, v in _lib.profile_collect().items(): print(f"   {k:12s} {v[0] / 10 * 1e3:8.1f} us per call ({v[1] // 10} launches)")
# per launch, in order (one call with the profiler's per-launch list is not exposed: the sums above are per kernel name)
