"""fps_sorted_kernel with and without the samples' coordinates, and inside rf_sample_and_group (same cloud, kernel times)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from rfnet_amd import _lib, _raw as R
def kern(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(it): fn()
    torch.cuda.synchronize(); _lib.profile_enable(False)
    return {k: round(v[0] / it, 4) for k, v in _lib.profile_collect().items()}
rng = np.random.RandomState(100)
x = torch.from_numpy(rng.random_sample((32, 16384, 3)).astype(np.float32)).cuda()
print("idx only      ", kern(lambda: R.farthest_point_sample_sorted(1024, x)))
print("idx + new_xyz ", kern(lambda: R.farthest_point_sample_sorted(1024, x, with_xyz=True)))
print("one call      ", kern(lambda: R.sample_and_group(1024, 0.1, 32, x)))
idx, nx = R.farthest_point_sample_sorted(1024, x, with_xyz=True)
h = R.nn_sort(x)
def both():
    R.farthest_point_sample_sorted(1024, x, with_xyz=True)
    R.query_ball_point(0.1, 32, x, nx, sorted1=h.buf)
print("fps then ball ", kern(both))
def both2():
    R.farthest_point_sample_sorted(1024, x, with_xyz=True)
    R.gather_point(x, idx)
print("fps then gather", kern(both2))
print("reg fps then ball", kern(lambda: (R.farthest_point_sample_reg(1024, x), R.query_ball_point(0.1, 32, x, nx, sorted1=h.buf))))
