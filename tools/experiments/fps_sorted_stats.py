"""Touched waves per iteration of fps_sorted_kernel at C3 (a build of sampling.hip with tools/experiments/fps_sorted_instrumented_builds.patch.txt applied and -DRF_FPS_STATS: python tools/build_variant.py fpsstats -DRF_FPS_STATS;
run with RFOPS_LIB=rfnet_amd/variants/librfops_fpsstats.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from rfnet_amd import _raw as R
rng = np.random.RandomState(100)
x = torch.from_numpy(rng.random_sample((32, 16384, 3)).astype(np.float32)).cuda()
dll = ctypes.CDLL(os.environ["RFOPS_LIB"])
buf = (ctypes.c_ulonglong * 4096)()
dll.rf_fps_stats_read(buf)  # clear
R.farthest_point_sample_sorted(1024, x); torch.cuda.synchronize()
dll.rf_fps_stats_read(buf)
t = np.array(buf[:1024], dtype=np.float64) / 32.0
print("touched waves of 16 per iteration, mean over 32 clouds: overall %.2f" % t[1:].mean())
for a, b in ((1, 8), (8, 32), (32, 128), (128, 512), (512, 1024)):
    print("  iterations %4d..%4d: %.2f" % (a, b, t[a:b].mean()))
