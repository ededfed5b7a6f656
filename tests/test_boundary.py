"""CPU: the drop-in boundary.  The C-ABI library loads and exports every symbol that
include/rfops.h declares; the Python mirrors keep the reference's module paths, names and error
wording; the product path has no CPU fallback and never touches the oracle."""
import ast
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "rfops.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rfnet_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 27
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"librfops.so lacks {s}"
        assert s in _lib.SIGNATURES, f"ctypes binding lacks {s}"
    assert sorted(_lib.SIGNATURES) == syms  # and binds nothing undeclared
    assert b"gfx950" in _lib.lib.rf_version()
    assert _lib.lib.rf_status_string(0) == b"ok"
    assert _lib.lib.rf_status_string(-2) == b"workspace too small"


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/rfops.h is the boundary for non-C++ hosts (cgo, JNI, ctypes): it must compile as
    strict C, and a C program must link against librfops.so and call the host-only entry points."""
    import shutil
    import subprocess
    from rfnet_amd import _lib
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    src = tmp_path / "use_rfops.c"
    src.write_text(
        '#include <stdio.h>\n#include "rfops.h"\n'
        "int main(void) {\n"
        "  size_t w = rf_nn_distance_workspace_bytes(32, 2048, 16384);\n"
        "  if (w == 0 || rf_earth_mover_workspace_bytes(2, 300, 300) == 0) return 2;\n"
        "  if (rf_auctionmatch_supported(1024) != 1 || rf_auctionmatch_supported(1500) != 0) return 3;\n"
        '  printf("%s|%s\\n", rf_version(), rf_status_string(RF_EWORKSPACE));\n'
        "  return rf_nn_distance(-1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0) == RF_EINVAL ? 0 : 4;\n"
        "}\n")
    exe = tmp_path / "use_rfops"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe), "-L", libdir, "-lrfops", f"-Wl,-rpath,{libdir}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "gfx950" in r.stdout and "workspace too small" in r.stdout


def test_workspace_queries_are_pure_host_functions():
    from rfnet_amd._lib import lib
    assert lib.rf_nn_distance_workspace_bytes(0, 10, 10) == 0
    # C2: 2048 queries need candidate splits -> partials; (dist,idx) pairs of 8 bytes
    w = lib.rf_nn_distance_workspace_bytes(32, 2048, 16384)
    assert w > 0 and w % 8 == 0
    assert lib.rf_approxmatch_workspace_bytes(32, 2048, 2048, 0) >= 32 * 4096 * 11 * 4
    assert lib.rf_approxmatch_workspace_bytes(1, 10, 20, 50) >= 30 * 51 * 4
    # round 6: the route pin and the cost-only form's column records (16 floats per column, padded per class) + class tables
    for fn in (lib.rf_approxmatch_mode_workspace_bytes, lib.rf_earth_mover_mode_workspace_bytes,
               lib.rf_grouppoint_grad_workspace_bytes, lib.rf_threeinterpolate_grad_workspace_bytes):
        fn.restype = ctypes.c_size_t
    amw = lambda b, n, m, lv, mode: lib.rf_approxmatch_mode_workspace_bytes(b, n, m, lv, mode)
    emw = lambda b, n, m, mode: lib.rf_earth_mover_mode_workspace_bytes(b, n, m, mode)
    assert amw(32, 2048, 2048, 0, 0) == lib.rf_approxmatch_workspace_bytes(32, 2048, 2048, 0)
    assert 0 < amw(32, 2048, 2048, 0, 1) <= amw(32, 2048, 2048, 0, 0)          # the swept route needs no sorted sets, lists or live sets
    assert amw(32, 2048, 2048, 0, 2) == 0 and emw(32, 2048, 2048, 7) == 0      # unknown modes
    for b, n, m in ((32, 2048, 2048), (4, 16384, 16384), (3, 700, 5000)):
        e, a = emw(b, n, m, 0), amw(b, n, m, 10, 0)
        assert e % 4 == 0 and e >= a + b * m * 16 * 4, (b, n, m, e, a)        # the levels' workspace + one 64-byte record per column
        assert e < a + b * (m + 256) * 16 * 4 + b * ((n + 255) // 256) * 64 * 4 + (1 << 20)
    assert emw(2, 64, 64, 0) > 0                                               # small clouds: match in the workspace
    assert lib.rf_grouppoint_grad_workspace_bytes(32, 16384, 64, 1024, 32) > 0
    assert lib.rf_grouppoint_grad_workspace_bytes(2, 16384, 16, 1024, 32) == 0   # below the threshold: the atomics, no workspace
    assert lib.rf_threeinterpolate_grad_workspace_bytes(32, 16384, 64, 4096) > 0
    assert lib.rf_farthestpointsampling_temp_floats(32, 16384) == 0
    assert lib.rf_farthestpointsampling_temp_floats(2, 20000) == 40000
    # round-2 entry points: sizes are pure functions of the shape (no device, no state)
    sb = lib.rf_nn_sort_bytes(32, 16384)
    assert sb >= 32 * 16384 * 16 and lib.rf_nn_sort_bytes(32, 16384) == sb       # records + indices (+ boxes)
    assert lib.rf_nn_sort_bytes(1, 65536) > 0 and lib.rf_nn_sort_bytes(1, 65537) == 0 and lib.rf_nn_sort_bytes(0, 10) == 0
    both = lib.rf_nn_distance_dir_workspace_bytes(32, 2048, 16384, 1, 1)
    assert both > 0 and lib.rf_nn_distance_dir_workspace_bytes(32, 2048, 16384, 0, 0) == 0
    # the one-call step on the culled path also holds the sweep's winner position + own gradient term (16 B per point) and bucket masks
    assert (lib.rf_chamfer_step_workspace_bytes(32, 2048, 16384)
            >= lib.rf_nn_distance_workspace_bytes(32, 2048, 16384) + 32 * (2048 + 16384) * 8 - 4096)
    assert lib.rf_chamfer_step_workspace_bytes(2, 300, 700) == lib.rf_nn_distance_workspace_bytes(2, 300, 700)
    # with both clouds pre-sorted the fused loss needs no scratch; with one, only the other's sort
    assert lib.rf_chamfer_loss_workspace_bytes(32, 16384, 16384, 1, 1, 1, 1) == 0
    one = lib.rf_chamfer_loss_workspace_bytes(32, 16384, 16384, 1, 1, 1, 0)
    assert 0 < one < lib.rf_chamfer_loss_workspace_bytes(32, 16384, 16384, 1, 1, 0, 0)
    assert lib.rf_merge_layer_workspace_bytes(32, 3000, 16384, 1) >= 32 * 16384 * 4
    assert lib.rf_auctionmatch_workspace_bytes(32, 4096) == 0  # no (n, n) cost matrix any more
    # the boxed three_nn sorts both sets into its scratch: two sorted sets; nothing outside 1..65536 points
    assert lib.rf_threenn_boxes_workspace_bytes(32, 16384, 1024) == lib.rf_nn_sort_bytes(32, 16384) + lib.rf_nn_sort_bytes(32, 1024)
    assert lib.rf_threenn_boxes_workspace_bytes(2, 100, 0) == 0 and lib.rf_threenn_boxes_workspace_bytes(2, 65537, 10) == 0
    assert lib.rf_point_affine_supported(128, 3) == 1 and lib.rf_point_affine_supported(126, 3) == 0
    assert lib.rf_device_check() in (0, -3)  # RF_OK on the MI355X box, RF_ENODEVICE here


def test_misaligned_workspace_is_an_argument_error():
    """include/rfops.h: workspaces and sorted-set handles must be 16-byte aligned (the kernels read them with 16-byte vector
    loads); a misaligned one is RF_EINVAL at the boundary -- checked before anything touches a device or the pointers."""
    from rfnet_amd._lib import lib
    p = 0x10000  # never dereferenced: every call below must return at its argument checks
    ws = 0x200000
    big = 1 << 32
    b, n, m = 32, 2048, 16384  # a culled shape
    assert lib.rf_chamfer_step(b, n, m, p, p, p, p, p, p, p, p, p, p, ws + 4, big, None) == -1
    assert lib.rf_nn_distance(b, n, m, p, p, p, p, p, p, ws + 8, big, None) == -1
    assert lib.rf_nn_sort(1, 4096, p, ws + 4, big, None) == -1
    assert lib.rf_nn_distance_sorted(1, 4096, 4096, ws + 4, ws, p, p, p, p, None) == -1
    assert lib.rf_approxmatch(1, 300, 300, p, p, p, ws + 12, big, None) == -1
    assert lib.rf_threenn_boxes(2, 500, 300, p, p, None, None, p, p, ws + 4, big, None) == -1
    assert lib.rf_threenn_boxes(2, 500, 300, p, p, ws + 8, None, p, p, ws, big, None) == -1
    assert lib.rf_earth_mover(1, 300, 300, p, p, p, None, None, ws + 4, big, None) == -1


def test_reference_module_paths_and_names():
    """vv_recon.py:8-20 imports these modules and calls these names."""
    import pc_distance.tf_approxmatch as am
    import pc_distance.tf_nndistance as nd2
    import tf_ops.CD.tf_nndistance as nd
    import tf_ops.emd.tf_auctionmatch as au
    import tf_ops.grouping.tf_grouping as gr
    import tf_ops.interpolation.tf_interpolate as ip
    import tf_ops.sampling.tf_sampling as sa
    for mod, names in ((nd, ["nn_distance"]), (nd2, ["nn_distance"]),
                       (am, ["approx_match", "match_cost"]),
                       (sa, ["farthest_point_sample", "gather_point"]),
                       (gr, ["query_ball_point", "group_point", "knn_point", "select_top_k"]),
                       (ip, ["three_nn", "three_interpolate"]), (au, ["auction_match"])):
        for n in names:
            assert callable(getattr(mod, n)), (mod.__name__, n)
    import inspect
    assert list(inspect.signature(sa.farthest_point_sample).parameters) == ["npoint", "inp"]
    assert list(inspect.signature(gr.query_ball_point).parameters) == ["radius", "nsample", "xyz1", "xyz2"]
    assert list(inspect.signature(am.match_cost).parameters) == ["xyz1", "xyz2", "match"]
    assert list(inspect.signature(ip.three_interpolate).parameters) == ["points", "idx", "weight"]


@pytest.mark.parametrize("call,msg", [
    (lambda: __import__("tf_ops.CD.tf_nndistance", fromlist=["x"]).nn_distance(
        np.zeros((2, 4), np.float32), np.zeros((2, 4, 3), np.float32)),
     "NnDistance requires xyz1 be of shape (batch,#points,3)"),
    (lambda: __import__("tf_ops.CD.tf_nndistance", fromlist=["x"]).nn_distance(
        np.zeros((2, 4, 3), np.float32), np.zeros((3, 4, 3), np.float32)),
     "NnDistance expects xyz1 and xyz2 have same batch size"),
    (lambda: __import__("pc_distance.tf_approxmatch", fromlist=["x"]).approx_match(
        np.zeros((2, 4, 2), np.float32), np.zeros((2, 4, 3), np.float32)),
     "ApproxMatch expects (batch_size,num_points,3) xyz1 shape"),
    (lambda: __import__("pc_distance.tf_approxmatch", fromlist=["x"]).match_cost(
        np.zeros((2, 4, 3), np.float32), np.zeros((2, 5, 3), np.float32), np.zeros((2, 4, 5), np.float32)),
     "MatchCost expects (batch_size,#query,#dataset) match shape"),
    (lambda: __import__("tf_ops.sampling.tf_sampling", fromlist=["x"]).farthest_point_sample(
        0, np.zeros((2, 4, 3), np.float32)), "FarthestPointSample expects positive npoint"),
    (lambda: __import__("tf_ops.sampling.tf_sampling", fromlist=["x"]).gather_point(
        np.zeros((2, 4, 3), np.float32), np.zeros((3, 4), np.int32)),
     "GatherPoint expects (batch_size,num_result) idx shape"),
    (lambda: __import__("tf_ops.grouping.tf_grouping", fromlist=["x"]).query_ball_point(
        0.1, 0, np.zeros((2, 4, 3), np.float32), np.zeros((2, 4, 3), np.float32)),
     "QueryBallPoint expects positive nsample"),
    (lambda: __import__("tf_ops.grouping.tf_grouping", fromlist=["x"]).group_point(
        np.zeros((2, 4), np.float32), np.zeros((2, 4, 1), np.int32)),
     "GroupPoint expects (batch_size, num_points, channel) points shape"),
    (lambda: __import__("tf_ops.interpolation.tf_interpolate", fromlist=["x"]).three_nn(
        np.zeros((2, 4, 3), np.float32), np.zeros((2, 4, 4), np.float32)),
     "ThreeNN expects (b,m,3) xyz2 shape."),
    (lambda: __import__("tf_ops.interpolation.tf_interpolate", fromlist=["x"]).three_interpolate(
        np.zeros((2, 4, 3), np.float32), np.zeros((2, 6, 2), np.int32), np.zeros((2, 6, 3), np.float32)),
     "ThreeInterpolate expects (b,n,3) idx shape"),
])
def test_argument_errors_use_the_reference_wording(call, msg):
    """OP_REQUIRES(... errors::InvalidArgument(msg)) in the reference OpKernels -> ValueError(msg),
    raised before anything touches the GPU."""
    with pytest.raises(ValueError) as e:
        call()
    assert str(e.value) == msg


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from rfnet_amd._lib import RfopsError
    from tf_ops.CD.tf_nndistance import nn_distance
    with pytest.raises(RfopsError, match="no CPU fallback"):
        nn_distance(np.zeros((1, 4, 3), np.float32), np.zeros((1, 4, 3), np.float32))


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under rfnet_amd/, tf_ops/, pc_distance/ may import,
    link or open it."""
    offenders = []
    for top in ("rfnet_amd", "tf_ops", "pc_distance"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                p = os.path.join(dirpath, f)
                if f.endswith(".py"):
                    tree = ast.parse(open(p).read())
                    for node in ast.walk(tree):
                        names = []
                        if isinstance(node, ast.Import):
                            names = [a.name for a in node.names]
                        elif isinstance(node, ast.ImportFrom):
                            names = [node.module or ""]
                        if any(n == "oracle" or n.startswith("oracle.") for n in names):
                            offenders.append(p)
                    if "liboracle" in open(p).read() or "libref" in open(p).read():
                        offenders.append(p)
                elif f.endswith((".hip", ".hpp", ".h", ".cpp")):
                    if re.search(r'#include\s+"[^"]*oracle', open(p).read()):
                        offenders.append(p)
    assert not offenders, offenders


def test_knn_point_is_pure_tensor_ops():
    """knn_point is pure TF ops in the reference (tf_grouping.py:64-73): val = top_k(-dist)."""
    import torch
    from tf_ops.grouping.tf_grouping import knn_point
    rng = np.random.RandomState(0)
    xyz1 = rng.rand(2, 50, 3).astype(np.float32)
    xyz2 = rng.rand(2, 7, 3).astype(np.float32)
    val, idx = knn_point(4, torch.from_numpy(xyz1), torch.from_numpy(xyz2))
    d = ((xyz1[:, None] - xyz2[:, :, None]) ** 2).sum(-1)
    assert np.array_equal(idx.numpy(), np.argsort(d, -1, kind="stable")[..., :4])
    assert np.allclose(val.numpy(), -np.sort(d, -1)[..., :4], atol=1e-6)


def test_one_hip_runtime_in_the_process():
    """rfnet_amd._lib must bind to the HIP runtime torch brings (import torch before dlopen): with
    librfops.so loaded first the process ends up with two runtimes -- /opt/rocm's and the wheel's --
    and on the GPU box whichever initialises second sees no device."""
    import subprocess
    code = ("import rfnet_amd._lib, sys\n"
            "assert 'torch' in sys.modules\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(libs)\n"
            "assert len(libs) == 1, libs\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr[-1500:]
