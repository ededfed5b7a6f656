"""ctypes binding of librfops.so (the C ABI declared in include/rfops.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing the
import-time loader raises, and if no HIP device is present every op raises at call time.
"""
import ctypes as C
import os

# PyTorch FIRST, deliberately: the torch wheel bundles its own HIP runtime (torch/lib/libamdhip64.so),
# and librfops.so merely NEEDS "libamdhip64.so.7".  Loaded after torch, the library binds to the
# runtime instance torch already brought (one runtime in the process: torch's streams and device
# pointers are the library's).  Loaded BEFORE torch it would pull in /opt/rocm's copy, torch would then
# add its own, and the process would hold two HIP runtimes that cannot share a device context (observed
# on the MI355X box: rf_device_check() = RF_ENODEVICE, or torch reporting no GPU, depending on who
# initialised first).  A host without torch gets the system runtime, alone, which is equally fine.
# (Importing this module changes nothing in the host process: no environment variable is written.  Hosts
# that capture HIP graphs holding torch reductions call rfnet_amd.enable_graph_safe_runtime() first.)
try:
    import torch  # noqa: F401  (import order is the point)
except ImportError:  # a host without PyTorch: the system HIP runtime, alone
    torch = None

_PKG = os.path.dirname(os.path.abspath(__file__))
# RFOPS_LIB: load another build of the same library (A/B of kernel variants, tools/ab_variants.py)
_VARIANT = bool(os.environ.get("RFOPS_LIB"))
LIB_PATH = os.environ.get("RFOPS_LIB") or os.path.join(_PKG, "librfops.so")

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/rfops.h declares
SIGNATURES = {
    "rf_version": (C.c_char_p, []),
    "rf_status_string": (C.c_char_p, [_i]),
    "rf_device_check": (_i, []),
    "rf_probe_memset_async": (_i, [_vp, _sz, _vp]),
    "rf_probe_exp2": (_i, [_vp, _vp, _i, _vp]),
    "rf_nn_distance_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_nn_distance": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_nn_distance_mode_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rf_nn_distance_mode": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp]),
    "rf_nn_distance_grad": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rf_nn_distance_dir_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "rf_nn_distance_dir": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _i]),
    "rf_nn_sort_bytes": (_sz, [_i, _i]),
    "rf_nn_sort": (_i, [_i, _i, _vp, _vp, _sz, _vp]),
    "rf_nn_distance_sorted": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rf_chamfer_step_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_chamfer_step": (_i, [_i, _i, _i] + [_vp] * 11 + [_sz, _vp]),
    "rf_chamfer_loss_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "rf_chamfer_loss": (_i, [_i, _i, _i] + [_vp] * 10 + [_sz, _vp]),
    "rf_chamfer_loss_grad": (_i, [_i, _i, _i] + [_vp] * 10),
    "rf_merge_layer_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rf_merge_layer": (_i, [_i, _i, _i] + [_vp] * 7 + [_sz, _vp]),
    "rf_merge_layer_grad": (_i, [_i, _i, _i] + [_vp] * 9),
    "rf_approxmatch_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rf_approxmatch": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_approxmatch_levels": (_i, [_i, _i, _i, _vp, _vp, _vp, C.POINTER(_f), _i, _vp, _sz, _vp]),
    "rf_approxmatch_mode_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "rf_approxmatch_mode": (_i, [_i, _i, _i, _vp, _vp, _vp, C.POINTER(_f), _i, _vp, _sz, _vp, _i]),
    "rf_matchcost_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_matchcost": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_matchcost_grad": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rf_farthestpointsampling_temp_floats": (_sz, [_i, _i]),
    "rf_farthestpointsampling": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "rf_farthestpointsampling_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_farthestpointsampling_ws": (_i, [_i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "rf_farthestpointsampling_sorted_workspace_bytes": (_sz, [_i, _i]),
    "rf_farthestpointsampling_sorted": (_i, [_i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "rf_gatherpoint": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "rf_scatteraddpoint": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "rf_queryballpoint": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "rf_queryballpoint_dev": (_i, [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "rf_queryballpoint_boxes_workspace_bytes": (_sz, [_i, _i]),
    "rf_queryballpoint_boxes": (_i, [_i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_sample_and_group_workspace_bytes": (_sz, [_i, _i]),
    "rf_sample_and_group": (_i, [_i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "rf_grouppoint": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rf_grouppoint_grad": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rf_grouppoint_grad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "rf_grouppoint_grad_ws": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_threenn": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rf_threenn_boxes_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_threenn_boxes": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_threeinterpolate": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rf_threeinterpolate_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rf_threeinterpolate_grad_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rf_threeinterpolate_grad_ws": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_auctionmatch_supported": (_i, [_i]),
    "rf_auctionmatch_workspace_bytes": (_sz, [_i, _i]),
    "rf_auctionmatch": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_selectionsort": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rf_probsample": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rf_earth_mover_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_earth_mover": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_earth_mover_mode_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "rf_earth_mover_mode": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i]),
    "rf_maxpool_points_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_maxpool_points": (_i, [_i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "rf_maxpool_points_idx_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_maxpool_points_idx": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rf_act_grad_colsum_workspace_bytes": (_sz, [_i, _i, _i]),
    "rf_act_grad_colsum": (_i, [_i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "rf_point_affine_supported": (_i, [_i, _i]),
    "rf_point_affine": (_i, [_i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "rf_profile_enable": (None, [_i]),
    "rf_profile_collect": (_i, [C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_long), _i]),
}


class RfopsError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise RfopsError(
            f"{LIB_PATH} is missing: build it with `python -m rfnet_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        except AttributeError:
            # the product library must export everything; an A/B variant named by RFOPS_LIB (tools/ab_*.py loading
            # an OLDER build) may lack entry points added since -- calling a missing one then raises there
            if _VARIANT:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status, what):
    if status != 0:
        raise RfopsError(f"{what} failed: status {status} ({lib.rf_status_string(status).decode()})")


def profile_enable(on=True):
    lib.rf_profile_enable(1 if on else 0)


def profile_collect(cap=64):
    """-> {kernel name: (total ms, launches)} since the previous collect."""
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    cnt = (C.c_long * cap)()
    k = lib.rf_profile_collect(names, ms, cnt, cap)
    return {names[i].decode(): (ms[i], cnt[i]) for i in range(k)}
