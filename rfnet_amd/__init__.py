"""rfnet_amd -- MI355X (gfx950) implementation of RFNet's point-cloud operator hot path.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/rfops.h) -> librfops.so
  _lib.py          ctypes binding of the C ABI (fails loudly if the .so is missing)
  _raw.py          one call per reference OpKernel: validation, allocation, launch
  tf_ops/, pc_distance/   mirrors of the reference's Python modules (same names/signatures),
                   with torch.autograd.Function in place of @RegisterGradient
  shard.py         batch sharding over the GPUs of a node (one process per GPU, RCCL)
The reference's import paths (`tf_ops.CD.tf_nndistance`, `pc_distance.tf_approxmatch`, ...)
exist at the repository root and re-export these modules.
"""
__version__ = "0.1.0"


def enable_graph_safe_runtime():
    """Opt-in, for hosts that capture HIP graphs holding torch reductions (`sum`, `max`, `amax`, ...).

    ROCm 7's HIP-graph "packet capture" replays a captured hipMemsetAsync of a small buffer with garbage
    from the second replay on (tools/experiments/graph_memset_probe.py).  librfops.so never issues a memset
    (rf::zero_async is a kernel), but torch's reduction kernels clear their semaphores that way, so such a
    graph goes stale.  The runtime switch DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 costs nothing in replay time
    (measured: 8.51 vs 8.52 ms per C5 step) but only counts if it is set BEFORE the HIP runtime starts
    (`torch.cuda.is_available()` starts it).  This function sets it (unless the host already chose a value)
    and returns a BEST-EFFORT hint: True when torch has not initialised CUDA/HIP yet.  The hint can be falsely
    True -- `torch.cuda.is_available()` / `device_count()` may already have started the runtime without torch
    counting as initialised -- so never branch on it: `rfnet_amd._host.graph_replay_ok()` asks the running
    runtime itself and is the authority (TrainStep / GraphedForward consult it and stay eager when it says no).

    `import rfnet_amd` does NOT call this: the binding leaves os.environ untouched.  The package's own
    entry points that capture such graphs (`python -m rfnet_amd.trainrun`, `python -m rfnet_amd.evalrun`,
    bench.py) call it first thing.
    """
    import os
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
    try:
        import torch
        return not torch.cuda.is_initialized()
    except ImportError:
        return True
