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
