"""Reference import path `tf_ops.CD.tf_nndistance` (vv_recon.py:8-20): re-exports the MI355X ops of
rfnet_amd.tf_ops.CD.tf_nndistance so reference-style callers run unchanged."""
from rfnet_amd.tf_ops.CD.tf_nndistance import *  # noqa: F401,F403
from rfnet_amd.tf_ops.CD.tf_nndistance import __doc__ as _impl_doc  # noqa: F401
