"""Reference import path `pc_distance.tf_nndistance` (vv_recon.py:8-20): re-exports the MI355X ops of
rfnet_amd.pc_distance.tf_nndistance so reference-style callers run unchanged."""
from rfnet_amd.pc_distance.tf_nndistance import *  # noqa: F401,F403
from rfnet_amd.pc_distance.tf_nndistance import __doc__ as _impl_doc  # noqa: F401
