"""Reference import path `tf_ops.grouping.tf_grouping` (vv_recon.py:8-20): re-exports the MI355X ops of
rfnet_amd.tf_ops.grouping.tf_grouping so reference-style callers run unchanged."""
from rfnet_amd.tf_ops.grouping.tf_grouping import *  # noqa: F401,F403
from rfnet_amd.tf_ops.grouping.tf_grouping import __doc__ as _impl_doc  # noqa: F401
