"""Reference import path `tf_ops.interpolation.tf_interpolate` (vv_recon.py:8-20): re-exports the MI355X ops of
rfnet_amd.tf_ops.interpolation.tf_interpolate so reference-style callers run unchanged."""
from rfnet_amd.tf_ops.interpolation.tf_interpolate import *  # noqa: F401,F403
from rfnet_amd.tf_ops.interpolation.tf_interpolate import __doc__ as _impl_doc  # noqa: F401
