"""Reference import path `tf_ops.emd.tf_auctionmatch` (vv_recon.py:8-20): re-exports the MI355X ops of
rfnet_amd.tf_ops.emd.tf_auctionmatch so reference-style callers run unchanged."""
from rfnet_amd.tf_ops.emd.tf_auctionmatch import *  # noqa: F401,F403
from rfnet_amd.tf_ops.emd.tf_auctionmatch import __doc__ as _impl_doc  # noqa: F401
