"""Reference import path; the implementation lives in rfnet_amd/tf_ops/grouping."""
