"""Reference import path; the implementation lives in rfnet_amd/pc_distance."""
