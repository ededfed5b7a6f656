"""Import surface of the reference module tf_ops/emd/tf_auctionmatch.py.  vv_recon.py:8 imports
it, but only dead code (emd_func, vv_recon.py:365-380) calls auction_match; the kernel is a
'next' row."""


def auction_match(xyz1, xyz2):
    raise NotImplementedError("auction_match is a 'next' row: SURVEY.md 8(f3); the RFNet losses "
                              "use pc_distance.tf_approxmatch instead")
