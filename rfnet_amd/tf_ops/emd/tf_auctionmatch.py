"""Drop-in for the reference module tf_ops/emd/tf_auctionmatch.py (auction_match :11-20,
ops.NoGradient :21).  vv_recon.py:8 imports it; only dead code (emd_func, :365-380) calls it."""
from ... import _raw


def auction_match(xyz1, xyz2):
    '''
input:
    xyz1 : batch_size * #points * 3
    xyz2 : batch_size * #points * 3
returns:
    matchl : batch_size * #npoints   (for each xyz1 point, the xyz2 point it is matched to)
    matchr : batch_size * #npoints   (for each xyz2 point, the xyz1 point it is matched to)
    '''
    return _raw.auction_match(xyz1, xyz2)
