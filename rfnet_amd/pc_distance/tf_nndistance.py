"""Drop-in for the reference module pc_distance/tf_nndistance.py: the same NnDistance op as
tf_ops/CD (the reference's C++ is byte-identical)."""
from ..tf_ops.CD.tf_nndistance import nn_distance, nn_distance_grad  # noqa: F401
