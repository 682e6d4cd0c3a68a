"""Import-path mirror of the reference's liif.py (implementation: anystereo/nn/liif.py)."""
from ...nn.liif import (MLP, AffinityFeature, PositionEncoder, SpatialEncoding, StructureFeature, convbn,  # noqa: F401
                        liif_feat_multiscale_train, liif_feat_multiscale_train_quater, liif_out_multi_scale_Training, make_coord)
