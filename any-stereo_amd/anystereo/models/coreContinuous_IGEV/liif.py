"""Import-path mirror of the reference's liif.py (implementation: anystereo/nn/liif.py)."""
from ...nn.liif import (MLP, AffinityFeature, StructureFeature, liif_feat_multiscale_train,  # noqa: F401
                        liif_out_multi_scale_Training, make_coord)
