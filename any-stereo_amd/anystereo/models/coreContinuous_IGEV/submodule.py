"""Import-path mirror of the reference's submodule.py (implementations: anystereo/nn/blocks.py,
anystereo/nn/functional.py)."""
from ...nn.blocks import *  # noqa: F401,F403
from ...nn.blocks import (BasicConv, BasicConv_IN, Conv2x, Conv2x_IN, FeatureAtt, LayerNorm2d,  # noqa: F401
                          HighRes_Aggregation, HighRes_Aggregation_LN, HighRes_Aggregation_LN_GeLU)
from ...nn.functional import (build_gwc_volume, context_upsample_multiscale_train,  # noqa: F401
                              context_upsample_multiscale_train_quaterp,
                              disparity_regression)
