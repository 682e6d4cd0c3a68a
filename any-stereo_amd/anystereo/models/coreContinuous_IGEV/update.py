"""Import-path mirror of models/coreContinuous_IGEV/update.py (implementation: anystereo/nn/update.py)."""
from ...nn.update import (BasicMotionEncoder, BasicMultiUpdateBlock, ConvGRU, DispHead, FlowHead,  # noqa: F401
                          interp, pool2x)
