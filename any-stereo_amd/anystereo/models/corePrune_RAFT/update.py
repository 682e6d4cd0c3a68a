"""Import-path mirror of models/corePrune_RAFT/update.py: same operators with the RAFT motion encoder
width (cor_planes = corr_levels*(2r+1), corePrune_RAFT/update.py:77)."""
from ...nn import update as _u
from ...nn.update import ConvGRU, DispHead, FlowHead, interp, pool2x  # noqa: F401


class BasicMotionEncoder(_u.BasicMotionEncoder):
    def __init__(self, args):
        super().__init__(args, geo_channels=0)


class BasicMultiUpdateBlock(_u.BasicMultiUpdateBlock):
    def __init__(self, args, hidden_dims=[]):
        super().__init__(args, hidden_dims=hidden_dims, geo_channels=0)
