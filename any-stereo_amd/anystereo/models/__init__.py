"""Model registry with the reference's keys (models/__init__.py:1-6)."""
from .coreContinuous_IGEV.continuous_IGEVstereo import continuous_IGEVStereo
from .corePrune_RAFT.prune_raft_stereo import continuous_RaftStereo
from .base import default_args

__models__ = {
    "continuous_IGEVStereo": continuous_IGEVStereo,
    "continuous_RAFTStereo": continuous_RaftStereo,
}
