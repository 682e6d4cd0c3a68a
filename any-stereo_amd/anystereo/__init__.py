"""anystereo — MI355X-native (gfx950) implementation of Any-Stereo's data-parallel hot path behind
the reference's own Python operator API.  Hot-path arithmetic lives in
lib/libanystereo_hip.so (C ABI: include/anystereo_hip.h); this package is the host-side mirror."""
from . import _lib  # noqa: F401
from . import torch_ops  # noqa: F401  (registers torch.ops.anystereo.*)

__all__ = ["_lib", "torch_ops", "ops", "nn", "models", "harness", "corr_sampler"]
