"""Drop-in for the reference's native extension module `corr_sampler`
(sampler/sampler.cpp:48-51: `forward(volume, coords, radius) -> [corr]`,
`backward(volume, coords, corr_grad, radius) -> [volume_grad]`), backed by
as_corr_sampler_fwd / as_corr_sampler_bwd.  Error behaviour follows CHECK_INPUT
(sampler.cpp:20-22): non-CUDA or non-contiguous inputs raise RuntimeError."""
from . import ops


def forward(volume, coords, radius):
    return [ops.corr_sampler_forward(volume, coords, int(radius))]


def backward(volume, coords, corr_grad, radius):
    return [ops.corr_sampler_backward(volume, coords, corr_grad, int(radius))]
