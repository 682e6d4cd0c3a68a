"""Top-level `corr_sampler`, the module name the reference's native extension installs under (sampler/setup.py, bound in
sampler/sampler.cpp:48-51): with `any-stereo_amd` on `sys.path`, `import corr_sampler; corr_sampler.forward(volume, coords, r)`
resolves here exactly as it resolves to the compiled extension in the reference.  Same two functions, same list returns, same
CHECK_INPUT errors; the work is done by libanystereo_hip.so (as_corr_sampler_fwd / as_corr_sampler_bwd)."""
from anystereo.corr_sampler import backward, forward  # noqa: F401

__all__ = ["forward", "backward"]
