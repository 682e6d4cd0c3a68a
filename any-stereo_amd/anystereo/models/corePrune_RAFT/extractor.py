"""Import-path mirror of the reference's extractor.py (implementation: anystereo/nn/encoders.py)."""
from ...nn.encoders import BasicEncoder, Feature, MultiBasicEncoder, ResidualBlock  # noqa: F401
