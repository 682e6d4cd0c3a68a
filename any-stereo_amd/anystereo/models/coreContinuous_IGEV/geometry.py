"""Import-path mirror of the reference's geometry.py (implementation: anystereo/nn/geometry.py)."""
from ...nn.geometry import Combined_Geo_Encoding_Volume, CorrBlock1D  # noqa: F401
