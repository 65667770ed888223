"""emd_amd -- MI355X-native street-Gaussian rasterizer behind EMD's `GaussianRasterizer` operator surface.

Only the hot path lives here: explicit-motion transform -> projection / covariance ->
tile duplication + radix sort -> per-tile alpha compositing with SH colour, forward and backward, as
hand-written HIP for gfx950 behind the C ABI of include/emd_raster.h.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, RasterCall, RasterConfig, RasterOptions  # noqa: F401

from .graphs import StepGraphs, StepInputs  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "RasterCall", "RasterConfig", "RasterOptions", "StepGraphs", "StepInputs"]
