"""emap_amd - MI355X-native render hot path of EMAP (cvg/EMAP) behind the reference's class API.

Public names mirror reference ``src/models``: UDFNetwork, SingleVarianceNetwork, BetaNetwork,
RenderingNetwork, UDFRendererBlending, sample_pdf, get_embedder, EdgeLoss.
"""
from .embedder import get_embedder, Embedder  # noqa: F401
from .loss import EdgeLoss  # noqa: F401
from .udf_model import UDFNetwork, SingleVarianceNetwork, BetaNetwork, RenderingNetwork  # noqa: F401
from .udf_renderer_blending import UDFRendererBlending, sample_pdf  # noqa: F401
from .ray_sampler import DeviceRaySampler  # noqa: F401

__all__ = ["UDFNetwork", "SingleVarianceNetwork", "BetaNetwork", "RenderingNetwork", "UDFRendererBlending",
           "sample_pdf", "get_embedder", "Embedder", "EdgeLoss", "DeviceRaySampler"]
