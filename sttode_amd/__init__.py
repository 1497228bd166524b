"""sttode_amd: MI355X-native (gfx950) implementation of the STTODE forward trajectory-forecasting hot path."""
from . import capi, packing, scenes, weights  # noqa: F401
from .model import STTODENet  # noqa: F401
from .sampler import Sampler  # noqa: F401
