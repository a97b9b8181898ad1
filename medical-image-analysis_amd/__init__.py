"""MI355X-native EfficientNet-B3 + CBAM hot path (package directory: medical-image-analysis_amd;
import name: mia_amd, see ../mia_amd.py)."""
from ._lib import MiaError, lib  # noqa: F401

__all__ = ["lib", "MiaError"]
