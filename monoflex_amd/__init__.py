"""monoflex_amd -- the MonoFlex detector hot path (DLA-34 + DCNv2 + heads + decode), MI355X-native.

    from monoflex_amd.config import get_cfg
    from monoflex_amd.model.detector import KeypointDetector
"""
__version__ = "0.1.0"
