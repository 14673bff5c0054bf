from .dla_dcn import build_backbone
