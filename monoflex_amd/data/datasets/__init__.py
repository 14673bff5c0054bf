from .kitti import KITTIDataset                            # noqa: F401

__all__ = ["KITTIDataset"]
