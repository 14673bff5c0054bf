"""Input pipeline (reference data/): KITTI files -> raw samples on the host, everything numeric on the device."""
from .collate_batch import DeviceBatchCollator, DeviceLoader   # noqa: F401
from .datasets.kitti import KITTIDataset                  # noqa: F401
from .encode import encode_targets, preprocess_images     # noqa: F401
from .samplers import InferenceSampler, IterationBatchSampler, TrainingSampler   # noqa: F401
from .build import DatasetCatalog, build_dataset, build_test_loader, make_data_loader   # noqa: F401,E402
