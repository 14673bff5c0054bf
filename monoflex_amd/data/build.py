"""Loader construction with the reference's names (data/build.py:17-180): `make_data_loader(cfg, is_train)` and
`build_test_loader(cfg)`.  Dataset roots come from `DatasetCatalog` (config/paths_catalog.py:3-27): DATA_DIR is
$MONOFLEX_DATA_DIR (default ./datasets) + `kitti/object/{training,testing}/`.  Workers only read files; the batch is encoded
on the device by DeviceLoader (collate_batch.py)."""
import os

from ..utils.comm import get_world_size
from .collate_batch import DeviceLoader
from .datasets.kitti import KITTIDataset
from .samplers import InferenceSampler, IterationBatchSampler, TrainingSampler


class DatasetCatalog:
    DATASETS = {"kitti_train": "kitti/object/training/", "kitti_test": "kitti/object/testing/"}

    @staticmethod
    def get(name):
        if name not in DatasetCatalog.DATASETS:
            raise RuntimeError("Dataset not available: {}".format(name))
        data_dir = os.environ.get("MONOFLEX_DATA_DIR", "./datasets")
        return dict(factory="KITTIDataset", args=dict(root=os.path.join(data_dir, DatasetCatalog.DATASETS[name])))


def build_dataset(cfg, is_train=True, device=None):
    names = cfg.DATASETS.TRAIN if is_train else cfg.DATASETS.TEST
    if not isinstance(names, (list, tuple)):
        raise RuntimeError("dataset_list should be a list of strings, got {}".format(names))
    return [KITTIDataset(cfg=cfg, is_train=is_train, transforms=None, device=device or cfg.MODEL.DEVICE, **DatasetCatalog.get(n)["args"])
            for n in names]


def make_data_loader(cfg, is_train=True):
    world = get_world_size()
    per_batch = cfg.SOLVER.IMS_PER_BATCH if is_train else cfg.TEST.IMS_PER_BATCH
    assert per_batch % world == 0, "IMS_PER_BATCH ({}) must be divisible by the number of GPUs ({}) used.".format(per_batch, world)
    loaders = []
    for ds in build_dataset(cfg, is_train):
        sampler = TrainingSampler(len(ds))                       # one shared endless permutation stream, rank r takes r::world
        bs = IterationBatchSampler(sampler, per_batch // world, num_iterations=int(cfg.SOLVER.MAX_ITERATION))
        loaders.append(DeviceLoader(ds, batch_sampler=bs, num_workers=int(cfg.get("DATALOADER", {}).get("NUM_WORKERS", 4))))
    if is_train:
        assert len(loaders) == 1
        return loaders[0]
    return loaders


def build_test_loader(cfg, is_train=False):
    return [DeviceLoader(ds, batch_size=cfg.TEST.IMS_PER_BATCH, sampler=InferenceSampler(len(ds)),
                         num_workers=int(cfg.get("DATALOADER", {}).get("NUM_WORKERS", 4)))
            for ds in build_dataset(cfg, is_train)]
