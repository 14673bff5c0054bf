"""Batch assembly (reference data/collate_batch.py:5-22, data/build.py make_data_loader).

The reference collates per-sample tensors that each worker already encoded on the CPU.  Here DataLoader workers only read
and parse files (`KITTIDataset.load_raw` through `RawView`, collated as a plain list), and `DeviceLoader` applies the
`DeviceBatchCollator` in the training process, which encodes the whole batch on the GPU in three launches."""
import torch

from ..structures.image_list import ImageList


class RawView(torch.utils.data.Dataset):
    """Dataset view whose items are the host-side RawSamples (safe to produce in worker processes: no GPU work)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        return self.dataset.load_raw(idx)


class DeviceBatchCollator:
    """[RawSample] -> the reference's dict(images, targets, img_ids) plus `fields`, the batch-stacked device tensors (what
    engine.trainer.prepare_targets would otherwise re-stack). Touches the GPU: call it in the training process, not as a
    worker-side collate_fn."""

    def __init__(self, dataset, check=True):
        self.dataset, self.check = dataset, check

    def __call__(self, batch):
        images, targets, ids, fields = self.dataset.encode_batch(list(batch), check=self.check)
        sizes = [(images.shape[-2], images.shape[-1])] * images.shape[0]
        return dict(images=ImageList(images, sizes), targets=tuple(targets), img_ids=tuple(ids), fields=fields)


def _as_list(batch):
    return list(batch)


class DeviceLoader:
    """Iterable of encoded batches: a torch DataLoader over RawView(dataset) for the file I/O (any sampler / worker count),
    with the device encoding applied to each list of raw samples as it arrives (reference data/build.py:61-120)."""

    def __init__(self, dataset, batch_size=1, sampler=None, batch_sampler=None, num_workers=0, shuffle=False, check=True):
        kw = dict(batch_sampler=batch_sampler) if batch_sampler is not None else dict(batch_size=batch_size, sampler=sampler,
                                                                                      shuffle=shuffle and sampler is None)
        self.loader = torch.utils.data.DataLoader(RawView(dataset), num_workers=num_workers, collate_fn=_as_list, **kw)
        self.collate = DeviceBatchCollator(dataset, check=check)
        self.dataset = dataset

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for raw in self.loader:
            yield self.collate(raw)
