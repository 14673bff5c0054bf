"""Batched-image container on the detector's call path (reference structures/image_list.py:6-70)."""
import torch


class ImageList:
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    """Tensor (3-d or 4-d), ImageList, or a list of CHW tensors (zero-padded to a common size)."""
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        if tensors.dim() == 3:
            tensors = tensors[None]
        assert tensors.dim() == 4
        return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if isinstance(tensors, (tuple, list)):
        max_size = [max(s) for s in zip(*[img.shape for img in tensors])]
        if size_divisible > 0:
            max_size[1] = -(-max_size[1] // size_divisible) * size_divisible
            max_size[2] = -(-max_size[2] // size_divisible) * size_divisible
        batched = tensors[0].new_zeros((len(tensors),) + tuple(max_size))
        for img, dst in zip(tensors, batched):
            dst[:img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        return ImageList(batched, [im.shape[-2:] for im in tensors])
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
