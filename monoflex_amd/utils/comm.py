"""Process-group helpers with the reference's names (utils/comm.py:20-260); one process per GPU, RCCL ("nccl") on the GPU box,
gloo in CPU tests."""
import functools
import os

import torch
import torch.distributed as dist

_LOCAL_PROCESS_GROUP = None


def _on():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _on() else 1


def get_rank():
    return dist.get_rank() if _on() else 0


def get_local_rank():
    if not _on():
        return 0
    if _LOCAL_PROCESS_GROUP is not None:
        return dist.get_rank(group=_LOCAL_PROCESS_GROUP)
    return int(os.environ.get("LOCAL_RANK", dist.get_rank()))


def get_local_size():
    if not _on():
        return 1
    return dist.get_world_size(group=_LOCAL_PROCESS_GROUP) if _LOCAL_PROCESS_GROUP is not None else int(os.environ.get("LOCAL_WORLD_SIZE", dist.get_world_size()))


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def all_gather(data, group=None):
    """Arbitrary picklable `data` from every rank -> list ordered by rank."""
    if get_world_size() == 1:
        return [data]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, data, group=group)
    return out


def gather(data, dst=0, group=None):
    got = all_gather(data, group)
    return got if get_rank() == dst else []


def shared_random_seed():
    from ..data.samplers import shared_random_seed as f
    return f()


def reduce_dict(input_dict, average=True):
    """{name: 0-d tensor} summed (or averaged) over ranks onto rank 0 (utils/comm.py:235-260)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world
        return dict(zip(names, values))
