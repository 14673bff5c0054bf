"""One process per GPU (reference engine/launch.py:23-89): `launch(main_func, num_gpus_per_machine, ...)` spawns the ranks,
joins them in an RCCL ("nccl") process group over 127.0.0.1 / `dist_url`, pins each to its device, then calls main_func."""
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ..utils import comm

__all__ = ["launch"]


def _find_free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=(), backend=None):
    world_size = num_machines * num_gpus_per_machine
    if world_size <= 1:
        return main_func(*args)
    if dist_url in (None, "auto"):
        assert num_machines == 1, "dist_url=auto cannot work with distributed training."
        dist_url = "tcp://127.0.0.1:%d" % _find_free_port()
    mp.spawn(_distributed_worker, nprocs=num_gpus_per_machine,
             args=(main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args, backend), daemon=False)


def _distributed_worker(local_rank, main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args, backend=None):
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        assert num_gpus_per_machine <= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size,
                            rank=machine_rank * num_gpus_per_machine + local_rank)
    comm.synchronize()
    for i in range(world_size // num_gpus_per_machine):              # ranks of one machine form the local group
        pg = dist.new_group(list(range(i * num_gpus_per_machine, (i + 1) * num_gpus_per_machine)))
        if i == machine_rank:
            comm._LOCAL_PROCESS_GROUP = pg
    try:
        main_func(*args)
    finally:
        dist.destroy_process_group()
