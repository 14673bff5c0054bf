"""Multi-GPU helpers: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

Inference/decode shards over images with no exchange step: every rank runs an independent replica
on its own shard and only the timing is reduced.  Training DP (gradient reduce-scatter/all-gather) comes with
the backward kernels.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's RANK/WORLD_SIZE/MASTER_* environment.
    Returns (rank, world, local_rank); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    return rank, world, local_rank


def image_shard(rank, world, global_batch):
    """Contiguous shard of a global batch, like the reference's InferenceSampler
    (data/samplers/distributed_sampler.py:193-196): returns (first image index, count)."""
    per = (global_batch + world - 1) // world
    first = min(rank * per, global_batch)
    return first, max(0, min(per, global_batch - first))


def shard_seed(base_seed, rank, per_rank_batch):
    """Seed of a rank's first synthetic image: ranks draw disjoint image streams."""
    return base_seed + rank * per_rank_batch


def aggregate_throughput(elapsed_s, images_local, device="cpu"):
    """Whole-job rate: (sum of images over ranks) / (max elapsed over ranks).  One tiny all-reduce each, outside
    the timed region.  Returns (images_per_s, max_elapsed_s, total_images)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
        n = torch.tensor([float(images_local)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        elapsed_s, images_local = float(t.item()), float(n.item())
    return images_local / elapsed_s, elapsed_s, int(round(images_local))


def ranks_seen(device="cpu"):
    """The ranks that answered an all-gather on the default process group (RCCL on GPUs), sorted: proof, inside the job, of how many processes took part
    (a record that says n_gpus = 8 should also say which eight ranks met).  [0] for a single process."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
        out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(out, mine)
        return sorted(int(t.item()) for t in out)
    return [0]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
