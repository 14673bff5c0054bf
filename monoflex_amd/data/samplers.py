"""Index streams for data-parallel runs (reference data/samplers/distributed_sampler.py:11-54, 172-199).

Training: every rank walks the SAME endless sequence `perm_0 + perm_1 + ...` (one seeded generator, identical on all ranks)
and keeps the entries rank, rank + world, rank + 2*world, ... -- disjoint shards with no communication after the seed is
agreed on.  Inference: contiguous shards of ceil(size / world) samples, the last ranks possibly shorter or empty."""
import itertools

import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shared_random_seed():
    """One random seed agreed on by all ranks: rank 0 draws it, everybody receives it (utils/comm.py shared_random_seed)."""
    seed = torch.randint(0, 2 ** 31, (1,), dtype=torch.int64)
    rank, world = _rank_world()
    if world > 1:
        if dist.get_backend() == "nccl":
            seed = seed.cuda()
        dist.broadcast(seed, src=0)
    return int(seed.item())


class TrainingSampler(Sampler):
    def __init__(self, size, shuffle=True, seed=None):
        if size <= 0:
            raise ValueError("TrainingSampler needs a non-empty dataset")
        self._size, self._shuffle = int(size), shuffle
        self._seed = int(shared_random_seed() if seed is None else seed)
        self._rank, self._world_size = _rank_world()

    def _stream(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            order = torch.randperm(self._size, generator=g) if self._shuffle else torch.arange(self._size)
            yield from order.tolist()

    def __iter__(self):
        return itertools.islice(self._stream(), self._rank, None, self._world_size)


class InferenceSampler(Sampler):
    def __init__(self, size):
        if size <= 0:
            raise ValueError("InferenceSampler needs a non-empty dataset")
        rank, world = _rank_world()
        shard = (size - 1) // world + 1
        self._local = range(min(shard * rank, size), min(shard * (rank + 1), size))

    def __iter__(self):
        return iter(self._local)

    def __len__(self):
        return len(self._local)


class IterationBatchSampler(Sampler):
    """Batches of `batch_size` indices from an endless sampler, `num_iterations` of them (reference
    data/samplers/iteration_based_batch_sampler.py / data/build.py: the training loader is iteration-based)."""

    def __init__(self, sampler, batch_size, num_iterations, start_iter=0):
        self.sampler, self.batch_size, self.num_iterations, self.start_iter = sampler, batch_size, num_iterations, start_iter

    def __iter__(self):
        it = iter(self.sampler)
        for _ in range(self.start_iter, self.num_iterations):
            yield list(itertools.islice(it, self.batch_size))

    def __len__(self):
        return self.num_iterations - self.start_iter
