"""Multi-GPU sharding of the align + classify path: one process per GPU, reads split into contiguous ranges, no
data-path collective; the per-amplicon count tensor is summed once with an all-reduce (RCCL over xGMI with the "nccl"
backend; gloo on CPU for the tests).  This replaces the reference's fork()ed worker pool over slices of the unique-read
dict (CRISPRessoCORE.py:1870-1898) and its JSON/TSV files on disk (:1222-1240, :1905-1950) as the exchange step.
"""
import os


def shard_boundaries(n_items, n_shards):
    """Same split as the reference's get_variant_cache_equal_boundaries (CRISPRessoCORE.py:1172-1195):
    n_shards-1 segments of n_items // n_shards, the last shard takes the remainder."""
    if n_items < n_shards:
        raise Exception("The number of unique sequences is less than the number of processes. Please reduce the number of processes.")
    boundaries = [0]
    segment = n_items // n_shards
    for _ in range(n_shards - 1):
        boundaries.append(boundaries[-1] + segment)
    boundaries.append(n_items)
    return boundaries


def my_shard(n_items, rank=None, world=None):
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    b = shard_boundaries(n_items, world)
    return b[rank], b[rank + 1]


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/MASTER_*); no-op for one process."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world


def reduce_counts(counts_tensor):
    """In-place sum of the count tensor over all ranks."""
    from .counts import all_reduce
    return all_reduce(counts_tensor)
