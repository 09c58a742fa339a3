"""Multi-GPU host logic: one process per GPU, each holding its shards of the popBWT.

Every query is searched in every shard; per-shard results are only concatenated (intervals) or
summed (counts) -- exactly what ReadServer's front-end does with the replies of its partitions
(src/service/server.cpp:184-197,404-410).  So the data path needs one collective per batch: a
gather of the (lower, upper) arrays to rank 0, or a sum-reduce of counts.  The functions take torch
tensors resident wherever the process group's backend wants them (HBM for nccl = RCCL, host for
gloo) and do not touch the search itself.
"""
import torch
import torch.distributed as dist


def shard_owner(shard, num_shards, world):
    """Rank holding suffix-shard `shard`: contiguous groups, shard s -> GPU s // (num_shards/world)."""
    per = (num_shards + world - 1) // world
    return shard // per


def local_shards(rank, num_shards, world):
    per = (num_shards + world - 1) // world
    return list(range(rank * per, min(num_shards, (rank + 1) * per)))


def gather_intervals(lower, upper, dst=0, group=None, out=None):
    """lower/upper: [S_local, Q] int64 per rank (uint64 bit patterns).  Rank `dst` receives
    [world, 2, S_local, Q]; other ranks None.  `out` may be a preallocated list of world tensors."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    pair = torch.stack([lower, upper], 0).contiguous()
    if world == 1:
        return pair.unsqueeze(0)
    if rank == dst:
        if out is None:
            out = [torch.empty_like(pair) for _ in range(world)]
        dist.gather(pair, out, dst=dst, group=group)
        return torch.stack(out, 0)
    dist.gather(pair, None, dst=dst, group=group)
    return None


def reduce_counts(counts, dst=0, group=None):
    """counts: [Q] int64 = this rank's sum over its shards.  Rank `dst` gets the sum over all shards."""
    if dist.get_world_size(group) > 1:
        dist.reduce(counts, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return counts if dist.get_rank(group) == dst else None


def broadcast_queries(kmers, src=0, group=None):
    """The query batch ([Q, k] uint8) goes to every rank (the reference PUBlishes each request to
    all partitions: src/service/server.cpp:124,578)."""
    if dist.get_world_size(group) > 1:
        dist.broadcast(kmers, src=src, group=group)
    return kmers
