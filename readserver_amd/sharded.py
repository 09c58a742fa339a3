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


class IntervalGatherer:
    """Gather of per-shard intervals to rank `dst`, pipelined behind the next batch's search.

    The searches of batch i write (lower, upper) straight into `pair(i)` = one of `depth` resident
    [2, S_local, Q] buffers; `submit(i)` starts the gather of that buffer without blocking (RCCL
    runs it on its own stream once the producing kernels are done), so batch i + 1 is searched
    while batch i travels.  A buffer is reused only after its gather has completed (`submit`
    waits for the one issued `depth` batches earlier); `drain()` completes everything outstanding.
    On rank `dst`, `result(i)` is the list of `world` tensors [2, S_local, Q] of batch i.
    """

    def __init__(self, s_local, q, device, depth=2, dst=0, group=None, dtype=torch.int64, interleaved=False):
        """interleaved: buffers are [S_local, Q, 2] = {lower, upper} pairs (what rsbwt_*_interval_pairs_dev
        writes) instead of [2, S_local, Q]."""
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst, self.group, self.depth = dst, group, depth
        shape = (s_local, q, 2) if interleaved else (2, s_local, q)
        self._pairs = [torch.empty(shape, dtype=dtype, device=device) for _ in range(depth)]
        self._out = None
        if self.world > 1 and self.rank == dst:
            self._out = [[torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world)]
                         for _ in range(depth)]
        self._work = [None] * depth

    def pair(self, i):
        return self._pairs[i % self.depth]

    def acquire(self, i):
        """Call before writing batch i into pair(i): waits until that buffer's previous gather is done."""
        w = self._work[i % self.depth]
        if w is not None:
            w.wait()
            self._work[i % self.depth] = None
        return self._pairs[i % self.depth]

    def submit(self, i):
        if self.world == 1:
            return
        j = i % self.depth
        self._work[j] = dist.gather(self._pairs[j], self._out[j] if self.rank == self.dst else None,
                                    dst=self.dst, group=self.group, async_op=True)

    def drain(self):
        for j, w in enumerate(self._work):
            if w is not None:
                w.wait()
                self._work[j] = None

    def result(self, i):
        if self.world == 1:
            return [self._pairs[i % self.depth]]
        return self._out[i % self.depth] if self.rank == self.dst else None
