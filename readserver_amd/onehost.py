"""The C++ host's shape, driven from Python: ONE process, every GPU of the node (csrc/sets.hip).

The other host (`sharded.py`, bench.py's default) is one process per GPU over torch.distributed.  This one is what
`BASELINE.json:north_star` words -- "host side in C++ calling HIP through a thin C-ABI ... per-shard interval results
gathered over RCCL/xGMI" -- and what a service process holding all 64 shards would run per batch:

    on every device g, on a stream of its own:   rsbwt_pack_kmers_dev
                                                 rsbwt_set_find_interval_pairs_dev   (ONE fused launch over g's shards)
                                                 rsbwt_pack_interval_pairs_dev       (16-byte pairs -> 10-byte records)
    then, on a second stream per device:         rsbwt_set_gather_intervals_dev      (ncclSend / ncclRecv in one group
                                                                                      onto the first device)

The gather of batch i travels while batch i + 1 is searched (double-buffered records, events between the two streams
of a device).  Nothing here touches torch.distributed; torch only provides device buffers, streams and events.  With
one device there is nothing to gather and a step is pack + search, as in the per-rank host.

Reference shape: every request goes to every partition and the front-end concatenates the replies
(src/service/server.cpp:124,184-197).
"""
import ctypes as C

import torch

from . import bwt as _bwt
from ._native import check, lib


def _p(t):
    return C.c_void_p(t.data_ptr())


class OneProcessHost:
    def __init__(self, shards_by_device, Q, k, wire_packed=True):
        """shards_by_device: one list of GpuBWT per device, device g's shards all on cuda:g' for one g' (any ids)."""
        self.L = lib()
        self.G = len(shards_by_device)
        self.Q, self.k = int(Q), int(k)
        self.devices = [int(self.L.rsbwt_device(sh[0].handle)) for sh in shards_by_device]
        self.S = [len(sh) for sh in shards_by_device]
        self.subsets = [_bwt.ShardSet(sh) for sh in shards_by_device]
        # the set that spans the devices owns the RCCL communicators the gather uses
        self.spanning = _bwt.ShardSet([g for sh in shards_by_device for g in sh]) if self.G > 1 else self.subsets[0]
        self.wire_packed = bool(wire_packed) and self.G > 1
        wpq = (self.k + 31) // 32
        self.search_st, self.comm_st = [], []
        self.packed, self.valid, self.pairs, self.rec = [], [], [], []
        self.searched, self.sent = [], []
        for g, d in enumerate(self.devices):
            dev = torch.device("cuda", d)
            with torch.cuda.device(dev):
                self.search_st.append(torch.cuda.Stream(device=dev))
                self.comm_st.append(torch.cuda.Stream(device=dev))
                self.packed.append(torch.empty((self.Q, wpq), dtype=torch.int64, device=dev))
                self.valid.append(torch.empty(self.Q, dtype=torch.uint8, device=dev))
                self.pairs.append([torch.empty((self.S[g], self.Q, 2), dtype=torch.int64, device=dev) for _ in range(2)])
                nb = self.block_bytes(g)
                self.rec.append([torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(2)] if self.wire_packed else None)
                self.searched.append([torch.cuda.Event(), torch.cuda.Event()])
                self.sent.append([torch.cuda.Event(), torch.cuda.Event()])
        self.root = None
        if self.G > 1:
            tot = sum(self.block_bytes(g) for g in range(self.G))
            self.root = [torch.empty(tot, dtype=torch.uint8, device=torch.device("cuda", self.devices[0])) for _ in range(2)]
        self.steps = 0

    def block_bytes(self, g):
        """bytes device g contributes to a gather: its [S_g][Q] pairs, as 10-byte records or as they are"""
        n = self.S[g] * self.Q
        return int(self.L.rsbwt_packed_pairs_bytes(n)) if self.wire_packed else 16 * n

    def attach_tables(self, depth, fmt=0):
        """fmt: RSBWT_KTAB_FORMAT_PLAIN (0) / _GROUPED (1) / _AUTO (2)"""
        for ss in self.subsets:
            check(self.L.rsbwt_set_attach_ktabs_format(ss._s, depth, fmt))

    def auto_table_depth(self):
        return min(int(self.L.rsbwt_set_auto_ktab_depth(ss._s)) for ss in self.subsets)

    def auto_tables(self, fmt=2, keep_free_bytes=0):
        """(depth, format) for the whole job by the library's rule (rsbwt_set_auto_ktab over the spanning set): the
        shallowest any device can hold, plain if any must.  fmt: RSBWT_KTAB_FORMAT_PLAIN (0) / _GROUPED (1) / _AUTO (2)"""
        d, f = C.c_uint32(), C.c_uint32()
        check(self.L.rsbwt_set_auto_ktab(self.spanning._s, fmt, keep_free_bytes, C.byref(d), C.byref(f)))
        return int(d.value), int(f.value)

    def step(self, d_kmers_by_device):
        """One batch: d_kmers_by_device[g] = the [Q][k] ASCII k-mers, resident on device g."""
        i, L, Q, k = self.steps, self.L, self.Q, self.k
        j = i % 2
        for g, d in enumerate(self.devices):
            st = self.search_st[g]
            sp = C.c_void_p(st.cuda_stream)
            with torch.cuda.device(d):
                if i >= 2:
                    st.wait_event(self.sent[g][j])  # batch i - 2's records and pairs have left these buffers
                check(L.rsbwt_pack_kmers_dev(_p(d_kmers_by_device[g]), Q, k, k, _p(self.packed[g]), _p(self.valid[g]), d, sp))
                check(L.rsbwt_set_find_interval_pairs_dev(self.subsets[g]._s, _p(self.packed[g]), _p(self.valid[g]), Q, k,
                                                          _p(self.pairs[g][j]), sp))
                if self.wire_packed:
                    check(L.rsbwt_pack_interval_pairs_dev(_p(self.pairs[g][j]), self.S[g] * Q, _p(self.rec[g][j]), None, d, sp))
                self.searched[g][j].record(st)
                self.comm_st[g].wait_event(self.searched[g][j])
        if self.G > 1:
            blocks = (C.c_void_p * self.G)(*[(self.rec[g][j] if self.wire_packed else self.pairs[g][j]).data_ptr() for g in range(self.G)])
            sizes = (C.c_size_t * self.G)(*[self.block_bytes(g) for g in range(self.G)])
            streams = (C.c_void_p * self.G)(*[self.comm_st[g].cuda_stream for g in range(self.G)])
            check(L.rsbwt_set_gather_intervals_dev(self.spanning._s, blocks, sizes, _p(self.root[j]), streams))
        for g, d in enumerate(self.devices):
            with torch.cuda.device(d):
                self.sent[g][j].record(self.comm_st[g])
        self.steps += 1

    def synchronize(self):
        for d in self.devices:
            torch.cuda.synchronize(d)

    def last_pairs(self, g):
        """[S_g][Q][2] of the last batch on device g (after synchronize())"""
        return self.pairs[g][(self.steps - 1) % 2]

    def verify_last_gather(self):
        """The root's blocks of the last batch are what the devices searched: None with one device."""
        if self.G == 1:
            return None
        from . import sharded
        self.synchronize()
        j = (self.steps - 1) % 2
        d0 = torch.device("cuda", self.devices[0])
        off, good = 0, True
        with torch.cuda.device(d0):
            for g in range(self.G):
                nb, n = self.block_bytes(g), self.S[g] * self.Q
                blk = self.root[j][off:off + nb]
                got = sharded.unpack_pairs(blk, n) if self.wire_packed else blk.view(torch.int64).reshape(n, 2)
                good = good and bool(torch.equal(got.reshape(self.S[g], self.Q, 2), self.pairs[g][j].to(d0)))
                off += nb
            torch.cuda.synchronize(d0)
        return good

    def close(self):
        if self.G > 1:
            self.spanning.close()
        for ss in self.subsets:
            ss.close()
