#!/usr/bin/env python3
"""How fast the service loop answers `ExactMatch` / `Reads` requests (find_reads + ReplyReads, csrc/service_slice.cpp;
reference: src/service/service.cpp:714-797,1260-1291): P suffix partitions of a synthetic read collection held by one
process, N requests pushed into the in-process transport, the loop run to the end of input, the replies popped and
counted (2 per request and partition: forward strand, reverse complement).  Queries: substrings of reads, a third each
shorter than min_read_length (every read containing the query: query()), of min..max (tiles of min_read_length as exact
reads + query()), and whole reads + flanks (tiles of both lengths: query_exactmatch()).
usage: tools/service_reads_probe.py [requests=20000] [partitions=4] [genome=300000] [coverage=8]   -> one JSON line
PROBE_COUNT=1: the same queries as CountReads requests (ReplyCount on the count socket): a window of mixed lengths in
one search (rsbwt_set_find_intervals_var)."""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import readserver_amd as rsb  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
GENOME = int(float(sys.argv[3])) if len(sys.argv) > 3 else 300000
COV = float(sys.argv[4]) if len(sys.argv) > 4 else 8.0
READ_LEN, MINL, MAXL = 100, 73, 100  # the reference's defaults (service.cpp:56-57)
L = rsb.lib()


def varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


COUNT = os.environ.get("PROBE_COUNT") is not None
CH = 1 if COUNT else 0  # the count socket / the reads socket (service.cpp:1499-1502,1568)


def request(q):  # Request{t = ExactMatch(2), rt = Reads(2), q}; PROBE_COUNT: {t = CountReads(1), rt = Count(1), q}
    b = q.encode()
    return (b"\x08\x01\x10\x01\x1a" if COUNT else b"\x08\x02\x10\x02\x1a") + varint(len(b)) + b


with tempfile.TemporaryDirectory() as td:
    kw = dict(seed=77, genome_len=GENOME, haplotypes=8, snp_rate=0.002, read_len=READ_LEN, coverage=COV)
    shards, reads = [], []
    for s in range(P):
        p, rd = os.path.join(td, f"s{s}.bwt"), os.path.join(td, f"s{s}.reads")
        rsb.synth_popbwt(p, rd, shard=s, num_shards=P, **kw)
        shards.append(rsb.GpuBWT(p, for_reads=True))
        reads += open(rd).read().split()
    rng = np.random.default_rng(5)
    qs = []
    for i in range(N):
        r = reads[int(rng.integers(len(reads)))]
        kind = i % 3
        if kind == 0:
            k = int(rng.integers(25, MINL))
            st = int(rng.integers(0, len(r) - k + 1))
            qs.append(r[st:st + k])
        elif kind == 1:
            k = int(rng.integers(MINL, MAXL))
            st = int(rng.integers(0, len(r) - k + 1))
            qs.append(r[st:st + k])
        else:
            qs.append("".join("ACGT"[x] for x in rng.integers(0, 4, 10)) + r + "".join("ACGT"[x] for x in rng.integers(0, 4, 10)))
    ss = rsb.ShardSet(shards)
    tr, svc = C.c_void_p(), C.c_void_p()
    assert L.rsbwt_transport_inproc(C.byref(tr)) == 0
    assert L.rsbwt_service_create(ss._s, tr, 2000, 4096, 1, C.byref(svc)) == 0
    L.rsbwt_service_set_reads(svc, 1, MINL, MAXL)
    if P == 4:
        suf = (C.c_char_p * 4)(b"A", b"C", b"G", b"T")
        assert L.rsbwt_service_set_suffixes(svc, suf, 4) == 0
    msgs = [request(q) for q in qs]
    threaded = os.environ.get("PROBE_ONE_THREAD") is None  # the loop as a pipeline (receiver, window workers, ordered sender): rsbwt_service_start
    if threaded:
        L.rsbwt_service_set_workers(svc, 8)
        assert L.rsbwt_service_start(svc) == 0
    flat = np.frombuffer(b"".join(msgs), np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(w) for w in msgs])]).astype(np.uint64)
    t0 = time.time()
    for i0 in range(0, N, 512):  # the in-process transport in bulk, as tools/service_bench.cpp
        m = min(512, N - i0)
        o = (offs[i0:i0 + m + 1] - offs[i0]).astype(np.uint64)
        assert L.rsbwt_transport_push_requests(tr, C.c_void_p(flat.ctypes.data + int(offs[i0])), C.c_void_p(o.ctypes.data), m) == 0
    cap = 1 << 26
    buf = (C.c_uint8 * cap)()
    n = C.c_size_t()
    replies = nbytes = 0
    if threaded:
        roff = np.zeros(8193, np.uint64)
        while replies < 2 * P * N:
            if L.rsbwt_transport_pop_replies(tr, CH, buf, cap, C.c_void_p(roff.ctypes.data), 8192, C.byref(n), 30_000_000) != 0 or n.value == 0:
                break
            replies += n.value
            nbytes += int(roff[n.value])
        dt = time.time() - t0
        L.rsbwt_transport_close(tr)
        L.rsbwt_service_stop(svc)
    else:
        L.rsbwt_transport_close(tr)
        assert L.rsbwt_service_run(svc) == 0
        dt = time.time() - t0
        while L.rsbwt_transport_pop_reply(tr, CH, buf, cap, C.byref(n), 1000) == 0:
            replies += 1
            nbytes += n.value
    st = (C.c_uint64 * 6)()
    L.rsbwt_service_stats(svc, st)
    print(json.dumps({"request_type": "CountReads -> ReplyCount" if COUNT else "ExactMatch / Reads -> ReplyReads", "requests": N, "partitions": P, "reads_in_the_collection": len(reads), "read_length": READ_LEN,
                      "min_read_length": MINL, "max_read_length": MAXL, "seconds": round(dt, 4), "requests_per_s": round(N / dt, 1),
                      "replies": replies, "replies_expected": 2 * P * N, "reply_bytes": nbytes,
                      "reply_MB_per_s": round(nbytes / dt / 1e6, 1), "windows": int(st[2]), "loop": "pipeline, 8 window workers" if threaded else "rsbwt_service_run on one thread",
                      "queries": "a third shorter than min_read_length, a third of min..max, a third whole reads with 10-base flanks"}))
    L.rsbwt_service_free(svc)
    L.rsbwt_transport_free(tr)
    ss.close()
    for g in shards:
        g.close()
    sys.exit(0 if replies == 2 * P * N else 1)
