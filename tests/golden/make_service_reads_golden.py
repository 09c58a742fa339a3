#!/usr/bin/env python3
"""Generates tests/golden/service_reads_v1.json: serialised ExactMatch `Request`s whose return type is Reads and, for
each, the bytes of the two `Reply` messages (forward strand, then reverse complement) a reference `service` process
holding the popbwt_v1 fixture as its one partition (suffix "") sends back: QueryTask::run
(src/service/service.cpp:1260-1291) around find_reads (:714-797), with min_read_length = 50 and max_read_length = 70
in its service.cfg (the fixture's reads are 70 long; the reference's defaults are 73 / 100, :56-57,1417-1420).

The service process itself cannot be built here (it needs ZeroMQ, protobuf, RocksDB and libconfig); its BWT work can:
every read in these replies comes out of the REAL reference's findInterval / extractPrefix / extractPostfix / query /
query_exactmatch (oracle/_ref/libref_bwt.so, compiled from /root/reference/src/bwt where it lies; build container
only).  find_reads' own control flow -- which of those it calls for which query length, and in what order it leaves
their results -- is restated below with the lines it follows; the order in which it visits a query's tiles is that of
std::unordered_set<std::string> on this C++ library (oracle/ref_harness.cpp, ref_tiles_order).  The Reply bytes are
the Python protobuf runtime's on the re-typed schema (tests/proto_schema.py = src/service/readserver.proto).
Only inputs and expected outputs are stored; replies longer than 4 KB as their length and SHA-256.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import proto_schema  # noqa: E402
import readserver_amd as rsb  # noqa: E402

MIN_READ_LENGTH, MAX_READ_LENGTH = 50, 70
LARGE = 2048  # large_match_size, service.cpp:86


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


class Ref:
    def __init__(self, bwt_path):
        L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bwt.so"))
        L.ref_open.restype = C.c_void_p
        L.ref_open.argtypes = [C.c_char_p]
        L.ref_find_interval.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ref_extract.restype = C.c_size_t
        L.ref_extract.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ref_query.restype = C.c_size_t
        L.ref_query.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ref_query_exactmatch.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.ref_tiles_order.restype = C.c_size_t
        L.ref_tiles_order.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        self.L, self.h = L, L.ref_open(bwt_path.encode())

    def interval(self, w):
        lo, up = C.c_uint64(), C.c_uint64()
        self.L.ref_find_interval(self.h, w.encode(), len(w), C.byref(lo), C.byref(up))
        return lo.value, up.value

    def extract(self, row):
        buf = C.create_string_buffer(4096)
        n = self.L.ref_extract(self.h, row, buf, 4096, None)
        return buf.raw[:n].decode()

    def query(self, w):
        cnt = C.c_size_t()
        need = self.L.ref_query(self.h, w.encode(), len(w), None, 0, C.byref(cnt))
        buf = C.create_string_buffer(max(need, 1))
        self.L.ref_query(self.h, w.encode(), len(w), buf, need, C.byref(cnt))
        return buf.raw[:need].decode().split("\n")[:-1] if need else []

    def exactmatch(self, w):
        return bool(self.L.ref_query_exactmatch(self.h, w.encode(), len(w)))

    def tiles(self, w, kmer):
        cnt = C.c_size_t()
        need = self.L.ref_tiles_order(w.encode(), len(w), kmer, None, 0, C.byref(cnt))
        buf = C.create_string_buffer(max(need, 1))
        self.L.ref_tiles_order(w.encode(), len(w), kmer, buf, need, C.byref(cnt))
        return buf.raw[:need].decode().split("\n")[:-1] if need else []


def find_reads(ref, w, s=""):
    """service.cpp:714-797, for one partition with suffix s."""
    seqs, sz = [], len(w)
    if sz < MIN_READ_LENGTH:                               # :718
        lo, up = ref.interval(w)                           # :719
        if up < lo:                                        # :720-722
            return seqs
        start, chunks = lo, []
        while up - start > 2 * LARGE:                      # :729: a chunk of 2,048 rows goes to the pool
            chunks.append((start, start + LARGE - 1))      # :735-736
            start += LARGE                                 # :739
        for row in range(start, up + 1):                   # :742-744: what is left, extracted on the spot, comes first
            seqs.append(ref.extract(row))
        for a, b in chunks:                                # :746-751: then the chunks' reads, in chunk order
            seqs.extend(ref.extract(row) for row in range(a, b + 1))
        return seqs
    ends = lambda t: s == "" or t.endswith(s)              # is_suffix_of, :228-230
    if sz < MAX_READ_LENGTH:                               # :755
        if sz != MIN_READ_LENGTH:                          # :756
            for t in ref.tiles(w, MIN_READ_LENGTH):        # :758-763
                if ends(t) and ref.exactmatch(t):
                    seqs.append(t)
        seqs.extend(ref.query(w))                          # :767-768
        return seqs
    for t in ref.tiles(w, MAX_READ_LENGTH):                # :776-781
        if ends(t) and ref.exactmatch(t):
            seqs.append(t)
    if MIN_READ_LENGTH != MAX_READ_LENGTH:                 # :784-791
        for t in ref.tiles(w, MIN_READ_LENGTH):
            if ends(t) and ref.exactmatch(t):
                seqs.append(t)
    return seqs


def main():
    meta = json.load(open(os.path.join(HERE, "popbwt_v1.json")))
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = os.environ.get("TMPDIR", "/tmp")
    bwt_path, reads_path = os.path.join(tmp, "popbwt_v1.bwt"), os.path.join(tmp, "popbwt_v1.reads")
    rsb.build()
    rsb.synth_popbwt(bwt_path, reads_path, **meta["synth"])
    assert hashlib.sha256(open(bwt_path, "rb").read()).hexdigest() == meta["bwt_sha256"]
    reads = open(reads_path).read().split()
    if os.path.exists(bwt_path + ".bpi2"):
        os.remove(bwt_path + ".bpi2")
    ref = Ref(bwt_path)
    Request, Reply = proto_schema.build()
    rng = np.random.default_rng(23)
    rnd = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, k))
    queries = []
    # the ExactMatch + Reads requests of service_v1.json (left unanswered there: the count path's golden file)
    for it in json.load(open(os.path.join(HERE, "service_v1.json")))["items"]:
        if it["t"] == 2 and it["rt"] == 2 and it["q"] and all(c in "ACGT" for c in it["q"]):
            queries.append(it["q"])
    # short queries: intervals of a few rows, of thousands (5-mers: more than 2 x 2,048 rows, the chunked order), none
    for k in (5, 6, 9, 14, 21, 31, 40, 49):
        for _ in range(3):
            r = reads[rng.integers(len(reads))]
            st = int(rng.integers(0, len(r) - k + 1))
            queries.append(r[st:st + k])
    queries += [rnd(25), rnd(49), "A" * 12, "ACGT" * 3]
    # MIN <= length < MAX: query(w), and the MIN-long tiles (none of them is a read: all reads are 70 long)
    for k in (50, 51, 58, 64, 69):
        for _ in range(3):
            r = reads[rng.integers(len(reads))]
            st = int(rng.integers(0, len(r) - k + 1))
            queries.append(r[st:st + k])
        queries.append(rnd(k))
    # length >= MAX: the MAX-long tiles that are reads -- a read itself, a read inside random flanks, two reads joined,
    # a read with one substitution (no tile is a read), a long random string
    for _ in range(4):
        queries.append(reads[rng.integers(len(reads))])
    for _ in range(4):
        queries.append(rnd(int(rng.integers(1, 30))) + reads[rng.integers(len(reads))] + rnd(int(rng.integers(0, 30))))
    queries.append(reads[rng.integers(len(reads))] + reads[rng.integers(len(reads))])
    r = reads[rng.integers(len(reads))]
    queries.append(r[:30] + ("A" if r[30] != "A" else "C") + r[31:])
    queries.append(rnd(150))
    queries.append(rc(reads[rng.integers(len(reads))]))  # (the reverse-complement strand finds it)
    items, big = [], 0
    for q in queries:
        rq = Request()
        rq.t, rq.rt, rq.q = 2, 2, q
        item = {"request": rq.SerializeToString().hex(), "t": 2, "rt": 2, "q": q, "replies": [], "reads": [], "channel": 0}
        for strand, w in ((0, q), (1, rc(q))):
            seqs = find_reads(ref, w)
            rep = Reply()
            rep.rt, rep.t, rep.q = 2, 2, q       # rt = ExactMatch, t = the request's return type, the ORIGINAL query (:1262-1265)
            rep.r.SetInParent()                   # mutable_r() (:1278): present even when nothing matched
            for sq in seqs:
                (rep.r.revcomp_matches if strand else rep.r.forward_matches).add().r = sq
            b = rep.SerializeToString()
            item["reads"].append(len(seqs))
            if len(b) > 4096:
                item["replies"].append({"len": len(b), "sha256": hashlib.sha256(b).hexdigest()})
                big += 1
            else:
                item["replies"].append(b.hex())
        items.append(item)
    json.dump(dict(what="ExactMatch Requests with return type Reads and the two Reply bytes (forward, reverse complement) a reference "
                        "service holding popbwt_v1 as its one partition sends for each (QueryTask::run + find_reads, "
                        "service.cpp:714-797,1260-1291) with min_read_length 50, max_read_length 70; reads from the compiled reference",
                   generator="tests/golden/make_service_reads_golden.py", fixture="popbwt_v1.json",
                   min_read_length=MIN_READ_LENGTH, max_read_length=MAX_READ_LENGTH, suffix="", items=items),
              open(os.path.join(HERE, "service_reads_v1.json"), "w"), indent=0)
    tot = sum(sum(x["reads"]) for x in items)
    print(f"wrote service_reads_v1.json: {len(items)} requests, {tot} reads in their replies, {big} replies stored as hashes, "
          f"{sum(1 for x in items if max(x['reads']) > 2 * LARGE + 1)} with a chunked interval, {sum(1 for x in items if sum(x['reads']) == 0)} without a read")


if __name__ == "__main__":
    main()
