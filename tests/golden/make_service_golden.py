#!/usr/bin/env python3
"""Generates tests/golden/service_v1.json: serialised `Request` messages and, for each, the exact
bytes of the two `Reply` messages a reference `service` process holding the popbwt_v1 fixture as its
one partition sends back (count_reads, src/service/service.cpp:279-315): forward strand, then reverse
complement.  Counts come from the REAL reference's findInterval (oracle/_ref/libref_bwt.so, build
container only); the bytes from the Python protobuf runtime on the re-typed schema
(tests/proto_schema.py = src/service/readserver.proto).  Only inputs and expected outputs are stored.
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


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def main():
    meta = json.load(open(os.path.join(HERE, "popbwt_v1.json")))
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bwt.so"))
    L.ref_open.restype = C.c_void_p
    L.ref_open.argtypes = [C.c_char_p]
    L.ref_find_interval.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    tmp = os.environ.get("TMPDIR", "/tmp")
    bwt_path = os.path.join(tmp, "popbwt_v1.bwt")
    reads_path = os.path.join(tmp, "popbwt_v1.reads")
    rsb.build()
    rsb.synth_popbwt(bwt_path, reads_path, **meta["synth"])
    assert hashlib.sha256(open(bwt_path, "rb").read()).hexdigest() == meta["bwt_sha256"]
    reads = open(reads_path).read().split()
    if os.path.exists(bwt_path + ".bpi2"):
        os.remove(bwt_path + ".bpi2")
    h = L.ref_open(bwt_path.encode())

    def count(w):  # service.cpp:299-304
        if not w or any(c not in "ACGT" for c in w):
            return 0
        lo, up = C.c_uint64(), C.c_uint64()
        L.ref_find_interval(h, w.encode(), len(w), C.byref(lo), C.byref(up))
        return up.value - lo.value + 1 if up.value >= lo.value else 0

    Request, Reply = proto_schema.build()
    rng = np.random.default_rng(17)
    items = []
    for i in range(240):
        r = Request()
        kind = i % 6
        r.t, r.rt = [(1, 1), (1, 1), (2, 1), (2, 2), (3, 1), (4, 3)][kind]
        rr = reads[rng.integers(len(reads))]
        k = int(rng.choice([12, 20, 31, 31, 45, 70]))
        st = int(rng.integers(0, len(rr) - k + 1))
        q = rr[st:st + k]
        if i % 7 == 0:
            q = rc(q)
        if i % 9 == 0:
            q = "".join("ACGT"[x] for x in rng.integers(0, 4, k))
        if i % 31 == 0:
            q = q[:5] + "N" + q[6:]
        if i == 14:
            q = ""
        r.q = q
        if kind >= 4:
            r.k, r.s = 31, 1
        item = {"request": r.SerializeToString().hex(), "t": r.t, "rt": r.rt, "q": q, "replies": []}
        if r.t == 1 or (r.t == 2 and r.rt == 1):
            for strand, w in ((0, q), (1, rc(q))):
                rep = Reply()
                rep.rt, rep.t, rep.q = r.t, 1, q  # the original query string (service.cpp:283)
                (rep.c.revcomp_matches if strand else rep.c.forward_matches).c = count(w)
                item["replies"].append(rep.SerializeToString().hex())
            item["channel"] = 1 if r.t == 1 else 0  # CountReads -> push_count, ExactMatch -> push
        items.append(item)
    json.dump(dict(what="Request bytes and the two Reply bytes a reference service holding popbwt_v1 sends for each "
                        "(count_reads, service.cpp:279-315); counts from the compiled reference",
                   generator="tests/golden/make_service_golden.py", fixture="popbwt_v1.json", items=items),
              open(os.path.join(HERE, "service_v1.json"), "w"), indent=0)
    n = sum(1 for x in items if x["replies"])
    nz = sum(1 for x in items if x["replies"] and any(len(r) > 2 * (8 + len(x["q"])) for r in x["replies"]))
    print(f"wrote service_v1.json: {len(items)} requests, {n} answered by the count path, {nz} with long replies")


if __name__ == "__main__":
    main()
