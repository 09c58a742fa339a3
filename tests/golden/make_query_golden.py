#!/usr/bin/env python3
"""Generates tests/golden/query_v1.npz from the REAL reference: the answers of `query` and
`query_exactmatch` (src/bwt/query.cpp:87-120) on the golden popBWT fixture (popbwt_v1.json's
synthesiser parameters, same SHA-256).

Runs only in the build container (needs /root/reference; oracle/Makefile `ref` compiles the
reference's own sources).  The reference was verified sound on this fixture by make_golden.py.
Only inputs and expected outputs are stored.
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

import readserver_amd as rsb  # noqa: E402


def main():
    meta = json.load(open(os.path.join(HERE, "popbwt_v1.json")))
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bwt.so"))
    vp = C.c_void_p
    L.ref_open.restype = vp
    L.ref_open.argtypes = [C.c_char_p]
    L.ref_close.argtypes = [vp]
    L.ref_query.restype = C.c_size_t
    L.ref_query.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.ref_query_exactmatch.restype = C.c_int
    L.ref_query_exactmatch.argtypes = [vp, C.c_char_p, C.c_size_t]
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
    rng = np.random.default_rng(11)
    rl = len(reads[0])

    # ---- query_exactmatch: whole reads, reads with one substitution, proper substrings, random, foreign symbol
    em = {}
    for L_ in (rl, rl - 1, 40):
        ws = []
        for _ in range(300):
            r = reads[rng.integers(len(reads))]
            kind = rng.integers(4)
            if L_ == rl and kind <= 1:
                w = r
            elif kind == 2:
                s = rng.integers(0, rl - L_ + 1)
                w = list(r[s:s + L_])
                p = rng.integers(L_)
                w[p] = "ACGT"[("ACGT".index(w[p]) + 1 + rng.integers(3)) % 4]
                w = "".join(w)
            elif kind == 3:
                w = "".join("ACGT"[x] for x in rng.integers(0, 4, L_))
            else:
                s = rng.integers(0, rl - L_ + 1)
                w = r[s:s + L_]
            ws.append(w)
        ws[7] = "N" + ws[7][1:]
        ans = np.array([L.ref_query_exactmatch(h, w.encode(), len(w)) for w in ws], np.uint8)
        em[L_] = (np.frombuffer("".join(ws).encode(), np.uint8).reshape(len(ws), L_), ans)
        print(f"query_exactmatch, length {L_}: {int(ans.sum())} of {len(ws)} are reads")

    # ---- query: k-mers with a handful of containing reads each
    qk = {}
    for k in (25, 31, 45):
        ws = []
        for _ in range(120):
            r = reads[rng.integers(len(reads))]
            s = rng.integers(0, rl - k + 1)
            ws.append(r[s:s + k])
        for _ in range(30):
            ws.append("".join("ACGT"[x] for x in rng.integers(0, 4, k)))
        ws[3] = ws[3][:5] + "N" + ws[3][6:]
        first = [0]
        flat = []
        buf = C.create_string_buffer(1 << 22)
        for w in ws:
            cnt = C.c_size_t()
            need = L.ref_query(h, w.encode(), len(w), buf, len(buf), C.byref(cnt))
            assert need <= len(buf)
            got = buf.raw[:need].decode().split("\n")[:-1]
            assert len(got) == cnt.value and all(w in g for g in got)
            flat += got
            first.append(len(flat))
        width = max(len(x) for x in flat)
        arr = np.zeros((len(flat), width), np.uint8)
        ln = np.zeros(len(flat), np.uint32)
        for i, x in enumerate(flat):
            arr[i, :len(x)] = np.frombuffer(x.encode(), np.uint8)
            ln[i] = len(x)
        qk[k] = (np.frombuffer("".join(ws).encode(), np.uint8).reshape(len(ws), k), np.array(first, np.uint64), arr, ln)
        print(f"query, k = {k}: {len(flat)} reads for {len(ws)} k-mers")
    L.ref_close(h)

    out = {}
    for L_, (a, ans) in em.items():
        out[f"em_w{L_}"] = a
        out[f"em_ans{L_}"] = ans
    for k, (a, first, arr, ln) in qk.items():
        out[f"q_w{k}"] = a
        out[f"q_first{k}"] = first
        out[f"q_reads{k}"] = arr
        out[f"q_len{k}"] = ln
    np.savez_compressed(os.path.join(HERE, "query_v1.npz"), **out)
    json.dump(dict(what="query / query_exactmatch answers of ReadServer's own src/bwt (query.cpp:87-120) on the popbwt_v1 fixture",
                   generator="tests/golden/make_query_golden.py", fixture="popbwt_v1.json", read_len=rl,
                   em_lengths=sorted(em), query_k=sorted(qk)),
              open(os.path.join(HERE, "query_v1.json"), "w"), indent=1)
    print("wrote query_v1.npz / query_v1.json")


if __name__ == "__main__":
    main()
