#!/usr/bin/env python3
"""Generates tests/golden/bpi2_v1.npz from the REAL reference (build container only).

For a few synthetic popBWTs (this repo's deterministic builder) the compiled reference
(oracle/_ref/libref_bwt.so, `make -C oracle ref`) loads the .bwt and writes its FM-index the way
src/util/index_rlebwt.cpp:19-22 does (RLEBWT::serialiseFMIndex, src/bwt/rlebwt.cpp:150-161).  The
bytes of those `.bpi2` files are the expected outputs of rsbwt_bpi2_write for the same inputs:
one, two and three counter levels (run counts below 2^10, below 2^20, above).
Only inputs (synthesis parameters) and outputs (file bytes) are stored.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import readserver_amd as rsb  # noqa: E402

CASES = {
    "one_level": dict(seed=11, genome_len=300, haplotypes=2, snp_rate=0.01, read_len=30, coverage=2.0),
    "two_levels": dict(seed=12, genome_len=20000, haplotypes=4, snp_rate=0.004, read_len=50, coverage=3.0),
    "golden_fixture": dict(seed=20261003, genome_len=400000, haplotypes=8, snp_rate=0.002, read_len=70,
                           coverage=3.0),
}

subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bwt.so"))
L.ref_open.restype = C.c_void_p
L.ref_open.argtypes = [C.c_char_p]
L.ref_close.argtypes = [C.c_void_p]
L.ref_serialise.argtypes = [C.c_void_p, C.c_char_p]

arrays, meta = {}, {}
with tempfile.TemporaryDirectory() as td:
    for name, p in CASES.items():
        bwt = os.path.join(td, name + ".bwt")
        rsb.synth_popbwt(bwt, None, shard=-1, num_shards=1, **p)
        h = L.ref_open(bwt.encode())
        out = os.path.join(td, name + ".bpi2")
        L.ref_serialise(h, out.encode())
        L.ref_close(h)
        b = np.fromfile(out, np.uint8)
        arrays[name] = b
        hdr = np.fromfile(bwt, np.uint8, 30)
        meta[name] = {"synth": p, "bpi2_bytes": int(b.size), "bpi2_sha256": hashlib.sha256(b.tobytes()).hexdigest(),
                      "bwt_sha256": hashlib.sha256(open(bwt, "rb").read()).hexdigest(),
                      "num_runs": int(hdr[18:26].view(np.uint64)[0]), "depth": int(b[:8].view(np.uint64)[0])}
        print(name, meta[name])
np.savez_compressed(os.path.join(HERE, "bpi2_v1.npz"), **arrays)
json.dump(meta, open(os.path.join(HERE, "bpi2_v1.json"), "w"), indent=1)
