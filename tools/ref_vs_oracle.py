#!/usr/bin/env python3
"""Build container only: throughput of the REAL reference (oracle/_ref) vs the oracle port on the
golden fixture, same queries, so CPU numbers taken on the GPU box (oracle only) can be related to
the reference.  Prints the ratio recorded in BASELINE.md."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding, readserver_amd as rsb
meta = json.load(open(os.path.join(ROOT, "tests/golden/popbwt_v1.json")))
path = "/tmp/popbwt_v1.bwt"
rsb.build(); rsb.synth_popbwt(path, None, **meta["synth"])
g = np.load(os.path.join(ROOT, "tests/golden/popbwt_v1.npz"))
km = np.ascontiguousarray(np.tile(g["kmers31"][:10000], (40, 1)))
Q, k = km.shape
L = C.CDLL(os.path.join(ROOT, "oracle/_ref/libref_bwt.so"))
L.ref_open.restype = C.c_void_p; L.ref_open.argtypes = [C.c_char_p]
L.ref_find_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
h = L.ref_open(path.encode())
lo = np.empty(Q, np.uint64); up = np.empty(Q, np.uint64)
oix = oracle_binding.load().load(path)
for threads in (1, 8):
    t = time.perf_counter(); L.ref_find_intervals(h, km.ctypes.data, Q, k, k, lo.ctypes.data, up.ctypes.data, threads); tr = time.perf_counter() - t
    t = time.perf_counter(); olo, oup = oix.find_intervals(km, nthreads=threads); to = time.perf_counter() - t
    assert np.array_equal(lo, olo) and np.array_equal(up, oup)
    print(f"threads {threads}: reference {Q/tr:.3e} q/s, oracle {Q/to:.3e} q/s, reference/oracle = {to/tr:.2f}")
