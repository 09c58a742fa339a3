#!/usr/bin/env python3
"""Throughput of the SURVEY section-8 "next" rows through the C-ABI's host entry points (host
buffers in and out, so PCIe and the host-side packing are inside the timings):
  f3  1-mismatch search by composition   (rsbwt_find_intervals_1mm, 3k+1 variants per k-mer)
  f2  batched read extraction            (rsbwt_extract: extractPrefix + extractPostfix per row)
  --  exact search, host interface       (rsbwt_find_intervals)
on a synthetic population BWT of real reads (csrc/synth.cpp).  Prints one JSON line.
usage: tools/bench_next_rows.py [genome_len=4000000] [coverage=8]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402

genome_len = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
coverage = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
read_len, k = 100, 31
with tempfile.TemporaryDirectory() as td:
    bwt, reads = os.path.join(td, "s.bwt"), os.path.join(td, "s.reads")
    t0 = time.time()
    rsb.synth_popbwt(bwt, reads, seed=7, genome_len=genome_len, haplotypes=8, snp_rate=0.002,
                     read_len=read_len, coverage=coverage)
    t_synth = time.time() - t0
    g = rsb.GpuBWT(bwt)
    rd = [l.strip() for l in open(reads) if l.strip()]
rng = np.random.default_rng(3)
out = {"n_symbols": int(g.getBWLen()), "n_runs": int(g.num_runs()), "reads": len(rd), "read_len": read_len,
       "synth_s": round(t_synth, 1), "ktab_depth": g.ktab_depth(), "window_span": g.window_span()}

# k-mers: half cut from the reads, half random
Q = 200000
km = np.empty((Q, k), np.uint8)
lut = np.frombuffer(b"ACGT", np.uint8)
for i in range(Q // 2):
    r = rd[rng.integers(len(rd))]
    o = rng.integers(0, len(r) - k + 1)
    km[i] = np.frombuffer(r[o:o + k].encode(), np.uint8)
km[Q // 2:] = lut[rng.integers(0, 4, (Q - Q // 2, k))]

def timed(f, reps=3):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t) / reps

# the C-ABI entry points themselves, on preallocated host buffers (no Python-side result building)
import ctypes as C  # noqa: E402
L = rsb.lib()
vp = lambda a: C.c_void_p(a.ctypes.data)
lo = np.empty(Q, np.uint64)
up = np.empty(Q, np.uint64)
dt = timed(lambda: L.rsbwt_find_intervals(g.handle, vp(km), Q, k, k, vp(lo), vp(up)))
out["exact_host_interface_qps"] = Q / dt
out["exact_hits"] = int((up >= lo).sum())
Q1 = 50000
V = 3 * k + 1
lo1 = np.empty((Q1, V), np.uint64)
up1 = np.empty((Q1, V), np.uint64)
dt = timed(lambda: L.rsbwt_find_intervals_1mm(g.handle, vp(km), Q1, k, k, vp(lo1), vp(up1)))
out["one_mismatch_kmers_per_s"] = Q1 / dt
out["one_mismatch_variant_searches_per_s"] = Q1 * V / dt
hits = np.zeros(8 * Q1, rsb.bwt.HIT_1MM)
nh = C.c_size_t()
dt = timed(lambda: L.rsbwt_hits_1mm(g.handle, vp(km), Q1, k, k, vp(hits), hits.size, C.byref(nh)))
out["one_mismatch_hit_list_kmers_per_s"] = Q1 / dt
out["one_mismatch_hit_list_variant_searches_per_s"] = Q1 * V / dt
out["one_mismatch_hits"] = int(nh.value)
rows = rng.integers(0, g.getBWLen(), 100000).astype(np.uint64)
stride = 128
buf = np.empty((rows.size, stride), np.uint8)
ln = np.empty(rows.size, np.uint32)
pl = np.empty(rows.size, np.uint32)
dt = timed(lambda: L.rsbwt_extract(g.handle, vp(rows), rows.size, vp(buf), stride, vp(ln), vp(pl)))
assert (ln != 0xFFFFFFFF).all()
out["extract_reads_per_s"] = rows.size / dt
out["extract_bases_per_s"] = float(ln.sum()) / dt
g.close()
print(json.dumps(out))
