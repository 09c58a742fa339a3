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
       "synth_s": round(t_synth, 1), "ktab_depth": g.ktab_depth(), "slot_span": g.slot_span()}

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

dt = timed(lambda: rsb.find_intervals(g, km))
out["exact_host_interface_qps"] = Q / dt
lo, up = rsb.find_intervals(g, km)
out["exact_hits"] = int((up >= lo).sum())
Q1 = 50000
dt = timed(lambda: rsb.find_intervals_1mm(g, km[:Q1]))
out["one_mismatch_kmers_per_s"] = Q1 / dt
out["one_mismatch_variant_searches_per_s"] = Q1 * (3 * k + 1) / dt
dt = timed(lambda: rsb.hits_1mm_batch(g, km[:Q1], cap=40 * Q1))
out["one_mismatch_hit_list_kmers_per_s"] = Q1 / dt
out["one_mismatch_hits"] = int(len(rsb.hits_1mm_batch(g, km[:Q1], cap=40 * Q1)))
rows = rng.integers(0, g.getBWLen(), 100000).astype(np.uint64)
dt = timed(lambda: rsb.extract_reads(g, rows, stride=256))
out["extract_reads_per_s"] = rows.size / dt
out["extract_bases_per_s"] = rows.size * read_len / dt
g.close()
print(json.dumps(out))
