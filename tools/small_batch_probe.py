#!/usr/bin/env python3
"""Times the service's count call at window size: rsbwt_set_count of Q 31-mers (default 8192 = a 4096-request window, both
strands) over P small shards opened like tools/service_bench.cpp opens them (each with its own table), host buffers, and the
single-shard rsbwt_count beside it.  RSBWT_LIB picks the library (A/B of two builds on one box).
usage: tools/small_batch_probe.py [P=8] [Q=8192] [run_bytes=2e8] [calls=200]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
R = int(float(sys.argv[3])) if len(sys.argv) > 3 else 200000000
N = int(sys.argv[4]) if len(sys.argv) > 4 else 200
L = rsb.lib()
shards = []
runs = np.empty(R, np.uint8)
for s in range(P):
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 4242 + s) == 0
    shards.append(rsb.GpuBWT(runs=runs))
ss = rsb.ShardSet(shards)
rng = np.random.default_rng(1)
km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (Q, 31))].copy()
out = {"lib": os.environ.get("RSBWT_LIB", "readserver_amd/lib/librsbwt.so"), "partitions": P, "kmers_per_call": Q, "run_bytes_per_shard": R,
       "ktab_depth": shards[0].ktab_depth()}
for name, fn in (("set_count", lambda: ss.count(km)), ("set_find_intervals", lambda: ss.find_intervals(km)),
                 ("count_one_shard", lambda: rsb.count_kmers(shards[0], km))):
    for _ in range(20):
        fn()
    ts = []
    for _ in range(N):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    out[name + "_us"] = {"mean": round(float(ts.mean()), 1), "p50": round(float(np.median(ts)), 1), "p99": round(float(np.percentile(ts, 99)), 1)}
print(json.dumps(out))
