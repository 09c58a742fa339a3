"""The body of tests/test_gpu_robustness.py::test_gpu_wild_counts_do_not_leave_the_index, runnable by
hand: RSBWT_LIB=tools/bin/librsbwt_wildocc.so RSBWT_SEARCH_KERNEL=pair|solo python tools/wildocc_probe.py DEPTH"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb


def say(*a):
    print(*a, flush=True)


rng = np.random.default_rng(11)
R = 3000000
runs = ((rng.integers(1, 5, R) << 5) | rng.integers(1, 32, R)).astype(np.uint8)
km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (400000, 31))]
t0 = time.time()
say("lib", rsb.lib_path(), "kernel", os.environ.get("RSBWT_SEARCH_KERNEL", "auto"))
with rsb.GpuBWT(runs=runs, ktab_depth=int(sys.argv[1])) as g:
    say("open done", round(time.time() - t0, 2), "T", g.ktab_depth())
    lo, up = rsb.find_intervals(g, km)
    say("intervals done", round(time.time() - t0, 2))
    c = rsb.count_kmers(g, km)
    say("counts done", round(time.time() - t0, 2))
say("survived", int((lo <= up).sum()))
