#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry point on bench.py's default shard: Q ASCII 31-mers in
host memory -> rsbwt_find_intervals -> (lower, upper) in host memory.  Never bench.py's `value`
(that one has its inputs resident in HBM); DESIGN.md section 5 quotes this number."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000000
k = 31
L = rsb.lib()
dev = torch.device("cuda", 0)
d_runs = torch.empty(R, dtype=torch.uint8, device=dev)
assert L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, 1000003, 0, None) == 0
torch.cuda.synchronize()
g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R))
del d_runs
rng = np.random.default_rng(5)
km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (Q, k))]
half = torch.empty((Q // 2, k), dtype=torch.uint8, device=dev)
assert L.rsbwt_sample_present_kmers_dev(g.handle, Q // 2, k, k, 8, C.c_void_p(half.data_ptr()), None) == 0
torch.cuda.synchronize()
km[::2][:Q // 2] = half.cpu().numpy()
km = np.ascontiguousarray(km)
vp = lambda a: C.c_void_p(a.ctypes.data)
res = {"queries": Q, "run_bytes": R, "bytes_over_pcie_per_query": k + 16}
for name, pin in (("pageable", False), ("pinned", True)):
    if pin:  # page-locked host buffers (what a caller that cares would pass)
        tk = torch.from_numpy(km).pin_memory()
        tl = torch.empty(Q, dtype=torch.int64).pin_memory()
        tu = torch.empty(Q, dtype=torch.int64).pin_memory()
        a, lo, up = tk.numpy(), tl.numpy().view(np.uint64), tu.numpy().view(np.uint64)
    else:
        a, lo, up = km, np.empty(Q, np.uint64), np.empty(Q, np.uint64)
    f = lambda: L.rsbwt_find_intervals(g.handle, vp(a), Q, k, k, vp(lo), vp(up))
    assert f() == 0
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        f()
    dt = (time.perf_counter() - t) / reps
    res[name + "_q_per_s"] = Q / dt
    res[name + "_ms_per_batch"] = dt * 1e3
    res["hits"] = int((up >= lo).sum())
print(json.dumps(res))
g.close()
