#!/usr/bin/env python3
"""Host entry points of the 1-mismatch search on the bench shard: the dense [m][3k+1] matrices
(rsbwt_find_intervals_1mm: 1.5 KB per 31-mer back over PCIe) against the hit list (rsbwt_hits_1mm: only
the variants that occur leave the search kernel).  Prints one dict."""
import ctypes as C, time, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb
L = rsb.lib()
R = 20000000000; k = 31; M = 400000
d_runs = torch.empty(R, dtype=torch.uint8, device="cuda:0")
assert L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, 1000003, 0, None) == 0
torch.cuda.synchronize()
g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R)); del d_runs; torch.cuda.empty_cache()
rng = np.random.default_rng(5)
km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (M, k))].copy()
half = torch.empty((M // 2, k), dtype=torch.uint8, device="cuda:0")
assert L.rsbwt_sample_present_kmers_dev(g.handle, M // 2, k, k, 8, C.c_void_p(half.data_ptr()), None) == 0
torch.cuda.synchronize(); km[::2] = half.cpu().numpy()
V = 3 * k + 1
lo = np.empty((M, V), np.uint64); up = np.empty((M, V), np.uint64)
hits = np.zeros(8 * M, rsb.bwt.HIT_1MM); nh = C.c_size_t()
vp = lambda a: C.c_void_p(a.ctypes.data)
def timed(f, n=3):
    f(); t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n
d1 = timed(lambda: L.rsbwt_find_intervals_1mm(g.handle, vp(km), M, k, k, vp(lo), vp(up)))
d2 = timed(lambda: L.rsbwt_hits_1mm(g.handle, vp(km), M, k, k, vp(hits), hits.size, C.byref(nh)))
print({"kmers": M, "dense_kmers_per_s": M / d1, "hit_list_kmers_per_s": M / d2, "hits": nh.value, "dense_ms": d1 * 1e3, "hit_list_ms": d2 * 1e3})
