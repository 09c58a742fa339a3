#!/usr/bin/env python3
"""Time of rsbwt_pack / unpack_interval_pairs_dev for one batch of the bench (8 x 10^7 pairs = 1.28 GB):
what a rank pays per batch, at N > 1, to send 0.8 GB instead."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb
L = rsb.lib()
n = 80000000
pairs = torch.randint(0, 1 << 39, (n, 2), dtype=torch.int64, device="cuda:0")
pairs[:, 1] = pairs[:, 0] + 5
pk = torch.empty(L.rsbwt_packed_pairs_bytes(n), dtype=torch.uint8, device="cuda:0")
back = torch.empty_like(pairs)
p = lambda t: C.c_void_p(t.data_ptr())
res = {}
for name, f in (("pack_ms", lambda: L.rsbwt_pack_interval_pairs_dev(p(pairs), n, p(pk), None, 0, None)),
                ("unpack_ms", lambda: L.rsbwt_unpack_interval_pairs_dev(p(pk), n, p(back), 0, None))):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    res[name] = e0.elapsed_time(e1) / 10
res["exact"] = bool(torch.equal(back, pairs))
res["pairs"] = n
print(json.dumps(res))
