#!/usr/bin/env python3
"""Open / use / close cycles of handles and shard sets, watching the device's free memory: whatever an
open handle allocates on the way (k-mer tables, select samples, scratch buffers, per-call contexts,
service objects) must be gone after its close.  Prints one JSON line; exit code 1 if free HBM ends
more than 64 MB below where it started.
usage: tools/leak_check.py [cycles=40] [run_bytes=2e8]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402

CYCLES = int(sys.argv[1]) if len(sys.argv) > 1 else 40
R = int(float(sys.argv[2])) if len(sys.argv) > 2 else 200000000
L = rsb.lib()
torch.cuda.init()
rng = np.random.default_rng(2)
k = 31
km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (50000, k))].copy()
runs = [np.empty(R, np.uint8) for _ in range(2)]
for i, r in enumerate(runs):
    assert L.rsbwt_synth_runs_host(r.ctypes.data, R, 77 + i) == 0
free = []


def cycle(i):
    shards = [rsb.GpuBWT(runs=r, ktab_depth=(None if i % 3 == 0 else 0), window_span=(0 if i % 2 else 1500)) for r in runs]
    ss = rsb.ShardSet(shards)
    if i % 3 == 0:
        assert L.rsbwt_set_attach_ktabs(ss._s, 0) == 0
    rsb.find_intervals(shards[0], km[: 1 + (i * 997) % 50000])
    ss.find_intervals(km[:20000])
    ss.count(km[:30000])
    rsb.find_intervals_1mm(shards[1], km[:300])
    rsb.hits_1mm_batch(shards[1], km[:3000])
    rows = rng.integers(0, shards[0].getBWLen(), 5000, dtype=np.uint64)
    out = np.zeros((rows.size, 1024), np.uint8)
    ln = np.empty(rows.size, np.uint32)
    pl = np.empty(rows.size, np.uint32)
    assert L.rsbwt_extract(shards[0].handle, rows.ctypes.data, rows.size, out.ctypes.data, 1024, ln.ctypes.data, pl.ctypes.data) == 0
    # the set-level forms of configs[3] / configs[4], host and device-resident (the fused 1-mismatch launches when the
    # tables have one depth, else the set's side streams; the walks side by side on them)
    ss.hits_1mm(km[:400])
    try:  # (the mirror raises when a read of this random stream is longer than the buffer: the call was made all the same)
        ss.extract(np.repeat(np.arange(2), 500), rows[:1000].astype(np.int64) % min(g.getBWLen() for g in shards), stride=1024)
    except rsb.RsbwtError:
        pass
    p_ = lambda t: C.c_void_p(t.data_ptr())
    m = 700
    d_km = torch.from_numpy(km[:m].copy()).cuda()
    d_pk = torch.empty(m, dtype=torch.int64, device="cuda")
    d_ok = torch.empty(m, dtype=torch.uint8, device="cuda")
    assert L.rsbwt_pack_kmers_dev(p_(d_km), m, k, k, p_(d_pk), p_(d_ok), 0, None) == 0
    d_h = torch.empty((2, 8 * m, 4), dtype=torch.int64, device="cuda")
    d_t = torch.empty(2, dtype=torch.int64, device="cuda")
    d_s = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(ss._s, m, k), dtype=torch.uint8, device="cuda")
    assert L.rsbwt_set_hits_1mm_dev(ss._s, p_(d_pk), p_(d_ok), m, k, p_(d_h), 8 * m, p_(d_t), p_(d_s), None) == 0
    d_r = torch.from_numpy((rows[:2000].astype(np.int64) % min(g.getBWLen() for g in shards)).reshape(2, 1000)).cuda()
    d_o = torch.empty((2, 1000, 256), dtype=torch.uint8, device="cuda")
    d_l = torch.empty((2, 1000), dtype=torch.int32, device="cuda")
    d_p = torch.empty((2, 1000), dtype=torch.int32, device="cuda")
    assert L.rsbwt_set_extract_dev(ss._s, p_(d_r), 1000, p_(d_o), 256, p_(d_l), p_(d_p), None) == 0
    torch.cuda.synchronize()
    del d_km, d_pk, d_ok, d_h, d_t, d_s, d_r, d_o, d_l, d_p
    torch.cuda.empty_cache()
    tr, svc = C.c_void_p(), C.c_void_p()
    assert L.rsbwt_transport_inproc(C.byref(tr)) == 0
    assert L.rsbwt_service_create(ss._s, tr, 100, 256, 1, C.byref(svc)) == 0
    assert L.rsbwt_service_start(svc) == 0
    msg = bytes([0x08, 0x01, 0x10, 0x01, 0x1A, k]) + km[0].tobytes()
    buf = (C.c_uint8 * len(msg)).from_buffer_copy(msg)
    for _ in range(50):
        L.rsbwt_transport_push_request(tr, buf, len(msg))
    rb = (C.c_uint8 * 512)()
    n = C.c_size_t()
    for _ in range(50 * 2 * 2):
        assert L.rsbwt_transport_pop_reply(tr, 1, rb, 512, C.byref(n), 20_000_000) == 0
    L.rsbwt_transport_close(tr)
    assert L.rsbwt_service_stop(svc) == 0
    L.rsbwt_service_free(svc)
    L.rsbwt_transport_free(tr)
    ss.close()
    for g in shards:
        g.close()


for i in range(CYCLES):
    cycle(i)
    torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info()[0])
    if i % 10 == 9:
        print(f"cycle {i + 1}: free {free[-1] / 1e9:.3f} GB", file=sys.stderr, flush=True)
drift = free[2] - free[-1]  # the first cycles pay one-time runtime allocations
print(json.dumps({"cycles": CYCLES, "run_bytes_per_shard": R, "free_after_cycle_3": free[2], "free_after_last": free[-1],
                  "drift_bytes": drift}))
sys.exit(1 if drift > (64 << 20) else 0)
