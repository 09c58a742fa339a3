#!/usr/bin/env python3
"""Soak of the C-ABI's thread-safety claim (include/rsbwt.h: every query entry point is re-entrant; the
reference answers from 8 + 64 pool threads on one shared index, src/service/service.cpp:88-89): N host
threads call the host entry points of two shards and their set at random for a fixed time, with batch
sizes from 1 to tens of thousands, and every answer is compared with the answer the same call gave
single-threaded before the soak.  Also records the slowest call of each kind (a stall of seconds in one
call in a few thousand is how the hipMallocAsync problem of profiles/r02d_latency.md showed).
Prints one JSON line; exit code 1 on any mismatch or error.
usage: tools/soak.py [seconds=120] [threads=8] [run_bytes=5e8]"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(float(sys.argv[3])) if len(sys.argv) > 3 else 500000000
L = rsb.lib()
k = 31
shards = []
for s in range(2):
    runs = np.empty(R, np.uint8)
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 31337 + s) == 0
    shards.append(rsb.GpuBWT(runs=runs, for_reads=(s == 1)))  # (one plain shard, one laid out for reads)
    del runs
sset = rsb.ShardSet(shards)
g = shards[0]
n = g.getBWLen()
rng = np.random.default_rng(1)
POOL = 60000
km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (POOL, k))].copy()
rows = rng.integers(0, n, 20000, dtype=np.uint64)
# answers of the whole pools, single-threaded
ref_lo, ref_up = rsb.find_intervals(g, km)
ref_cnt = rsb.count_kmers(g, km)
ref_slo, ref_sup = sset.find_intervals(km)
ref_scnt = sset.count(km)
out = np.zeros((rows.size, 1024), np.uint8)
ln = np.empty(rows.size, np.uint32)
pl = np.empty(rows.size, np.uint32)
assert L.rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, 1024, ln.ctypes.data, pl.ctypes.data) == 0
ref_out, ref_ln, ref_pl = out, ln, pl
ref_1lo, ref_1up = rsb.find_intervals_1mm(g, km[:2000])
# the device-resident hit lists of the set (both shards in one traced and one resumed launch when their tables have one
# depth: csrc/sets.hip, set_hits_1mm_fused): the lists of the first 2000 k-mers, single-threaded
import torch  # noqa: E402
V = 3 * k + 1
DEV = torch.device("cuda", 0)
ptr = lambda t: C.c_void_p(t.data_ptr())
d_km = torch.from_numpy(km[:2000].copy()).to(DEV)
d_pk = torch.empty(2000, dtype=torch.int64, device=DEV)
d_ok = torch.empty(2000, dtype=torch.uint8, device=DEV)
assert L.rsbwt_pack_kmers_dev(ptr(d_km), 2000, k, k, ptr(d_pk), ptr(d_ok), 0, None) == 0
CAP1 = 8 * 2000
_h = torch.zeros((2, CAP1, 4), dtype=torch.int64, device=DEV)
_t = torch.zeros(2, dtype=torch.int64, device=DEV)
_s = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(sset._s, 2000, k), dtype=torch.uint8, device=DEV)
assert L.rsbwt_set_hits_1mm_dev(sset._s, ptr(d_pk), ptr(d_ok), 2000, k, ptr(_h), CAP1, ptr(_t), ptr(_s), None) == 0
torch.cuda.synchronize()
ref_hits = [_h[si, :int(_t[si].item())].cpu().numpy().view(np.uint64) for si in range(2)]
fused_1mm = int(L.rsbwt_set_hits_1mm_is_fused(sset._s, 500, k))
del _h, _t, _s
# the set's device-resident extraction (ONE launch sequence walks both shards: csrc/extract_lines.hip): 4000 rows of each
n_min = min(int(x.getBWLen()) for x in shards)
rows2 = np.stack([rng.integers(0, n_min, 4000), rng.integers(0, n_min, 4000)]).astype(np.uint64)
d_rows2 = torch.from_numpy(rows2.view(np.int64)).to(DEV)
_o = torch.zeros((2, 4000, 512), dtype=torch.uint8, device=DEV)
_l = torch.empty((2, 4000), dtype=torch.int32, device=DEV)
_pl = torch.empty((2, 4000), dtype=torch.int32, device=DEV)
assert L.rsbwt_set_extract_dev(sset._s, ptr(d_rows2), 4000, ptr(_o), 512, ptr(_l), ptr(_pl), None) == 0
torch.cuda.synchronize()
ref_x_len = _l.cpu().numpy().view(np.uint32).copy()
_keep = (torch.arange(512, device=DEV)[None, None, :] < _l.clamp(min=0)[..., None])
ref_x_out = (_o * _keep).cpu().numpy()
del _o, _l, _pl, _keep

stats = {}
errors = []
lock = threading.Lock()
stop_at = time.time() + SECONDS


def note(kind, dt):
    with lock:
        c = stats.setdefault(kind, {"calls": 0, "worst_ms": 0.0, "total_s": 0.0})
        c["calls"] += 1
        c["total_s"] += dt
        c["worst_ms"] = max(c["worst_ms"], dt * 1e3)


def worker(seed):
    r = np.random.default_rng(seed)
    st = torch.cuda.Stream(device=DEV)
    sp = C.c_void_p(st.cuda_stream)
    w_h = torch.zeros((2, 8 * 500, 4), dtype=torch.int64, device=DEV)
    w_t = torch.zeros(2, dtype=torch.int64, device=DEV)
    w_s = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(sset._s, 500, k), dtype=torch.uint8, device=DEV)
    try:
        while time.time() < stop_at:
            op = int(r.integers(0, 8))
            size = int(2 ** r.uniform(0, 15.5))
            a = int(r.integers(0, POOL - size))
            t0 = time.perf_counter()
            if op == 0:
                lo, up = rsb.find_intervals(g, km[a:a + size])
                ok = np.array_equal(lo, ref_lo[a:a + size]) and np.array_equal(up, ref_up[a:a + size])
                kind = "find_intervals"
            elif op == 1:
                ok = np.array_equal(rsb.count_kmers(g, km[a:a + size]), ref_cnt[a:a + size])
                kind = "count"
            elif op == 2:
                lo, up = sset.find_intervals(km[a:a + size])
                ok = np.array_equal(lo, ref_slo[:, a:a + size]) and np.array_equal(up, ref_sup[:, a:a + size])
                kind = "set_find_intervals"
            elif op == 3:
                ok = np.array_equal(sset.count(km[a:a + size]), ref_scnt[a:a + size])
                kind = "set_count"
            elif op == 4:
                m = min(size, 5000)
                b = int(r.integers(0, rows.size - m))
                o = np.zeros((m, 1024), np.uint8)
                l1 = np.empty(m, np.uint32)
                p1 = np.empty(m, np.uint32)
                rc = L.rsbwt_extract(g.handle, rows[b:b + m].ctypes.data, m, o.ctypes.data, 1024, l1.ctypes.data, p1.ctypes.data)
                ok = rc == 0 and np.array_equal(l1, ref_ln[b:b + m]) and np.array_equal(p1, ref_pl[b:b + m])
                fits = l1 != 0xFFFFFFFF
                ok = ok and all(np.array_equal(o[i, :l1[i]], ref_out[b + i, :l1[i]]) for i in np.nonzero(fits)[0][:200])
                kind = "extract"
            elif op == 7:
                m = min(size, 2000)
                b = int(r.integers(0, 4000 - m))
                with torch.cuda.stream(st):
                    x_r = d_rows2[:, b:b + m].contiguous()
                    x_o = torch.zeros((2, m, 512), dtype=torch.uint8, device=DEV)
                    x_l = torch.empty((2, m), dtype=torch.int32, device=DEV)
                    x_p = torch.empty((2, m), dtype=torch.int32, device=DEV)
                    rc = L.rsbwt_set_extract_dev(sset._s, ptr(x_r), m, ptr(x_o), 512, ptr(x_l), ptr(x_p), sp)
                    keep = (torch.arange(512, device=DEV)[None, None, :] < x_l.clamp(min=0)[..., None])
                    got_o = (x_o * keep).cpu().numpy()
                st.synchronize()
                ok = rc == 0 and np.array_equal(x_l.cpu().numpy().view(np.uint32), ref_x_len[:, b:b + m]) and np.array_equal(got_o, ref_x_out[:, b:b + m])
                kind = "set_extract_dev"
            elif op == 6:
                m = min(size, 500)
                b = int(r.integers(0, 2000 - m))
                rc = L.rsbwt_set_hits_1mm_dev(sset._s, C.c_void_p(d_pk.data_ptr() + 8 * b), C.c_void_p(d_ok.data_ptr() + b), m, k,
                                              ptr(w_h), 8 * 500, ptr(w_t), ptr(w_s), sp)
                st.synchronize()
                ok = rc == 0
                for si in range(2):
                    want = ref_hits[si][(ref_hits[si][:, 2] >= b * V) & (ref_hits[si][:, 2] < (b + m) * V)].copy()
                    want[:, 2] -= b * V
                    got = w_h[si, :int(w_t[si].item())].cpu().numpy().view(np.uint64)
                    ok = ok and np.array_equal(got, want)
                kind = "set_hits_1mm_dev"
            else:
                m = min(size, 500)
                b = int(r.integers(0, 2000 - m))
                lo, up = rsb.find_intervals_1mm(g, km[b:b + m])
                ok = np.array_equal(lo, ref_1lo[b:b + m]) and np.array_equal(up, ref_1up[b:b + m])
                kind = "find_intervals_1mm"
            note(kind, time.perf_counter() - t0)
            if not ok:
                with lock:
                    errors.append(f"{kind}: answers differ (size {size}, offset {a})")
                return
    except Exception as e:  # noqa: BLE001
        with lock:
            errors.append(repr(e))


threads = [threading.Thread(target=worker, args=(100 + i,)) for i in range(NT)]
t0 = time.time()
for t in threads:
    t.start()
while any(t.is_alive() for t in threads):  # a line every half minute: a silent GPU run is taken to be hung
    time.sleep(1.0)
    if int(time.time() - t0) % 30 == 0:
        with lock:
            print(f"soak: {int(time.time() - t0)} s, {sum(v['calls'] for v in stats.values())} calls, {len(errors)} errors",
                  file=sys.stderr, flush=True)
for t in threads:
    t.join()
res = {"seconds": round(time.time() - t0, 1), "threads": NT, "run_bytes_per_shard": R, "symbols_per_shard": int(n),
       "set_hits_1mm_dev_runs_as_the_fused_launches": fused_1mm,
       "calls": {k_: {"calls": v["calls"], "mean_ms": round(v["total_s"] / v["calls"] * 1e3, 3), "worst_ms": round(v["worst_ms"], 2)}
                 for k_, v in sorted(stats.items())},
       "errors": errors[:5]}
print(json.dumps(res))
sset.close()
for s in shards:
    s.close()
sys.exit(1 if errors else 0)
