#!/usr/bin/env python3
"""SURVEY section-8 "next" rows on the bench's 20 GB shard, device-resident (inputs and outputs in
HBM, as bench.py measures the exact search), with a roofline each:
  f3  1-mismatch search    rsbwt_find_intervals_1mm_dev   (3k+1 variants per 31-mer, traced + resumed)
  f2  read extraction      rsbwt_extract_dev              (extractPrefix + extractPostfix per row)
Algorithmic bytes: 128 B per distinct window line read by an Occ lookup (+ 40 B per search) for f3,
from the search kernel's own counters; 128 B per LF / select step for f2 (one line holds what a step
needs), counted from the lengths extracted.  Prints one JSON line.
With shards > 1 (BASELINE configs[3]/[4]: the 8 shards of one GPU) the 1-mismatch search runs over a
shard set (rsbwt_set_find_intervals_1mm_dev, tables sized for the set) and rows are extracted from every
shard in turn.
With stride < 512 (5th argument) walks longer than the row buffer are cut where the buffer ends and reported as
not fitting (the reference's reads are at most 100 long; the synthetic stream's are geometric with mean
81 per direction): their steps still count, from the walk kernels' own counters.
usage: tools/bench_rows.py [run_bytes=2e10] [kmers=400000] [rows=2000000] [shards=1] [stride=512]"""
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
M = int(float(sys.argv[2])) if len(sys.argv) > 2 else 400000
NR = int(float(sys.argv[3])) if len(sys.argv) > 3 else 2000000
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
STRIDE = int(sys.argv[5]) if len(sys.argv) > 5 else 512
k, PEAK = 31, 8000.0
L = rsb.lib()
dev = torch.device("cuda", 0)
p = lambda t: C.c_void_p(t.data_ptr())


def ok(rc):
    if rc != 0:
        raise RuntimeError(L.rsbwt_last_error().decode())


shards = []
for s in range(S):
    d_runs = torch.empty(R, dtype=torch.uint8, device=dev)
    ok(L.rsbwt_synth_runs_dev(p(d_runs), R, 1000003 + s, 0, None))
    torch.cuda.synchronize()
    shards.append(rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), ktab_depth=(0 if S == 1 else None)))
    del d_runs
    torch.cuda.empty_cache()
g = shards[0]
sset = rsb.ShardSet(shards)
if S > 1:
    # the select samples of every shard first (built by a shard's first extraction), then one k-mer
    # table depth for the GPU's shards out of the HBM that is left, interleaved (as bench.py)
    one = torch.zeros(1, dtype=torch.int64, device=dev)
    o1 = torch.empty((1, 512), dtype=torch.uint8, device=dev)
    l1 = torch.empty(2, dtype=torch.int32, device=dev)
    for h in shards:
        ok(L.rsbwt_extract_dev(h.handle, p(one), 1, p(o1), 512, p(l1), p(l1[1:]), None))
    torch.cuda.synchronize()
    T = L.rsbwt_set_auto_ktab_depth(sset._s)
    if T >= 2:
        ok(L.rsbwt_set_attach_ktabs(sset._s, T))
n = min(h.getBWLen() for h in shards)
out = {"run_bytes": R, "shards": S, "symbols": int(n), "ktab_depth": g.ktab_depth(), "window_span": g.window_span()}

# ---- f3: 1-mismatch, half of the k-mers drawn from the index ---------------------------------------
gen = torch.Generator(device=dev)
gen.manual_seed(5)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
d_km = lut[torch.randint(0, 4, (M, k), generator=gen, device=dev).long()].contiguous()
half = torch.empty((M // 2, k), dtype=torch.uint8, device=dev)
ok(L.rsbwt_sample_present_kmers_dev(g.handle, M // 2, k, k, 8, p(half), None))
torch.cuda.synchronize()
d_km[::2][:M // 2] = half
V = 3 * k + 1
d_pk = torch.empty(M, dtype=torch.int64, device=dev)
d_ok = torch.empty(M, dtype=torch.uint8, device=dev)
d_lo = torch.empty((S, M, V), dtype=torch.int64, device=dev)
d_up = torch.empty((S, M, V), dtype=torch.int64, device=dev)
d_scr = torch.empty(L.rsbwt_set_1mm_scratch_bytes(sset._s, M, k), dtype=torch.uint8, device=dev)
ok(L.rsbwt_pack_kmers_dev(p(d_km), M, k, k, p(d_pk), p(d_ok), 0, None))
run1 = lambda: ok(L.rsbwt_set_find_intervals_1mm_dev(sset._s, p(d_pk), p(d_ok), M, k, p(d_lo), p(d_up), p(d_scr), None))
w = [0] * 16
# a set whose shards share a table depth searches them all in one traced and one resumed launch, metered by the set
fused = bool(L.rsbwt_set_hits_1mm_is_fused(sset._s, M, k))
for h in shards:  # counters of every shard's resumed search of the m x (3k+1) variants
    ok(L.rsbwt_set_counting(h.handle, 1))
ok(L.rsbwt_set_set_counting(sset._s, 1))
run1()
torch.cuda.synchronize()
for h in shards:
    wi = (C.c_uint64 * 16)()
    ok(L.rsbwt_last_search_counters(h.handle, wi))
    ok(L.rsbwt_set_counting(h.handle, 0))
    w = [a + int(b) for a, b in zip(w, wi)]
ws = (C.c_uint64 * 16)()
ok(L.rsbwt_set_last_search_counters(sset._s, ws))
ok(L.rsbwt_set_set_counting(sset._s, 0))
if fused:
    w = [int(x) for x in ws]
for _ in range(2):
    run1()
torch.cuda.synchronize()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    run1()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
kms = 0.0
for h in ([] if fused else shards):
    buf = (C.c_float * 64)()
    cnt = C.c_size_t()
    ok(L.rsbwt_search_history_ms(h.handle, buf, 2 * reps, C.byref(cnt)))
    kms += sum(buf[:cnt.value]) / reps  # traced + resumed search kernels of one call, every shard
if fused:
    buf = (C.c_float * 64)()
    cnt = C.c_size_t()
    ok(L.rsbwt_set_search_history_ms(sset._s, buf, 2 * reps, C.byref(cnt)))
    kms = sum(buf[:cnt.value]) / reps
alg = w[2] * 128 + S * M * V * 40
hits = int((d_up >= d_lo).sum().item())
out["one_mismatch"] = {
    "kmers": M, "variants_per_kmer": V, "kmers_per_s": M / dt, "variant_searches_per_s": S * M * V / dt,
    "ms_per_call": dt * 1e3, "search_kernels_ms_per_call": kms, "hits": hits,
    "lf_steps_per_variant": w[0] / (S * M * V), "line_reads": w[2], "continuation_line_reads": w[11],
    "roofline": {"bound": "hbm", "achieved": alg / (kms * 1e-3) / 1e9, "peak": PEAK, "unit": "GB/s",
                 "frac": alg / (kms * 1e-3) / 1e9 / PEAK, "kernel": "search_lines_kernel (variants resumed from the trace)",
                 "algorithmic_bytes": alg},
}
# the same search leaving only the variants that occur (rsbwt_hits_1mm_dev), first shard
room = 8 * M
d_hl = torch.empty((room, 4), dtype=torch.int64, device=dev)
d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
d_hscr = torch.empty(L.rsbwt_hits_1mm_scratch_bytes(g.handle, M, k), dtype=torch.uint8, device=dev)
def run_hits():
    ok(L.rsbwt_hits_1mm_dev(g.handle, p(d_pk), p(d_ok), M, k, p(d_hl), room, p(d_tot), p(d_hscr), None))
ok(L.rsbwt_set_counting(g.handle, 1))
run_hits()
torch.cuda.synchronize()
wh = (C.c_uint64 * 16)()
ok(L.rsbwt_last_search_counters(g.handle, wh))
ok(L.rsbwt_set_counting(g.handle, 0))
for _ in range(2):
    run_hits()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    run_hits()
torch.cuda.synchronize()
dth = (time.perf_counter() - t0) / reps
buf = (C.c_float * 64)()
cnt = C.c_size_t()
ok(L.rsbwt_search_history_ms(g.handle, buf, 2 * reps, C.byref(cnt)))
kmh = sum(buf[:cnt.value]) / reps
algh = wh[2] * 128 + M * V * 24 + int(d_tot.item()) * 48  # lines + start record and packed word per search + 16 + 32 B per hit
out["one_mismatch_hit_list_first_shard"] = {
    "kmers": M, "kmers_per_s": M / dth, "variant_searches_per_s": M * V / dth, "ms_per_call": dth * 1e3,
    "search_kernels_ms_per_call": kmh, "hits": int(d_tot.item()),
    "roofline": {"bound": "hbm", "achieved": algh / (kmh * 1e-3) / 1e9, "peak": PEAK, "unit": "GB/s",
                 "frac": algh / (kmh * 1e-3) / 1e9 / PEAK, "algorithmic_bytes": algh},
}
del d_lo, d_up, d_scr, d_hl, d_hscr

# ---- f2: extraction of random rows (the run stream is not a valid BWT: walks end at the '$' they meet) -
stride = STRIDE
rows = torch.randint(0, n, (NR,), generator=gen, device=dev, dtype=torch.int64)
d_out = torch.empty((NR, stride), dtype=torch.uint8, device=dev)
d_len = torch.empty(NR, dtype=torch.int32, device=dev)
d_pl = torch.empty(NR, dtype=torch.int32, device=dev)
run2 = lambda h: ok(L.rsbwt_extract_dev(h.handle, p(rows), NR, p(d_out), stride, p(d_len), p(d_pl), None))
steps, nfit, bases = 0, 0, 0
for h in shards:  # NR rows of every shard (row numbers are per shard: one call per shard)
    run2(h)
    torch.cuda.synchronize()
    ln = d_len.cpu().numpy().view(np.uint32)
    fits = ln != 0xFFFFFFFF
    steps += int(ln[fits].astype(np.int64).sum()) + 2 * int(fits.sum())  # one line per symbol + the two '$' steps
    nfit += int(fits.sum())
    bases += int(ln[fits].astype(np.int64).sum())
# the walk kernels' own counters, from one counting run on the first shard
ok(L.rsbwt_set_counting(g.handle, 1))
run2(g)
torch.cuda.synchronize()
xw = (C.c_uint64 * 16)()
ok(L.rsbwt_last_search_counters(g.handle, xw))
ok(L.rsbwt_set_counting(g.handle, 0))
names = ["passes", "lanes_with_a_row", "steps", "lanes_on_a_continuation", "lines_fetched", "cycles", "cycles_fetch_to_landed", "steps_from_line_hint"]
walk_counters = {"prefix": dict(zip(names, [int(v) for v in xw[:8]])), "postfix": dict(zip(names, [int(v) for v in xw[8:]]))}
if STRIDE != 512:  # walks are cut at the buffer's end: the steps actually taken, from the kernels' counters
    steps = S * (walk_counters["prefix"]["steps"] + walk_counters["postfix"]["steps"])
reps = 3
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps):
    for h in shards:
        run2(h)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
out["extract"] = {
    "rows": S * NR, "stride": stride, "rows_fitting_stride": nfit, "mean_read_length": bases / max(nfit, 1),
    "reads_per_s": S * NR / (ms * 1e-3), "bases_per_s": bases / (ms * 1e-3), "ms_per_call": ms,
    "roofline": {"bound": "hbm", "achieved": steps * 128 / (ms * 1e-3) / 1e9, "peak": PEAK, "unit": "GB/s",
                 "frac": steps * 128 / (ms * 1e-3) / 1e9 / PEAK, "kernel": "extract_prefix_wave_kernel + extract_postfix_wave_kernel",
                 "algorithmic_bytes": steps * 128, "steps": steps},
    "walk_counters_first_shard": walk_counters,
}
sset.close()
for h in shards:
    h.close()
print(json.dumps(out))
