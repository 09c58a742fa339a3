#!/usr/bin/env python3
"""bench.py -- 31-mer exact backward search on a population BWT resident in HBM.

One "step" = one pass of the hot path over one batch: Q ASCII 31-mers already in HBM are packed
to 2 bits and searched (findInterval, src/bwt/query.cpp:24-41) in every shard this rank holds;
with N > 1 ranks the per-shard (lower, upper) arrays are then gathered on rank 0 over RCCL
(every query goes to every shard, results are only concatenated); the gather of batch i
runs behind the search of batch i + 1 (two resident result buffers).

N = 1 default workload = BASELINE.json configs[1]: one ~20 GB shard (2e10 run bytes from the
direct run-stream synthesiser), 1e7 31-mers, half drawn from the index (all 30 LF steps), half
uniform random (terminate early).  `value` counts (query x shard) searches per second.

    python bench.py [--gpus N --steps K --warmup W] [--runs R --queries Q --shards-per-gpu S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BLOCK_BYTES = 128      # algorithmic bytes per distinct block read by an Occ lookup


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--runs", type=float, default=2e10, help="run bytes per shard")
    ap.add_argument("--queries", type=float, default=1e7, help="31-mers per batch")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--shards-per-gpu", type=int, default=1)
    ap.add_argument("--present-frac", type=float, default=0.5)
    ap.add_argument("--cpu-sample", type=float, default=1e6, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(host cores, 32)")
    ap.add_argument("--dir-shift", type=int, default=0)
    ap.add_argument("--ktab-depth", type=int, default=0, help="k-mer table depth (0 = auto, -1 = none)")
    ap.add_argument("--slots", choices=["auto", "on", "off"], default=None,
                    help="single-request search layout (default auto: built while index + slots fit 45 %% of HBM)")
    ap.add_argument("--slot-span", type=int, default=0, help="symbols per slot (0 = auto)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="HIP streams the batches alternate on.  2: packing, start records and the head of "
                         "batch i + 1 overlap the tail of batch i (+2..4 %% searches/s), but two search kernels "
                         "then share the GPU and their event-timed durations no longer price one launch, so "
                         "the roofline line is quoted at 1")
    ap.add_argument("--seed", type=int, default=1)
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import readserver_amd as rsb

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU path", file=sys.stderr)
        sys.exit(1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    L = rsb.lib()
    R, Q, k, S = int(a.runs), int(a.queries), a.k, a.shards_per_gpu
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def ok(rc):
        if rc != 0:
            raise RuntimeError(L.rsbwt_last_error().decode())

    # ---- resident index: S shards per rank, each R run bytes synthesised in HBM ---------------
    t_build0 = time.time()
    shards, host_runs = [], None
    for s in range(S):
        seed = a.seed * 1000003 + (rank * S + s)
        d_runs = torch.empty(R, dtype=torch.uint8, device=dev)
        ok(L.rsbwt_synth_runs_dev(ptr(d_runs), R, seed, local, sp))
        torch.cuda.synchronize()
        slots = a.slots or "auto"  # the library builds slots only while index + slots stay within 45 % of HBM
        g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), device=local, dir_shift=a.dir_shift,
                       ktab_depth=None if a.ktab_depth < 0 else a.ktab_depth,
                       slots={"auto": "auto", "on": True, "off": False}[slots], slot_span=a.slot_span)
        if rank == 0 and s == 0 and world == 1 and a.cpu_sample > 0:
            host_runs = d_runs.cpu().numpy()
        del d_runs
        shards.append(g)
    torch.cuda.empty_cache()
    t_build = time.time() - t_build0
    n_sym = shards[0].getBWLen()

    # ---- the query batch (identical on every rank) ----------------------------------------------
    n_present = int(Q * a.present_frac)
    gen = torch.Generator(device=dev)
    gen.manual_seed(a.seed + 12345)
    codes = torch.randint(0, 4, (Q, k), generator=gen, device=dev, dtype=torch.uint8)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    d_kmers = _lut_chunks(lut, codes)
    del codes
    # present k-mers: every rank draws its share from its own first shard; shares are concatenated
    share = n_present // world
    if share:
        mine = torch.empty((share, k), dtype=torch.uint8, device=dev)
        ok(L.rsbwt_sample_present_kmers_dev(shards[0].handle, share, k, k, a.seed + 7 + rank, ptr(mine), sp))
        torch.cuda.synchronize()
        if world > 1:
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            mine = torch.cat(parts, 0)
        # interleave present and random k-mers so every wave sees the mix
        idx = torch.arange(mine.shape[0], device=dev) * (Q // mine.shape[0])
        d_kmers[idx] = mine
        del mine, idx

    wpq = (k + 31) // 32
    from readserver_amd import sharded
    # Batches can alternate between two HIP streams (--streams 2), each with its own packed-query
    # and result buffers: packing and start records of batch i + 1 then run beside the search of
    # batch i.  With N > 1 the gather of batch i to rank 0 (RCCL) travels behind the search of
    # batch i + 1 either way.
    nst = a.streams
    streams = [torch.cuda.Stream(device=dev) for _ in range(nst)] if nst > 1 else [stream]
    for st in streams:
        st.wait_stream(stream)
    d_packed = [torch.empty((Q, wpq), dtype=torch.int64, device=dev) for _ in range(nst)]
    d_valid = [torch.empty(Q, dtype=torch.uint8, device=dev) for _ in range(nst)]
    gat = sharded.IntervalGatherer(S, Q, dev, depth=2)
    d_lower, d_upper = gat.pair(0)[0], gat.pair(0)[1]
    step_no = [0]

    def step():
        i = step_no[0]
        step_no[0] += 1
        j = i % nst
        with torch.cuda.stream(streams[j]):
            spj = C.c_void_p(streams[j].cuda_stream)
            pair = gat.acquire(i)
            ok(L.rsbwt_pack_kmers_dev(ptr(d_kmers), Q, k, k, ptr(d_packed[j]), ptr(d_valid[j]), local, spj))
            for s, g in enumerate(shards):
                ok(L.rsbwt_find_intervals_dev(g.handle, ptr(d_packed[j]), ptr(d_valid[j]), Q, k,
                                              ptr(pair[0][s]), ptr(pair[1][s]), spj))
            gat.submit(i)

    def barrier():
        gat.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- exact work of one step (counting mode, untimed) -----------------------------------------
    for g in shards:
        ok(L.rsbwt_set_counting(g.handle, 1))
    step()
    torch.cuda.synchronize()
    lf = oc = bl = 0
    for g in shards:
        x, y, z = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ok(L.rsbwt_last_search_work(g.handle, C.byref(x), C.byref(y), C.byref(z)))
        lf, oc, bl = lf + x.value, oc + y.value, bl + z.value
        ph = (C.c_uint64 * 6)()
        npass = C.c_uint64()
        ok(L.rsbwt_last_search_phases(g.handle, ph, C.byref(npass)))
        phases = {"passes": npass.value, "cycles_per_pass": [round(v / max(npass.value, 1)) for v in ph],
                  "names": ["setup", "issue", "wait+park", "rank", "overflow", "update"]}
        ok(L.rsbwt_set_counting(g.handle, 0))

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # per-launch search-kernel time over the timed region: HIP events the library records on the
    # launch stream around every search launch (it keeps the last 64 pairs per handle)
    k_ms = []
    for g in shards:
        buf = (C.c_float * 64)()
        cnt = C.c_size_t()
        ok(L.rsbwt_search_history_ms(g.handle, buf, min(a.steps, 64), C.byref(cnt)))
        k_ms += list(buf[:cnt.value])
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    searches = world * S * Q * a.steps
    value = searches / dt
    ms_per_step = dt / a.steps * 1e3
    avg_kernel_ms = float(np.mean(k_ms))
    alg_bytes = (bl / S) * BLOCK_BYTES + Q * (8 * wpq + 16)  # per launch (one shard)
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9

    out = {
        "metric": "31-mer backward-search queries/sec on popBWT",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": ("configs[1]: single BWT shard resident in one MI355X's HBM, batched 31-mer exact backward search"
                         if world == 1 and S == 1 else
                         f"configs[2]-style: {world * S} shards over {world} GPUs, RCCL gather of intervals"),
            "run_bytes_per_shard": R, "symbols_per_shard": int(n_sym), "shards_per_gpu": S,
            "queries_per_batch": Q, "k": k, "present_fraction": a.present_frac,
            "mean_lf_steps_per_search": lf / (S * Q), "dir_shift": shards[0].dir_shift(),
            "ktab_depth": shards[0].ktab_depth(), "slot_span": shards[0].slot_span(),
            "slot_overflow_blocks": int(shards[0].slot_overflow_blocks()),
            "index_hbm_bytes_per_shard": int(shards[0].hbm_bytes()), "index_build_s": round(t_build, 2),
            "value_counts": "query x shard searches (= queries at 1 shard)",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic(R, Q),
            "kernel": ("search_wave_kernel" if (shards[0].slot_span() or shards[0].dir_shift() == 8)
                       and os.environ.get("RSBWT_SEARCH_KERNEL", "w")[0] != "o" else "search_kernel"),
            "kernel_ms": avg_kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes, "block_reads_per_launch": bl / S,
            "occ_lookups_per_launch": oc / S, "phase_stamps": phases,
        },
    }

    if rank == 0 and world == 1 and host_runs is not None:
        out["cpu_baseline"] = cpu_baseline(a, host_runs, d_kmers, d_lower[0], d_upper[0], Q, k)
    if rank == 0:
        print(json.dumps(out), flush=True)
    for g in shards:
        g.close()
    if world > 1:
        dist.destroy_process_group()


def _lut_chunks(lut, codes):
    import torch
    out = torch.empty_like(codes)
    step = 1 << 22
    for i in range(0, codes.shape[0], step):
        out[i:i + step] = lut[codes[i:i + step].long()]
    return out


def _pmc_traffic(R, Q):
    """HBM bytes per search launch from the committed rocprofv3 PMC pass of this same command
    (profiles/<round>_pmc.json, corrected as MI355X_MICROARCH.md prescribes), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        if int(d["run_bytes_per_shard"]) == R and int(d["queries_per_batch"]) == Q:
            return d["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(a, host_runs, d_kmers, d_lower, d_upper, Q, k):
    """The oracle (CPU restatement of the reference algorithm) timed on this host, on a bounded
    sample of the same batch against the same shard; also re-checks the GPU answers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    orc = oracle_binding.load()
    t0 = time.time()
    ix = orc.from_runs(host_runs)
    t_index = time.time() - t0
    m = min(int(a.cpu_sample), Q)
    sel = np.linspace(0, Q - 1, m).astype(np.int64)
    import torch
    sel_t = torch.from_numpy(sel).to(d_kmers.device)
    km = d_kmers[sel_t].cpu().numpy()
    glo = d_lower[sel_t].cpu().numpy().view(np.uint64)
    gup = d_upper[sel_t].cpu().numpy().view(np.uint64)
    threads = a.cpu_threads or min(os.cpu_count() or 1, 32)
    t0 = time.perf_counter()
    lo, up = ix.find_intervals(km, nthreads=threads)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    m1 = max(1, m // 16)
    ix.find_intervals(km[:m1], nthreads=1)
    dt1 = time.perf_counter() - t1
    match = bool(np.array_equal(lo, glo) and np.array_equal(up, gup))
    return {
        "value": m / dt, "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"{m} of the batch's {Q} k-mers (evenly spaced) on the same shard; oracle/rlebwt_oracle.c, "
                  f"{threads} POSIX threads sharing one index",
        "single_thread_value": m1 / dt1, "host_cores": os.cpu_count(),
        "index_build_s": round(t_index, 2), "gpu_matches_oracle_on_sample": match,
    }


if __name__ == "__main__":
    main()
