#!/usr/bin/env python3
"""bench.py -- 31-mer exact backward search on a population BWT resident in HBM.

One "step" = one pass of the hot path over one batch: Q ASCII 31-mers already in HBM are packed
to 2 bits (once) and searched (findInterval, src/bwt/query.cpp:24-41) in EVERY shard this rank
holds by one fused launch ((query, shard) pairs drawn from per-shard pools); with N > 1 ranks the
per-shard (lower, upper) arrays are then gathered on rank 0 over RCCL (every query goes
to every shard, results are only concatenated); the gather of batch i runs behind the search of
batch i + 1 (two resident result buffers).

Default workload = the per-GPU load of BASELINE.json configs[2]: 8 shards of ~20 GB (2e10 run bytes
each, from the direct run-stream synthesiser) resident in one MI355X, 1e7 31-mers per batch, half
of them drawn from the local shards (all 30 LF steps in the shard they come from), half uniform
random (terminate early).  `value` counts (query x shard) searches per second = S/s; Q/s = S/s /
shards.  `--shards-per-gpu 1` is configs[1] (one shard, deepest k-mer table).

    python bench.py [--gpus N --steps K --warmup W] [--runs R --queries Q --shards-per-gpu S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks
(self_launch below: the parent makes no GPU call, starts torch.distributed.run as a child, relays
rank 0's line and exits with the child's code).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LINE_BYTES = 128       # algorithmic bytes per distinct window line read by an Occ lookup
SEARCH_BYTES = 40      # per (query, shard) search: 16 B start record + 8 B packed word read, 16 B result written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--runs", type=float, default=2e10, help="run bytes per shard")
    ap.add_argument("--queries", type=float, default=1e7, help="31-mers per batch")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--shards-per-gpu", type=int, default=8)
    ap.add_argument("--present-frac", type=float, default=0.5)
    ap.add_argument("--same-shards", action="store_true",
                    help="every shard of a rank holds the same stream, so a present k-mer is present in all of them "
                         "(the work profile of a real population: each shard sees configs[1]'s mix)")
    ap.add_argument("--stream", choices=["mixed", "long"], default="mixed",
                    help="run-length mix of the synthetic stream: mixed = mean ~10.4 symbols per unit; "
                         "long = mostly 31-symbol units of long runs (mean ~25), as in a deep population BWT")
    ap.add_argument("--cpu-sample", type=float, default=1e6, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every CPU this process may run on")
    ap.add_argument("--ktab-depth", type=int, default=0, help="k-mer table depth (0 = auto, -1 = none)")
    ap.add_argument("--window-span", type=int, default=0, help="symbols per window line (0 = from the data)")
    ap.add_argument("--separate-arrays", action="store_true",
                    help="results as lower[S][Q] and upper[S][Q] (two scattered 8-byte stores per search) instead of "
                         "{lower, upper}[S][Q] pairs (one 16-byte store)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal where only one GPU exists: every rank uses GPU 0 and the collectives run over "
                         "gloo through host copies (RCCL refuses two ranks on one device).  Exercises this script's "
                         "multi-rank logic, not xGMI: its numbers mean nothing")
    ap.add_argument("--verify-all-shards", action="store_true",
                    help="after the timed region, hold EVERY resident shard's answers to the oracle on a sample of the "
                         "batch (default: shard 0 only, inside cpu_baseline): regenerates each shard's run bytes, copies "
                         "them to the host and builds the oracle's index over them, ~30 s per 20 GB shard")
    ap.add_argument("--no-single-check", action="store_true",
                    help="skip the single-shard (configs[1]) launches after the timed region: profile passes want "
                         "only the fused launches under the kernel's name")
    ap.add_argument("--gather-unpacked", action="store_true",
                    help="N > 1: gather the 16-byte {lower, upper} pairs as they are instead of their 10-byte form")
    ap.add_argument("--counts", action="store_true",
                    help="variant (N = 1 only, not the headline): the service's count path -- rsbwt_set_count_dev, one u64 "
                         "count per (query, shard) instead of the interval pairs; checked against the pairs of one launch")
    ap.add_argument("--hold-gb", type=float, default=0.0,
                    help="rehearsal aid: hold this much HBM while the k-mer tables are sized, as rank 0 of an N-GPU "
                         "job holds the gathered intervals (12.8 GB of 10-byte records at N = 8; 20.5 GB unpacked)")
    ap.add_argument("--seed", type=int, default=1)
    return ap.parse_args()


def kernel_source_sha():
    """Identifies the search kernel + layout a PMC traffic figure was measured on."""
    h = hashlib.sha256()
    for f in ("search_lines.hip", "search_solo.h", "wave_lines.h", "line_format.h", "rank_device.h"):
        h.update(open(os.path.join(ROOT, "readserver_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def search_kernel_name(nshards, n=0, ktab_depth=0, span=0):
    """The kernel launch_search (csrc/search_lines.hip) picks: one lane per search for a full batch on
    a single shard behind a deep k-mer table (4 n / 4^T <= S: capi.hip, search_dev), lane pairs
    otherwise; RSBWT_SEARCH_KERNEL overrides."""
    e = os.environ.get("RSBWT_SEARCH_KERNEL", "auto")
    narrow = ktab_depth >= 2 and ((n >> (2 * ktab_depth)) << 2) <= span
    solo = e == "solo" or (e not in ("pair", "solo") and nshards == 1 and narrow)
    return "search_solo_kernel" if solo else "search_lines_kernel"


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:  # cgroup v2 quota
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch_cmd(argv, n, port):
    """The command a plain `python bench.py --gpus N ...` turns itself into: one rank per GPU under
    torch.distributed.run on this node, every rank running this same script with the same arguments
    (the reference's shape: every request goes to every partition, src/service/server.cpp:124,578)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv):
    """Parent of an N > 1 run started without a launcher.  It must not have touched the GPU (no
    torch.cuda.*, no rsb.lib()): a process that has initialised HIP may not be replaced or forked into
    the ranks, so the ranks are children started from a clean process and this one only waits."""
    import subprocess
    assert "torch" not in sys.modules and "readserver_amd" not in sys.modules, "the launcher parent imports no GPU code"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks here
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = self_launch_cmd(argv, a.gpus, int(os.environ.get("BENCH_MASTER_PORT", "0")) or free_port())
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, env=env)  # stdout/stderr inherited: rank 0's JSON line arrives on our stdout
    try:
        rc = p.wait()
    except KeyboardInterrupt:
        p.terminate()
        rc = p.wait()
    sys.exit(rc if rc >= 0 else 128 - rc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a, sys.argv[1:])
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
    if a.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if a.rehearse_on_one_gpu else dev  # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
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
    style = (1 << 63) if a.stream == "long" else 0
    for s in range(S):
        seed = style | (a.seed * 1000003 + (rank * S + (0 if a.same_shards else s)))
        d_runs = torch.empty(R, dtype=torch.uint8, device=dev)
        ok(L.rsbwt_synth_runs_dev(ptr(d_runs), R, seed, local, sp))
        torch.cuda.synchronize()
        # tables are sized afterwards, for all shards of the GPU together (explicit depth: now)
        g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), device=local,
                       ktab_depth=(a.ktab_depth if a.ktab_depth > 0 else None), window_span=a.window_span)
        if rank == 0 and world == 1 and s == 0 and a.cpu_sample > 0:  # the CPU baseline is an N = 1 leg
            host_runs = d_runs.cpu().numpy()
        del d_runs
        torch.cuda.empty_cache()
        shards.append(g)
    sset = rsb.ShardSet(shards)
    n_sym = shards[0].getBWLen()
    # the batch's buffers first (rank 0 also holds the gathered intervals of all ranks), then the k-mer
    # tables out of what HBM is left: one depth for every shard of the job
    wpq = (k + 31) // 32
    from readserver_amd import sharded
    d_packed = torch.empty((Q, wpq), dtype=torch.int64, device=dev)
    d_valid = torch.empty(Q, dtype=torch.uint8, device=dev)
    d_kmers = torch.empty((Q, k), dtype=torch.uint8, device=dev)
    # N > 1: the pairs travel as 10-byte {lower:40, width:40} records (exact for every interval; 5/8 of the
    # bytes): an xGMI link moves ~77 GB/s per direction, so 1.28 GB of 16-byte pairs per peer and batch would
    # take longer than the 12 ms search that produced them
    wire_packed = world > 1 and not a.separate_arrays and not a.gather_unpacked
    gat = sharded.IntervalGatherer(S, Q, cdev, depth=2, interleaved=not a.separate_arrays, packed=wire_packed, wire_device=cdev)
    d_res = [torch.empty_like(gat.pair(i), device=dev) for i in range(2)] if cdev != dev else None
    hold = torch.empty(int(a.hold_gb * (1 << 30)), dtype=torch.uint8, device=dev) if a.hold_gb > 0 else None
    if a.ktab_depth == 0:
        # The library's own rule leaves two thirds of the free HBM to a caller it knows nothing about.
        # Here every buffer of the job already exists (rank 0's gathered intervals included), so the
        # tables may take what is left but a reserve for the search's start records (1.4 GB), the
        # single-shard check and RCCL's own buffers: the same depth then fits rank 0 of an 8-GPU job
        # (13-20 GB of gathered intervals) and a lone GPU, and the scaling curve compares like with like.
        T = L.rsbwt_set_auto_ktab_depth(sset._s)
        free_b = torch.cuda.mem_get_info(dev)[0]
        if a.rehearse_on_one_gpu:  # the ranks share one GPU: each sizes its tables out of its share
            if world > 1:
                dist.barrier()  # every rank's buffers exist before anyone looks at what is free
                free_b = torch.cuda.mem_get_info(dev)[0]
            free_b //= world
            while T >= 2 and S * 8 * 4 ** T > free_b // 2:
                T -= 1
        while T < 16 and T >= 2 and S * 8 * 4 ** (T + 1) <= free_b - (8 << 30) and 4 ** (T + 1) <= n_sym:
            T += 1
        if world > 1:
            tt = torch.tensor([T], dtype=torch.int64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            T = int(tt.item())
        if T >= 2:
            ok(L.rsbwt_set_attach_ktabs(sset._s, T))
    t_build = time.time() - t_build0

    # ---- the query batch (identical on every rank) ----------------------------------------------
    n_present = int(Q * a.present_frac)
    gen = torch.Generator(device=dev)
    gen.manual_seed(a.seed + 12345)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for i in range(0, Q, 1 << 22):
        j = min(Q, i + (1 << 22))
        d_kmers[i:j] = lut[torch.randint(0, 4, (j - i, k), generator=gen, device=dev, dtype=torch.uint8).long()]
    # present k-mers: every rank draws its share evenly from its shards; shares are concatenated
    share = n_present // world
    if share:
        per = [share // S + (1 if i < share % S else 0) for i in range(S)]
        parts = []
        for s, m in enumerate(per):
            if m:
                t = torch.empty((m, k), dtype=torch.uint8, device=dev)
                ok(L.rsbwt_sample_present_kmers_dev(shards[s].handle, m, k, k, a.seed + 7 + rank * S + s, ptr(t), sp))
                parts.append(t)
        torch.cuda.synchronize()
        mine = torch.cat(parts, 0)
        mine = mine[torch.randperm(mine.shape[0], device=dev, generator=gen)]  # the shards' k-mers interleaved
        if world > 1:
            mine_c = mine.to(cdev)
            allp = [torch.empty_like(mine_c) for _ in range(world)]
            dist.all_gather(allp, mine_c)
            mine = torch.cat(allp, 0).to(dev)
        # interleave present and random k-mers so every wave sees the mix
        idx = torch.arange(mine.shape[0], device=dev) * (Q // mine.shape[0])
        d_kmers[idx] = mine
        del mine, idx, parts

    step_no = [0]
    d_counts = None
    if a.counts:
        if world != 1 or a.separate_arrays:
            raise SystemExit("bench.py --counts: one GPU, default result layout")
        a.cpu_sample, a.no_single_check = 0, True
        d_counts = torch.empty((S, Q), dtype=torch.int64, device=dev)

    def step():
        i = step_no[0]
        step_no[0] += 1
        pair = gat.acquire(i)
        host_pair = None
        if d_res is not None:  # rehearsal: search into HBM, gather from a host copy
            host_pair, pair = pair, d_res[i % 2]
        ok(L.rsbwt_pack_kmers_dev(ptr(d_kmers), Q, k, k, ptr(d_packed), ptr(d_valid), local, sp))
        if a.counts:
            ok(L.rsbwt_set_count_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(d_counts), sp))
        elif a.separate_arrays:
            ok(L.rsbwt_set_find_intervals_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(pair[0]), ptr(pair[1]), sp))
        else:
            ok(L.rsbwt_set_find_interval_pairs_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(pair), sp))
        if host_pair is not None and not wire_packed:
            host_pair.copy_(pair)
        gat.submit(i, source=pair if wire_packed else None)

    def barrier():
        gat.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- exact work of one step (counting mode, untimed) -----------------------------------------
    ok(L.rsbwt_set_set_counting(sset._s, 1))
    step()
    torch.cuda.synchronize()
    w = (C.c_uint64 * 16)()
    ok(L.rsbwt_set_last_search_counters(sset._s, w))
    ok(L.rsbwt_set_set_counting(sset._s, 0))
    lf, oc, ln, kt, hops, npass = w[0], w[1], w[2], w[3], w[11], w[10]
    phases = {"passes": npass, "cycles_per_pass": [round(w[4 + i] / max(npass, 1)) for i in range(6)],
              "names": ["setup", "issue", "wait", "rank", "exchange", "update"]}

    if a.counts:  # the counts of a launch against the interval pairs of the same batch
        pr = gat.pair(0)
        ok(L.rsbwt_set_find_interval_pairs_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(pr), sp))
        step()
        torch.cuda.synchronize()
        want = torch.where(pr[..., 1] >= pr[..., 0], pr[..., 1] - pr[..., 0] + 1, torch.zeros_like(pr[..., 0]))
        if not torch.equal(want, d_counts):
            raise SystemExit("bench.py --counts: counts differ from upper - lower + 1 of the same batch")
        del pr, want
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # per-launch search-kernel time over the timed region: HIP events the library records on the
    # launch stream around every search launch (it keeps the last 64 pairs)
    buf = (C.c_float * 64)()
    cnt = C.c_size_t()
    ok(L.rsbwt_set_search_history_ms(sset._s, buf, min(a.steps, 64), C.byref(cnt)))
    k_ms = list(buf[:cnt.value])
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the gathered blocks are what the ranks sent: checksums of the last batch, rank by rank
    gather_verified = None
    if world > 1:
        last = step_no[0] - 1
        sent = gat.wire(last) if wire_packed else gat.pair(last)
        mine_sum = sent.sum(dtype=torch.int64).reshape(1).to(cdev)
        exact = torch.ones(1, dtype=torch.int64, device=cdev)
        if wire_packed:  # the 10-byte form carries this rank's pairs exactly
            src = d_res[last % 2] if d_res is not None else gat.pair(last)
            sharded.pack_pairs(src, check=True)  # raises if a pair of this rank does not fit the 10-byte record
            back = gat.unpack_block(sent.to(src.device))
            exact[0] = int(torch.equal(back, src))
            del back
        sums = [torch.empty_like(mine_sum) for _ in range(world)]
        dist.all_gather(sums, mine_sum)
        dist.all_reduce(exact, op=dist.ReduceOp.MIN)
        if rank == 0:
            got = gat.result(last)
            gather_verified = bool(exact.item()) and all(int(got[r].sum(dtype=torch.int64).item()) == int(sums[r].item()) for r in range(world))

    searches = world * S * Q * a.steps
    value = searches / dt
    ms_per_step = dt / a.steps * 1e3
    avg_kernel_ms = float(np.mean(k_ms))
    alg_bytes = ln * LINE_BYTES + S * Q * SEARCH_BYTES  # per launch, the search kernel's own reads and writes
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
    # the whole step: + packing (k + 8 wpq + 1 B per query) and the start records (8-byte table read +
    # 16 B written per search)
    step_bytes = alg_bytes + Q * (k + 8 * wpq + 1) + S * Q * 24
    hbm = sum(int(g.hbm_bytes()) for g in shards)

    # ---- configs[1] on the same resident data: shard 0 alone (same kernel, one shard) -----------
    single = None
    if S > 1 and not a.no_single_check:
        g0 = shards[0]
        lo1 = torch.empty(Q, dtype=torch.int64, device=dev)
        up1 = torch.empty(Q, dtype=torch.int64, device=dev)
        ok(L.rsbwt_set_counting(g0.handle, 1))
        ok(L.rsbwt_find_intervals_dev(g0.handle, ptr(d_packed), ptr(d_valid), Q, k, ptr(lo1), ptr(up1), sp))
        torch.cuda.synchronize()
        w1 = (C.c_uint64 * 16)()
        ok(L.rsbwt_last_search_counters(g0.handle, w1))
        ok(L.rsbwt_set_counting(g0.handle, 0))
        for _ in range(2):
            ok(L.rsbwt_find_intervals_dev(g0.handle, ptr(d_packed), ptr(d_valid), Q, k, ptr(lo1), ptr(up1), sp))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 10
        for _ in range(n1):
            ok(L.rsbwt_find_intervals_dev(g0.handle, ptr(d_packed), ptr(d_valid), Q, k, ptr(lo1), ptr(up1), sp))
        torch.cuda.synchronize()
        d1 = (time.perf_counter() - t1) / n1
        b1 = (C.c_float * 64)()
        ok(L.rsbwt_search_history_ms(g0.handle, b1, n1, C.byref(cnt)))
        km1 = float(np.mean(list(b1[:cnt.value])))
        single = {"searches_per_s": Q / d1, "kernel": search_kernel_name(1, int(g0.getBWLen()), g0.ktab_depth(), g0.window_span()), "kernel_ms": km1, "mean_lf_steps_per_search": w1[0] / Q,
                  "roofline_frac": (w1[2] * LINE_BYTES + Q * SEARCH_BYTES) / (km1 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del lo1, up1

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
                         f"configs[2]: {world * S} of 64 suffix-shards over {world} GPU(s), {S} per GPU, exact match, "
                         "one fused launch per GPU and batch" + (", RCCL gather of intervals to rank 0" if world > 1 else
                                                                 " (one GPU: no gather)")),
            "value_counts": "S/s = (query x shard) searches per second; Q/s = value / shards",
            "queries_per_s_all_shards": value / (world * S),
            "shards_per_gpu": S, "shards": world * S, "run_bytes_per_shard": R, "symbols_per_shard": int(n_sym),
            "stream": a.stream + (", the same in every shard of a GPU" if a.same_shards else ""),
            "queries_per_batch": Q, "k": k, "present_fraction": a.present_frac,
            "mean_lf_steps_per_search": lf / (S * Q), "ktab_depth": shards[0].ktab_depth(),
            "window_span": shards[0].window_span(), "far_lines_per_shard": int(shards[0].far_lines()),
            "spilled_position_fraction": shards[0].spilled_symbols() / max(int(n_sym), 1),
            "index_hbm_bytes_per_gpu": hbm, "index_bytes_per_run_byte": (hbm - sum(8 * 4 ** g.ktab_depth() for g in shards)) / (S * R),
            "index_build_s": round(t_build, 2),
            "gather_verified": gather_verified,
            "gathered_as": (None if world == 1 else "10-byte {lower:40, width:40} records (rsbwt_pack_interval_pairs_dev), exact"
                            if wire_packed else "16-byte pairs"),
            "results": ("counts[S][Q] (rsbwt_set_count_dev: the service's count path, NOT the headline)" if a.counts else
                        "lower[S][Q], upper[S][Q]" if a.separate_arrays else "{lower, upper}[S][Q] pairs (BWTInterval)"),
            "multi_gpu": ("REHEARSAL on one GPU over gloo: not a measurement" if a.rehearse_on_one_gpu else "measured" if world > 1 else "one GPU; the N > 1 gather path is covered by the gloo world-2 test only"),
            "single_shard_check": single,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic(R, Q, S, k, a.stream),
            "kernel": search_kernel_name(S, int(n_sym), shards[0].ktab_depth(), shards[0].window_span()), "kernel_ms": avg_kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes, "line_reads_per_launch": ln,
            "continuation_line_reads_per_launch": hops, "occ_lookups_per_launch": oc,
            "ktab_starts_per_launch": kt, "phase_stamps": phases,
            "frac_of_step": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
    }

    if rank == 0 and host_runs is not None:
        pair = gat.pair(step_no[0] - 1).to(dev)
        lo0, up0 = (pair[0][0], pair[1][0]) if a.separate_arrays else (pair[0, :, 0], pair[0, :, 1])
        out["cpu_baseline"] = cpu_baseline(a, host_runs, d_kmers, lo0, up0, Q, k)
    elif rank == 0:
        out["cpu_baseline"] = None  # timed at N = 1 only (or switched off with --cpu-sample 0)
    if rank == 0 and a.verify_all_shards:
        out["config"]["shards_matching_oracle"] = verify_shards(a, L, gat.pair(step_no[0] - 1).to(dev), d_kmers, S, R, Q, k, style,
                                                              rank, local, dev, sp)
    if rank == 0:
        print(json.dumps(out), flush=True)
    sset.close()
    for g in shards:
        g.close()
    if world > 1:
        dist.destroy_process_group()


def _pmc_traffic(R, Q, S, k, stream):
    """HBM bytes per search launch from the committed rocprofv3 PMC pass of this same command
    (profiles/pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes) -- only if that pass
    was measured on this very kernel and layout source; otherwise None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        if (d["kernel_source_sha"] == kernel_source_sha() and int(d["run_bytes_per_shard"]) == R
                and int(d["queries_per_batch"]) == Q and int(d["shards_per_gpu"]) == S and int(d["k"]) == k
                and d.get("stream", "mixed") == stream):
            return d["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def verify_shards(a, L, pair, d_kmers, S, R, Q, k, style, rank, local, dev, sp):
    """Every resident shard against the oracle: [true/false per shard] on 1e5 evenly spaced k-mers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_binding
    import torch
    orc = oracle_binding.load()
    m = min(100000, Q)
    sel_t = torch.from_numpy(np.linspace(0, Q - 1, m).astype(np.int64)).to(dev)
    km = d_kmers[sel_t].cpu().numpy()
    res = []
    for s in range(S):
        seed = style | (a.seed * 1000003 + (rank * S + (0 if a.same_shards else s)))
        d_runs = torch.empty(R, dtype=torch.uint8, device=dev)
        assert L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, seed, local, sp) == 0
        torch.cuda.synchronize()
        runs = d_runs.cpu().numpy()
        del d_runs
        torch.cuda.empty_cache()
        ix = orc.from_runs(runs)
        lo, up = ix.find_intervals(km, nthreads=a.cpu_threads or usable_cpus())
        if a.separate_arrays:
            glo, gup = pair[0][s][sel_t], pair[1][s][sel_t]
        else:
            glo, gup = pair[s, :, 0][sel_t], pair[s, :, 1][sel_t]
        res.append(bool(np.array_equal(lo, glo.cpu().numpy().view(np.uint64)) and
                        np.array_equal(up, gup.cpu().numpy().view(np.uint64))))
        print(f"bench.py: shard {s} {'matches' if res[-1] else 'DIFFERS FROM'} the oracle on {m} k-mers", file=sys.stderr, flush=True)
        del ix, runs
    return res


def cpu_baseline(a, host_runs, d_kmers, d_lower, d_upper, Q, k):
    """The oracle (CPU restatement of the reference algorithm) timed on this host, on a bounded
    sample of the same batch against the same shard (shard 0); also re-checks the GPU answers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    import torch
    orc = oracle_binding.load()
    t0 = time.time()
    ix = orc.from_runs(host_runs)
    t_index = time.time() - t0
    m = min(int(a.cpu_sample), Q)
    sel = np.linspace(0, Q - 1, m).astype(np.int64)
    sel_t = torch.from_numpy(sel).to(d_kmers.device)
    km = d_kmers[sel_t].cpu().numpy()
    glo = d_lower[sel_t].cpu().numpy().view(np.uint64)
    gup = d_upper[sel_t].cpu().numpy().view(np.uint64)
    # one thread first, cold: a disjoint sample, before the index has been walked at all
    m1 = max(1, m // 16)
    sel1 = torch.from_numpy((np.linspace(0, Q - 1, m1).astype(np.int64) + 7) % Q).to(d_kmers.device)
    km1 = d_kmers[sel1].cpu().numpy()
    t1 = time.perf_counter()
    ix.find_intervals(km1, nthreads=1)
    dt1 = time.perf_counter() - t1
    threads = a.cpu_threads or usable_cpus()
    t0 = time.perf_counter()
    lo, up = ix.find_intervals(km, nthreads=threads)
    dt = time.perf_counter() - t0
    match = bool(np.array_equal(lo, glo) and np.array_equal(up, gup))
    return {
        "value": m / dt, "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"{m} of the batch's {Q} k-mers (evenly spaced) on shard 0 (one of the resident shards: per-shard "
                  f"searches/s, to be set against S/s per shard); oracle/rlebwt_oracle.c, {threads} POSIX threads sharing "
                  f"one index = every CPU this process may run on",
        "single_thread_value": m1 / dt1,
        "single_thread_sample": f"{m1} other k-mers, run first on the cold index",
        "host_cpus_visible": os.cpu_count(), "index_build_s": round(t_index, 2),
        "gpu_matches_oracle_on_sample": match,
    }


if __name__ == "__main__":
    main()
