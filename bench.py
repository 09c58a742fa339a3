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
# when this run began: the launcher parent's start where there is one (it hands its own down), else this process's
T_START = float(os.environ.get("BENCH_T0", "0") or 0) or time.time()

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
    ap.add_argument("--mode", choices=["exact", "1mm", "extract"], default="exact",
                    help="exact = configs[1]/[2], the headline; 1mm = configs[3] (1-mismatch hit lists of every shard); "
                         "extract = configs[4] (locate + read extraction in every shard)")
    ap.add_argument("--mix", choices=["population", "disjoint"], default="population",
                    help="population (headline): every shard holds the same stream, a present k-mer is present in every shard; "
                         "disjoint: every shard its own stream, a present k-mer is in one shard only (round 2's default). "
                         "A one-GPU run measures the other mix too (config.mixes) unless --no-second-mix")
    ap.add_argument("--same-shards", action="store_true", help="(old name of --mix population)")
    ap.add_argument("--no-second-mix", action="store_true", help="time only --mix (profile passes; also skips the valid-popBWT leg)")
    ap.add_argument("--valid-popbwt-symbols", type=float, default=-1,
                    help="third leg of a one-GPU run: the same search on a VALID population BWT built on the GPU (tools/popbwt_bench.py), "
                         "this many symbols per shard before dedup (8 shards); 0 = skip; default: 3e9 (1.7e10 symbols in all, ~80 s to "
                         "build) on a full-size run, skipped when --runs is below 1e10")
    ap.add_argument("--deep-popbwt-symbols", type=float, default=-1,
                    help="fourth leg of a one-GPU run: the same search on a VALID population BWT of the DEPTH north_star names -- the "
                         "reads of ~2,700 genomes in every shard (512 haplotypes, 420x per shard: tools/popbwt_bench.py --depth 420 "
                         "--haplotypes 512), a genomic 31-mer's final interval tens of rows wide -- this many symbols per shard (8 "
                         "shards); 0 = skip; default: 3e9 (1.4e10 symbols in all after dedup, ~55 s to build) on a full-size run")
    ap.add_argument("--two-streams", action="store_true",
                    help="N = 1: batches alternate between two streams (buffers of their own), so that batch i + 1's packing, start "
                         "records and ramp run under batch i's tail -- what two of a service's pool threads calling the handle do")
    ap.add_argument("--piped-start", action="store_true",
                    help="prepare batch i + 1 (packing, start records) on a second stream while batch i is searched, instead of "
                         "pack, start records and search one after the other on one stream (measured slower on the headline mix)")
    ap.add_argument("--stream", choices=["mixed", "long", "pop"], default="pop",
                    help="run-length mix of the synthetic stream: pop = the unit-length histogram measured on a valid population "
                         "BWT (5.8 symbols per unit; tools/popbwt_gpu.py); mixed = ~10.4 symbols per unit; long = mostly "
                         "31-symbol units of long runs (~25)")
    ap.add_argument("--kmers", type=float, default=4e5, help="--mode 1mm: 31-mers per batch")
    ap.add_argument("--rows", type=float, default=2e6, help="--mode extract: rows per shard and batch")
    ap.add_argument("--row-run", type=int, default=8,
                    help="--mode extract: rows come as runs of this many consecutive SA rows (the rows of an interval: "
                         "query.cpp:94-96 extracts lower..upper; 8 = the final width measured on the valid popBWT)")
    ap.add_argument("--stride", type=int, default=256, help="--mode extract: bytes per read buffer")
    ap.add_argument("--verify-rows", type=int, default=200,
                    help="--mode 1mm / extract at N = 1: after the timed region, this many k-mers' hit lists / this many reads of shard 0 "
                         "are held to the oracle (its index over shard 0's 2e10 run bytes takes ~30 s to build); 0 = skip")
    ap.add_argument("--cpu-sample", type=float, default=1e6, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every CPU this process may run on")
    ap.add_argument("--ktab-depth", type=int, default=0, help="k-mer table depth (0 = auto, -1 = none)")
    ap.add_argument("--ktab-format", choices=["auto", "plain", "grouped"], default="auto",
                    help="k-mer table format (include/rsbwt.h, rsbwt_attach_ktab_format): plain = 8 B per T-mer; grouped = 12 B per "
                         "four sibling T-mers, one level deeper out of the same HBM; auto = grouped where that is deeper")
    ap.add_argument("--window-span", type=int, default=0, help="symbols per window line (0 = from the data)")
    ap.add_argument("--ref-out-of-cache-runs", type=float, default=2e9,
                    help="run bytes of the index the compiled reference is timed on OUT OF CACHE beside the port (cpu_baseline.reference_beside_port.out_of_cache; 0 = skip)")
    ap.add_argument("--host", choices=["ranks", "cxx"], default="ranks",
                    help="ranks = one process per GPU over torch.distributed / RCCL (the driver's launch shape); cxx = the C++ host's "
                         "shape: ONE process drives all --gpus devices (a fused launch per device on a stream of its own, the 10-byte "
                         "records gathered onto device 0 by rsbwt_set_gather_intervals_dev: ncclSend / ncclRecv), no launcher.  An "
                         "N > 1 run of the default host adds this as a second leg (config.cxx_host) unless --no-cxx-leg")
    ap.add_argument("--no-cxx-leg", action="store_true", help="N > 1: skip the one-process (C++ host) leg after the per-rank one")
    ap.add_argument("--cxx-leg-timeout", type=float, default=420.0, help="seconds the second leg may take before it is given up")
    ap.add_argument("--total-budget", type=float, default=540.0,
                    help="seconds the WHOLE run may take (the driver's limit is 600): rank 0 prints the headline line BEFORE the second "
                         "leg starts, bounds the leg by what is left of this budget, and prints the line again with config.cxx_host "
                         "after it -- the first line stands whatever happens to the leg")
    ap.add_argument("--layout", choices=["auto", "plain", "reads"], default="auto",
                    help="reads = RSBWT_OPEN_READS: a psi hint in every window line, built with the index (~9 %% more lines): "
                         "what a shard that serves read extraction is opened with; auto = reads for --mode extract, plain otherwise")
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
    a = ap.parse_args()
    if a.same_shards:
        a.mix = "population"
    return a


def kernel_source_sha():
    """Identifies the search kernel + layout a PMC traffic figure was measured on."""
    h = hashlib.sha256()
    for f in ("search_lines.hip", "search_solo.h", "wave_lines.h", "line_format.h", "rank_device.h"):
        h.update(open(os.path.join(ROOT, "readserver_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def total(t):
    """Sum of a tensor's elements as a Python int, in slices: `t.sum(dtype=int64)` of a byte tensor makes an int64 copy of
    it first -- 8 x its size, beside shards that fill the HBM."""
    f = t.reshape(-1)
    return sum(int(f[i:i + (1 << 27)].sum(dtype=f.dtype if f.dtype.is_floating_point else __import__("torch").int64).item())
               for i in range(0, max(f.numel(), 1), 1 << 27)) if f.numel() else 0


def mode_source_sha():
    """Identifies the kernels a PMC traffic figure of --mode 1mm / extract was measured on: every device source under
    csrc/ and the host code that launches it (*.hip, *.h; the service loop, the file readers and the synthesiser -- *.cpp --
    launch nothing these modes time)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "readserver_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _pmc_traffic_mode(mode, R, S, units):
    """HBM bytes per step of the kernels roofline.kernel_ms covers in --mode 1mm / extract, from the committed PMC passes of
    the same command (profiles/pmc_traffic_modes.json, tools/collect_mode_profile.py) -- only if they were measured on these
    very sources and sizes; otherwise None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_modes.json")))[mode]
        if (d["source_sha"] == mode_source_sha() and int(d["run_bytes_per_shard"]) == R and int(d["shards_per_gpu"]) == S
                and int(d["units_per_batch"]) == units):
            return d["hbm_bytes_per_step_covered_kernels"]
    except Exception:
        pass
    return None


def pick_tables(a, free_b, S, n_sym, T):
    """(depth, format) of the job's k-mer tables: S tables out of the free HBM less 8 GiB, by the LIBRARY's rule
    (include/rsbwt.h, rsbwt_auto_ktab_for_budget: the deepest plain table that fits and whose T-mers are still expected to
    occur; the grouped format, 3 B per T-mer, where it gets a level deeper, its groups of four siblings fit their records
    and a T-mer still has 64 rows on average).  T: the depth the set reported (below 2: no tables at all)."""
    if T < 2:
        return T, 0
    import readserver_amd as rsb
    fmt_in = {"plain": 0, "grouped": 1, "auto": 2}[a.ktab_format]
    d, f = C.c_uint32(), C.c_uint32()
    rc = rsb.lib().rsbwt_auto_ktab_for_budget(max(0, free_b - (8 << 30)) // max(S, 1), n_sym, fmt_in, C.byref(d), C.byref(f))
    if rc != 0:
        raise RuntimeError(rsb.lib().rsbwt_last_error().decode())
    return int(d.value), int(f.value)


def ktab_config(shards):
    """config keys that describe the tables the shards got"""
    fmt, nbytes, left = shards[0].ktab_info()
    T = shards[0].ktab_depth()
    return {"ktab_depth": T, "ktab_format": "grouped (12 B per 4 sibling T-mers)" if fmt == 1 else "plain (8 B per T-mer)",
            "ktab_bytes_per_shard": nbytes, "ktab_untabulated_frac": (left / 4 ** T) if T else 0.0}


def search_kernel_name(nshards, n=0, ktab_depth=0, span=0):
    """The kernel launch_search (csrc/search_lines.hip) picks: one lane per search for a full batch on
    a single shard behind a deep k-mer table (4 n / 4^T <= S: capi.hip, search_dev), lane pairs
    otherwise; RSBWT_SEARCH_KERNEL overrides."""
    e = os.environ.get("RSBWT_SEARCH_KERNEL", "auto")
    narrow = ktab_depth >= 2 and ((n >> (2 * ktab_depth)) << 2) <= span
    solo = e == "solo" or (e not in ("pair", "solo") and narrow)
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
    env.setdefault("BENCH_T0", repr(T_START))  # the ranks count the run's budget from here
    cmd = self_launch_cmd(argv, a.gpus, int(os.environ.get("BENCH_MASTER_PORT", "0")) or free_port())
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, env=env)  # stdout/stderr inherited: rank 0's JSON line arrives on our stdout
    try:
        rc = p.wait()
    except KeyboardInterrupt:
        p.terminate()
        rc = p.wait()
    sys.exit(rc if rc >= 0 else 128 - rc)


class Ctx:
    """What every leg of the bench shares: the library, this rank's place in the job, its device."""


def setup(a):
    import torch
    import torch.distributed as dist
    import readserver_amd as rsb

    c = Ctx()
    c.torch, c.dist, c.rsb = torch, dist, rsb
    c.rank = int(os.environ.get("RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.local = int(os.environ.get("LOCAL_RANK", "0"))
    if c.world != a.gpus:
        if c.rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={c.world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU path", file=sys.stderr)
        sys.exit(1)
    if a.rehearse_on_one_gpu:
        c.local = 0
    torch.cuda.set_device(c.local)
    c.dev = torch.device("cuda", c.local)
    c.cdev = torch.device("cpu") if a.rehearse_on_one_gpu else c.dev  # where the collectives' tensors live
    if c.world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=c.dev)
            # a CPU-side group for the waits around the second leg (--host cxx as rank 0's child): an RCCL barrier would
            # leave a kernel spinning on every GPU for as long as the leg runs on them
            c.cpu_group = dist.new_group(backend="gloo")
    c.L = rsb.lib()
    c.stream = torch.cuda.current_stream()
    c.sp = C.c_void_p(c.stream.cuda_stream)
    return c


def ptr(t):
    return C.c_void_p(t.data_ptr())


def ok(c, rc):
    if rc != 0:
        raise RuntimeError(c.L.rsbwt_last_error().decode())


STREAM_STYLE = {"mixed": 0, "long": 1 << 63, "pop": 1 << 62}
STREAM_NOTE = {
    "mixed": "mixed = 20 % full units, 20 % 6..21, 60 % 1..4 symbols: 10.4 symbols per run byte",
    "long": "long = mostly 31-symbol units of long runs: ~25 symbols per run byte",
    "pop": "pop = the unit-length histogram measured on a valid 1.1e9-symbol population BWT (64 haplotypes, 64 suffix shards, "
           "28x depth per shard, 1 % base errors: tools/popbwt_gpu.py, profiles/r03_popbwt_calibration.json): 5.8 symbols per run byte",
}


def shard_seed(a, mix, rank, S, s):
    """population: every shard of the job holds the same stream, so that a present k-mer is present in every shard
    (a valid popBWT holds a genomic 31-mer in 75 % of its 64 shards and keeps searching for most steps in the others:
    profiles/r03_popbwt_calibration.json); disjoint: every shard its own stream (a present k-mer is in one shard only)."""
    style = STREAM_STYLE[a.stream]
    return style | (a.seed * 1000003 + (0 if mix == "population" else rank * S + s))


def build_shards(a, c, mix, want_host_runs=False):
    torch, L, rsb = c.torch, c.L, c.rsb
    R, S = int(a.runs), a.shards_per_gpu
    shards, host_runs = [], None
    for s in range(S):
        d_runs = torch.empty(R, dtype=torch.uint8, device=c.dev)
        ok(c, L.rsbwt_synth_runs_dev(ptr(d_runs), R, shard_seed(a, mix, c.rank, S, s), c.local, c.sp))
        torch.cuda.synchronize()
        # tables are sized afterwards, for all shards of the GPU together (explicit depth: now)
        g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), device=c.local,
                       ktab_depth=(a.ktab_depth if a.ktab_depth > 0 else None), window_span=a.window_span,
                       for_reads=(a.layout == "reads" or (a.layout == "auto" and a.mode == "extract")))
        if want_host_runs and s == 0:
            host_runs = d_runs.cpu().numpy()
        del d_runs
        torch.cuda.empty_cache()
        shards.append(g)
    return shards, rsb.ShardSet(shards), host_runs


def size_tables(a, c, sset, shards, S):
    """One k-mer table depth for every shard of the job, out of the HBM that is free once every buffer of the
    job exists (rank 0's gathered results included), less an 8 GB reserve."""
    torch, dist, L = c.torch, c.dist, c.L
    if a.ktab_depth != 0:
        return
    n_sym = shards[0].getBWLen()
    T = L.rsbwt_set_auto_ktab_depth(sset._s)
    free_b = torch.cuda.mem_get_info(c.dev)[0]
    if a.rehearse_on_one_gpu:  # the ranks share one GPU: each sizes its tables out of its share
        if c.world > 1:
            dist.barrier()  # every rank's buffers exist before anyone looks at what is free
            free_b = torch.cuda.mem_get_info(c.dev)[0]
        free_b //= c.world
        while T >= 2 and S * 8 * 4 ** T > free_b // 2:
            T -= 1
    T, fmt = pick_tables(a, free_b, S, n_sym, T)
    if c.world > 1:  # one depth and one format for the whole job: the shallowest any rank can hold (plain if any must)
        tt = torch.tensor([T, fmt], dtype=torch.int64, device=c.cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        if int(tt[0].item()) != T or int(tt[1].item()) != fmt:
            T, fmt = int(tt[0].item()), int(tt[1].item())
    if T >= 2:
        ok(c, L.rsbwt_set_attach_ktabs_format(sset._s, T, fmt))


def make_batch(a, c, shards, mix, Q, k, d_kmers):
    """The query batch, identical on every rank: uniform random k-mers with `present_frac` of them replaced by
    k-mers drawn from the index by LF walks (all k - 1 steps in the shards that hold them)."""
    torch, dist, L = c.torch, c.dist, c.L
    S = len(shards)
    gen = torch.Generator(device=c.dev)
    gen.manual_seed(a.seed + 12345)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=c.dev)
    for i in range(0, Q, 1 << 22):
        j = min(Q, i + (1 << 22))
        d_kmers[i:j] = lut[torch.randint(0, 4, (j - i, k), generator=gen, device=c.dev, dtype=torch.uint8).long()]
    n_present = int(Q * a.present_frac)
    if not n_present:
        return
    if mix == "population":  # every shard of the job is the same stream: drawn once, the same on every rank
        mine = torch.empty((n_present, k), dtype=torch.uint8, device=c.dev)
        ok(c, L.rsbwt_sample_present_kmers_dev(shards[0].handle, n_present, k, k, a.seed + 7, ptr(mine), c.sp))
        torch.cuda.synchronize()
    else:  # every rank draws its share evenly from its shards; shares are concatenated
        share = n_present // c.world
        if not share:
            return
        per = [share // S + (1 if i < share % S else 0) for i in range(S)]
        parts = []
        for s, m in enumerate(per):
            if m:
                t = torch.empty((m, k), dtype=torch.uint8, device=c.dev)
                ok(c, L.rsbwt_sample_present_kmers_dev(shards[s].handle, m, k, k, a.seed + 7 + c.rank * S + s, ptr(t), c.sp))
                parts.append(t)
        torch.cuda.synchronize()
        mine = torch.cat(parts, 0)
        mine = mine[torch.randperm(mine.shape[0], device=c.dev, generator=gen)]  # the shards' k-mers interleaved
        if c.world > 1:
            mine_c = mine.to(c.cdev)
            allp = [torch.empty_like(mine_c) for _ in range(c.world)]
            dist.all_gather(allp, mine_c)
            mine = torch.cat(allp, 0).to(c.dev)
    # interleave present and random k-mers so every wave sees the mix
    idx = torch.arange(mine.shape[0], device=c.dev) * (Q // mine.shape[0])
    d_kmers[idx] = mine


def run_exact(a, c, mix, steps, warmup, headline):
    """One mix of the exact search (configs[1] / configs[2]): builds the resident shards, times `steps` batches."""
    torch, dist, L, rsb = c.torch, c.dist, c.L, c.rsb
    from readserver_amd import sharded
    rank, world, local, dev, cdev, sp = c.rank, c.world, c.local, c.dev, c.cdev, c.sp
    R, Q, k, S = int(a.runs), int(a.queries), a.k, a.shards_per_gpu
    t_build0 = time.time()
    want_cpu = headline and rank == 0 and world == 1 and a.cpu_sample > 0  # the CPU baseline is an N = 1 leg
    # what this rank will hold, item by item, against what the device has free -- BEFORE the first shard is built: a
    # job that cannot fit (rank 0 of an N-GPU job also holds every rank's gathered blocks) stops here with the table,
    # not with an out-of-memory error between two shards or in the first gather (readserver_amd/sharded.py, hbm_plan)
    wire_packed = world > 1 and not a.separate_arrays and not a.gather_unpacked
    for_reads = a.layout == "reads" or (a.layout == "auto" and a.mode == "extract")
    plan_kw = dict(wire_packed=wire_packed, separate=a.separate_arrays, out_depth=(2 if world <= 2 else 1))
    if not a.rehearse_on_one_gpu:  # (a rehearsal's ranks share one GPU: no plan holds there)
        free0 = torch.cuda.mem_get_info(dev)[0]
        est = sharded.hbm_plan(world, rank, S, Q, k, int(R * (1.73 if for_reads else 1.64)), R, **plan_kw)
        try:
            sharded.check_hbm_plan(est, free0)
        except MemoryError as e:
            raise SystemExit("bench.py: " + str(e))
    shards, sset, host_runs = build_shards(a, c, mix, want_host_runs=want_cpu)
    n_sym = shards[0].getBWLen()
    # the batch's buffers first (rank 0 also holds the gathered intervals of all ranks), then the k-mer
    # tables out of what HBM is left: one depth for every shard of the job
    wpq = (k + 31) // 32
    d_packed = torch.empty((Q, wpq), dtype=torch.int64, device=dev)
    d_valid = torch.empty(Q, dtype=torch.uint8, device=dev)
    d_kmers = torch.empty((Q, k), dtype=torch.uint8, device=dev)
    # N > 1: the pairs travel as 10-byte {lower:40, width:40} records (exact for every interval; 5/8 of the
    # bytes): an xGMI link moves ~77 GB/s per direction, so 1.28 GB of 16-byte pairs per peer and batch would
    # take longer than the 12 ms search that produced them (wire_packed, above)
    # (from 4 ranks on rank 0 keeps ONE batch's gathered blocks: at N = 8 that is 6.4 GB of HBM the k-mer tables get --
    # the grouped 15-mer tables then fit beside the shards on rank 0 too, and the job's one depth is the N = 1 depth)
    gat = sharded.IntervalGatherer(S, Q, cdev, depth=2, interleaved=not a.separate_arrays, packed=wire_packed, wire_device=cdev,
                                   out_depth=(2 if world <= 2 else 1))
    d_res = [torch.empty_like(gat.pair(i), device=dev) for i in range(2)] if cdev != dev else None
    hold = torch.empty(int(a.hold_gb * (1 << 30)), dtype=torch.uint8, device=dev) if a.hold_gb > 0 else None
    # --piped-start: a batch's packing and start records (they depend on the k-mers and the k-mer tables only) are
    # computed on a second stream while the previous batch is searched, so a step of the main stream is the search
    # kernel alone.  Measured (profiles/r03_piped_start.json): the search kernel is at the memory system's request
    # ceiling, the start-record kernel beside it takes requests from it and the pair runs LONGER than one after the
    # other on the population mix (24.3 vs 23.2 ms per step), a little shorter on the disjoint mix (12.0 vs 12.4):
    # the default stays one stream.
    piped = a.piped_start and not (a.counts or a.separate_arrays)
    if piped:
        side = torch.cuda.Stream(device=dev, priority=0)  # (the lowest priority there is; the search's stream is the default one)
        side_p = C.c_void_p(side.cuda_stream)
        d_packed2 = [d_packed, torch.empty_like(d_packed)]
        d_valid2 = [d_valid, torch.empty_like(d_valid)]
        d_rec2 = [torch.empty(L.rsbwt_set_records_bytes(sset._s, Q), dtype=torch.uint8, device=dev) for _ in range(2)]
        prep_done = [torch.cuda.Event(), torch.cuda.Event()]
        searched = [torch.cuda.Event(), torch.cuda.Event()]
        prepared_for = [None, None]
    two = a.two_streams and world == 1 and not piped and not (a.counts or a.separate_arrays)
    if two:
        two_st = [c.stream, torch.cuda.Stream(device=dev)]
        two_pk = [d_packed, torch.empty_like(d_packed)]
        two_ok = [d_valid, torch.empty_like(d_valid)]
    size_tables(a, c, sset, shards, S)
    t_build = time.time() - t_build0
    # the plan again with what was built (exact line bytes, the tables chosen), and what the device says is left
    lines_b = max(int(g.hbm_bytes()) - int(g.ktab_info()[1]) for g in shards)
    plan = sharded.hbm_plan(world, rank, S, Q, k, lines_b, R, ktab_bytes_per_shard=max(int(g.ktab_info()[1]) for g in shards), **plan_kw)
    plan["free_after_allocation"] = int(torch.cuda.mem_get_info(dev)[0])
    plan = {kk: (round(v / 1e9, 3) if kk not in ("out_depth", "world", "rank") else v) for kk, v in plan.items()} | {"unit": "GB"}
    make_batch(a, c, shards, mix, Q, k, d_kmers)
    torch.cuda.synchronize()

    step_no = [0]
    d_counts = None

    def prepare(i):  # batch i's packed k-mers and start records, on the side stream
        j = i % 2
        if prepared_for[j] == i:
            return
        with torch.cuda.stream(side):
            if i >= 2:
                side.wait_event(searched[j])  # the search of batch i - 2 read these buffers
            ok(c, L.rsbwt_pack_kmers_dev(ptr(d_kmers), Q, k, k, ptr(d_packed2[j]), ptr(d_valid2[j]), local, side_p))
            ok(c, L.rsbwt_set_prepare_dev(sset._s, ptr(d_packed2[j]), ptr(d_valid2[j]), Q, k, ptr(d_rec2[j]), side_p))
            prep_done[j].record(side)
        prepared_for[j] = i
    if a.counts:
        if world != 1 or a.separate_arrays:
            raise SystemExit("bench.py --counts: one GPU, default result layout")
        a.cpu_sample, a.no_single_check = 0, True
        want_cpu = False
        d_counts = torch.empty((S, Q), dtype=torch.int64, device=dev)

    def step():
        i = step_no[0]
        step_no[0] += 1
        pair = gat.acquire(i)
        host_pair = None
        if d_res is not None:  # rehearsal: search into HBM, gather from a host copy
            host_pair, pair = pair, d_res[i % 2]
        if piped:
            j = i % 2
            prepare(i)  # (only the very first batch of a sequence is not prepared yet)
            c.stream.wait_event(prep_done[j])
            ok(c, L.rsbwt_set_find_interval_pairs_prepared_dev(sset._s, ptr(d_packed2[j]), ptr(d_valid2[j]), ptr(d_rec2[j]), Q, k,
                                                               ptr(pair), sp))
            searched[j].record(c.stream)
            prepare(i + 1)
            if host_pair is not None and not wire_packed:
                host_pair.copy_(pair)
            gat.submit(i, source=pair if wire_packed else None)
            return
        if two:
            jj = i % 2
            spj = C.c_void_p(two_st[jj].cuda_stream)
            ok(c, L.rsbwt_pack_kmers_dev(ptr(d_kmers), Q, k, k, ptr(two_pk[jj]), ptr(two_ok[jj]), local, spj))
            ok(c, L.rsbwt_set_find_interval_pairs_dev(sset._s, ptr(two_pk[jj]), ptr(two_ok[jj]), Q, k, ptr(pair), spj))
            gat.submit(i, source=None)
            return
        ok(c, L.rsbwt_pack_kmers_dev(ptr(d_kmers), Q, k, k, ptr(d_packed), ptr(d_valid), local, sp))
        if a.counts:
            ok(c, L.rsbwt_set_count_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(d_counts), sp))
        elif a.separate_arrays:
            ok(c, L.rsbwt_set_find_intervals_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(pair[0]), ptr(pair[1]), sp))
        else:
            ok(c, L.rsbwt_set_find_interval_pairs_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(pair), sp))
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
    ok(c, L.rsbwt_set_set_counting(sset._s, 1))
    step()
    torch.cuda.synchronize()
    w = (C.c_uint64 * 16)()
    ok(c, L.rsbwt_set_last_search_counters(sset._s, w))
    ok(c, L.rsbwt_set_set_counting(sset._s, 0))
    lf, oc, ln, kt, hops, npass = w[0], w[1], w[2], w[3], w[11], w[10]
    phases = {"passes": npass, "cycles_per_pass": [round(w[4 + i] / max(npass, 1)) for i in range(6)],
              "names": ["setup", "issue", "wait", "rank", "exchange", "update"]}

    if a.counts:  # the counts of a launch against the interval pairs of the same batch
        pr = gat.pair(0)
        ok(c, L.rsbwt_set_find_interval_pairs_dev(sset._s, ptr(d_packed), ptr(d_valid), Q, k, ptr(pr), sp))
        step()
        torch.cuda.synchronize()
        want = torch.where(pr[..., 1] >= pr[..., 0], pr[..., 1] - pr[..., 0] + 1, torch.zeros_like(pr[..., 0]))
        if not torch.equal(want, d_counts):
            raise SystemExit("bench.py --counts: counts differ from upper - lower + 1 of the same batch")
        del pr, want
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # per-launch search-kernel time over the timed region: HIP events the library records on the
    # launch stream around every search launch (it keeps the last 64 pairs)
    buf = (C.c_float * 64)()
    cnt = C.c_size_t()
    ok(c, L.rsbwt_set_search_history_ms(sset._s, buf, min(steps, 64), C.byref(cnt)))
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
        mine_sum = torch.tensor([total(sent)], dtype=torch.int64, device=cdev)
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
            gather_verified = bool(exact.item()) and all(total(got[r]) == int(sums[r].item()) for r in range(world))

    searches = world * S * Q * steps
    value = searches / dt
    ms_per_step = dt / steps * 1e3
    avg_kernel_ms = float(np.mean(k_ms))
    alg_bytes = ln * LINE_BYTES + S * Q * SEARCH_BYTES  # per launch, the search kernel's own reads and writes
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
    # the whole step: + packing (k + 8 wpq + 1 B per query) and the start records (8-byte table read +
    # 16 B written per search)
    step_bytes = alg_bytes + Q * (k + 8 * wpq + 1) + S * Q * 24
    hbm = sum(int(g.hbm_bytes()) for g in shards)

    # ---- configs[1] on the same resident data: shard 0 alone (same kernel, one shard) -----------
    single = None
    if headline and S > 1 and not a.no_single_check:
        g0 = shards[0]
        pr1 = torch.empty((Q, 2), dtype=torch.int64, device=dev)  # {lower, upper} pairs, as the headline's launch writes them
        ok(c, L.rsbwt_set_counting(g0.handle, 1))
        ok(c, L.rsbwt_find_interval_pairs_dev(g0.handle, ptr(d_packed), ptr(d_valid), Q, k, ptr(pr1), sp))
        torch.cuda.synchronize()
        w1 = (C.c_uint64 * 16)()
        ok(c, L.rsbwt_last_search_counters(g0.handle, w1))
        ok(c, L.rsbwt_set_counting(g0.handle, 0))
        for _ in range(2):
            ok(c, L.rsbwt_find_interval_pairs_dev(g0.handle, ptr(d_packed), ptr(d_valid), Q, k, ptr(pr1), sp))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 10
        for _ in range(n1):
            ok(c, L.rsbwt_find_interval_pairs_dev(g0.handle, ptr(d_packed), ptr(d_valid), Q, k, ptr(pr1), sp))
        torch.cuda.synchronize()
        d1 = (time.perf_counter() - t1) / n1
        b1 = (C.c_float * 64)()
        ok(c, L.rsbwt_search_history_ms(g0.handle, b1, n1, C.byref(cnt)))
        km1 = float(np.mean(list(b1[:cnt.value])))
        single = {"searches_per_s": Q / d1, "kernel": search_kernel_name(1, int(g0.getBWLen()), g0.ktab_depth(), g0.window_span()), "kernel_ms": km1, "mean_lf_steps_per_search": w1[0] / Q,
                  "roofline_frac": (w1[2] * LINE_BYTES + Q * SEARCH_BYTES) / (km1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "what": "shard 0 of the resident set alone, behind the SET's table (its share of the 8 shards' interleaved tables): the "
                          "one-shard kernel on the headline's data, not configs[1]'s own line -- `python bench.py --shards-per-gpu 1` sizes a "
                          "table for one shard (plain 16-mer: a step less per search) and is profiles/rNN_bench_1shard.json"}
        del pr1

    res = {
        "mix": mix, "value": value, "queries_per_s_all_shards": value / (world * S), "ms_per_step": ms_per_step,
        "steps": steps, "warmup": warmup,
        "mean_lf_steps_per_search": lf / (S * Q), "lines_per_lf_step": ln / max(lf, 1),
        **ktab_config(shards), "window_span": shards[0].window_span(),
        "symbols_per_shard": int(n_sym), "far_lines_per_shard": int(shards[0].far_lines()),
        "spilled_position_fraction": shards[0].spilled_symbols() / max(int(n_sym), 1),
        "index_hbm_bytes_per_gpu": hbm,
        "index_bytes_per_run_byte": (hbm - sum(g.ktab_info()[1] for g in shards)) / (S * R),
        "index_build_s": round(t_build, 2), "gather_verified": gather_verified, "wire_packed": wire_packed,
        "hbm_plan": plan,
        "single_shard_check": single,
        "step": ("search kernel on the main stream; the next batch's packing and start records on a second stream beside it "
                 "(rsbwt_set_prepare_dev / rsbwt_set_find_interval_pairs_prepared_dev)" if piped else
                 "pack + start records + search, batches alternating between two streams" if two else
                 "pack + start records + search, one after the other on one stream"),
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": _pmc_traffic(R, Q, S, k, a.stream, mix) if headline else None,
            "kernel": search_kernel_name(S, int(n_sym), shards[0].ktab_depth(), shards[0].window_span()), "kernel_ms": avg_kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes, "line_reads_per_launch": ln,
            "continuation_line_reads_per_launch": hops, "occ_lookups_per_launch": oc,
            "ktab_starts_per_launch": kt, "phase_stamps": phases,
            "frac_of_step": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
    }
    if want_cpu and host_runs is not None:
        pair = gat.pair(step_no[0] - 1).to(dev)
        lo0, up0 = (pair[0][0], pair[1][0]) if a.separate_arrays else (pair[0, :, 0], pair[0, :, 1])
        res["cpu_baseline"] = cpu_baseline(a, host_runs, d_kmers, lo0, up0, Q, k)
    else:
        res["cpu_baseline"] = None  # timed at N = 1 only (or switched off with --cpu-sample 0)
    if headline and rank == 0 and a.verify_all_shards:
        res["shards_matching_oracle"] = verify_shards(a, L, gat.pair(step_no[0] - 1).to(dev), d_kmers, S, R, Q, k, mix,
                                                      rank, local, dev, sp)
    sset.close()
    for g in shards:
        g.close()
    del shards, sset, gat, d_res, d_kmers, d_packed, d_valid, hold, d_counts
    if piped:
        del d_packed2, d_valid2, d_rec2
    torch.cuda.empty_cache()
    return res


MIX_NOTE = {
    "population": "every shard of the job holds the same stream, so a 31-mer drawn from the index is present in EVERY shard and "
                  "runs all its steps there (a valid 64-shard popBWT holds a genomic 31-mer in 75 % of its shards, final interval "
                  "~8 rows, and the searches that end empty still run most of their steps: profiles/r03_popbwt_calibration.json)",
    "disjoint": "every shard its own stream: a 31-mer drawn from the index is present in ONE shard and dies within a few steps "
                "of the k-mer table in the others (round 2's default; flatters S/s by ~2x)",
}


def cxx_leg_cmd(a):
    """The command of the second leg of an N > 1 run: this script as the one-process host over the same devices."""
    return [sys.executable, os.path.abspath(__file__), "--host", "cxx", "--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup),
            "--runs", str(a.runs), "--queries", str(a.queries), "--k", str(a.k), "--shards-per-gpu", str(a.shards_per_gpu),
            "--stream", a.stream, "--mix", a.mix, "--seed", str(a.seed), "--ktab-depth", str(a.ktab_depth), "--ktab-format", a.ktab_format, "--window-span", str(a.window_span)]


def cxx_leg_budget(a, elapsed):
    """Seconds the second leg may take: what is left of --total-budget after `elapsed` seconds of the run, less 20 s for
    the second line and the ranks' teardown, and no more than --cxx-leg-timeout.  None (with the reason) when less than
    90 s would be left -- the leg builds 8 shards per GPU (25 s) before its first batch: not worth starting."""
    left = a.total_budget - elapsed - 20.0
    if left < 90.0:
        return None, f"{max(left, 0):.0f} s left of the run's {a.total_budget:.0f} s budget after {elapsed:.0f} s: not started"
    return min(a.cxx_leg_timeout, left), None


def second_leg(a, out, rehearsal=False, emit=None, runner=None, clock=time.time):
    """Rank 0, after every rank has closed its shards: the headline line goes out NOW (config.cxx_host says the leg is
    still to come), then the one-process leg runs as a child within the budget, then the line goes out again with the
    leg's record.  A reader that takes the last JSON line of stdout gets the most complete one that was reached."""
    import subprocess
    emit = emit or (lambda line: print(line, flush=True))
    runner = runner or (lambda cmd, timeout: subprocess.run(cmd, capture_output=True, text=True, timeout=timeout))
    out["config"]["cxx_host"] = {"pending": "this line was printed before the one-process (--host cxx) leg started; a later line of the same run carries its record"}
    emit(json.dumps(out))
    limit, why_not = cxx_leg_budget(a, clock() - T_START)
    if limit is None:
        out["config"]["cxx_host"] = {"skipped": why_not}
    else:
        try:
            cmd = cxx_leg_cmd(a)
            if rehearsal:  # (one device: the leg's control flow, not its gather)
                cmd[cmd.index("--gpus") + 1] = "1"
            t0 = clock()
            r = runner(cmd, limit)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            rec = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": (r.stderr or r.stdout)[-800:]}
            rec["leg_seconds"], rec["leg_limit_seconds"] = round(clock() - t0, 1), round(limit, 1)
            out["config"]["cxx_host"] = rec
        except Exception as e:  # noqa: BLE001  (a time-out included: subprocess.TimeoutExpired)
            out["config"]["cxx_host"] = {"error": repr(e), "leg_limit_seconds": round(limit, 1)}
    emit(json.dumps(out))


def main():
    a = parse()
    if a.gpus > 1 and not a.rehearse_on_one_gpu:
        # N > 1: the previous batch's intervals are gathered (RCCL) while this batch is searched.  A search launch is
        # persistent and fills every CU (registers, LDS), so the collective's kernels would wait for its tail: the
        # library leaves 32 of its 1,024 workgroups out (csrc/search_lines.hip, RSBWT_SEARCH_SPARE_WGS: measured at
        # N = 1, no cost: 18.51 against 18.60 ms per launch, profiles/r05q_*) -- a workgroup slot on 32 CUs.
        os.environ.setdefault("RSBWT_SEARCH_SPARE_WGS", "32")
    if a.host == "cxx":
        if a.mode != "exact":
            raise SystemExit("bench.py --host cxx: the exact search (configs[1] / configs[2])")
        print(json.dumps(run_exact_cxx(a)), flush=True)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a, sys.argv[1:])
    c = setup(a)
    if a.mode != "exact":
        out = run_rows(a, c)
    else:
        S, world = a.shards_per_gpu, c.world
        head = run_exact(a, c, a.mix, a.steps, a.warmup, headline=True)
        mixes = {a.mix: {k: head[k] for k in ("value", "queries_per_s_all_shards", "ms_per_step", "mean_lf_steps_per_search",
                                                 "lines_per_lf_step", "ktab_depth")} | {"roofline_frac": head["roofline"]["frac"],
                                                                                        "kernel_ms": head["roofline"]["kernel_ms"], "what": MIX_NOTE[a.mix]}}
        other = "disjoint" if a.mix == "population" else "population"
        if world == 1 and not a.no_second_mix and not a.counts and S > 1:
            o = run_exact(a, c, other, max(3, a.steps // 2), 1, headline=False)
            mixes[other] = {k: o[k] for k in ("value", "queries_per_s_all_shards", "ms_per_step", "mean_lf_steps_per_search",
                                              "lines_per_lf_step", "ktab_depth")} | {"roofline_frac": o["roofline"]["frac"],
                                                                                     "kernel_ms": o["roofline"]["kernel_ms"], "what": MIX_NOTE[other]}
        vsym = a.valid_popbwt_symbols if a.valid_popbwt_symbols >= 0 else (3e9 if a.runs >= 1e10 else 0)
        if world == 1 and vsym > 0 and not a.no_second_mix and not a.counts and S > 1:
            # the same search on a VALID population BWT (the shards above are a random run stream): reads of 64 haplotypes,
            # RLO sort + dedup, suffix shards, per-shard suffix sort on the GPU -- and the oracle on a sample of it
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import popbwt_bench
                va = argparse.Namespace(symbols_per_shard=vsym, depth=28.0, haplotypes=64, snp=1e-3, err=0.01, queries=a.queries,
                                        steps=max(3, a.steps // 2), seed=11, check=False)
                v = popbwt_bench.run(va)
                mixes["valid_popbwt"] = {
                    "value": v["searches_per_s"], "queries_per_s_all_shards": v["queries_per_s"], "ms_per_step": v["ms_per_step"],
                    "mean_lf_steps_per_search": v["mean_lf_steps_per_search"], "lines_per_lf_step": v["lines_per_lf_step"],
                    "ktab_depth": v["ktab_depth"], "roofline_frac": v["roofline"]["frac"], "kernel_ms": v["roofline"]["kernel_ms"],
                    "symbols": v["symbols"], "run_bytes": v["run_bytes"], "index_hbm_bytes": v["index_hbm_bytes"],
                    "fraction_of_shards_holding_a_genomic_31mer": v["genomic_31mers"]["fraction_of_shards_holding_one"],
                    "gpu_matches_oracle_on_sample": v["oracle"]["gpu_matches_oracle"], "build_s": v["seconds"],
                    "what": "8 shards of a VALID 64-shard population BWT built on the GPU (tools/popbwt_bench.py: 64 haplotypes, 28x per "
                            "shard, 1 % base errors), the same fused search, half genomic / half random 31-mers; out of cache, a tenth of "
                            "configs[2]'s size per shard",
                }
            except Exception as e:  # the headline stands whatever happens to this leg
                mixes["valid_popbwt"] = {"error": repr(e)}
        dsym = a.deep_popbwt_symbols if a.deep_popbwt_symbols >= 0 else (3e9 if a.runs >= 1e10 else 0)
        if world == 1 and dsym > 0 and not a.no_second_mix and not a.counts and S > 1:
            # the population north_star quotes its target on (~2.7k genomes): every shard holds the reads of 512 haplotypes at
            # 420x, so a genomic 31-mer ends on an interval tens of rows wide and the two ends of an interval part ways more
            # often (more lines per LF step) -- the regime VERDICT r04 (missing #3) asks to see in the driver's own line
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import popbwt_bench
                va = argparse.Namespace(symbols_per_shard=dsym, depth=420.0, haplotypes=512, snp=1e-3, err=0.01, queries=a.queries,
                                        steps=max(3, a.steps // 2), seed=13, check=False)
                v = popbwt_bench.run(va)
                mixes["deep_population"] = {
                    "value": v["searches_per_s"], "queries_per_s_all_shards": v["queries_per_s"], "ms_per_step": v["ms_per_step"],
                    "mean_lf_steps_per_search": v["mean_lf_steps_per_search"], "lines_per_lf_step": v["lines_per_lf_step"],
                    "ktab_depth": v["ktab_depth"], "roofline_frac": v["roofline"]["frac"], "kernel_ms": v["roofline"]["kernel_ms"],
                    "kernel": v["roofline"]["kernel"],
                    "symbols": v["symbols"], "run_bytes": v["run_bytes"], "index_hbm_bytes": v["index_hbm_bytes"],
                    "haplotypes": v["haplotypes"], "depth_per_shard": v["depth_per_shard"], "genome_len": v["genome_len"],
                    "final_interval_width_mean": v["genomic_31mers"]["final_width_mean"],
                    "final_interval_width_median": v["genomic_31mers"]["final_width_median"],
                    "fraction_of_shards_holding_a_genomic_31mer": v["genomic_31mers"]["fraction_of_shards_holding_one"],
                    "gpu_matches_oracle_on_sample": v["oracle"]["gpu_matches_oracle"], "oracle_sample_kmers": v["oracle"]["kmers"],
                    "build_s": v["seconds"],
                    "what": "8 shards of a VALID 64-shard population BWT of the depth north_star names: 512 haplotypes at 420x per shard "
                            "(the reads of ~2,700 genomes at 10x in every shard; tools/popbwt_bench.py --depth 420 --haplotypes 512), on a "
                            "genome short enough that the build fits a minute; the same fused search, half genomic / half random 31-mers",
                }
            except Exception as e:  # the headline stands whatever happens to this leg
                mixes["deep_population"] = {"error": repr(e)}
        out = {
            # BASELINE.json's metric with what `value` counts spelled out: one backward search = one
            # findInterval = one query on one shard; queries resolved against every resident shard per second = queries_per_s
            "metric": "31-mer backward-search queries/sec on popBWT, counted as (query x shard) backward searches/s",
            "value": head["value"],
            "unit": "searches/s",
            "queries_per_s": head["queries_per_s_all_shards"],
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"],
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
                "value_counts": "value = searches/s = (query x shard) backward searches per second (one findInterval each); "
                                "queries_per_s = value / shards = queries resolved against every resident shard",
                "queries_per_s_all_shards": head["queries_per_s_all_shards"],
                "mix": a.mix + ": " + MIX_NOTE[a.mix],
                "mixes": mixes,
                "shards_per_gpu": S, "shards": world * S, "run_bytes_per_shard": int(a.runs), "symbols_per_shard": head["symbols_per_shard"],
                "stream": STREAM_NOTE[a.stream],
                "queries_per_batch": int(a.queries), "k": a.k, "present_fraction": a.present_frac,
                "mean_lf_steps_per_search": head["mean_lf_steps_per_search"], "lines_per_lf_step": head["lines_per_lf_step"],
                "ktab_depth": head["ktab_depth"], "ktab_format": head.get("ktab_format"), "ktab_bytes_per_shard": head.get("ktab_bytes_per_shard"),
                "ktab_untabulated_frac": head.get("ktab_untabulated_frac"),
                "window_span": head["window_span"], "far_lines_per_shard": head["far_lines_per_shard"],
                "spilled_position_fraction": head["spilled_position_fraction"],
                "index_hbm_bytes_per_gpu": head["index_hbm_bytes_per_gpu"], "index_bytes_per_run_byte": head["index_bytes_per_run_byte"],
                "index_build_s": head["index_build_s"],
                "hbm_plan": head["hbm_plan"],
                "gather_verified": head["gather_verified"],
                "gathered_as": (None if world == 1 else "10-byte {lower:40, width:40} records (rsbwt_pack_interval_pairs_dev), exact"
                                if head["wire_packed"] else "16-byte pairs"),
                "results": ("counts[S][Q] (rsbwt_set_count_dev: the service's count path, NOT the headline)" if a.counts else
                            "lower[S][Q], upper[S][Q]" if a.separate_arrays else "{lower, upper}[S][Q] pairs (BWTInterval)"),
                "multi_gpu": ("REHEARSAL on one GPU over gloo: not a measurement" if a.rehearse_on_one_gpu else "measured" if world > 1 else "one GPU; the N > 1 gather path is covered by the gloo world-2 test only"),
                "single_shard_check": head["single_shard_check"],
                "step": head["step"],
            },
            "roofline": head["roofline"],
            "cpu_baseline": head["cpu_baseline"],
        }
        if "shards_matching_oracle" in head:
            out["config"]["shards_matching_oracle"] = head["shards_matching_oracle"]
        if world > 1 and not a.no_cxx_leg:
            # second leg: the same workload driven by ONE process over all the GPUs (the C++ host's shape).  Every rank has
            # closed its shards; rank 0 PRINTS THE HEADLINE LINE FIRST, then starts the leg as a child bounded by what is
            # left of the run's budget, and prints the line again with the leg's record: the first line stands whatever
            # happens to the leg -- a time-out of the leg, or the driver's own clock running out on it.
            c.torch.cuda.synchronize()
            c.dist.barrier(group=getattr(c, "cpu_group", None))
            if c.rank == 0:
                second_leg(a, out, rehearsal=a.rehearse_on_one_gpu)
                out = None  # (both lines are out)
            c.dist.barrier(group=getattr(c, "cpu_group", None))
    if c.rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    if c.world > 1:
        c.dist.destroy_process_group()


def oracle_of_shard0(a, c, mix):
    """The oracle's index over shard 0's run bytes (test infrastructure, the checker of the rows modes' outputs): the
    stream is synthesised again slice by slice on the GPU (HBM is full of shards: no room for 20 GB of run bytes) and
    copied to the host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding
    torch, L = c.torch, c.L
    R, S = int(a.runs), a.shards_per_gpu
    seed = shard_seed(a, mix, c.rank, S, 0)
    host = np.empty(R, np.uint8)
    CH = min(R, 1 << 29)
    d = torch.empty(CH, dtype=torch.uint8, device=c.dev)
    for i0 in range(0, R, CH):
        n = min(CH, R - i0)
        ok(c, L.rsbwt_synth_runs_dev_at(ptr(d), i0, n, seed, c.local, c.sp))
        torch.cuda.synchronize()
        host[i0:i0 + n] = d[:n].cpu().numpy()
    del d
    return oracle_binding.load().from_runs(host)


def run_rows(a, c):
    """--mode 1mm (configs[3]) and --mode extract (configs[4]) over the shards of every GPU of the job: what
    each rank's shards give is gathered on rank 0 and laid side by side in global shard order, the way the
    reference's front-end concatenates its partitions' replies (src/service/server.cpp:199-261)."""
    torch, dist, L, rsb = c.torch, c.dist, c.L, c.rsb
    from readserver_amd import sharded
    rank, world, local, dev, cdev, sp = c.rank, c.world, c.local, c.dev, c.cdev, c.sp
    R, k, S = int(a.runs), a.k, a.shards_per_gpu
    mix = a.mix
    t0 = time.time()
    shards, sset, _ = build_shards(a, c, mix)
    n_sym = min(int(g.getBWLen()) for g in shards)
    if a.mode == "extract" and not L.rsbwt_opened_for_reads(shards[0].handle):
        # the plain layout: the owner's open-time step for a shard that will serve reads all the same -- its select samples
        # and the psi hints its lines have room for (include/rsbwt.h, rsbwt_prepare_extraction), before the timed region
        for h in shards:
            ok(c, L.rsbwt_prepare_extraction(h.handle))
        torch.cuda.synchronize()
    size_tables(a, c, sset, shards, S)
    t_build = time.time() - t0
    common = {"shards_per_gpu": S, "shards": world * S, "run_bytes_per_shard": R, "symbols_per_shard": n_sym,
              "stream": STREAM_NOTE[a.stream], "mix": mix + ": " + MIX_NOTE[mix], "k": k, **ktab_config(shards),
              "window_span": shards[0].window_span(), "index_build_s": round(t_build, 2),
              "layout": ("reads: a psi hint in every window line (RSBWT_OPEN_READS)" if L.rsbwt_opened_for_reads(shards[0].handle) else "plain"),
              "index_hbm_bytes_per_gpu": int(sum(int(L.rsbwt_hbm_bytes(g.handle)) for g in shards)),
              "multi_gpu": ("REHEARSAL on one GPU over gloo: not a measurement" if a.rehearse_on_one_gpu else "measured" if world > 1 else "one GPU")}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return dt

    if a.mode == "1mm":
        M, V = int(a.kmers), 3 * k + 1
        d_km = torch.empty((M, k), dtype=torch.uint8, device=dev)
        make_batch(a, c, shards, mix, M, k, d_km)
        d_pk = torch.empty(M, dtype=torch.int64, device=dev)
        d_ok = torch.empty(M, dtype=torch.uint8, device=dev)
        cap = 4 * M  # records per shard: a present 31-mer leaves itself and the odd variant
        d_scr = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(sset._s, M, k), dtype=torch.uint8, device=dev)
        # what travels at N > 1: every rank's [S][cap] record buffers and [S] counts (fixed size, so the gather is one
        # collective issued behind the next batch's search); rank 0 lays the lists out in global shard order
        gat_h = sharded.BlockGatherer((S, cap, 4), torch.int64, cdev, depth=2)
        gat_t = sharded.BlockGatherer((S,), torch.int64, cdev, depth=2)
        d_h = [torch.empty((S, cap, 4), dtype=torch.int64, device=dev) for _ in range(2)] if cdev != dev else None
        d_t = [torch.empty(S, dtype=torch.int64, device=dev) for _ in range(2)] if cdev != dev else None
        step_no = [0]

        def step():
            i = step_no[0]
            step_no[0] += 1
            hb, tb = gat_h.acquire(i), gat_t.acquire(i)
            h_dev, t_dev = (d_h[i % 2], d_t[i % 2]) if d_h is not None else (hb, tb)
            ok(c, L.rsbwt_pack_kmers_dev(ptr(d_km), M, k, k, ptr(d_pk), ptr(d_ok), local, sp))
            ok(c, L.rsbwt_set_hits_1mm_dev(sset._s, ptr(d_pk), ptr(d_ok), M, k, ptr(h_dev), cap, ptr(t_dev), ptr(d_scr), sp))
            if d_h is not None:
                hb.copy_(h_dev)
                tb.copy_(t_dev)
            gat_h.submit(i)
            gat_t.submit(i)

        for h in shards:
            ok(c, L.rsbwt_set_counting(h.handle, 1))
        ok(c, L.rsbwt_set_set_counting(sset._s, 1))
        step()
        torch.cuda.synchronize()
        w = [0] * 16
        for h in shards:
            wi = (C.c_uint64 * 16)()
            ok(c, L.rsbwt_last_search_counters(h.handle, wi))  # the resumed search of the variants, per shard
            ok(c, L.rsbwt_set_counting(h.handle, 0))
            w = [x + int(y) for x, y in zip(w, wi)]
        # ... or of all the set's shards in ONE launch (csrc/sets.hip, set_hits_1mm_fused): the set's own counters
        ws = (C.c_uint64 * 16)()
        ok(c, L.rsbwt_set_last_search_counters(sset._s, ws))
        ok(c, L.rsbwt_set_set_counting(sset._s, 0))
        fused = bool(L.rsbwt_set_hits_1mm_is_fused(sset._s, M, k))
        if fused:
            w = [int(x) for x in ws]
        for _ in range(a.warmup):
            step()
        gat_h.drain(); gat_t.drain(); barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        gat_h.drain(); gat_t.drain(); barrier()
        dt = max_over_ranks(time.perf_counter() - t1)
        kms = 0.0
        for h in shards:  # traced + resumed search kernels of the timed calls
            buf = (C.c_float * 64)()
            cnt = C.c_size_t()
            ok(c, L.rsbwt_search_history_ms(h.handle, buf, min(64, 2 * a.steps), C.byref(cnt)))
            kms += sum(buf[:cnt.value]) / max(cnt.value / 2, 1)
        last = step_no[0] - 1
        tot_local = (d_t[last % 2] if d_t is not None else gat_t.result(last)[0] if world == 1 else gat_t.acquire(last)).clone()
        hits_local = int(tot_local.sum().item())
        verified = None
        if world > 1:  # rank 0: the lists side by side in global shard order; every rank's checksum must be in it
            sent_h = gat_h.acquire(last)
            mine = torch.tensor([total(sent_h), int(tot_local.sum().item())], dtype=torch.int64, device=cdev)
            sums = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(sums, mine)
            if rank == 0:
                blocks, totals = gat_h.result(last), gat_t.result(last)
                rec, first = sharded.concat_hit_lists(blocks, [t.cpu() for t in totals])
                verified = (all(total(blocks[r]) == int(sums[r][0].item()) for r in range(world))
                            and int(first[-1]) == sum(int(x[1].item()) for x in sums) and rec.shape[0] == int(first[-1]))
        overflowed = bool((tot_local > cap).any().item())  # (a list longer than its buffer keeps its count and drops records)
        cpu_base = None
        if world == 1 and a.verify_rows > 0:
            # shard 0's list of the last batch against the oracle's exact search of every variant of the first k-mers -- and
            # that search, timed, is the CPU baseline of this mode (the composition SURVEY 8 f3 defines: 3k + 1 findInterval
            # calls per 31-mer, query.cpp:24-41, on every core this process may use)
            oix = oracle_of_shard0(a, c, mix)
            nv = int(min(max(a.verify_rows, 20000 if a.cpu_sample > 0 else 0), M))
            acgt = np.frombuffer(b"ACGT", np.uint8)
            kmh = d_km[:nv].cpu().numpy()
            sp_ = np.repeat(kmh[:, None, :], V, axis=1)
            alt_tab = np.array([[x for x in acgt if x != o][:3] for o in acgt], np.uint8)  # the bases other than A / C / G / T, in order
            code = np.searchsorted(acgt, kmh)  # 0..3 (the batch holds ACGT only)
            for pos in range(k):
                sp_[:, 1 + 3 * pos:4 + 3 * pos, pos] = alt_tab[code[:, pos]]
            nthreads = a.cpu_threads if a.cpu_threads > 0 else usable_cpus()
            t_c = time.perf_counter()
            vlo, vup = oix.find_intervals(sp_.reshape(-1, k), nthreads=nthreads)
            t_c = time.perf_counter() - t_c
            n0 = int(oix.bwlen())
            widx = np.nonzero((vup >= vlo) & (vup < np.uint64(n0)))[0]
            h0 = (d_h[last % 2] if d_h is not None else gat_h.acquire(last))[0]
            rec0 = h0[:min(int(tot_local[0].item()), cap)].cpu().numpy().view(np.uint64)
            rec0 = rec0[rec0[:, 2] < np.uint64(nv * V)]
            verified = bool(rec0.shape[0] == widx.size and np.array_equal(rec0[:, 2], widx.astype(np.uint64))
                            and np.array_equal(rec0[:, 0], vlo[widx]) and np.array_equal(rec0[:, 1], vup[widx])) and not overflowed
            cpu_base = {"value": nv / t_c, "unit": "(31-mer x shard) 1-mismatch searches/s", "cores": nthreads, "kind": "port",
                        "sample": f"the first {nv} 31-mers of the batch on shard 0's index ({int(a.runs)} run bytes): {V} exact searches each "
                                  f"(oracle/rlebwt_oracle.c, {nthreads} POSIX threads, {t_c:.1f} s), the same lists the GPU's are held to",
                        "exact_searches_per_s": nv * V / t_c, "gpu_matches_oracle_on_sample": verified}
            oix.close()
        hits_local = int(torch.minimum(tot_local, torch.full_like(tot_local, cap)).sum().item())  # (what the lists hold)
        alg = w[2] * LINE_BYTES + S * M * V * 24 + hits_local * 48
        # below 2^26 variant searches per shard the set's shards work side by side on streams of their own
        # (csrc/sets.hip): their kernels overlap, so the sum of their durations says nothing -- the step is priced
        side_by_side = (not fused and S > 1 and M * V < (1 << int(os.environ.get("RSBWT_SET_1MM_SIDE_LOG2", "26")))
                        and "RSBWT_SET_1MM_TURNS" not in os.environ)
        if side_by_side:
            kms = dt / a.steps * 1e3
        if fused:  # the traced and the resumed launch of every timed call, from the set's HIP events
            buf = (C.c_float * 64)()
            cnt = C.c_size_t()
            ok(c, L.rsbwt_set_search_history_ms(sset._s, buf, min(64, 2 * a.steps), C.byref(cnt)))
            kms = sum(buf[:cnt.value]) / max(cnt.value / 2, 1)
        out = {
            "metric": "31-mer 1-mismatch backward searches/sec on popBWT (BASELINE configs[3])",
            "value": world * S * M / (dt / a.steps), "unit": "(31-mer x shard) 1-mismatch searches/s",
            "kmers_per_s_all_shards": M / (dt / a.steps), "variant_searches_per_s": world * S * M * V / (dt / a.steps),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": dict(common, workload=f"configs[3]: 1-mismatch branching search of {M} 31-mers in each of {world * S} shards "
                           f"({S} per GPU), {V} variants per 31-mer resumed from the k-mer's traced search; output = every shard's "
                           "ordered hit list" + (", gathered on rank 0 and concatenated in shard order" if world > 1 else ""),
                           kmers_per_batch=M, variants_per_kmer=V, hits_per_batch_this_rank=hits_local,
                           lf_steps_per_variant=w[0] / (S * M * V), hit_lists_verified=verified,
                           travels=(None if world == 1 else f"[{S}][{cap}] 32-byte records + {S} counts per rank and batch")),
            "roofline": {"bound": "hbm", "achieved": alg / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": _pmc_traffic_mode("1mm", int(a.runs), S, M) if fused else None,
                         "frac_of_step": alg / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
                         "kernel": ("search_solo_kernel<WALK> (the k-mers walked, the three substitutions of every traced position stepped off the same "
                                    "line) + search_solo_kernel<WL> (the surviving variants and those inside the tables' reach): all the rank's shards "
                                    "in one launch each" if fused and "RSBWT_SET_1MM_NO_WALK" not in os.environ and "RSBWT_SET_1MM_NO_WORKLIST" not in os.environ else
                                    "search_lines_kernel (the k-mers traced, their variants resumed): all the rank's shards in one launch each"
                                    if fused else
                                    "search_lines_kernel (the k-mers traced, their variants resumed): the shards side by side, priced on the whole step"
                                    if side_by_side else
                                    "search_lines_kernel (the k-mers traced, their variants resumed), summed over the rank's shards"),
                         "kernel_ms": kms, "algorithmic_bytes_per_launch": alg, "line_reads_per_launch": w[2]},
            "cpu_baseline": cpu_base,
        }
    else:
        NR, stride, run = int(a.rows), a.stride, max(1, a.row_run)
        # N > 1: rank 0 also holds every rank's 2-bit reads of a batch; with 8 x 20 GB shards opened for reads (295.6 GB) what
        # is left of the HBM bounds the batch: one gather in flight from 4 ranks on, and fewer rows per shard if even
        # that does not fit (the same on every rank: the smallest that fits anywhere)
        gdepth = 2 if world <= 2 else 1
        if world > 1 and cdev == dev:
            free_b = torch.cuda.mem_get_info(dev)[0] - (2 << 30)
            need = lambda nr: 2 * S * nr * (stride + 4) + gdepth * S * nr * (stride // 4 + 4) * (1 + (world if rank == 0 else 0)) + S * nr * 16
            while NR > 1 << 16 and need(NR) > free_b:
                NR //= 2
            tt = torch.tensor([NR], dtype=torch.int64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            NR = int(tt.item())
        gen = torch.Generator(device=dev)
        gen.manual_seed(a.seed + 99)
        # rows as the reference extracts them: the rows of intervals (query.cpp:94-96 walks lower..upper), here runs
        # of `run` consecutive SA rows at random places of every shard
        starts = torch.randint(0, max(n_sym - run, 1), (S, (NR + run - 1) // run), generator=gen, device=dev, dtype=torch.int64)
        rows = (starts[:, :, None] + torch.arange(run, device=dev)[None, None, :]).reshape(S, -1)[:, :NR].contiguous()
        if stride % 16:
            raise SystemExit("bench.py --mode extract: --stride must be a multiple of 16")
        # results stay in HBM as [S][NR][stride] ASCII + lengths; what travels at N > 1 is their 2-bit form
        # (rsbwt_pack_reads_dev: a quarter of the bytes) + the lengths, gathered behind the next batch's walks
        d_full = [torch.empty((S, NR, stride), dtype=torch.uint8, device=dev) for _ in range(2)]
        d_lenb = [torch.empty((S, NR), dtype=torch.int32, device=dev) for _ in range(2)]
        gat_o = sharded.BlockGatherer((S, NR, stride // 4), torch.uint8, cdev, depth=gdepth) if world > 1 else None
        gat_l = sharded.BlockGatherer((S, NR), torch.int32, cdev, depth=gdepth) if world > 1 else None
        d_pl = torch.empty((S, NR), dtype=torch.int32, device=dev)
        step_no = [0]

        def step():
            i = step_no[0]
            step_no[0] += 1
            o_dev, l_dev = d_full[i % 2], d_lenb[i % 2]
            ok(c, L.rsbwt_set_extract_dev(sset._s, ptr(rows), NR, ptr(o_dev), stride, ptr(l_dev), ptr(d_pl), sp))
            if world > 1:
                ob, lb = gat_o.acquire(i), gat_l.acquire(i)
                if cdev == dev:
                    sharded.pack_reads(o_dev, l_dev, out=ob)
                    lb.copy_(l_dev)
                else:  # rehearsal: packed in HBM, gathered from host copies
                    ob.copy_(sharded.pack_reads(o_dev, l_dev))
                    lb.copy_(l_dev)
                gat_o.submit(i)
                gat_l.submit(i)

        def drain():
            if world > 1:
                gat_o.drain()
                gat_l.drain()

        ok(c, L.rsbwt_set_set_counting(sset._s, 1))
        step()
        torch.cuda.synchronize()
        xw = (C.c_uint64 * 16)()
        ok(c, L.rsbwt_set_last_search_counters(sset._s, xw))  # the walk kernels' counters: one launch sequence walks every shard of the GPU
        ok(c, L.rsbwt_set_set_counting(sset._s, 0))
        names = ["passes", "lanes_with_a_row", "steps", "lanes_on_a_continuation", "lines_fetched", "cycles", "cycles_fetch_to_landed", "steps_from_line_hint"]
        walk = {"prefix": dict(zip(names, [int(v) for v in xw[:8]])), "postfix": dict(zip(names, [int(v) for v in xw[8:]]))}
        lens0 = d_lenb[0].reshape(-1)
        ln = lens0.cpu().numpy().view(np.uint32)
        fits = ln != 0xFFFFFFFF
        bases = int(ln[fits].astype(np.int64).sum())
        # one line per symbol and the two '$' steps of every read that fits; walks cut at the buffer's end took `stride` steps
        steps_alg = bases + 2 * int(fits.sum()) + int((~fits).sum()) * stride
        for _ in range(a.warmup):
            step()
        drain(); barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        ev0.record()
        for _ in range(a.steps):
            step()
        ev1.record()
        drain(); barrier()
        dt = max_over_ranks(time.perf_counter() - t1)
        k_ms = ev0.elapsed_time(ev1) / a.steps  # the walk kernels of one batch (current stream; the gather runs beside them)
        verified = None
        cpu_base = None
        if world == 1 and a.verify_rows > 0:
            # reads of shard 0 of the last batch against the oracle's extractPrefix + extractPostfix of the same rows -- and
            # that walk, timed, is the CPU baseline of this mode (query.cpp:43-85 per row)
            oix = oracle_of_shard0(a, c, mix)
            lastb = (step_no[0] - 1) % 2
            # (the sample: every CPU this process may run on, >= 1e5 rows -- rso_extract_batch, POSIX threads sharing the
            # index as the reference's pool threads share one BWT*, service.cpp:1513 -- about 5 s)
            threads = a.cpu_threads or usable_cpus()
            pick = np.unique(np.linspace(0, NR - 1, int(min(max(a.verify_rows, 5e5 if a.cpu_sample > 0 else 0), NR))).astype(np.int64))
            pick_t = torch.from_numpy(pick).to(dev)
            o_h = d_full[lastb][0][pick_t].cpu().numpy()
            l_h = d_lenb[lastb][0][pick_t].cpu().numpy().view(np.uint32)
            r_h = rows[0][pick_t].cpu().numpy().astype(np.uint64)
            one = np.ascontiguousarray(r_h[::max(1, r_h.size // 4000)][:4000])
            t_1 = time.perf_counter()
            oix.extract_batch(one, stride=stride, nthreads=1)  # one thread, a sample of the sample
            t_1 = time.perf_counter() - t_1
            t_c = time.perf_counter()
            c_out, c_len, _c_pl = oix.extract_batch(r_h, stride=stride, nthreads=threads)
            t_c = time.perf_counter() - t_c
            # every read of the sample: the same length (UINT32_MAX where it does not fit the buffer) and the same bytes
            fit_c = c_len != 0xFFFFFFFF
            verified = bool(np.array_equal(c_len, l_h))
            if verified:
                keep = np.arange(stride)[None, :] < np.where(fit_c, c_len, 0)[:, None]
                verified = bool(np.array_equal(np.where(keep, c_out, 0), np.where(keep, o_h, 0)))
            cpu_base = {"value": pick.size / t_c, "unit": "reads/s", "cores": threads, "kind": "port",
                        "sample": f"{pick.size} evenly spaced rows of the batch on shard 0's index ({int(a.runs)} run bytes): extractPrefix + "
                                  f"extractPostfix each (oracle/rlebwt_oracle.c, rso_extract_batch: {threads} POSIX threads sharing one index = "
                                  f"every CPU this process may run on, {t_c:.1f} s), the same reads the GPU's are held to",
                        "single_thread_value": one.size / t_1, "single_thread_sample": f"{one.size} of those rows, {t_1:.1f} s",
                        "host_cpus_visible": os.cpu_count(),
                        "bases_per_s": int(c_len[fit_c].astype(np.int64).sum()) / t_c, "gpu_matches_oracle_on_sample": verified}
            oix.close()
        if world > 1:
            last = step_no[0] - 1
            sent, sent_l = gat_o.acquire(last), gat_l.acquire(last)
            # the 2-bit form carries this rank's reads exactly: unpacked, it is the ASCII bytes up to every read's length
            full, ln_d = d_full[last % 2], d_lenb[last % 2]
            back = sharded.unpack_reads(sent.to(dev), ln_d)
            lnc = torch.where(ln_d < 0, torch.zeros_like(ln_d), ln_d).to(torch.int64)
            masked = torch.where(torch.arange(stride, device=dev)[None, None, :] < lnc[..., None], full, torch.zeros_like(full))
            exact = torch.tensor([int(torch.equal(back, masked))], dtype=torch.int64, device=cdev)
            del back, masked
            mine = torch.tensor([total(sent), total(sent_l)], dtype=torch.int64, device=cdev)
            sums = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(sums, mine)
            dist.all_reduce(exact, op=dist.ReduceOp.MIN)
            if rank == 0:
                reads_all, lens_all = sharded.concat_reads(gat_o.result(last), gat_l.result(last))
                verified = (bool(exact.item()) and reads_all.shape[0] == world * S
                            and total(reads_all) == sum(int(x[0].item()) for x in sums)
                            and total(lens_all) == sum(int(x[1].item()) for x in sums))
        out = {
            "metric": "reads located and extracted per second on popBWT (BASELINE configs[4])",
            "value": world * S * NR / (dt / a.steps), "unit": "reads/s",
            "bases_per_s": world * bases / (dt / a.steps),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": dict(common, workload=f"configs[4]: extractPrefix + extractPostfix (query.cpp:43-85) of {NR} SA rows of each of "
                           f"{world * S} shards ({S} per GPU), rows in runs of {run} consecutive rows (the rows of an interval)"
                           + (", reads gathered on rank 0 and concatenated in shard order" if world > 1 else ""),
                           rows_per_shard=NR, row_run=run, stride=stride, rows_fitting_stride=int(fits.sum()) / max(ln.size, 1),
                           mean_read_length=bases / max(int(fits.sum()), 1), reads_verified=verified, walk_counters=walk,
                           window_lines_with_a_psi_hint=int(L.rsbwt_psi_hint_lines(shards[0].handle)) / max(int(shards[0].num_lines()) * 16 // 17, 1),
                           travels=(None if world == 1 else f"[{S}][{NR}][{stride // 4}] bytes of 2-bit bases (rsbwt_pack_reads_dev) + [{S}][{NR}] lengths per rank and batch")),
            "roofline": {"bound": "hbm", "achieved": steps_alg * LINE_BYTES / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": steps_alg * LINE_BYTES / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": _pmc_traffic_mode("extract", int(a.runs), S, NR),
                         "kernel": "extract_prefix_wave_kernel + move_prefix16_kernel + extract_postfix_wave_kernel: one launch each over all the rank's shards",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": steps_alg * LINE_BYTES, "lf_and_psi_steps": steps_alg},
            "cpu_baseline": cpu_base,
        }
    sset.close()
    for g in shards:
        g.close()
    return out


def _pmc_traffic(R, Q, S, k, stream, mix):
    """HBM bytes per search launch from the committed rocprofv3 PMC pass of this same command
    (profiles/pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes) -- only if that pass
    was measured on this very kernel and layout source; otherwise None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        if (d["kernel_source_sha"] == kernel_source_sha() and int(d["run_bytes_per_shard"]) == R
                and int(d["queries_per_batch"]) == Q and int(d["shards_per_gpu"]) == S and int(d["k"]) == k
                and d.get("stream", "mixed") == stream and d.get("mix", "disjoint") == mix):
            return d["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def verify_shards(a, L, pair, d_kmers, S, R, Q, k, mix, rank, local, dev, sp):
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
        seed = shard_seed(a, mix, rank, S, s)
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
    ref = reference_beside_port(threads)
    if isinstance(ref, dict) and "error" not in ref and a.ref_out_of_cache_runs > 0:
        ref["out_of_cache"] = reference_out_of_cache(a, threads, d_kmers.device)
    return {
        "value": m / dt, "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"{m} of the batch's {Q} k-mers (evenly spaced) on shard 0 (one of the resident shards: per-shard "
                  f"searches/s, to be set against S/s per shard); oracle/rlebwt_oracle.c, {threads} POSIX threads sharing "
                  f"one index = every CPU this process may run on",
        "single_thread_value": m1 / dt1,
        "single_thread_sample": f"{m1} other k-mers, run first on the cold index",
        "host_cpus_visible": os.cpu_count(), "index_build_s": round(t_index, 2),
        "gpu_matches_oracle_on_sample": match,
        "reference_beside_port": ref,
    }


def reference_beside_port(threads):
    """The REAL reference (oracle/_ref/libref_bwt.so: ReadServer's own src/bwt sources compiled in the build
    container, shipped with the working tree) beside the port (oracle/rlebwt_oracle.c) on this host's cores: the
    golden popBWT fixture (9.1e6 symbols, cache-resident: the reference is sound on it, tests/golden/make_golden.py),
    4e5 31-mers, same answers required.  Relates the port's numbers above to the reference; None where the compiled
    reference did not travel."""
    import ctypes as C
    import tempfile
    lib = os.path.join(ROOT, "oracle", "_ref", "libref_bwt.so")
    gold = os.path.join(ROOT, "tests", "golden")
    if not os.path.exists(lib):
        return None
    try:
        import oracle_binding
        import readserver_amd as rsb
        meta = json.load(open(os.path.join(gold, "popbwt_v1.json")))
        g = np.load(os.path.join(gold, "popbwt_v1.npz"))
        km = np.ascontiguousarray(np.tile(g["kmers31"][:10000], (40, 1)))
        Q, k = km.shape
        R = C.CDLL(lib)
        R.ref_open.restype = C.c_void_p
        R.ref_open.argtypes = [C.c_char_p]
        R.ref_find_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        out = {"index": "tests/golden popBWT fixture, %d symbols" % meta["num_symbols"], "kmers": Q}
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "popbwt_v1.bwt")
            rsb.synth_popbwt(path, None, **meta["synth"])
            h = R.ref_open(path.encode())
            oix = oracle_binding.load().load(path)
            lo, up = np.empty(Q, np.uint64), np.empty(Q, np.uint64)
            for t in (1, threads):
                t0 = time.perf_counter()
                R.ref_find_intervals(h, km.ctypes.data, Q, k, k, lo.ctypes.data, up.ctypes.data, t)
                tr = time.perf_counter() - t0
                t0 = time.perf_counter()
                olo, oup = oix.find_intervals(km, nthreads=t)
                to = time.perf_counter() - t0
                same = bool(np.array_equal(lo, olo) and np.array_equal(up, oup))
                out["%d_threads" % t] = {"reference_queries_per_s": Q / tr, "port_queries_per_s": Q / to, "reference_over_port": to / tr,
                                         "same_answers": same}
        return out
    except Exception as e:  # the checker's checker must not take the bench line down
        return {"error": repr(e)}


def run_exact_cxx(a):
    """--host cxx: configs[1] / configs[2] driven by ONE process over --gpus devices (readserver_amd/onehost.py: the
    sequence of C-ABI calls the C++ host makes per batch).  Same workload, same batch, same step definition as the
    per-rank host; one JSON line."""
    import threading
    import torch
    import readserver_amd as rsb
    from readserver_amd import onehost
    L = rsb.lib()
    G, S, R, Q, k = a.gpus, a.shards_per_gpu, int(a.runs), int(a.queries), a.k
    if not torch.cuda.is_available() or torch.cuda.device_count() < G:
        raise SystemExit(f"bench.py --host cxx --gpus {G}: {torch.cuda.device_count() if torch.cuda.is_available() else 0} device(s) visible")
    t0 = time.time()
    by_dev = [[None] * S for _ in range(G)]
    errs = []

    def build(g):  # a device's shards, one after the other; the devices side by side (a host thread each)
        try:
            dev = torch.device("cuda", g)
            with torch.cuda.device(dev):
                sp_ = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for s_ in range(S):
                    d_runs = torch.empty(R, dtype=torch.uint8, device=dev)
                    if L.rsbwt_synth_runs_dev(ptr(d_runs), R, shard_seed(a, a.mix, g, S, s_), g, sp_) != 0:
                        raise RuntimeError(L.rsbwt_last_error().decode())
                    torch.cuda.synchronize(dev)
                    by_dev[g][s_] = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), device=g,
                                               ktab_depth=(a.ktab_depth if a.ktab_depth > 0 else None), window_span=a.window_span)
                    del d_runs
                    torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=build, args=(g,)) for g in range(G)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise SystemExit("bench.py --host cxx: " + errs[0])
    host = onehost.OneProcessHost(by_dev, Q, k)
    if a.ktab_depth == 0:  # one depth for the whole job, out of the HBM that is free now that every buffer exists
        n_sym = by_dev[0][0].getBWLen()
        T = host.auto_table_depth()
        free_b = min(torch.cuda.mem_get_info(g)[0] for g in range(G))
        T, fmt = pick_tables(a, free_b, S, n_sym, T)
        if T >= 2:
            host.attach_tables(T, fmt)
    t_build = time.time() - t0
    # the batch (device 0), then a copy on every device
    ctx0 = Ctx()
    ctx0.torch, ctx0.dist, ctx0.L, ctx0.rsb = torch, None, L, rsb
    ctx0.rank, ctx0.world, ctx0.local = 0, 1, 0
    ctx0.dev = ctx0.cdev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx0.stream = torch.cuda.current_stream()
    ctx0.sp = C.c_void_p(ctx0.stream.cuda_stream)
    d_km0 = torch.empty((Q, k), dtype=torch.uint8, device=ctx0.dev)
    if a.mix == "population":
        make_batch(a, ctx0, by_dev[0], a.mix, Q, k, d_km0)
    else:  # (every device's shards are their own streams: present k-mers drawn from device 0's, as rank 0's share would be)
        make_batch(a, ctx0, by_dev[0], "population", Q, k, d_km0)
    torch.cuda.synchronize()
    d_km = [d_km0 if g == 0 else d_km0.to(torch.device("cuda", g)) for g in range(G)]
    # exact work of one step on device 0 (counting mode, untimed)
    ok(ctx0, L.rsbwt_set_set_counting(host.subsets[0]._s, 1))
    host.step(d_km)
    host.synchronize()
    w = (C.c_uint64 * 16)()
    ok(ctx0, L.rsbwt_set_last_search_counters(host.subsets[0]._s, w))
    ok(ctx0, L.rsbwt_set_set_counting(host.subsets[0]._s, 0))
    for _ in range(a.warmup):
        host.step(d_km)
    host.synchronize()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        host.step(d_km)
    host.synchronize()
    dt = time.perf_counter() - t1
    buf = (C.c_float * 64)()
    cnt = C.c_size_t()
    ok(ctx0, L.rsbwt_set_search_history_ms(host.subsets[0]._s, buf, min(a.steps, 64), C.byref(cnt)))
    k_ms = float(np.mean(list(buf[:cnt.value]))) if cnt.value else float("nan")
    verified = host.verify_last_gather()
    lf, oc, ln, kt = w[0], w[1], w[2], w[3]
    alg = ln * LINE_BYTES + S * Q * 40
    ms = dt / a.steps * 1e3
    out = {
        "metric": "31-mer backward-search queries/sec on popBWT, counted as (query x shard) backward searches/s", "value": G * S * Q / (dt / a.steps), "unit": "searches/s",
        "queries_per_s": Q / (dt / a.steps), "n_gpus": G, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": f"configs[2]: {G * S} of 64 suffix-shards over {G} GPU(s), {S} per GPU, exact match, one fused launch per GPU and batch"
                        + (", interval records gathered onto device 0 over RCCL (ncclSend / ncclRecv)" if G > 1 else " (one GPU: no gather)"),
            "host": "cxx: ONE process drives every device through the C-ABI (rsbwt_pack_kmers_dev, rsbwt_set_find_interval_pairs_dev, "
                    "rsbwt_pack_interval_pairs_dev on a stream per device; rsbwt_set_gather_intervals_dev on a second one, batch i "
                    "travelling while batch i + 1 is searched): readserver_amd/onehost.py, csrc/sets.hip",
            "mix": a.mix, "shards_per_gpu": S, "shards": G * S, "run_bytes_per_shard": R, "queries_per_batch": Q, "k": k,
            **ktab_config(by_dev[0]), "window_span": by_dev[0][0].window_span(), "index_build_s": round(t_build, 2),
            "mean_lf_steps_per_search": lf / max(S * Q, 1), "gather_verified": verified,
            "gathered_as": None if G == 1 else "10-byte {lower:40, width:40} records (rsbwt_pack_interval_pairs_dev), exact",
            "multi_gpu": "measured" if G > 1 else "one GPU",
        },
        "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "search_lines_kernel (device 0's launches)",
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg, "line_reads_per_launch": ln, "occ_lookups_per_launch": oc,
                     "ktab_starts_per_launch": kt,
                     "frac_of_step": (alg + Q * (k + 8 * ((k + 31) // 32) + 1) + S * Q * 24) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "cpu_baseline": None,
    }
    host.close()
    for sh in by_dev:
        for g_ in sh:
            g_.close()
    return out


def reference_out_of_cache(a, threads, dev):
    """The REAL reference (oracle/_ref/libref_bwt.so) beside the port in the regime the headline runs in: an index
    that fits no cache.  A run stream of the bench's kind (--ref-out-of-cache-runs run bytes, 2e9: 1.2e10 symbols) is
    written to local disk as an SGA .bwt (src/bwt/rlebwt_reader.cpp:27-48), checked for the two conditions under
    which the reference is sound (tests/golden/make_golden.py's: D1 the last 65,536-symbol window lies inside
    one 2^20-run bucket, D2 n does not reach a multiple of 65,536 inside the last run), opened by the reference's own
    RLEBWT(filename) and by the port; 1e6 DISTINCT 31-mers -- half drawn from the index by LF walks on the GPU (all 30
    steps), half random -- are searched by both on all usable threads, a tenth of them on one thread first, and the
    answers must agree.  Here the reference's ~6-7 dependent DRAM misses per Occ (BPTree::rank's levels,
    include/bwt/BPTree.h:69-129) show; the cache-resident fixture above hides them."""
    import ctypes as C
    import tempfile
    import torch
    import oracle_binding
    import readserver_amd as rsb
    L = rsb.lib()
    out = {}
    try:
        R2 = int(a.ref_out_of_cache_runs)
        seed = STREAM_STYLE[a.stream] | (a.seed * 1000003 + 424242)
        d_runs = torch.empty(R2, dtype=torch.uint8, device=dev)
        assert L.rsbwt_synth_runs_dev(ptr(d_runs), R2, seed, dev.index or 0, None) == 0
        torch.cuda.synchronize()
        tail = (d_runs[-400000:] & 31).cpu().numpy().astype(np.int64)
        # (in slices: the shards and their tables leave ~12 GB of HBM, and a whole-stream int64 temporary would be 15)
        n_all = sum(total(d_runs[i:i + (1 << 27)] & 31) for i in range(0, R2, 1 << 27))
        # drop trailing runs until D2 holds, then see that D1 does
        drop = 0
        while drop < 1000:
            n2, last = n_all - int(tail[tail.size - drop:].sum()), int(tail[tail.size - 1 - drop])
            if n2 // 65536 == (n2 - last) // 65536:
                break
            drop += 1
        R2 -= drop
        n2 = n_all - int(tail[tail.size - drop:].sum())
        back = np.cumsum(tail[:tail.size - drop][::-1])
        runs_in_last_window = int(np.searchsorted(back, n2 - ((n2 - 1) // 65536) * 65536, side="left")) + 1
        d1 = ((R2 - runs_in_last_window) >> 20) == ((R2 - 1) >> 20)
        out.update({"run_bytes": R2, "symbols": n2, "sound_for_the_reference": {"D1_last_window_in_one_2^20_run_bucket": bool(d1), "D2_no_65536_multiple_inside_the_last_run": True}})
        if not d1:
            out["error"] = "the last window straddles a 2^20-run bucket: the reference is unsound on this stream (SURVEY 8c)"
            return out
        Q, k = 1_000_000, 31
        gen = torch.Generator(device=dev).manual_seed(a.seed + 4711)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        d_km = lut[torch.randint(0, 4, (Q, k), generator=gen, device=dev, dtype=torch.uint8).long()]
        present = 0
        try:  # half of them drawn from the index (needs ~5 GB of HBM for a moment)
            with rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R2), device=dev.index or 0, ktab_depth=None) as g2:
                d_p = torch.empty((Q // 2, k), dtype=torch.uint8, device=dev)
                assert L.rsbwt_sample_present_kmers_dev(g2.handle, Q // 2, k, k, a.seed + 99, ptr(d_p), None) == 0
                torch.cuda.synchronize()
                d_km[::2] = d_p
                present = Q // 2
        except Exception as e:  # noqa: BLE001  (HBM is full of shards: random k-mers only)
            out["present_kmers_skipped"] = repr(e)
        km = np.unique(d_km.cpu().numpy(), axis=0)  # distinct
        km = km[np.random.default_rng(a.seed).permutation(km.shape[0])]
        Q = km.shape[0]
        host = d_runs[:R2].cpu().numpy()
        del d_runs
        torch.cuda.empty_cache()
        R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bwt.so"))
        R.ref_open.restype = C.c_void_p
        R.ref_open.argtypes = [C.c_char_p]
        R.ref_close.argtypes = [C.c_void_p]
        R.ref_find_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "out_of_cache.bwt")
            with open(path, "wb") as f:  # the 30-byte SGA header + the run bytes (rlebwt_reader.cpp:27-48)
                f.write((0xCACA).to_bytes(2, "little") + (0).to_bytes(8, "little") + n2.to_bytes(8, "little") + R2.to_bytes(8, "little") + (0).to_bytes(4, "little"))
                f.write(memoryview(host))
            del host
            t0 = time.perf_counter()
            h = R.ref_open(path.encode())
            out["reference_open_s"] = round(time.perf_counter() - t0, 2)
            t0 = time.perf_counter()
            oix = oracle_binding.load().load(path)
            out["port_open_s"] = round(time.perf_counter() - t0, 2)
        out["kmers"] = {"distinct": Q, "drawn_from_the_index": present}
        lo, up = np.empty(Q, np.uint64), np.empty(Q, np.uint64)
        for t, m in ((1, Q // 10), (threads, Q)):
            sub = np.ascontiguousarray(km[:m] if t != 1 else km[Q - m:])  # (the one-thread run on k-mers nobody has searched yet)
            t0 = time.perf_counter()
            R.ref_find_intervals(h, sub.ctypes.data, m, k, k, lo.ctypes.data, up.ctypes.data, t)
            tr = time.perf_counter() - t0
            t0 = time.perf_counter()
            olo, oup = oix.find_intervals(sub, nthreads=t)
            to = time.perf_counter() - t0
            out["%d_threads" % t] = {"kmers": m, "reference_queries_per_s": m / tr, "port_queries_per_s": m / to, "reference_over_port": to / tr,
                                     "same_answers": bool(np.array_equal(lo[:m], olo) and np.array_equal(up[:m], oup))}
        R.ref_close(h)
        oix.close()
        return out
    except Exception as e:  # the checker's checker must not take the bench line down
        out["error"] = repr(e)
        return out


if __name__ == "__main__":
    main()
