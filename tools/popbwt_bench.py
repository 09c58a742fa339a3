#!/usr/bin/env python3
"""The headline search on a VALID population BWT that does not fit the caches: the 8 shards one GPU holds of a
64-shard popBWT, each built on the GPU from its own reads (tools/popbwt_gpu.py's construction, bucketed so that a
shard of several 1e9 symbols sorts in HBM), then bench.py's measurement on them: 1e7 31-mers per batch, half cut
from the genome, half uniform random, every k-mer searched in all 8 shards by one fused launch.

Collection: a random genome of G bases, 64 haplotypes with shared SNPs, reads of 100 bases from both strands at
`--depth` x per shard and haplotype set, 1 % base errors, reverse-lexicographic sort + dedup, shards by the reversed
last three bases (src/util/load_data_into_rocksdb.cpp:45).  Only the reads of shards 0..7 are made (the per-GPU
load of BASELINE configs[2]: shard s -> GPU s / 8); reads whose last three bases carry a sequencing error are
dropped (they would change shard), which lowers the depth by 3 %.

Prints one JSON line (profiles/r03_popbwt_bench.json).
usage: tools/popbwt_bench.py [--symbols-per-shard 4e9 --depth 28 --queries 1e7 --steps 10]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402

dev = torch.device("cuda", 0)
L = rsb.lib()
RL, W = 100, 101
HBM_PEAK_GBS, LINE_BYTES, SEARCH_BYTES = 8000.0, 128, 40


def p(t):
    return C.c_void_p(t.data_ptr())


def ok(rc):
    if rc != 0:
        raise RuntimeError(L.rsbwt_last_error().decode())


def reads_of_first_shards(G, H, reads_per_hap, snp, err, seed, nshards=8):
    """[N, 100] uint8 codes 1..4 of the reads whose shard key (reversed last three bases) is < nshards, and their keys."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    genome = torch.randint(1, 5, (G,), generator=gen, device=dev, dtype=torch.uint8)
    nsites = int(G * snp)
    sites = torch.randint(0, G, (nsites,), generator=gen, device=dev)
    alt = ((genome[sites].long() - 1 + torch.randint(1, 4, (nsites,), generator=gen, device=dev)) % 4 + 1).to(torch.uint8)
    freq = torch.rand(nsites, generator=gen, device=dev) * 0.5
    ar = torch.arange(RL, device=dev)
    comp = torch.tensor([0, 4, 3, 2, 1], dtype=torch.uint8, device=dev)
    out, keys = [], []
    for h in range(H):
        hap = genome.clone()
        carries = torch.rand(nsites, generator=gen, device=dev) < freq
        hap[sites[carries]] = alt[carries]
        st = torch.randint(0, G - RL + 1, (reads_per_hap,), generator=gen, device=dev)
        rev = torch.rand(reads_per_hap, generator=gen, device=dev) < 0.5
        # the read's last three bases, straight from the haplotype: forward hap[st+99], [st+98], [st+97];
        # reverse strand: the complements of hap[st], hap[st+1], hap[st+2]
        b1 = torch.where(rev, comp[hap[st].long()], hap[st + RL - 1])
        b2 = torch.where(rev, comp[hap[st + 1].long()], hap[st + RL - 2])
        b3 = torch.where(rev, comp[hap[st + 2].long()], hap[st + RL - 3])
        key = (b1.long() - 1) * 16 + (b2.long() - 1) * 4 + (b3.long() - 1)
        keep = key < nshards
        st, rev = st[keep], rev[keep]
        r = hap[st[:, None] + ar[None, :]]
        r[rev] = comp[r[rev].long()].flip(1)
        if err > 0:
            e = torch.rand(r.shape, generator=gen, device=dev) < err
            sub = torch.randint(1, 4, r.shape, generator=gen, device=dev, dtype=torch.uint8)
            r = torch.where(e, ((r - 1 + sub) % 4 + 1).to(torch.uint8), r)
            key2 = (r[:, RL - 1].long() - 1) * 16 + (r[:, RL - 2].long() - 1) * 4 + (r[:, RL - 3].long() - 1)
            same = key2 == key[keep]
            r = r[same]
            k2 = key2[same]
        else:
            k2 = key[keep]
        out.append(r)
        keys.append(k2.to(torch.uint8))
        del hap, st, rev, r
    return torch.cat(out, 0), torch.cat(keys, 0), genome


def nonzero_big(flag_of_chunk, n, chunk=1 << 30):
    """Positions i in [0, n) with flag_of_chunk(i0, i1)[i - i0] set, as int64 (torch.nonzero takes at most 2^31 - 1
    elements a call)."""
    parts = []
    for i in range(0, n, chunk):
        j = min(n, i + chunk)
        parts.append(torch.nonzero(flag_of_chunk(i, j)).squeeze(1) + i)
    return torch.cat(parts) if len(parts) > 1 else parts[0]


def rlo_sort_dedup(reads):
    revd = reads.flip(1)
    n = reads.shape[0]
    perm = torch.arange(n, device=dev)
    nw = (RL + 20) // 21
    for w in range(nw - 1, -1, -1):
        k = torch.zeros(n, dtype=torch.int64, device=dev)
        for j in range(21):
            c = 21 * w + j
            if c < RL:
                k = k * 8 + revd[perm, c].long()
            else:
                k = k * 8
        _, idx = torch.sort(k, stable=True)
        perm = perm[idx]
        del k, idx
    reads = reads[perm]
    keep = torch.ones(n, dtype=torch.bool, device=dev)
    keep[1:] = (reads[1:] != reads[:-1]).any(1)
    return reads[keep]


def bwt_runs_bucketed(reads):
    """RLUnit bytes of the multi-string BWT of `reads` ([N, 100] codes, RLO-sorted): the suffixes are split by
    their first two symbols (25 buckets, in lexicographic order) and every bucket is sorted by itself -- five
    stable radix passes over 21-symbol keys -- so that nothing larger than a bucket's keys is ever in flight."""
    N = reads.shape[0]
    text = torch.zeros((N, W), dtype=torch.uint8, device=dev)
    text[:, :RL] = reads
    flat = text.reshape(-1)
    n = flat.numel()
    padded = torch.cat([flat, torch.zeros(21 * 5 + 2, dtype=torch.uint8, device=dev)])
    del text
    prev_all = torch.empty(n, dtype=torch.uint8, device=dev)
    at = 0
    # the second symbol of a suffix that is just '$' is padding (0): bucket code = 8 * first + second', second' = 0 after '$'
    first = padded[:n]
    CH = 1 << 28
    code = torch.empty(n, dtype=torch.uint8, device=dev)
    for i in range(0, n, CH):
        j = min(n, i + CH)
        f = first[i:j]
        s2 = torch.where(f == 0, torch.zeros((), dtype=torch.uint8, device=dev), padded[i + 1:j + 1])
        code[i:j] = f * 8 + s2
    for b in range(40):
        pos = nonzero_big(lambda i, j: code[i:j] == b, n)
        m = pos.numel()
        if m == 0:
            continue
        left = W - (pos % W)
        perm = pos
        nw = (W + 20) // 21
        for w in range(nw - 1, -1, -1):
            lf = left if w == nw - 1 else (W - (perm % W))
            k = torch.zeros(m, dtype=torch.int64, device=dev)
            for j in range(21):
                c = 21 * w + j
                k = k * 8 + torch.where(lf > c, padded[perm + c].long(), torch.zeros((), dtype=torch.int64, device=dev))
            _, idx = torch.sort(k, stable=True)
            perm = perm[idx]
            del k, idx, lf
        prev_all[at:at + m] = torch.where(perm % W == 0, torch.zeros((), dtype=torch.uint8, device=dev), padded[perm - 1])
        at += m
        del pos, perm, left
    assert at == n
    del code, padded
    prev = prev_all
    def run_start(i, j):  # position i starts a run: the first one, or a symbol that differs from the one before
        f = torch.ones(j - i, dtype=torch.bool, device=dev)
        lo = max(i, 1)
        f[lo - i:] = prev[lo:j] != prev[lo - 1:j - 1]
        return f
    starts = nonzero_big(run_start, n)
    lens = torch.diff(torch.cat([starts, torch.tensor([n], device=dev)]))
    syms = prev[starts]
    del prev, prev_all, starts
    units = (lens + 30) // 31
    first_unit = torch.cumsum(units, 0) - units
    R = int(units.sum().item())
    run_of_unit = torch.repeat_interleave(torch.arange(lens.numel(), device=dev), units)
    k_in_run = torch.arange(R, device=dev) - first_unit[run_of_unit]
    ulen = torch.minimum(lens[run_of_unit] - 31 * k_in_run, torch.tensor(31, device=dev))
    runs = ((syms[run_of_unit].long() << 5) | ulen).to(torch.uint8)
    return runs, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--symbols-per-shard", type=float, default=4e9)
    ap.add_argument("--depth", type=float, default=28.0, help="read bases per genome position and shard")
    ap.add_argument("--haplotypes", type=int, default=64)
    ap.add_argument("--snp", type=float, default=1e-3)
    ap.add_argument("--err", type=float, default=0.01)
    ap.add_argument("--queries", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--check", action="store_true", help="shard 0's run bytes against the unbucketed builder (small sizes)")
    ap.add_argument("--rows", type=float, default=0, help="also: extraction of this many rows per shard (the rows of the genomic 31-mers' intervals)")
    ap.add_argument("--kmers-1mm", type=float, default=0, help="also: 1-mismatch hit lists of this many genomic 31-mers, half with a planted substitution")
    a = ap.parse_args()
    print(json.dumps(run(a)))


def run(a, log=sys.stderr):
    """a: symbols_per_shard, depth, haplotypes, snp, err, queries, steps, seed, check.  Returns the record."""
    S, H, k, Q = 8, a.haplotypes, 31, int(a.queries)
    reads_per_shard = int(a.symbols_per_shard / W)
    G = int(reads_per_shard * RL / a.depth)
    reads_per_hap = int(reads_per_shard * 64 / H)  # of the whole 64-shard collection; an eighth of them is made
    t0 = time.time()
    reads, keys, genome = reads_of_first_shards(G, H, reads_per_hap, a.snp, a.err, a.seed)
    torch.cuda.synchronize()
    t_reads = time.time() - t0
    shards, nsym, nreads, runs_total, t_bwt = [], [], [], 0, 0.0
    units_hist = np.zeros(32, np.int64)
    for s in range(S):
        t1 = time.time()
        rs = rlo_sort_dedup(reads[keys == s])
        runs, n = bwt_runs_bucketed(rs)
        if a.check and s == 0:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import popbwt_gpu
            ref, n2, _ = popbwt_gpu.bwt_runs(rs)
            assert n2 == n and torch.equal(ref, runs), "bucketed builder differs from the plain one"
            del ref
        torch.cuda.synchronize()
        t_bwt += time.time() - t1
        units_hist += torch.bincount((runs & 31).long(), minlength=32).cpu().numpy()
        runs_total += int(runs.numel())
        if s == 0:
            host_runs0 = runs.cpu().numpy()  # for the oracle, after the timed region
        g = rsb.GpuBWT(device_runs=(runs.data_ptr(), int(runs.numel())), num_strings=int(rs.shape[0]), ktab_depth=None)
        assert g.getBWLen() == n
        shards.append(g)
        nsym.append(n)
        nreads.append(int(rs.shape[0]))
        del runs, rs
        torch.cuda.empty_cache()
        print(f"popbwt_bench: shard {s}: {n} symbols, {shards[-1].num_runs()} run bytes, {time.time() - t1:.1f} s", file=log, flush=True)
    del reads, keys
    torch.cuda.empty_cache()
    sset = rsb.ShardSet(shards)
    # validity: LF walks of random rows of shard 3 give reads whose k-mers the shard finds; every extracted read is 100 long
    rng = np.random.default_rng(2)
    rows = rng.integers(0, nsym[3], 2000).astype(np.uint64)
    got, _ = rsb.extract_reads(shards[3], rows, stride=128)
    valid_reads = all(len(r) == RL and set(r) <= set("ACGT") for r in got)
    km_chk = np.frombuffer("".join(r[30:61] for r in got).encode(), np.uint8).reshape(-1, 31)
    lo, up = rsb.find_intervals(shards[3], km_chk)
    valid_found = bool((up >= lo).all())
    # tables: one depth for the 8 shards out of what is free
    T = L.rsbwt_set_auto_ktab_depth(sset._s)
    free_b = torch.cuda.mem_get_info(dev)[0]
    while T < 16 and T >= 2 and S * 8 * 4 ** (T + 1) <= free_b - (16 << 30) and 4 ** (T + 1) <= min(nsym):
        T += 1
    T = min(T, 14)
    ok(L.rsbwt_set_attach_ktabs(sset._s, T))
    # the batch: half genomic 31-mers (either strand), half uniform random, interleaved
    gen = torch.Generator(device=dev)
    gen.manual_seed(a.seed + 5)
    asc = torch.tensor(list(b"$ACGT"), dtype=torch.uint8, device=dev)
    comp = torch.tensor([0, 4, 3, 2, 1], dtype=torch.uint8, device=dev)
    codes = torch.randint(1, 5, (Q, k), generator=gen, device=dev, dtype=torch.uint8)
    gs = torch.randint(0, G - k + 1, (Q // 2,), generator=gen, device=dev)
    gk = genome[gs[:, None] + torch.arange(k, device=dev)[None, :]]
    rv = torch.rand(Q // 2, generator=gen, device=dev) < 0.5
    gk[rv] = comp[gk[rv].long()].flip(1)
    codes[0::2][:Q // 2] = gk
    d_km = asc[codes.long()].contiguous()
    del codes, gk
    d_pk = torch.empty(Q, dtype=torch.int64, device=dev)
    d_ok = torch.empty(Q, dtype=torch.uint8, device=dev)
    d_pairs = torch.empty((S, Q, 2), dtype=torch.int64, device=dev)

    def step():
        ok(L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, None))
        ok(L.rsbwt_set_find_interval_pairs_dev(sset._s, p(d_pk), p(d_ok), Q, k, p(d_pairs), None))

    ok(L.rsbwt_set_set_counting(sset._s, 1))
    step()
    torch.cuda.synchronize()
    w = (C.c_uint64 * 16)()
    ok(L.rsbwt_set_last_search_counters(sset._s, w))
    ok(L.rsbwt_set_set_counting(sset._s, 0))
    lf, oc, ln = int(w[0]), int(w[1]), int(w[2])
    kernel_name = "search_solo_kernel" if int(w[12]) else "search_lines_kernel"  # (WORK_SOLO: which kernel the launch ran on)
    width = torch.clamp(d_pairs[..., 1] - d_pairs[..., 0] + 1, min=0)
    gen_present = (width[:, 0::2] > 0).float()
    frac_shards = float(gen_present.mean().item())
    wpos = width[:, 0::2][width[:, 0::2] > 0].float()
    rand_present = float((width[:, 1::2] > 0).float().mean().item())
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / a.steps
    buf = (C.c_float * 64)()
    cnt = C.c_size_t()
    ok(L.rsbwt_set_search_history_ms(sset._s, buf, min(a.steps, 64), C.byref(cnt)))
    kms = float(np.mean(list(buf[:cnt.value])))
    alg = ln * LINE_BYTES + S * Q * SEARCH_BYTES
    hbm = sum(int(g.hbm_bytes()) for g in shards)
    # the oracle (CPU restatement of the reference algorithm: the checker) on a sample of the batch, shard 0
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_binding
    oix = oracle_binding.load().from_runs(host_runs0, num_strings=nreads[0])
    sel = torch.from_numpy(np.linspace(0, Q - 1, min(Q, 1000000)).astype(np.int64)).to(dev)
    km = d_km[sel].cpu().numpy()
    t2 = time.perf_counter()
    elo, eup = oix.find_intervals(km, nthreads=len(os.sched_getaffinity(0)))
    t_cpu = time.perf_counter() - t2
    got = d_pairs[0][sel].cpu().numpy().view(np.uint64)
    matches = bool(np.array_equal(got[:, 0], elo) and np.array_equal(got[:, 1], eup))
    units = units_hist[1:].sum()
    rows_rec = mm_rec = None
    if getattr(a, "rows", 0):
        # ---- configs[4] on the valid BWT: locate + extract = the reads of every row of the intervals of genomic 31-mers
        # (query.cpp:87-100 over every shard): rows in interval order, NR per shard, 128-byte buffers (reads are 100 long)
        NR, stride = int(a.rows), 128
        lo_all, up_all = d_pairs[..., 0], d_pairs[..., 1]
        rows_t = torch.empty((S, NR), dtype=torch.int64, device=dev)
        for si in range(S):
            wd = torch.clamp(up_all[si] - lo_all[si] + 1, min=0)
            wd = torch.where(wd > 64, torch.zeros_like(wd), wd)  # (leave the few very wide intervals out)
            first = torch.cumsum(wd, 0) - wd
            total = int(wd.sum().item())
            owner = torch.repeat_interleave(torch.arange(Q, device=dev), wd)
            rr = lo_all[si][owner] + (torch.arange(total, device=dev) - first[owner])
            reps = (NR + total - 1) // max(total, 1)
            rows_t[si] = rr.repeat(reps)[:NR]
            del wd, first, owner, rr
        d_out = torch.empty((S, NR, stride), dtype=torch.uint8, device=dev)
        d_len = torch.empty((S, NR), dtype=torch.int32, device=dev)
        d_pl = torch.empty((S, NR), dtype=torch.int32, device=dev)
        run_x = lambda: ok(L.rsbwt_set_extract_dev(sset._s, p(rows_t), NR, p(d_out), stride, p(d_len), p(d_pl), None))
        run_x()  # builds the select samples and psi hints of every shard
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run_x()
        ev0.record()
        for _ in range(3):
            run_x()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 3
        lens = d_len.cpu().numpy()
        all100 = bool((lens == RL).all())
        steps = int(lens.astype(np.int64).sum()) + 2 * lens.size
        # the oracle's extraction of a sample of shard 0's rows
        r0 = rows_t[0].cpu().numpy()
        o0, p0 = d_out[0].cpu().numpy(), d_pl[0].cpu().numpy()
        same = True
        for i in range(0, NR, max(1, NR // 3000)):
            pre, post = oix.extract(int(r0[i]), cap=512)
            same = same and o0[i, :lens[0, i]].tobytes().decode() == pre + post and int(p0[i]) == len(pre)
        rows_rec = {"rows_per_shard": NR, "stride": stride, "every_read_is_100_bases": all100, "reads_per_s": S * NR / (ms * 1e-3),
                    "ms_per_batch": ms, "psi_hint_lines_fraction": int(L.rsbwt_psi_hint_lines(shards[0].handle)) / max(shards[0].num_lines() * 16 // 17, 1),
                    "gpu_matches_oracle_on_sample": bool(same),
                    "roofline": {"bound": "hbm", "achieved": steps * LINE_BYTES / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": steps * LINE_BYTES / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "lf_and_psi_steps": steps}}
        del rows_t, d_out, d_len, d_pl
    if getattr(a, "kmers_1mm", 0):
        # ---- configs[3] on the valid BWT: genomic 31-mers, half of them with one planted substitution
        M, V = int(a.kmers_1mm), 3 * k + 1
        km1 = d_km[0::2][:M].clone()
        pos = torch.randint(0, k, (M,), generator=gen, device=dev)
        flip = torch.rand(M, generator=gen, device=dev) < 0.5
        cur = km1[torch.arange(M, device=dev), pos]
        alt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[(torch.searchsorted(torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev), cur) + 1 + torch.randint(0, 3, (M,), generator=gen, device=dev)) % 4]
        km1[torch.arange(M, device=dev)[flip], pos[flip]] = alt[flip]
        pk1 = torch.empty(M, dtype=torch.int64, device=dev)
        ok1 = torch.empty(M, dtype=torch.uint8, device=dev)
        ok(L.rsbwt_pack_kmers_dev(p(km1), M, k, k, p(pk1), p(ok1), 0, None))
        cap = 8 * M
        d_hits = torch.empty((S, cap, 4), dtype=torch.int64, device=dev)
        d_tot = torch.zeros(S, dtype=torch.int64, device=dev)
        d_scr = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(sset._s, M, k), dtype=torch.uint8, device=dev)
        run_m = lambda: ok(L.rsbwt_set_hits_1mm_dev(sset._s, p(pk1), p(ok1), M, k, p(d_hits), cap, p(d_tot), p(d_scr), None))
        run_m()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(5):
            run_m()
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t3) / 5
        tot = d_tot.cpu().numpy()
        # the oracle's exact search of every variant of a sample of the k-mers, shard 0
        kmh = km1[:200].cpu().numpy()
        rec0 = d_hits[0, :int(tot[0])].cpu().numpy().view(np.uint64)
        same1 = True
        for qi in range(200):
            w = kmh[qi].tobytes().decode()
            want = []
            lo_, up_ = oix.find_interval(w)
            if up_ >= lo_:
                want.append((qi * V, lo_, up_))
            v = 1
            for pp in range(k):
                for al in [c for c in "ACGT" if c != w[pp]]:
                    lo_, up_ = oix.find_interval(w[:pp] + al + w[pp + 1:])
                    if up_ >= lo_:
                        want.append((qi * V + v, lo_, up_))
                    v += 1
            mine = rec0[(rec0[:, 2] >= qi * V) & (rec0[:, 2] < (qi + 1) * V)]
            same1 = same1 and [(int(r[2]), int(r[0]), int(r[1])) for r in mine] == want
        mm_rec = {"kmers": M, "with_a_planted_substitution": 0.5, "hits_per_shard_mean": float(tot.mean()),
                  "kmer_x_shard_searches_per_s": S * M / dtm, "variant_searches_per_s": S * M * V / dtm, "ms_per_batch": dtm * 1e3,
                  "gpu_matches_oracle_on_sample": bool(same1)}
        del d_hits, d_scr
    out = {
        "what": "bench.py's exact search on the 8 shards one GPU holds of a VALID 64-shard population BWT built on the GPU "
                "(tools/popbwt_bench.py): 1e7 31-mers per batch, half genomic (either strand), half uniform random",
        "genome_len": G, "haplotypes": H, "depth_per_shard": a.depth, "snp_rate": a.snp, "error_rate": a.err,
        "shards": S, "of_a_collection_of": 64, "symbols_per_shard": {"min": min(nsym), "max": max(nsym)}, "reads_per_shard": nreads,
        "symbols": int(sum(nsym)), "run_bytes": runs_total, "symbols_per_run_byte": float(sum(nsym) / units),
        "unit_length_histogram_fraction": {str(i): round(float(units_hist[i] / units), 4) for i in (1, 2, 3, 4, 5, 8, 16, 31)},
        "index_hbm_bytes": hbm, "window_span": shards[0].window_span(), "ktab_depth": T,
        "seconds": {"reads": round(t_reads, 1), "sort_dedup_and_bwt_8_shards": round(t_bwt, 1)},
        "validity": {"extracted_reads_are_100_bases_of_ACGT": valid_reads, "their_31mers_are_found": valid_found},
        "queries_per_batch": Q,
        "genomic_31mers": {"fraction_of_shards_holding_one": frac_shards, "final_width_mean": float(wpos.mean().item()),
                           "final_width_median": float(wpos.median().item())},
        "random_31mers_present_fraction": rand_present,
        "oracle": {"shard": 0, "kmers": int(sel.numel()), "gpu_matches_oracle": matches,
                   "oracle_queries_per_s": float(sel.numel() / t_cpu), "threads": len(os.sched_getaffinity(0))},
        "mean_lf_steps_per_search": lf / (S * Q), "lines_per_lf_step": ln / max(lf, 1), "occ_lookups": oc,
        "extraction_of_the_rows_of_the_genomic_intervals": rows_rec, "one_mismatch_hit_lists": mm_rec,
        "searches_per_s": S * Q / dt, "queries_per_s": Q / dt, "ms_per_step": dt * 1e3,
        "roofline": {"bound": "hbm", "achieved": alg / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": kernel_name, "kernel_ms": kms,
                     "algorithmic_bytes_per_launch": alg, "line_reads_per_launch": ln},
    }
    sset.close()
    for g in shards:
        g.close()
    del d_pairs, d_km, d_pk, d_ok, genome
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
