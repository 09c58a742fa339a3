#!/usr/bin/env python3
"""A VALID population BWT of >= 1e9 symbols, built on the GPU, and what backward search looks like on it.

Bench infrastructure: the reference's pipeline (demo/build_bwt.sh: BFC, `sga index -a ropebwt`)
cannot run here, csrc/synth.cpp sorts suffixes on the host (<= 1e8 symbols), and bench.py's full-size shards
are a random run stream -- not a BWT.  This tool makes the real thing at the size a GPU allows in seconds and
MEASURES what the bench stream and query sampler have to look like:

  reads   : a random genome, H haplotypes carrying shared SNPs (allele frequencies uniform in (0, 1/2]),
            L-base reads from both strands at coverage c per haplotype, optional per-base error rate;
            reverse-lexicographic sort + dedup (src/util/rlosort_seq_and_convert_sample_names.cpp:16-20,42-63);
  shards  : 64 partitions by the reversed last three bases (src/util/load_data_into_rocksdb.cpp:45,
            demo/permutations-3.txt) = 64 contiguous ranges of the RLO-sorted reads;
  BWT     : per shard, every suffix of every read ($-terminated) sorted by LSD radix over five 21-symbol
            keys (torch.sort, stable; ties = identical suffixes stay in read order: $_i ordered by read
            index, the convention of csrc/synth.cpp and of the golden fixture), BWT[i] = the symbol before
            suffix i, run-length encoded as RLUnit bytes (include/bwt/rlunit.h:8-11) in HBM;
  index   : rsbwt_open_device_runs per shard, one shard set on the GPU.

Checked: shard 0's BWT against csrc/synth.cpp's host builder on a small instance (--selftest), LF-walks of
random rows give back reads of the collection, 31-mers cut from reads are found in their own shard.

Measured (one JSON line; profiles/r03_popbwt_calibration.json): run-length histogram of the RLUnit bytes and
symbols per unit; for 31-mers cut from the indexed reads, the fraction of the 64 shards that hold them, the
interval width after every step in the shards that do, the LF steps a (k-mer, shard) search takes, and the
distinct window lines per step counted by the search kernel itself.

usage: tools/popbwt_gpu.py [--genome 1e7 --haplotypes 64 --coverage 2 --read-len 100 --snp 1e-3 --err 0 --queries 2e5]
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


def make_reads(G, H, cov, L, snp, err, seed):
    """[N, L] uint8 codes 1..4 (A, C, G, T), RLO-sorted and de-duplicated."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    genome = torch.randint(1, 5, (G,), generator=gen, device=dev, dtype=torch.uint8)
    nsites = int(G * snp)
    sites = torch.randint(0, G, (nsites,), generator=gen, device=dev)
    alt = ((genome[sites].long() - 1 + torch.randint(1, 4, (nsites,), generator=gen, device=dev)) % 4 + 1).to(torch.uint8)
    freq = torch.rand(nsites, generator=gen, device=dev) * 0.5
    per_hap = int(G * cov / L)
    reads = torch.empty((H * per_hap, L), dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev)
    comp = torch.tensor([0, 4, 3, 2, 1], dtype=torch.uint8, device=dev)
    for h in range(H):
        hap = genome.clone()
        carries = torch.rand(nsites, generator=gen, device=dev) < freq
        hap[sites[carries]] = alt[carries]
        st = torch.randint(0, G - L + 1, (per_hap,), generator=gen, device=dev)
        r = hap[st[:, None] + ar[None, :]]
        rev = torch.rand(per_hap, generator=gen, device=dev) < 0.5
        r[rev] = comp[r[rev].long()].flip(1)
        reads[h * per_hap:(h + 1) * per_hap] = r
    if err > 0:
        e = torch.rand(reads.shape, generator=gen, device=dev) < err
        sub = torch.randint(1, 4, reads.shape, generator=gen, device=dev, dtype=torch.uint8)
        reads = torch.where(e, ((reads - 1 + sub) % 4 + 1).to(torch.uint8), reads)
    # RLO sort + dedup: lexicographic on the REVERSED read = LSD radix over words of 21 symbols of it
    revd = reads.flip(1)
    perm = torch.arange(reads.shape[0], device=dev)
    nw = (L + 20) // 21
    keys = []
    for w in range(nw):
        k = torch.zeros(reads.shape[0], dtype=torch.int64, device=dev)
        for j in range(21):
            c = 21 * w + j
            k = k * 8 + (revd[:, c].long() if c < L else 0)
        keys.append(k)
    for w in range(nw - 1, -1, -1):
        _, idx = torch.sort(keys[w][perm], stable=True)
        perm = perm[idx]
    reads = reads[perm]
    keep = torch.ones(reads.shape[0], dtype=torch.bool, device=dev)
    keep[1:] = (reads[1:] != reads[:-1]).any(1)
    return reads[keep], genome


def bwt_runs(reads):
    """RLUnit bytes (uint8 tensor in HBM) of the multi-string BWT of `reads` ([N, L] codes 1..4), and n."""
    N, L = reads.shape
    W = L + 1
    text = torch.zeros((N, W), dtype=torch.uint8, device=dev)
    text[:, :L] = reads
    flat = text.reshape(-1)
    n = flat.numel()
    # suffix at text position p (read p // W, offset p % W): symbols up to and including its '$', then 0s
    padded = torch.cat([flat, torch.zeros(21 * 5 + 1, dtype=torch.uint8, device=dev)])
    pos = torch.arange(n, device=dev)
    left = W - (pos % W)  # symbols of the suffix, '$' included
    perm = pos.clone()
    nw = (W + 20) // 21
    for w in range(nw - 1, -1, -1):
        p = perm
        lf = left[p]
        k = torch.zeros(n, dtype=torch.int64, device=dev)
        for j in range(21):
            c = 21 * w + j
            k = k * 8 + torch.where(lf > c, padded[p + c].long(), torch.zeros((), dtype=torch.int64, device=dev))
        _, idx = torch.sort(k, stable=True)
        perm = p[idx]
        del k, idx
    # BWT[i] = the symbol before suffix i; a read's first suffix is preceded by '$' (its own terminator's role)
    prev = torch.where(perm % W == 0, torch.zeros((), dtype=torch.uint8, device=dev), padded[perm - 1])
    # run-length encode: runs of equal symbols, split into units of at most 31
    change = torch.ones(n, dtype=torch.bool, device=dev)
    change[1:] = prev[1:] != prev[:-1]
    starts = torch.nonzero(change).squeeze(1)
    lens = torch.diff(torch.cat([starts, torch.tensor([n], device=dev)]))
    syms = prev[starts]
    units = (lens + 30) // 31
    first_unit = torch.cumsum(units, 0) - units
    R = int(units.sum().item())
    run_of_unit = torch.repeat_interleave(torch.arange(starts.numel(), device=dev), units)
    k_in_run = torch.arange(R, device=dev) - first_unit[run_of_unit]
    ulen = torch.minimum(lens[run_of_unit] - 31 * k_in_run, torch.tensor(31, device=dev))
    runs = ((syms[run_of_unit].long() << 5) | ulen).to(torch.uint8)
    return runs, n, lens


def selftest():
    """The GPU builder against csrc/synth.cpp's host builder: same reads -> same run bytes."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        bwt, rd = os.path.join(d, "a.bwt"), os.path.join(d, "a.reads")
        rsb.synth_popbwt(bwt, rd, seed=3, genome_len=40000, haplotypes=5, snp_rate=0.003, read_len=80, coverage=3.0)
        reads = open(rd).read().split()
        code = np.zeros(256, np.uint8)
        for i, c in enumerate(b"ACGT"):
            code[c] = i + 1
        arr = torch.from_numpy(code[np.frombuffer("".join(reads).encode(), np.uint8)].reshape(len(reads), 80)).to(dev)
        runs, n, _ = bwt_runs(arr)
        raw = open(bwt, "rb").read()
        host = np.frombuffer(raw[30:], np.uint8)
        same = np.array_equal(host, runs.cpu().numpy())
        print(json.dumps({"selftest": "GPU builder vs csrc/synth.cpp on the same reads", "reads": len(reads), "symbols": n,
                          "run_bytes": int(runs.numel()), "identical_run_bytes": bool(same)}))
        return 0 if same else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=float, default=1e7)
    ap.add_argument("--haplotypes", type=int, default=64)
    ap.add_argument("--coverage", type=float, default=2.0, help="per haplotype")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--snp", type=float, default=1e-3)
    ap.add_argument("--err", type=float, default=0.0)
    ap.add_argument("--queries", type=float, default=2e5)
    ap.add_argument("--kmers-from", choices=["genome", "reads"], default="genome",
                    help="where the present 31-mers are cut from: the base genome (what a user looks up) or the indexed "
                         "reads themselves (with --err > 0 a quarter of those carry a sequencing error)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--selftest", action="store_true")
    a = ap.parse_args()
    L = rsb.lib()
    if a.selftest:
        sys.exit(selftest())
    G, H, RL, Q, k = int(a.genome), a.haplotypes, a.read_len, int(a.queries), 31
    t0 = time.time()
    reads, genome = make_reads(G, H, a.coverage, RL, a.snp, a.err, a.seed)
    torch.cuda.synchronize()
    t_reads = time.time() - t0
    N = reads.shape[0]
    # shard key = the last three bases reversed: reads are RLO-sorted, so shards are contiguous ranges
    key = (reads[:, RL - 1].long() - 1) * 16 + (reads[:, RL - 2].long() - 1) * 4 + (reads[:, RL - 3].long() - 1)
    assert bool((key[1:] >= key[:-1]).all()), "RLO order = shard order"
    bounds = torch.searchsorted(key, torch.arange(65, device=dev)).cpu().numpy()
    t0 = time.time()
    shards, nsym, hist_units, hist_runs = [], [], np.zeros(32, np.int64), np.zeros(4096, np.int64)
    runs_total = 0
    for s in range(64):
        rs = reads[bounds[s]:bounds[s + 1]]
        runs, n, lens = bwt_runs(rs)
        hist_units += torch.bincount((runs & 31).long(), minlength=32).cpu().numpy()
        hist_runs += torch.bincount(torch.clamp(lens, max=4095), minlength=4096).cpu().numpy()
        runs_total += int(runs.numel())
        g = rsb.GpuBWT(device_runs=(runs.data_ptr(), int(runs.numel())), num_strings=int(rs.shape[0]), ktab_depth=None)
        assert g.getBWLen() == n
        shards.append(g)
        nsym.append(n)
        del runs
    torch.cuda.synchronize()
    t_bwt = time.time() - t0
    sset = rsb.ShardSet(shards)
    T = 8
    assert L.rsbwt_set_attach_ktabs(sset._s, T) == 0
    # ---- validity: LF walks of random rows give back reads of the collection
    rng = np.random.default_rng(1)
    lut = np.frombuffer(b"$ACGT", np.uint8)
    ok_reads = 0
    for s in (0, 17, 63):
        rows = rng.integers(0, nsym[s], 400).astype(np.uint64)
        got, _ = rsb.extract_reads(shards[s], rows, stride=128)
        blob = lut[reads[bounds[s]:bounds[s + 1]].cpu().numpy()].tobytes()
        mine = set(blob[i:i + RL] for i in range(0, len(blob), RL))
        ok_reads += sum(1 for r in got if r.encode() in mine)
    # ---- queries: 31-mers cut from reads of the whole collection
    gen = torch.Generator(device=dev)
    gen.manual_seed(a.seed + 1)
    ri = torch.randint(0, N, (Q,), generator=gen, device=dev)
    st = torch.randint(0, RL - k + 1, (Q,), generator=gen, device=dev)
    km_codes = reads[ri[:, None], st[:, None] + torch.arange(k, device=dev)[None, :]]
    if a.kmers_from == "genome":
        gs = torch.randint(0, G - k + 1, (Q,), generator=gen, device=dev)
        km_codes = genome[gs[:, None] + torch.arange(k, device=dev)[None, :]]
    asc = torch.tensor(list(b"$ACGT"), dtype=torch.uint8, device=dev)
    d_km = asc[km_codes.long()].contiguous()
    home = torch.bucketize(ri, torch.from_numpy(bounds[1:]).to(dev), right=True).cpu().numpy()  # the shard a k-mer's read lives in
    p = lambda t: C.c_void_p(t.data_ptr())
    d_pk = torch.empty(Q, dtype=torch.int64, device=dev)
    d_ok = torch.empty(Q, dtype=torch.uint8, device=dev)
    assert L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, None) == 0
    d_pairs = torch.empty((64, Q, 2), dtype=torch.int64, device=dev)
    assert L.rsbwt_set_set_counting(sset._s, 1) == 0
    assert L.rsbwt_set_find_interval_pairs_dev(sset._s, p(d_pk), p(d_ok), Q, k, p(d_pairs), None) == 0
    torch.cuda.synchronize()
    w = (C.c_uint64 * 16)()
    assert L.rsbwt_set_last_search_counters(sset._s, w) == 0
    assert L.rsbwt_set_set_counting(sset._s, 0) == 0
    width = torch.clamp(d_pairs[..., 1] - d_pairs[..., 0] + 1, min=0)  # [64][Q]
    present = (width > 0)
    found_home = bool(present[torch.from_numpy(home).to(dev), torch.arange(Q, device=dev)].all()) if a.kmers_from == "reads" else None
    shards_holding = present.sum(0).float()
    wpos = width[present].float()
    # interval width after every step: the suffix of length j of each k-mer, j = 1..31, in every shard
    sub = min(Q, 20000)
    per_step = []
    for j in range(1, k + 1):
        dk = d_km[:sub, k - j:].contiguous()
        pk = torch.empty(sub, dtype=torch.int64, device=dev)
        okb = torch.empty(sub, dtype=torch.uint8, device=dev)
        assert L.rsbwt_pack_kmers_dev(p(dk), sub, j, j, p(pk), p(okb), 0, None) == 0
        pr = torch.empty((64, sub, 2), dtype=torch.int64, device=dev)
        assert L.rsbwt_set_find_interval_pairs_dev(sset._s, p(pk), p(okb), sub, j, p(pr), None) == 0
        torch.cuda.synchronize()
        wd = torch.clamp(pr[..., 1] - pr[..., 0] + 1, min=0).float()
        alive = wd > 0
        per_step.append({"suffix_len": j, "fraction_of_searches_alive": float(alive.float().mean().item()),
                         "mean_width_alive": float(wd[alive].mean().item()) if bool(alive.any()) else 0.0,
                         "median_width_alive": float(wd[alive].median().item()) if bool(alive.any()) else 0.0})
    lf, oc, ln = int(w[0]), int(w[1]), int(w[2])
    units = hist_units[1:].sum()
    out = {
        "what": "valid population BWT built on the GPU (tools/popbwt_gpu.py) and backward search measured on it",
        "genome_len": G, "haplotypes": H, "coverage_per_haplotype": a.coverage, "read_len": RL, "snp_rate": a.snp,
        "error_rate": a.err, "present_kmers_cut_from": a.kmers_from, "reads_after_dedup": int(N), "symbols": int(sum(nsym)), "shards": 64,
        "symbols_per_shard": {"min": int(min(nsym)), "max": int(max(nsym))},
        "distinct_reads_per_genome_position_per_shard": N / G / 64, "depth_per_shard": N * RL / G / 64,
        "seconds": {"reads": round(t_reads, 1), "bwt_64_shards": round(t_bwt, 1)},
        "validity": {"lf_walk_reads_found_in_their_shard": f"{ok_reads} of 1200", "kmers_found_in_their_reads_shard": found_home},
        "run_units": {"run_bytes": int(runs_total), "symbols_per_unit": float(sum(nsym) / max(units, 1)),
                      "unit_length_histogram_fraction": {str(i): round(float(hist_units[i] / units), 4) for i in range(1, 32)},
                      "run_length_quantiles_symbols": {q: int(np.searchsorted(np.cumsum(hist_runs) / hist_runs.sum(), float(q)))
                                                       for q in ("0.5", "0.9", "0.99")},
                      "mean_run_symbols": float(sum(nsym) / hist_runs.sum())},
        "window_span": shards[0].window_span(), "ktab_depth": T,
        "present_31mers": {"queries": Q, "mean_fraction_of_shards_holding_one": float((shards_holding / 64).mean().item()),
                           "shards_holding_quantiles": [float(torch.quantile(shards_holding, q).item()) for q in (0.1, 0.5, 0.9)],
                           "final_width_where_present": {"mean": float(wpos.mean().item()), "median": float(wpos.median().item()),
                                                         "p90": float(torch.quantile(wpos[:4000000], 0.9).item())},
                           "lf_steps_per_search_behind_T8": lf / (64 * Q), "occ_lookups": oc, "distinct_lines": ln,
                           "lines_per_lf_step": ln / max(lf, 1)},
        "width_by_step": per_step,
    }
    print(json.dumps(out))
    sset.close()
    for g in shards:
        g.close()


if __name__ == "__main__":
    main()
