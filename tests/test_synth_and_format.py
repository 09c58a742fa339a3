"""CPU suite: the popBWT synthesiser and the SGA .bwt format (host-only product code)."""
import os

import numpy as np

import oracle_binding as ob


def _naive_bwt(reads):
    """Multi-string BWT by brute force: $ < A < C < G < T, $_i < $_j for i < j."""
    sufs = []
    for i, r in enumerate(reads):
        for o in range(len(r) + 1):
            sufs.append((r[o:], i, o))
    # the empty remainder ('$') sorts first because '' < any string; ties by read index
    sufs.sort(key=lambda s: (s[0] + "\x00", s[1]) if False else (tuple(" ACGT".index(c) for c in s[0]) + (0,), s[1]))
    return "".join(reads[i][o - 1] if o else "$" for _, i, o in sufs)


def test_synth_is_a_valid_bwt(rsb, tmp_path):
    bwt, rd = str(tmp_path / "s.bwt"), str(tmp_path / "s.reads")
    rsb.synth_popbwt(bwt, rd, seed=3, genome_len=400, haplotypes=3, snp_rate=0.01, read_len=25,
                     coverage=2.0)
    reads = open(rd).read().split()
    assert reads == sorted(set(reads), key=lambda r: r[::-1]), "reads must be RLO-sorted and unique"
    nstr, nsym, runs = ob.read_bwt_file(bwt)
    assert nstr == len(reads) and nsym == sum(len(r) + 1 for r in reads)
    got = "".join("$ACGT"[c] for c in ob.expand_runs(runs))
    assert got == _naive_bwt(reads)
    assert all(1 <= (u & 31) <= 31 for u in runs)


def test_synth_is_deterministic_and_sharded(rsb, tmp_path):
    kw = dict(seed=11, genome_len=3000, haplotypes=4, snp_rate=0.005, read_len=40, coverage=3.0)
    a, b = str(tmp_path / "a.bwt"), str(tmp_path / "b.bwt")
    rsb.synth_popbwt(a, None, **kw)
    rsb.synth_popbwt(b, None, **kw)
    assert open(a, "rb").read() == open(b, "rb").read()
    # 4 suffix shards partition the read set by the reversed last three bases
    all_reads = str(tmp_path / "all.reads")
    rsb.synth_popbwt(a, all_reads, **kw)
    whole = open(all_reads).read().split()
    parts = []
    for s in range(4):
        p = str(tmp_path / f"p{s}.reads")
        rsb.synth_popbwt(str(tmp_path / f"p{s}.bwt"), p, shard=s, num_shards=4, **kw)
        rs = open(p).read().split()
        for r in rs:
            key = "ACGT".index(r[-1]) * 16 + "ACGT".index(r[-2]) * 4 + "ACGT".index(r[-3])
            assert key * 4 // 64 == s
        parts += rs
    assert parts == whole


def test_interval_counts_match_read_occurrences(rsb, oracle, tmp_path):
    """findInterval's count equals the number of occurrences of w in the read set."""
    bwt, rd = str(tmp_path / "s.bwt"), str(tmp_path / "s.reads")
    rsb.synth_popbwt(bwt, rd, seed=5, genome_len=2000, haplotypes=4, snp_rate=0.01, read_len=50,
                     coverage=4.0)
    reads = open(rd).read().split()
    ix = oracle.load(bwt)
    rng = np.random.default_rng(0)
    for _ in range(200):
        r = reads[rng.integers(len(reads))]
        k = int(rng.integers(1, 30))
        s = int(rng.integers(0, len(r) - k + 1))
        w = r[s:s + k]
        lo, up = ix.find_interval(w)
        occ = sum(sum(1 for i in range(len(x) - k + 1) if x[i:i + k] == w) for x in reads)
        assert up - lo + 1 == occ
