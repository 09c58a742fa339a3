"""GPU suite (-m gpu): the HIP engine, called through the C-ABI, against the oracle and the
reference's golden vectors.  Bit-exact: every value is an integer or a byte."""
import os

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _random_runs(rng, R, with_dollar=True):
    sym = rng.integers(0 if with_dollar else 1, 5, R).astype(np.uint8)
    ln = rng.integers(1, 32, R).astype(np.uint8)
    return (sym << 5) | ln


def _random_kmers(rng, Q, k):
    return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (Q, k))]


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "popbwt_v1.npz"))


@pytest.fixture(scope="module")
def gix(rsb, fixture_bwt):
    path, meta = fixture_bwt
    g = rsb.GpuBWT(path)
    assert g.getBWLen() == meta["num_symbols"]
    assert g.num_runs() == meta["num_runs"]
    assert g.num_strings() == meta["num_strings"]
    yield g
    g.close()


# ---- golden vectors of the real reference ------------------------------------------------------

def test_gpu_golden_pc(gix, golden):
    assert [gix.getPC(c) for c in "$ACGT"] == golden["pc"].tolist()


def test_gpu_golden_31mers(rsb, gix, golden):
    lo, up = rsb.find_intervals(gix, golden["kmers31"])
    assert np.array_equal(lo, golden["lower31"])
    assert np.array_equal(up, golden["upper31"])
    cnt = rsb.count_kmers(gix, golden["kmers31"])
    exp = np.where(golden["upper31"] >= golden["lower31"], golden["upper31"] - golden["lower31"] + 1, 0)
    assert np.array_equal(cnt, exp.astype(np.uint64))


def test_gpu_golden_ladder(rsb, gix, golden, fixture_bwt):
    for kk in fixture_bwt[1]["ladder"]:
        lo, up = rsb.find_intervals(gix, golden[f"kmers{kk}"])
        assert np.array_equal(lo, golden[f"lower{kk}"]), kk
        assert np.array_equal(up, golden[f"upper{kk}"]), kk


@pytest.mark.parametrize("T", [4, 6, 9])
def test_gpu_golden_vectors_behind_a_grouped_table(rsb, golden, fixture_bwt, T):
    """The reference's own answers on the golden popBWT (a VALID BWT of '$'-terminated reads) from searches that start
    in the grouped k-mer table (rsbwt_attach_ktab_format): the four T-mers that differ in their last symbol tile one
    stretch of rows there too -- rows that begin with a shorter suffix and '$' sort before them -- so the records
    answer for the T-mers that occur, and the others are searched from initInterval: same intervals either way."""
    path, meta = fixture_bwt
    with rsb.GpuBWT(path, ktab_depth=T, ktab_grouped=True) as g:
        fmt, nbytes, left = g.ktab_info()
        assert (fmt, nbytes, g.ktab_depth()) == (1, 3 * 4 ** T, T)
        n = g.getBWLen()
        if 64 * 4 ** T <= n and 4 * n <= 16383 * 4 ** T:  # many rows per T-mer, a group's rows fit its record
            assert left < 0.5 * 4 ** T, (left, n)
        lo, up = rsb.find_intervals(g, golden["kmers31"])
        assert np.array_equal(lo, golden["lower31"]) and np.array_equal(up, golden["upper31"])
        for kk in meta["ladder"]:
            lo, up = rsb.find_intervals(g, golden[f"kmers{kk}"])
            assert np.array_equal(lo, golden[f"lower{kk}"]) and np.array_equal(up, golden[f"upper{kk}"]), kk


def test_gpu_golden_occ_char_occ_at(gix, golden):
    pos = golden["occ_pos"]
    for c, ch in enumerate("$ACGT"):
        assert np.array_equal(gix.occ_batch(ch, pos), golden["occ_tab"][c]), ch
    syms = bytes(b"$ACGT"[c] for c in golden["sel_sym"])
    assert np.array_equal(gix.occ_at_batch(syms, golden["sel_bc"]), golden["sel_idx"])
    # BWT[OccAt(b, bc)] == b
    ch = gix.char_batch(golden["sel_idx"])
    assert ch.tobytes() == syms


def test_gpu_golden_extract(rsb, gix, golden):
    # scalar mirrors (one GPU round trip per virtual call), as the C++ shim would drive them
    for r, e, n in list(zip(golden["rows"], golden["ext"], golden["ext_len"]))[:20]:
        s = rsb.extractPrefix(gix, int(r)) + rsb.extractPostfix(gix, int(r))
        assert s.encode() == e[:n].tobytes()
    # batched extraction kernels: every golden row
    reads, plen = rsb.extract_reads(gix, golden["rows"], stride=160)
    for s, pl, e, n, gpl in zip(reads, plen, golden["ext"], golden["ext_len"], golden["ext_prefix_len"]):
        assert s.encode() == e[:n].tobytes() and pl == gpl


@pytest.mark.parametrize("k", [1, 4, 5, 31, 32, 33])  # 301 k-mers: with even k the [m][3k+1] parts of the scratch have odd sizes
def test_gpu_one_mismatch_equals_composition_of_exact_searches(rsb, oracle, tmp_path, k):
    """configs[3] is not in the reference: the result is defined as the oracle's exact findInterval
    of every single-substitution variant."""
    bwt, rd = str(tmp_path / "s.bwt"), str(tmp_path / "s.reads")
    rsb.synth_popbwt(bwt, rd, seed=9, genome_len=30000, haplotypes=6, snp_rate=0.01, read_len=60, coverage=3.0)
    reads = open(rd).read().split()
    oix = oracle.load(bwt)
    rng = np.random.default_rng(k)
    kmers = []
    for _ in range(300):
        r = reads[rng.integers(len(reads))]
        s = rng.integers(0, len(r) - k + 1)
        w = list(r[s:s + k])
        if rng.random() < 0.5:  # plant one mismatch
            p = rng.integers(k)
            w[p] = "ACGT"[("ACGT".index(w[p]) + 1 + rng.integers(3)) % 4]
        kmers.append("".join(w))
    kmers.append("N" * k)
    with rsb.GpuBWT(bwt) as g:
        lo, up = rsb.find_intervals_1mm(g, kmers)
        assert lo.shape == (len(kmers), 3 * k + 1)
        # a k-mer with a symbol outside ACGT is invalid as a whole: every column is (1, 0)
        assert (lo[-1] == 1).all() and (up[-1] == 0).all()
        for qi, w in enumerate(kmers[:-1]):
            assert (int(lo[qi, 0]), int(up[qi, 0])) == oix.find_interval(w)
            v = 1
            for pos in range(k):
                for alt in [c for c in "ACGT" if c != w[pos]]:
                    var = w[:pos] + alt + w[pos + 1:]
                    assert (int(lo[qi, v]), int(up[qi, v])) == oix.find_interval(var), (w, pos, alt)
                    v += 1
        hits = rsb.hits_1mm(kmers[0], lo[0], up[0])
        assert hits == sorted(hits) and all(h[3] >= h[2] for h in hits)
        planted = sum(1 for qi in range(len(kmers) - 1) if len(rsb.hits_1mm(kmers[qi], lo[qi], up[qi])) > 0)
        assert planted >= 290 or k < 5
        # the compacted list (SURVEY 8 f3's output) = the non-empty cells of the matrices, in order
        hl = rsb.hits_1mm_batch(g, kmers, cap=16)  # too small a buffer: grown from the count returned
        want = []
        for qi, w in enumerate(kmers):
            want += [(qi,) + h for h in rsb.hits_1mm(w, lo[qi], up[qi])]
        got = [(int(r["query"]), int(r["pos"]), r["base"].decode(), int(r["lower"]), int(r["upper"])) for r in hl]
        assert got == want
        assert len(rsb.hits_1mm_batch(g, ["N" * k])) == 0


@pytest.mark.parametrize("T,span,k", [(2, 0, 31), (6, 300, 31), (12, 0, 31), (12, 2944, 40), (9, 0, 64),
                                      (8, 0, 9), (None, 0, 20)])
def test_gpu_one_mismatch_resumed_from_the_trace_equals_searching_every_variant(rsb, tmp_path, T, span, k):
    """Variants left of the k-mer table's reach resume from their k-mer's traced search; the result
    must be what the exact search gives for every variant spelled out (itself held to the oracle
    elsewhere), whatever the table depth (T = 2: wide entries, fallback starts), the window span
    (2944: most lookups continue in far lines), and with k-mers that span several packed words."""
    bwt, rd = str(tmp_path / "s.bwt"), str(tmp_path / "s.reads")
    rsb.synth_popbwt(bwt, rd, seed=21, genome_len=60000, haplotypes=4, snp_rate=0.005, read_len=80, coverage=4.0)
    reads = open(rd).read().split()
    rng = np.random.default_rng(100 * k + (T or 0))
    kmers = []
    for i in range(700):
        r = reads[rng.integers(len(reads))]
        s0 = rng.integers(0, len(r) - k + 1)
        w = list(r[s0:s0 + k])
        for _ in range(i % 3):  # 0, 1 or 2 planted substitutions
            p = rng.integers(k)
            w[p] = "ACGT"[("ACGT".index(w[p]) + 1 + rng.integers(3)) % 4]
        kmers.append("".join(w))
    kmers += ["A" * k, "T" * k, "ACGT" * (k // 4) + "A" * (k % 4), "N" + "A" * (k - 1)]
    with rsb.GpuBWT(bwt, ktab_depth=T, window_span=span) as g:
        lo, up = rsb.find_intervals_1mm(g, kmers)
        spelled = []
        for w in kmers:
            spelled.append(w)
            for pos in range(k):
                for alt in [c for c in "ACGT" if c != w[pos]][:3]:
                    spelled.append(w[:pos] + alt + w[pos + 1:])
        elo, eup = rsb.find_intervals(g, spelled)
        elo, eup = elo.reshape(len(kmers), -1), eup.reshape(len(kmers), -1)
        elo[-1], eup[-1] = 1, 0  # a k-mer with a foreign symbol is invalid as a whole
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)
        assert (up[:-1] >= lo[:-1]).sum() > len(kmers) // 2


def test_gpu_extract_vs_oracle_and_limits(rsb, oracle):
    import ctypes as C
    L = rsb.lib()
    runs = np.empty(400000, np.uint8)
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, runs.size, 77) == 0
    oix = oracle.from_runs(runs)
    rng = np.random.default_rng(5)
    with rsb.GpuBWT(runs=runs) as g:
        n = g.getBWLen()
        rows = np.concatenate([rng.integers(0, n, 3000), [0, 1, n - 1, n, n + 5]]).astype(np.uint64)
        stride = 192
        out = np.zeros((rows.size, stride), np.uint8)
        ln = np.empty(rows.size, np.uint32)
        pl = np.empty(rows.size, np.uint32)
        assert L.rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, stride,
                               ln.ctypes.data, pl.ctypes.data) == 0
        ok = 0
        for i, r in enumerate(rows):
            if r >= n:
                assert ln[i] == 0xFFFFFFFF
                continue
            buf = C.create_string_buffer(4096)
            a = oix.L.rso_extract_prefix(oix.h, int(r), buf, 4096)
            pre = buf.raw[:a] if a != C.c_size_t(-1).value else None
            b = oix.L.rso_extract_postfix(oix.h, int(r), buf, 4096)
            post = buf.raw[:b] if b != C.c_size_t(-1).value else None
            if pre is None or post is None or len(pre) + len(post) > stride or len(pre) > stride:
                if pre is not None and post is not None and len(pre) + len(post) <= stride:
                    pass
                else:
                    assert ln[i] == 0xFFFFFFFF or ln[i] == len(pre or b"") + len(post or b"")
                    continue
            assert ln[i] == len(pre) + len(post) and pl[i] == len(pre), (i, r)
            assert out[i, :ln[i]].tobytes() == pre + post
            ok += 1
        assert ok > 1000


def test_gpu_single_query_mirrors(rsb, gix, golden):
    w = golden["kmers31"][0].tobytes().decode()
    itv = rsb.findInterval(gix, w)
    assert (itv.lower, itv.upper) == (int(golden["lower31"][0]), int(golden["upper31"][0]))
    assert gix.getOcc("A", -1) == 0
    assert gix.getF(0) == "$" and gix.getF(gix.getBWLen() - 1) == "T"
    reads = rsb.query(gix, w)
    assert len(reads) == itv.upper - itv.lower + 1 and all(w in r for r in reads)
    assert rsb.query_exactmatch(gix, reads[0])
    assert not rsb.query_exactmatch(gix, w)
    assert rsb.query(gix, "ACGN") == []


# ---- oracle parity on seeded random run streams -------------------------------------------------

@pytest.mark.parametrize("R,span", [(1, 0), (2, 0), (95, 0), (96, 0), (97, 0), (1000, 0), (5000, 2),
                                    (5000, 37), (70000, 0), (70000, 2944), (300000, 700), (300000, 1800)])
def test_gpu_occ_char_vs_oracle(rsb, oracle, R, span):
    """The class BWT mirrors on every position, for windows of 2 symbols up to spans whose windows
    continue in spill chunks and chains of far lines."""
    rng = np.random.default_rng(R + span)
    runs = _random_runs(rng, R)
    oix = oracle.from_runs(runs)
    with rsb.GpuBWT(runs=runs, window_span=span) as g:
        n = g.getBWLen()
        assert n == oix.bwlen()
        if span:
            assert g.window_span() == span
        assert [g.getPC(c) for c in "$ACGT"] == [oix.pc(c) for c in "$ACGT"]
        pos = np.arange(n, dtype=np.uint64) if n <= 200000 else np.unique(np.concatenate([
            rng.integers(0, n, 100000), np.arange(n - 5000, n), np.arange(5000)])).astype(np.uint64)
        nv = ob.NaiveIndex(runs)
        for c, ch in enumerate("$ACGT"):
            got = g.occ_batch(ch, pos)
            assert np.array_equal(got, nv.cum[c, 1:][pos.astype(np.int64)].astype(np.uint64)), ch
            for p in pos[:: max(1, pos.size // 50)]:
                assert got[np.searchsorted(pos, p)] == oix.occ(ch, int(p))
        assert g.char_batch(pos).tobytes() == bytes(b"$ACGT"[x] for x in nv.bwt[pos.astype(np.int64)])
        for c, ch in enumerate("$ACGT"):
            where = np.nonzero(nv.bwt == c)[0]
            if where.size:
                bc = rng.integers(1, where.size + 1, min(2000, where.size)).astype(np.uint64)
                assert np.array_equal(g.occ_at_batch(ch, bc), where[bc.astype(np.int64) - 1].astype(np.uint64))


def test_gpu_dense_runs_force_far_line_chains(rsb, oracle):
    # all runs of length 1: a window of S symbols has S pieces, so spans of 300 and 2944 put every
    # window into chains of 3 and 32 far lines, which a lookup follows one line per pass
    rng = np.random.default_rng(9)
    runs = ((rng.integers(1, 5, 200000).astype(np.uint8)) << 5) | 1
    oix = oracle.from_runs(runs)
    for span in (0, 300, 2944):
        with rsb.GpuBWT(runs=runs, window_span=span) as g:
            assert (g.far_lines() > 0) == (span > 0)
            km = _random_kmers(rng, 20000, 9)
            lo, up = rsb.find_intervals(g, km)
            elo, eup = oix.find_intervals(km)
            assert np.array_equal(lo, elo) and np.array_equal(up, eup)


@pytest.mark.parametrize("span", [0, 1900])
@pytest.mark.parametrize("R,with_dollar", [(1, True), (2, False), (17, True), (50, True), (65, False), (3000, False),
                                           (200000, True), (4000000, True)])
def test_gpu_find_intervals_vs_oracle(rsb, oracle, R, with_dollar, span):
    rng = np.random.default_rng(77 + R)
    runs = _random_runs(rng, R, with_dollar)
    oix = oracle.from_runs(runs)
    with rsb.GpuBWT(runs=runs, window_span=span) as g:
        assert g.window_span() == span or span == 0
        for k in (1, 2, 7, 16, 31, 32, 33, 64, 65, 100):
            Q = 3000 if R < 1000000 else 20000
            km = _random_kmers(rng, Q, k)
            lo, up = rsb.find_intervals(g, km)
            elo, eup = oix.find_intervals(km, nthreads=8)
            assert np.array_equal(lo, elo), (R, k)
            assert np.array_equal(up, eup), (R, k)
            cnt = rsb.count_kmers(g, km)
            assert np.array_equal(cnt, np.where(eup >= elo, eup - elo + 1, 0).astype(np.uint64))


@pytest.mark.parametrize("span", [0, 1000])
def test_gpu_interval_at_the_top_of_a_bwt_without_terminators(rsb, oracle, span):
    """No '$' => C[A] = 0, so searches can sit at lower == 0, and a step that finds no b there gives
    upper = 0 + 0 - 1 = 2^64 - 1.  The reference does not see that as empty (unsigned compare,
    query.cpp:35) and takes one more step with Occ(b, 2^64 - 1) = 0 (rlebwt.cpp:269); so must we."""
    rng = np.random.default_rng(12)
    runs = (rng.integers(1, 5, 60000).astype(np.uint8) << 5) | 31   # long runs: many such cases
    oix = oracle.from_runs(runs)
    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=None) as g:
        km = _random_kmers(rng, 30000, 24)
        km[:, 12:23] = ord("A")      # ...AAAAAAAAAAAX: the poly-A suffix keeps lower at 0
        lo, up, steps = oix.find_intervals(km, want_steps=True)
        assert (steps > 12).any()
        glo, gup = rsb.find_intervals(g, km)
        assert np.array_equal(glo, lo) and np.array_equal(gup, up)


def test_gpu_invalid_and_ragged_inputs(rsb, oracle):
    rng = np.random.default_rng(3)
    runs = _random_runs(rng, 10000)
    oix = oracle.from_runs(runs)
    with rsb.GpuBWT(runs=runs) as g:
        km = _random_kmers(rng, 1000, 31)
        km[::7, 5] = ord("N")
        km[3, 30] = ord("$")
        km[4, 0] = ord("a")
        lo, up = rsb.find_intervals(g, km)
        elo, eup = oix.find_intervals(km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)
        assert (lo[::7] == 1).all() and (up[::7] == 0).all()
        assert (rsb.count_kmers(g, km)[::7] == 0).all()
        # empty batch and k == 0
        lo, up = rsb.find_intervals(g, np.zeros((0, 31), np.uint8))
        assert lo.size == 0
        lo, up = rsb.find_intervals(g, np.zeros((5, 0), np.uint8))
        assert (lo == 1).all() and (up == 0).all()


def test_gpu_present_kmers_survive_every_step(rsb, oracle):
    """rsbwt_sample_present_kmers_dev: LF-walk k-mers keep a non-empty interval for all k-1 steps."""
    import ctypes as C
    import torch
    L = rsb.lib()
    runs = np.empty(500000, np.uint8)  # the bench's run stream: ~1 % '$', so most walks survive
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, runs.size, 21) == 0
    oix = oracle.from_runs(runs)
    with rsb.GpuBWT(runs=runs) as g:
        Q, k = 5000, 31
        d = torch.empty((Q, k), dtype=torch.uint8, device="cuda:0")
        rc = L.rsbwt_sample_present_kmers_dev(g.handle, Q, k, k, 99, C.c_void_p(d.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        km = d.cpu().numpy()
        assert set(np.unique(km)) <= set(b"ACGT")
        lo, up, steps = oix.find_intervals(km, want_steps=True)
        assert (up >= lo).all() and (steps == k - 1).all()
        glo, gup = rsb.find_intervals(g, km)
        assert np.array_equal(glo, lo) and np.array_equal(gup, up)


def test_gpu_device_synth_equals_host_synth(rsb):
    import ctypes as C
    import torch
    L = rsb.lib()
    n = 1000003
    d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_synth_runs_dev(C.c_void_p(d.data_ptr()), n, 1234, 0, None) == 0
    torch.cuda.synchronize()
    h = np.empty(n, np.uint8)
    assert L.rsbwt_synth_runs_host(h.ctypes.data, n, 1234) == 0
    assert np.array_equal(d.cpu().numpy(), h)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100])
def test_gpu_pack_kernels_against_numpy(rsb, k):
    """2 bits per base, 32 bases per u64 (base i at bits 2i), validity = only ACGT (callers test
    find_first_not_of("ACGT"), service.cpp:299): dense input (stride == k, the 4-symbols-per-step
    kernel, every byte alignment of a k-mer's start) and strided input (the byte-wise kernel)."""
    import ctypes as C
    import torch
    L = rsb.lib()
    rng = np.random.default_rng(k)
    Q = 3001
    km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (Q, k))].copy()
    bad = rng.random(Q) < 0.2
    for q in np.nonzero(bad)[0]:
        km[q, rng.integers(k)] = rng.choice(np.frombuffer(b"N$acgtBDUZ@[`{\x00\xff", np.uint8))
    wpq = (k + 31) // 32
    code = np.zeros((256,), np.uint64)
    code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
    want = np.zeros((Q, wpq), np.uint64)
    for i in range(k):
        want[:, i // 32] |= code[km[:, i]] << np.uint64(2 * (i % 32))
    ok_want = np.isin(km, np.frombuffer(b"ACGT", np.uint8)).all(1)
    p = lambda t: C.c_void_p(t.data_ptr())
    for stride, off in ((k, 0), (k, 1), (k, 2), (k, 3), (k + 5, 0)):
        flat = torch.zeros(off + Q * stride + 64, dtype=torch.uint8, device="cuda:0")
        view = flat[off:off + Q * stride].view(Q, stride)
        view[:, :k] = torch.from_numpy(km).cuda()
        d_pk = torch.zeros((Q, wpq), dtype=torch.int64, device="cuda:0")
        d_ok = torch.zeros(Q, dtype=torch.uint8, device="cuda:0")
        assert L.rsbwt_pack_kmers_dev(C.c_void_p(flat.data_ptr() + off), Q, k, stride, p(d_pk), p(d_ok), 0, None) == 0
        torch.cuda.synchronize()
        got_ok = d_ok.cpu().numpy().astype(bool)
        assert np.array_equal(got_ok, ok_want), (stride, off)
        got = d_pk.cpu().numpy().view(np.uint64)
        assert np.array_equal(got[ok_want], want[ok_want]), (stride, off)


def test_gpu_host_interface_pipelines_big_batches_over_two_streams(rsb):
    """rsbwt_find_intervals / rsbwt_count cut a host batch into 2M-query slices that alternate
    between two streams and staging halves: 5.3M k-mers (three slices, the last one ragged) must
    come back exactly as one device-resident launch over the same k-mers gives them."""
    import ctypes as C
    import torch
    L = rsb.lib()
    R = 3000000
    d_runs = torch.empty(R, dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, 41, 0, None) == 0
    torch.cuda.synchronize()
    g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R))
    Q, k = 5300001, 20
    d_km = torch.empty((Q, k), dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_sample_present_kmers_dev(g.handle, Q, k, k, 3, C.c_void_p(d_km.data_ptr()), None) == 0
    torch.cuda.synchronize()
    rnd = torch.randint(0, 4, (Q // 3, k), device="cuda:0", dtype=torch.uint8)
    d_km[::3][:Q // 3] = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda:0")[rnd.long()]
    d_km[12345, 7] = ord("N")
    p = lambda t: C.c_void_p(t.data_ptr())
    d_pk = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    d_ok = torch.empty(Q, dtype=torch.uint8, device="cuda:0")
    d_lo = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    d_up = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    assert L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, None) == 0
    assert L.rsbwt_find_intervals_dev(g.handle, p(d_pk), p(d_ok), Q, k, p(d_lo), p(d_up), None) == 0
    torch.cuda.synchronize()
    km = d_km.cpu().numpy()
    lo, up = rsb.find_intervals(g, km)
    assert np.array_equal(lo, d_lo.cpu().numpy().view(np.uint64))
    assert np.array_equal(up, d_up.cpu().numpy().view(np.uint64))
    assert (lo[12345], up[12345]) == (1, 0) and (up >= lo).sum() > Q // 2
    cnt = rsb.count_kmers(g, km)
    assert np.array_equal(cnt, np.where(up >= lo, up - lo + 1, 0).astype(np.uint64))
    g.close()


def test_gpu_device_entry_points_and_work_counters(rsb, oracle):
    """The *_dev forms bench.py uses: pack + search on device buffers, HIP-event timing, and the
    exact LF-step / Occ / block counters against the oracle's step counts."""
    import ctypes as C
    import torch
    L = rsb.lib()
    rng = np.random.default_rng(8)
    R = 2000000
    d_runs = torch.empty(R, dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, 5, 0, None) == 0
    torch.cuda.synchronize()
    runs = d_runs.cpu().numpy()
    oix = oracle.from_runs(runs)
    g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), ktab_depth=None)  # every LF step is taken
    assert g.ktab_depth() == 0
    Q, k = 50000, 31
    km = _random_kmers(rng, Q, k)
    d_km = torch.from_numpy(km).cuda()
    d_pk = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    d_ok = torch.empty(Q, dtype=torch.uint8, device="cuda:0")
    d_lo = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    d_up = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    p = lambda t: C.c_void_p(t.data_ptr())
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, s) == 0
    assert L.rsbwt_set_counting(g.handle, 1) == 0
    assert L.rsbwt_find_intervals_dev(g.handle, p(d_pk), p(d_ok), Q, k, p(d_lo), p(d_up), s) == 0
    ms = C.c_float()
    assert L.rsbwt_last_search_ms(g.handle, C.byref(ms)) == 0 and ms.value > 0
    st, oc, bl = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert L.rsbwt_last_search_work(g.handle, C.byref(st), C.byref(oc), C.byref(bl)) == 0
    elo, eup, steps = oix.find_intervals(km, nthreads=8, want_steps=True)
    assert np.array_equal(d_lo.cpu().numpy().view(np.uint64), elo)
    assert np.array_equal(d_up.cpu().numpy().view(np.uint64), eup)
    assert st.value == int(steps.sum())
    assert oc.value == 2 * st.value and st.value <= bl.value <= oc.value
    assert bl.value > st.value  # wide intervals early in a search do span windows
    # distinct window lines, independently: lower-1 and upper share a line iff they share a window
    w = (C.c_uint64 * 16)()
    assert L.rsbwt_last_search_counters(g.handle, w) == 0
    assert (w[0], w[1], w[2]) == (st.value, oc.value, bl.value)
    assert w[11] <= 0.05 * w[2]  # continuation lines: the few lookups past their window's own pieces
    assert L.rsbwt_set_counting(g.handle, 0) == 0
    d_cnt = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    assert L.rsbwt_count_dev(g.handle, p(d_pk), p(d_ok), Q, k, p(d_cnt), s) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint64), np.where(eup >= elo, eup - elo + 1, 0).astype(np.uint64))
    g.close()


def test_gpu_file_open_and_shard_set(rsb, oracle, tmp_path):
    kw = dict(seed=13, genome_len=20000, haplotypes=4, snp_rate=0.004, read_len=60, coverage=3.0)
    shards, oixs = [], []
    for s in range(4):
        p = str(tmp_path / f"s{s}.bwt")
        rsb.synth_popbwt(p, None, shard=s, num_shards=4, **kw)
        shards.append(rsb.GpuBWT(p))
        oixs.append(oracle.load(p))
    whole = str(tmp_path / "whole.bwt")
    rd = str(tmp_path / "whole.reads")
    rsb.synth_popbwt(whole, rd, **kw)
    reads = open(rd).read().split()
    rng = np.random.default_rng(2)
    km = np.array([np.frombuffer(reads[i][j:j + 31].encode(), np.uint8)
                   for i, j in zip(rng.integers(0, len(reads), 2000), rng.integers(0, 29, 2000))])
    ss = rsb.ShardSet(shards)
    lo, up = ss.find_intervals(km)
    for s in range(4):
        elo, eup = oixs[s].find_intervals(km)
        assert np.array_equal(lo[s], elo) and np.array_equal(up[s], eup)
    # per-shard counts add up to the unsharded index's (the front-end sums partitions:
    # src/service/server.cpp:184-197)
    with rsb.GpuBWT(whole) as gw:
        assert np.array_equal(ss.count(km), rsb.count_kmers(gw, km))
    ss.close()
    for g in shards:
        g.close()


@pytest.mark.parametrize("style,span", [("synth", 0), ("synth", 256), ("synth", 1536), ("synth", 2944),
                                        ("dense", 0), ("dense", 2944), ("long", 0), ("long", 256),
                                        ("mixed", 0), ("mixed", 768)])
def test_gpu_window_layout_is_bit_exact(rsb, oracle, style, span):
    """The window-line layout: runs split at window borders, spill chunks for windows of 97..120
    pieces, chains of far lines beyond (forced by large spans over short runs), half-empty lines
    (small spans over long runs) -- the same intervals as the oracle, and the builder on the GPU
    lays the index out exactly as the same code does on the host (layout self-test)."""
    import ctypes as C
    L = rsb.lib()
    rng = np.random.default_rng(len(style) * 1000 + span)
    R = 300000
    if style == "synth":
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 555) == 0
    elif style == "dense":   # every run of length 1
        runs = (rng.integers(1, 5, R).astype(np.uint8) << 5) | 1
    elif style == "long":    # every unit full: 31 symbols per piece
        runs = (rng.integers(1, 5, R).astype(np.uint8) << 5) | 31
    else:                    # alternating dense and sparse stretches
        ln = np.where((np.arange(R) // 5000) % 2 == 0, 1, 31).astype(np.uint8)
        runs = (rng.integers(0, 5, R).astype(np.uint8) << 5) | ln
    oix = oracle.from_runs(runs)
    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=None) as g:
        S = g.window_span()
        assert 2 <= S <= 2944 and (S == span or span == 0)
        if (style == "dense" and span) or (style in ("synth", "mixed") and span >= 1536):
            assert g.far_lines() > 0
        if style == "long":
            assert g.far_lines() == 0 and g.spilled_symbols() == 0
        if span == 0 and style != "mixed":  # (no single span suits stretches of 1- and 31-symbol runs)
            assert g.spilled_symbols() <= 0.025 * g.getBWLen()
        # the same layout decisions as the host-side run of the builder's code
        st = (C.c_uint64 * 6)()
        bad = C.c_uint64()
        assert L.rsbwt_layout_selftest_host(runs.ctypes.data, R, S, st, C.byref(bad)) == 0
        assert (st[0], st[1], st[2], st[5]) == (S, g.num_lines(), g.far_lines(), g.spilled_symbols())
        for k in (1, 2, 9, 31, 40):
            km = _random_kmers(rng, 20000, k)
            km[::11] = km[0]
            lo, up = rsb.find_intervals(g, km)
            elo, eup = oix.find_intervals(km, nthreads=8)
            assert np.array_equal(lo, elo) and np.array_equal(up, eup), (style, span, k)
        # k-mers that exist (long matches walk through many windows)
        import torch
        d = torch.empty((4000, 31), dtype=torch.uint8, device="cuda:0")
        if L.rsbwt_sample_present_kmers_dev(g.handle, 4000, 31, 31, 5, C.c_void_p(d.data_ptr()), None) == 0:
            torch.cuda.synchronize()
            km = d.cpu().numpy()
            lo, up = rsb.find_intervals(g, km)
            elo, eup = oix.find_intervals(km, nthreads=8)
            assert np.array_equal(lo, elo) and np.array_equal(up, eup)


@pytest.mark.parametrize("grouped", [False, True])
@pytest.mark.parametrize("T", [2, 3, 5, 8, 11])
def test_gpu_kmer_table_is_bit_exact(rsb, oracle, T, grouped):
    """Searches that start from the k-mer table return exactly what the step-by-step search does
    (early exits inside the tabulated suffix included), for k below, at and above T, and for
    k-mers whose last T symbols straddle two packed words.  grouped: the table's 3-bytes-per-T-mer format
    (rsbwt_attach_ktab_format) -- T = 2, 3: every group is too wide for its record (all left to the search), T = 11:
    most T-mers do not occur (left to the search too), in between the records answer."""
    import ctypes as C
    L = rsb.lib()
    runs = np.empty(400000, np.uint8)
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, runs.size, 100 + T) == 0
    oix = oracle.from_runs(runs)
    rng = np.random.default_rng(T)
    with rsb.GpuBWT(runs=runs, ktab_depth=T, ktab_grouped=grouped) as g, rsb.GpuBWT(runs=runs, ktab_depth=None) as plain:
        assert g.ktab_depth() == T and plain.ktab_depth() == 0
        assert g.hbm_bytes() == plain.hbm_bytes() + (3 if grouped else 8) * 4 ** T  # both carry the same lines
        fmt, nbytes, left = g.ktab_info()
        assert (fmt, nbytes) == (1 if grouped else 0, (3 if grouped else 8) * 4 ** T) and plain.ktab_info() == (0, 0, 0)
        if not grouped:
            assert left == 0
        elif T <= 3:
            assert left == 4 ** T
        elif T == 8:
            assert 0 < left < 0.03 * 4 ** T  # (35 rows per 8-mer on average: nearly all occur)
        elif T == 11:
            assert left > 0.4 * 4 ** T
        for k in sorted({1, T - 1, T, T + 1, 12, 31, 32, 33, 32 + T // 2, 64, 65, 97}):
            if k < 1:
                continue
            km = _random_kmers(rng, 4000, k)
            km[::9] = ord("A")  # poly-A: long matches
            km[5, k // 2] = ord("N")
            lo, up = rsb.find_intervals(g, km)
            elo, eup = oix.find_intervals(km, nthreads=8)
            assert np.array_equal(lo, elo) and np.array_equal(up, eup), (T, k)
            plo, pup = rsb.find_intervals(plain, km)
            assert np.array_equal(lo, plo) and np.array_equal(up, pup), (T, k)


def test_gpu_kmer_table_auto_depth_and_counters(rsb, oracle):
    import ctypes as C
    import torch
    L = rsb.lib()
    R = 3000000
    runs = np.empty(R, np.uint8)
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 31) == 0
    oix = oracle.from_runs(runs)
    g = rsb.GpuBWT(runs=runs)
    T = g.ktab_depth()
    # auto depth: the table takes no more HBM than 5/4 of the index itself, and 4^T <= n
    assert 2 <= T <= 16 and 8 * 4 ** T <= 1.25 * (g.hbm_bytes() - 8 * 4 ** T) and 4 ** T <= g.getBWLen()
    Q, k = 30000, 31
    rng = np.random.default_rng(1)
    km = _random_kmers(rng, Q, k)
    d_km = torch.from_numpy(km).cuda()
    d_pk = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    d_ok = torch.empty(Q, dtype=torch.uint8, device="cuda:0")
    d_lo = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    d_up = torch.empty(Q, dtype=torch.int64, device="cuda:0")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, None) == 0
    assert L.rsbwt_set_counting(g.handle, 1) == 0
    assert L.rsbwt_find_intervals_dev(g.handle, p(d_pk), p(d_ok), Q, k, p(d_lo), p(d_up), None) == 0
    torch.cuda.synchronize()
    st, oc, bl, kt = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert L.rsbwt_last_search_work(g.handle, C.byref(st), C.byref(oc), C.byref(bl)) == 0
    assert L.rsbwt_last_search_ktab_lookups(g.handle, C.byref(kt)) == 0
    elo, eup, steps = oix.find_intervals(km, nthreads=8, want_steps=True)
    assert np.array_equal(d_lo.cpu().numpy().view(np.uint64), elo)
    assert kt.value == Q
    # the table replaces the first min(steps, T-1) LF steps of every query
    assert st.value == int(np.maximum(steps.astype(np.int64) - (T - 1), 0).sum())
    g.close()


# ---- full-size properties (sizes the oracle cannot sweep exhaustively) --------------------------

def test_gpu_large_index_properties(rsb, oracle):
    """R = 6e8 run bytes = 6.2e9 symbols: positions, counts and intervals beyond 2^32 (the high byte
    of the 40-bit counts, 64-bit window arithmetic).  sum_b Occ(b, p) == p + 1, Occ monotone,
    BWT[OccAt(b, c)] == b, and Occ / 31-mer intervals against the oracle built over the same bytes,
    sampled where p > 2^32."""
    import ctypes as C
    import torch
    L = rsb.lib()
    R = 600_000_000
    d_runs = torch.empty(R, dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, 77, 0, None) == 0
    torch.cuda.synchronize()
    g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R))
    n = g.getBWLen()
    assert n > (1 << 32) + (1 << 30) and g.ktab_depth() >= 12
    rng = np.random.default_rng(4)
    pos = np.sort(np.concatenate([rng.integers(0, n, 100000), rng.integers(1 << 32, n, 100000),
                                  [(1 << 32) - 1, 1 << 32, (1 << 32) + 1, n - 1]])).astype(np.uint64)
    tot = np.zeros(pos.size, np.uint64)
    occ = {}
    for ch in "$ACGT":
        o = g.occ_batch(ch, pos)
        assert (np.diff(o.astype(np.int64)) >= 0).all()
        tot += o
        occ[ch] = o
    assert np.array_equal(tot, pos + 1)
    assert occ["A"][-1] > (1 << 30) and g.getPC("T") > (1 << 32)
    for ch in "ACGT":
        last = g.getOcc(ch, n - 1)
        assert g.getPC(ch) + last == (g.getPC("ACGT"["ACGT".index(ch) + 1]) if ch != "T" else n)
        bc = rng.integers(1, last + 1, 5000).astype(np.uint64)
        idx = g.occ_at_batch(ch, bc)
        assert (g.char_batch(idx) == ord(ch)).all()
        assert np.array_equal(g.occ_batch(ch, idx), bc)
    runs = d_runs.cpu().numpy()
    del d_runs
    oix = oracle.from_runs(runs)
    hi = pos[pos > (1 << 32)][::400]
    for ch in "$ACGT":
        got = occ[ch][pos > (1 << 32)][::400]
        assert all(int(x) == oix.occ(ch, int(p)) for x, p in zip(got, hi)), ch
    # 31-mers: random (die early) and drawn from the index (all 30 steps, intervals beyond 2^32)
    km = _random_kmers(rng, 100000, 31)
    d = torch.empty((100000, 31), dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_sample_present_kmers_dev(g.handle, 100000, 31, 31, 5, C.c_void_p(d.data_ptr()), None) == 0
    torch.cuda.synchronize()
    km = np.concatenate([km, d.cpu().numpy()])
    lo, up = rsb.find_intervals(g, km)
    elo, eup = oix.find_intervals(km, nthreads=16)
    assert np.array_equal(lo, elo) and np.array_equal(up, eup)
    assert (lo[100000:] > (1 << 32)).sum() > 10000 and (up[100000:] >= lo[100000:]).all()
    g.close()


# ---- query / query_exactmatch behind the C-ABI (a13), pinned by the compiled reference ------------

def test_gpu_golden_query_and_query_exactmatch(rsb, gix, golden_dir):
    """rsbwt_query / rsbwt_query_exactmatch against the reference's own answers (query.cpp:87-120;
    tests/golden/make_query_golden.py): the same reads in the same order, the same booleans."""
    gq = np.load(os.path.join(golden_dir, "query_v1.npz"))
    for L_ in (70, 69, 40):
        got = rsb.query_exactmatch_batch(gix, gq[f"em_w{L_}"])
        assert np.array_equal(got, gq[f"em_ans{L_}"].astype(bool)), L_
    assert rsb.query_exactmatch_batch(gix, gq["em_w70"]).sum() > 100
    for k in (25, 31, 45):
        want_first, want_reads, want_len = gq[f"q_first{k}"], gq[f"q_reads{k}"], gq[f"q_len{k}"]
        got = rsb.query_batch(gix, gq[f"q_w{k}"], read_stride=96)
        assert len(got) == want_first.size - 1
        for q, reads in enumerate(got):
            exp = [want_reads[r, :want_len[r]].tobytes().decode() for r in range(int(want_first[q]), int(want_first[q + 1]))]
            assert reads == exp, (k, q)
    # single-string mirrors
    w = gq["q_w31"][0].tobytes().decode()
    assert rsb.query(gix, w) == rsb.query_batch(gix, [w])[0] and len(rsb.query(gix, w)) > 0
    assert rsb.query(gix, "ACGN") == [] and not rsb.query_exactmatch(gix, "ACGN")


# ---- thread safety (include/rsbwt.h): one handle, concurrent host callers -------------------------

def test_gpu_concurrent_callers_share_one_handle(rsb, oracle):
    """The reference answers from 8 + 64 pool threads on one shared BWT* (service.cpp:88-89,
    1532-1569).  Eight host threads hammer one handle with every host entry point at once; each must
    get exactly what a lone caller gets."""
    import threading
    L = rsb.lib()
    runs = np.empty(1500000, np.uint8)
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, runs.size, 404) == 0
    oix = oracle.from_runs(runs)
    g = rsb.GpuBWT(runs=runs, ktab_depth=8)
    n = g.getBWLen()
    work = []
    for t in range(8):
        rng = np.random.default_rng(1000 + t)
        km = _random_kmers(rng, 30000 + 1000 * t, 31 if t % 2 else 20)
        pos = rng.integers(0, n, 20000).astype(np.uint64)
        rows = rng.integers(0, n, 300).astype(np.uint64)
        work.append((km, pos, rows))
    expect = []
    for km, pos, rows in work:  # serial pass
        expect.append((rsb.find_intervals(g, km), rsb.count_kmers(g, km), g.occ_batch("C", pos), g.char_batch(pos),
                       rsb.extract_reads(g, rows, stride=4096)[0], rsb.find_intervals_1mm(g, km[:200])))
    elo, eup = oix.find_intervals(work[0][0], nthreads=8)
    assert np.array_equal(expect[0][0][0], elo) and np.array_equal(expect[0][0][1], eup)
    errors = []

    def run(t):
        try:
            km, pos, rows = work[t]
            for _ in range(3):
                lo, up = rsb.find_intervals(g, km)
                assert np.array_equal(lo, expect[t][0][0]) and np.array_equal(up, expect[t][0][1])
                assert np.array_equal(rsb.count_kmers(g, km), expect[t][1])
                assert np.array_equal(g.occ_batch("C", pos), expect[t][2])
                assert np.array_equal(g.char_batch(pos), expect[t][3])
                assert rsb.extract_reads(g, rows, stride=4096)[0] == expect[t][4]
                l1, u1 = rsb.find_intervals_1mm(g, km[:200])
                assert np.array_equal(l1, expect[t][5][0]) and np.array_equal(u1, expect[t][5][1])
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=run, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    g.close()


# ---- shard sets: one fused launch over the shards of a device --------------------------------------

def test_gpu_shard_set_fused_launch_device_resident(rsb, oracle):
    """rsbwt_set_find_intervals_dev / _count_dev: 5 shards of different sizes and table depths on
    device 0 searched by ONE launch ((query, shard) pairs drawn from per-shard pools), [S][Q]
    results against each shard's oracle; the set-level work counters add up."""
    import ctypes as C
    import torch
    L = rsb.lib()
    sizes = [200000, 1, 70000, 1200000, 333333]
    shards, oixs = [], []
    for i, R in enumerate(sizes):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 900 + i) == 0
        oixs.append(oracle.from_runs(runs))
        shards.append(rsb.GpuBWT(runs=runs, ktab_depth=[None, None, 5, 9, 0][i], window_span=[0, 0, 300, 0, 1700][i]))
    ss = rsb.ShardSet(shards)
    assert L.rsbwt_set_devices(ss._s) == 1
    rng = np.random.default_rng(6)
    for k, Q in ((31, 50000), (12, 20000), (40, 7000)):
        km = _random_kmers(rng, Q, k)
        km[5, 3] = ord("N")
        d_km = torch.from_numpy(km).cuda()
        wpq = (k + 31) // 32
        d_pk = torch.empty((Q, wpq), dtype=torch.int64, device="cuda:0")
        d_ok = torch.empty(Q, dtype=torch.uint8, device="cuda:0")
        d_lo = torch.empty((len(sizes), Q), dtype=torch.int64, device="cuda:0")
        d_up = torch.empty((len(sizes), Q), dtype=torch.int64, device="cuda:0")
        p = lambda t: C.c_void_p(t.data_ptr())
        assert L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, None) == 0
        assert L.rsbwt_set_set_counting(ss._s, 1) == 0
        assert L.rsbwt_set_find_intervals_dev(ss._s, p(d_pk), p(d_ok), Q, k, p(d_lo), p(d_up), None) == 0
        torch.cuda.synchronize()
        w = (C.c_uint64 * 16)()
        assert L.rsbwt_set_last_search_counters(ss._s, w) == 0
        assert L.rsbwt_set_set_counting(ss._s, 0) == 0
        lo, up = d_lo.cpu().numpy().view(np.uint64), d_up.cpu().numpy().view(np.uint64)
        steps = 0
        for s, oix in enumerate(oixs):
            elo, eup, st = oix.find_intervals(km, nthreads=8, want_steps=True)
            assert np.array_equal(lo[s], elo) and np.array_equal(up[s], eup), (k, s)
            T = shards[s].ktab_depth()
            steps += int(np.maximum(st.astype(np.int64) - (max(T, 1) - 1 if k >= T else 0), 0).sum())
        assert w[0] == steps and w[2] <= w[1] <= 2 * w[0]
        # the same intervals as {lower, upper} pairs (one 16-byte store per search), set and single handle
        d_pr = torch.empty((len(sizes), Q, 2), dtype=torch.int64, device="cuda:0")
        assert L.rsbwt_set_find_interval_pairs_dev(ss._s, p(d_pk), p(d_ok), Q, k, p(d_pr), None) == 0
        d_p3 = torch.empty((Q, 2), dtype=torch.int64, device="cuda:0")
        assert L.rsbwt_find_interval_pairs_dev(shards[3].handle, p(d_pk), p(d_ok), Q, k, p(d_p3), None) == 0
        torch.cuda.synchronize()
        pr = d_pr.cpu().numpy().view(np.uint64)
        assert np.array_equal(pr[:, :, 0], lo) and np.array_equal(pr[:, :, 1], up)
        assert np.array_equal(d_p3.cpu().numpy().view(np.uint64), pr[3])
        d_cnt = torch.empty((len(sizes), Q), dtype=torch.int64, device="cuda:0")
        assert L.rsbwt_set_count_dev(ss._s, p(d_pk), p(d_ok), Q, k, p(d_cnt), None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_cnt.cpu().numpy().view(np.uint64), np.where(up >= lo, up - lo + 1, 0).astype(np.uint64))
        # host entry points of the set: the same intervals, counts summed over the shards
        hlo, hup = ss.find_intervals(km)
        assert np.array_equal(hlo, lo) and np.array_equal(hup, up)
        assert np.array_equal(ss.count(km), np.where(up >= lo, up - lo + 1, 0).sum(0).astype(np.uint64))
    ms = (C.c_float * 8)()
    cnt = C.c_size_t()
    assert L.rsbwt_set_search_history_ms(ss._s, ms, 8, C.byref(cnt)) == 0 and cnt.value == 8 and min(ms) > 0
    # the gather entry of the C++ host: with one device it is a device-to-device copy of the [S][Q] block
    # (several devices: ncclSend / ncclRecv in one group -- no multi-GPU box here)
    assert L.rsbwt_rccl_available() in (0, 1)
    blk = torch.stack([d_lo, d_up]).contiguous()
    root = torch.zeros_like(blk)
    blocks = (C.c_void_p * 1)(blk.data_ptr())
    sizes = (C.c_size_t * 1)(blk.numel() * 8)
    streams = (C.c_void_p * 1)(None)
    assert L.rsbwt_set_gather_intervals_dev(ss._s, blocks, sizes, C.c_void_p(root.data_ptr()), streams) == 0
    torch.cuda.synchronize()
    assert torch.equal(root, blk)
    ss.close()
    for g in shards:
        g.close()


def test_gpu_set_open_sizes_the_tables_per_device(rsb, oracle, tmp_path):
    kw = dict(seed=14, genome_len=30000, haplotypes=4, snp_rate=0.004, read_len=60, coverage=3.0)
    import ctypes as C
    L = rsb.lib()
    paths = []
    for s in range(4):
        p = str(tmp_path / f"s{s}.bwt")
        rsb.synth_popbwt(p, None, shard=s, num_shards=4, **kw)
        paths.append(p)
    arr = (C.c_char_p * 4)(*[p.encode() for p in paths])
    h = C.c_void_p()
    assert L.rsbwt_set_open(arr, 4, None, 0, C.byref(h)) == 0
    depths = [L.rsbwt_ktab_depth(L.rsbwt_set_shard(h, i)) for i in range(4)]
    assert len(set(depths)) == 1 and depths[0] >= 6  # one depth for the device's shards
    rng = np.random.default_rng(3)
    km = _random_kmers(rng, 5000, 31)
    lo = np.empty((4, 5000), np.uint64)
    up = np.empty((4, 5000), np.uint64)
    assert L.rsbwt_set_find_intervals(h, km.ctypes.data, 5000, 31, 31, lo.ctypes.data, up.ctypes.data) == 0
    for s in range(4):
        elo, eup = oracle.load(paths[s]).find_intervals(km)
        assert np.array_equal(lo[s], elo) and np.array_equal(up[s], eup)
    L.rsbwt_set_close(h)


def test_gpu_reference_query_cpp_runs_on_the_shim(rsb, oracle, fixture_bwt, golden_dir):
    """oracle/_ref/shim_demo = the reference's unmodified src/bwt/query.cpp linked with
    `class GpuBWT : public BWT` (include/rsbwt_gpubwt.hpp), built in the build container.  Here every
    virtual call of the reference's own findInterval / query / query_exactmatch is one GPU round trip;
    the answers must be the reference's (golden) ones."""
    import subprocess
    exe = os.path.join(os.path.dirname(golden_dir), "..", "oracle", "_ref", "shim_demo")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_demo is built only where the reference tree exists")
    path, _ = fixture_bwt
    g31 = np.load(os.path.join(golden_dir, "popbwt_v1.npz"))
    gq = np.load(os.path.join(golden_dir, "query_v1.npz"))
    ws = [g31["kmers31"][i].tobytes().decode() for i in (0, 1, 5003, 10005)]
    ws += [gq["q_w45"][i].tobytes().decode() for i in (0, 1)] + [gq["em_w70"][i].tobytes().decode() for i in (0, 1, 2, 3)]
    p = subprocess.run([exe, path] + ws, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.strip().split("\n")
    assert len(lines) == len(ws)
    for j, i in enumerate((0, 1, 5003, 10005)):
        f = lines[j].split()
        assert (int(f[0]), int(f[1])) == (int(g31["lower31"][i]), int(g31["upper31"][i]))
    for j, i in enumerate((0, 1)):
        f = lines[4 + j].split()
        a, b = int(gq["q_first45"][i]), int(gq["q_first45"][i + 1])
        want = [gq["q_reads45"][r, :gq["q_len45"][r]].tobytes().decode() for r in range(a, b)]
        assert int(f[3]) == len(want) and f[4:] == want
    for j, i in enumerate((0, 1, 2, 3)):
        assert int(lines[6 + j].split()[2]) == int(gq["em_ans70"][i])


def test_gpu_set_interleaved_tables_and_detach(rsb, oracle):
    """rsbwt_set_attach_ktabs gives the shards of a device ONE interleaved k-mer table (entry of shard s
    for T-mer c at [c * S + s]); searches through the set and through each handle use it; when the set
    goes while the handles live on, they lose the table that lived in the set and still answer right."""
    import ctypes as C
    L = rsb.lib()
    shards, oixs = [], []
    for i, R in enumerate([150000, 400000, 90000]):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 700 + i) == 0
        oixs.append(oracle.from_runs(runs))
        shards.append(rsb.GpuBWT(runs=runs, ktab_depth=None))
    ss = rsb.ShardSet(shards)
    T = L.rsbwt_set_auto_ktab_depth(ss._s)
    assert 2 <= T <= 16
    base = [g.hbm_bytes() for g in shards]
    assert L.rsbwt_set_attach_ktabs(ss._s, 7) == 0
    assert [g.ktab_depth() for g in shards] == [7, 7, 7]
    assert [g.hbm_bytes() - b for g, b in zip(shards, base)] == [8 * 4 ** 7] * 3
    rng = np.random.default_rng(12)
    for k in (5, 7, 8, 31, 40):
        km = _random_kmers(rng, 6000, k)
        km[::7] = ord("C")
        lo, up = ss.find_intervals(km)
        for s, oix in enumerate(oixs):
            elo, eup = oix.find_intervals(km, nthreads=8)
            assert np.array_equal(lo[s], elo) and np.array_equal(up[s], eup), (k, s)
            hlo, hup = rsb.find_intervals(shards[s], km)  # the handle alone, through its stride-3 table
            assert np.array_equal(hlo, elo) and np.array_equal(hup, eup), (k, s)
    lo1, up1 = rsb.find_intervals_1mm(shards[1], km[:100])  # traced / resumed searches on a strided table
    ss.close()
    assert [g.ktab_depth() for g in shards] == [0, 0, 0] and [g.hbm_bytes() for g in shards] == base
    lo, up = rsb.find_intervals(shards[1], km)
    elo, eup = oixs[1].find_intervals(km, nthreads=8)
    assert np.array_equal(lo, elo) and np.array_equal(up, eup)
    lo2, up2 = rsb.find_intervals_1mm(shards[1], km[:100])
    assert np.array_equal(lo1, lo2) and np.array_equal(up1, up2)
    for g in shards:
        g.close()


def _spell_variants(km):
    """[m][3k+1][k]: every k-mer followed by its single substitutions, position by position, the
    alternatives in ACGT order (variants_kernel's order)."""
    m, k = km.shape
    out = np.repeat(km[:, None, :], 3 * k + 1, axis=1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for pos in range(k):
        for q in range(m):
            alts = [c for c in acgt if c != km[q, pos]][:3]
            out[q, 1 + 3 * pos:4 + 3 * pos, pos] = alts
    return out


@pytest.mark.gpu
def test_gpu_set_one_mismatch_over_the_shards(rsb, oracle):
    """rsbwt_set_find_intervals_1mm_dev: the [S][m][3k+1] variant intervals of a one-device set =
    the oracle's exact search of every spelled-out variant in every shard (tables of different
    depths, so the shards' traces differ in length; one shard without a table)."""
    import ctypes as C
    import torch
    L = rsb.lib()
    sizes = [300000, 40000, 900000]
    shards, oixs = [], []
    for i, R in enumerate(sizes):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 1700 + i) == 0
        oixs.append(oracle.from_runs(runs))
        shards.append(rsb.GpuBWT(runs=runs, ktab_depth=[6, 0, 9][i]))
    ss = rsb.ShardSet(shards)
    rng = np.random.default_rng(16)
    p = lambda t: C.c_void_p(t.data_ptr())
    for k, m in ((31, 300), (12, 500), (33, 100)):
        km = _random_kmers(rng, m, k)
        d_half = torch.empty((m // 2, k), dtype=torch.uint8, device="cuda:0")
        assert L.rsbwt_sample_present_kmers_dev(shards[2].handle, m // 2, k, k, 5, p(d_half), None) == 0
        torch.cuda.synchronize()
        km[: m // 2] = d_half.cpu().numpy()
        km[7, k // 2] = ord("N")
        V = 3 * k + 1
        wpq = (k + 31) // 32
        d_km = torch.from_numpy(km).cuda()
        d_pk = torch.empty((m, wpq), dtype=torch.int64, device="cuda:0")
        d_ok = torch.empty(m, dtype=torch.uint8, device="cuda:0")
        d_lo = torch.empty((len(sizes), m, V), dtype=torch.int64, device="cuda:0")
        d_up = torch.empty((len(sizes), m, V), dtype=torch.int64, device="cuda:0")
        need = L.rsbwt_set_1mm_scratch_bytes(ss._s, m, k)
        assert need >= max(L.rsbwt_1mm_scratch_bytes(g.handle, m, k) for g in shards)
        d_scr = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        assert L.rsbwt_pack_kmers_dev(p(d_km), m, k, k, p(d_pk), p(d_ok), 0, None) == 0
        assert L.rsbwt_set_find_intervals_1mm_dev(ss._s, p(d_pk), p(d_ok), m, k, p(d_lo), p(d_up), p(d_scr), None) == 0
        torch.cuda.synchronize()
        lo, up = d_lo.cpu().numpy().view(np.uint64), d_up.cpu().numpy().view(np.uint64)
        variants = _spell_variants(km)
        assert variants.shape == (m, V, k)
        for s, oix in enumerate(oixs):
            elo, eup = oix.find_intervals(variants.reshape(m * V, k), nthreads=8)
            elo, eup = elo.reshape(m, V), eup.reshape(m, V)
            elo[7], eup[7] = 1, 0  # a k-mer with a foreign symbol is invalid as a whole
            assert np.array_equal(lo[s], elo) and np.array_equal(up[s], eup), (k, s)
    ss.close()
    for g in shards:
        g.close()





@pytest.mark.gpu
@pytest.mark.parametrize("span,long_runs,R,rows", [(0, 0, 30_000_000, 300_000), (2944, 0, 6_000_000, 300_000),
                                                  (0, 1, 20_000_000, 200_000)])
def test_gpu_extraction_equals_the_mirrors_walks(rsb, span, long_runs, R, rows):
    """The wave-cooperative walks against the same walks taken step by step with the class-BWT
    mirrors (readserver_amd/selfcheck.py), on indexes and row counts large enough for rare layout
    cases: a select argument that outlasts a far line's 92 pieces must not find its symbol in the
    link that follows them (seen once in 10^5 rows on a 20 GB shard before the hit was held to the
    line's span; tools/check_extract_at_scale.py runs the same check there)."""
    import ctypes as C
    import torch
    from readserver_amd import selfcheck
    L = rsb.lib()
    d_runs = torch.empty(R, dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, ((1 << 63) if long_runs else 0) | 77, 0, None) == 0
    torch.cuda.synchronize()
    with rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), ktab_depth=None, window_span=span) as g:
        del d_runs
        if span:
            assert g.far_lines() > 0
        r = np.random.default_rng(3).integers(0, g.getBWLen(), rows, dtype=np.uint64)
        res = selfcheck.extraction_vs_mirrors(g, r, stride=1024)
        assert res["rows_differing"] == 0, res


@pytest.mark.gpu
@pytest.mark.parametrize("k,T,span,grouped", [(31, None, 0, False), (31, 8, 2944, False), (40, 10, 0, False), (12, 12, 300, False),
                                              (31, 10, 0, True), (31, 12, 0, True), (40, 9, 0, True), (12, 12, 300, True)])
def test_gpu_one_lane_per_search_on_a_batch_that_fills_the_launch(rsb, oracle, k, T, span, grouped):
    """A single shard and a batch of >= 262,144 k-mers: launch_search takes the one-lane-per-search
    kernel (search_solo.h; smaller batches and shard sets stay on lane pairs).  Intervals, counts and
    {lower, upper} pairs against the oracle; the work counters of a counting launch add up.  grouped: the k-mer table's
    12-byte records (the kernel that makes its own start records reads them: FUSED; at T = 12 most 12-mers do not
    occur and are searched from initInterval)."""
    import ctypes as C
    import torch
    L = rsb.lib()
    R, Q = 2_500_000, 300_000
    runs = np.empty(R, np.uint8)
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 4100 + k) == 0
    oix = oracle.from_runs(runs)
    rng = np.random.default_rng(k)
    p = lambda t: C.c_void_p(t.data_ptr())
    with rsb.GpuBWT(runs=runs, ktab_depth=(0 if T is None else T), window_span=span, ktab_grouped=grouped) as g:
        assert g.ktab_info()[0] == (1 if grouped else 0)
        km = _random_kmers(rng, Q, k)
        d_half = torch.empty((Q // 2, k), dtype=torch.uint8, device="cuda:0")
        assert L.rsbwt_sample_present_kmers_dev(g.handle, Q // 2, k, k, 5, p(d_half), None) == 0
        torch.cuda.synchronize()
        km[::2] = d_half.cpu().numpy()
        km[11, 0] = ord("N")
        km[12] = ord("A")
        km[13] = ord("T")
        elo, eup, st = oix.find_intervals(km, nthreads=8, want_steps=True)
        wpq = (k + 31) // 32
        d_km = torch.from_numpy(km).cuda()
        d_pk = torch.empty((Q, wpq), dtype=torch.int64, device="cuda:0")
        d_ok = torch.empty(Q, dtype=torch.uint8, device="cuda:0")
        d_lo = torch.empty(Q, dtype=torch.int64, device="cuda:0")
        d_up = torch.empty(Q, dtype=torch.int64, device="cuda:0")
        d_pr = torch.empty((Q, 2), dtype=torch.int64, device="cuda:0")
        assert L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, None) == 0
        assert L.rsbwt_set_counting(g.handle, 1) == 0
        assert L.rsbwt_find_intervals_dev(g.handle, p(d_pk), p(d_ok), Q, k, p(d_lo), p(d_up), None) == 0
        torch.cuda.synchronize()
        w = (C.c_uint64 * 16)()
        assert L.rsbwt_last_search_counters(g.handle, w) == 0
        assert L.rsbwt_set_counting(g.handle, 0) == 0
        lo, up = d_lo.cpu().numpy().view(np.uint64), d_up.cpu().numpy().view(np.uint64)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)
        Td = g.ktab_depth()
        steps = int(np.maximum(st.astype(np.int64) - (max(Td, 1) - 1 if k >= Td else 0), 0).sum())
        if grouped:  # a T-mer its record leaves to the search takes the steps the table would have saved
            assert steps <= w[0] <= int(st.astype(np.int64).sum()) and w[2] <= w[1] <= 2 * w[0]
            if T == 12 and k > T:
                assert w[0] > steps  # (most 12-mers of 1.4e7 symbols do not occur)
        else:
            assert w[0] == steps and w[2] <= w[1] <= 2 * w[0]
        assert w[12] == 1  # one lane per search
        assert L.rsbwt_find_interval_pairs_dev(g.handle, p(d_pk), p(d_ok), Q, k, p(d_pr), None) == 0
        assert L.rsbwt_count_dev(g.handle, p(d_pk), p(d_ok), Q, k, p(d_lo), None) == 0
        torch.cuda.synchronize()
        pr = d_pr.cpu().numpy().view(np.uint64)
        assert np.array_equal(pr[:, 0], elo) and np.array_equal(pr[:, 1], eup)
        assert np.array_equal(d_lo.cpu().numpy().view(np.uint64), np.where(eup >= elo, eup - elo + 1, 0).astype(np.uint64))
        # the host entry point of the same batch (slices of 64K k-mers: lane pairs) agrees
        hlo, hup = rsb.find_intervals(g, km[:70000])
        assert np.array_equal(hlo, elo[:70000]) and np.array_equal(hup, eup[:70000])


@pytest.mark.gpu
def test_gpu_hit_list_that_outgrows_its_first_buffer(rsb, tmp_path):
    """rsbwt_hits_1mm takes the variants that occur straight out of the search kernel as a list
    (search_extra::d_hit_ctl).  Short k-mers on a small index: nearly every variant occurs, so a slice
    leaves more hits than the room first given to it and is searched once more with enough; the list
    must equal the non-empty cells of the dense matrices, in (k-mer, position, base) order."""
    bwt, rd = str(tmp_path / "s.bwt"), str(tmp_path / "s.reads")
    rsb.synth_popbwt(bwt, rd, seed=5, genome_len=40000, haplotypes=3, snp_rate=0.01, read_len=60, coverage=5.0)
    rng = np.random.default_rng(8)
    k = 5
    kmers = ["".join("ACGT"[i] for i in rng.integers(0, 4, k)) for _ in range(3000)] + ["ANCGT"]
    with rsb.GpuBWT(bwt) as g:
        lo, up = rsb.find_intervals_1mm(g, kmers)
        want = []
        for qi, w in enumerate(kmers):
            want += [(qi,) + h for h in rsb.hits_1mm(w, lo[qi], up[qi])]
        assert len(want) > 2 * 16384  # room first given to the slice: max(4 m, 16384) records
        hl = rsb.hits_1mm_batch(g, kmers)
        got = [(int(r["query"]), int(r["pos"]), r["base"].decode(), int(r["lower"]), int(r["upper"])) for r in hl]
        assert got == want
        # the device-resident form: the same records ordered by index = q * (3k+1) + v; a list that is too
        # short is filled to its capacity and the total still says how many there are
        import ctypes as C
        import torch
        L = rsb.lib()
        km = np.frombuffer("".join(kmers).encode(), np.uint8).reshape(len(kmers), k)
        m, V = len(kmers), 3 * k + 1
        p = lambda t: C.c_void_p(t.data_ptr())
        d_km = torch.from_numpy(km.copy()).cuda()
        d_pk = torch.empty(m, dtype=torch.int64, device="cuda:0")
        d_ok = torch.empty(m, dtype=torch.uint8, device="cuda:0")
        assert L.rsbwt_pack_kmers_dev(p(d_km), m, k, k, p(d_pk), p(d_ok), 0, None) == 0
        d_scr = torch.empty(L.rsbwt_hits_1mm_scratch_bytes(g.handle, m, k), dtype=torch.uint8, device="cuda:0")
        dense = [(qi * V + v, int(lo[qi, v]), int(up[qi, v])) for qi in range(m) for v in range(V) if lo[qi, v] <= up[qi, v]]
        for room in (len(dense) + 100, 1000, 0):
            d_hits = torch.full((max(room, 1), 4), -1, dtype=torch.int64, device="cuda:0")
            d_tot = torch.zeros(1, dtype=torch.int64, device="cuda:0")
            assert L.rsbwt_hits_1mm_dev(g.handle, p(d_pk), p(d_ok), m, k, p(d_hits), room, p(d_tot), p(d_scr), None) == 0
            torch.cuda.synchronize()
            assert int(d_tot.item()) == len(dense)
            n = min(room, len(dense))
            rec = d_hits.cpu().numpy().view(np.uint64)
            assert [(int(r[2]), int(r[0]), int(r[1])) for r in rec[:n]] == dense[:n] and (rec[:n, 3] == 0).all()
            assert (rec[n:] == np.uint64(2**64 - 1)).all()  # nothing past the capacity


@pytest.mark.gpu
def test_gpu_interval_pairs_for_the_wire(rsb, oracle):
    """rsbwt_pack / unpack_interval_pairs_dev: {lower:40, width:40} records carry every interval exactly --
    real search output (hits, empty intervals, an invalid k-mer's (1, 0)), the reference's (0, 2^64-1)
    corner and the 2^40-1 extremes -- byte for byte what the torch reference in sharded.py packs, for
    every tail length; a pair that does not fit is counted."""
    import ctypes as C
    import torch
    from readserver_amd import sharded
    L = rsb.lib()
    R = 400000
    runs = np.empty(R, np.uint8)
    assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 606) == 0
    rng = np.random.default_rng(9)
    with rsb.GpuBWT(runs=runs) as g:
        km = _random_kmers(rng, 50001, 12)  # 4^12 = 1.7e7 against 4e6 symbols: hits and misses
        km[3, 5] = ord("N")
        lo, up = rsb.find_intervals(g, km)
        assert (up >= lo).any() and (up < lo).any()
    pairs = np.stack([lo, up], 1)
    pairs[7] = (0, 2**64 - 1)
    pairs[8] = (2**40 - 1, 2**40 - 2)
    pairs[9] = (1, 2**40 - 1)
    t = torch.from_numpy(pairs.view(np.int64))
    p = lambda x: C.c_void_p(x.data_ptr())
    for n in (1, 2, 3, 4, 5, 7, 1003, 50001):
        sub = t[:n].contiguous()
        want = sharded.pack_pairs(sub)
        assert want.numel() == L.rsbwt_packed_pairs_bytes(n) == sharded.packed_pairs_bytes(n)
        d_pairs = sub.cuda()
        d_pk = torch.full((want.numel() + 16,), 0xAB, dtype=torch.uint8, device="cuda:0")
        d_bad = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        assert L.rsbwt_pack_interval_pairs_dev(p(d_pairs), n, p(d_pk), p(d_bad), 0, None) == 0
        torch.cuda.synchronize()
        got = d_pk.cpu()
        assert torch.equal(got[: 10 * n], want[: 10 * n]) and int(d_bad.item()) == 0
        assert (got[want.numel():] == 0xAB).all()  # nothing past the buffer's size
        d_back = torch.full((n + 2, 2), -7, dtype=torch.int64, device="cuda:0")
        assert L.rsbwt_unpack_interval_pairs_dev(p(d_pk), n, p(d_back), 0, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(d_back[:n].cpu(), sub) and (d_back[n:] == -7).all()
        assert torch.equal(sharded.unpack_pairs(want, n), sub)
        # the wrappers on GPU tensors go through the same kernels
        assert torch.equal(sharded.unpack_pairs(sharded.pack_pairs(d_pairs), n).cpu(), sub)
    bad = torch.tensor([[1 << 40, (1 << 40) + 5], [5, 4], [0, 1 << 41]], dtype=torch.int64, device="cuda:0")
    d_pk = torch.zeros(32, dtype=torch.uint8, device="cuda:0")
    d_bad = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    assert L.rsbwt_pack_interval_pairs_dev(p(bad), 3, p(d_pk), p(d_bad), 0, None) == 0
    torch.cuda.synchronize()
    assert int(d_bad.item()) == 1  # one thread saw pairs that do not fit
