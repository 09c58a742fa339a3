"""GPU suite (-m gpu): BASELINE configs[3] / configs[4] over a shard set -- 1-mismatch hit lists, extraction
of (shard, row) pairs and locate + extract (query) in every shard, the per-shard results laid side by side the
way the reference's front-end concatenates its partitions' replies (src/service/server.cpp:199-261).  One
device here; the same entry points drive the devices of a set concurrently (csrc/sets.hip)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def four_shards(rsb, oracle, tmp_path_factory):
    d = tmp_path_factory.mktemp("set")
    kw = dict(seed=31, genome_len=25000, haplotypes=6, snp_rate=0.004, read_len=70, coverage=3.0)
    shards, oixs = [], []
    for s in range(4):
        p = str(d / f"s{s}.bwt")
        rsb.synth_popbwt(p, None, shard=s, num_shards=4, **kw)
        shards.append(rsb.GpuBWT(p, ktab_depth=(8 if s % 2 else 6)))
        oixs.append(oracle.load(p))
    rd = str(d / "whole.reads")
    rsb.synth_popbwt(str(d / "whole.bwt"), rd, **kw)
    reads = open(rd).read().split()
    ss = rsb.ShardSet(shards)
    yield ss, shards, oixs, reads
    ss.close()
    for g in shards:
        g.close()


def _kmers_from(reads, rng, n, k, mutate=0.0):
    out = []
    for _ in range(n):
        r = reads[rng.integers(len(reads))]
        s = rng.integers(0, len(r) - k + 1)
        w = list(r[s:s + k])
        if rng.random() < mutate:
            p = rng.integers(k)
            w[p] = "ACGT"[("ACGT".index(w[p]) + 1 + rng.integers(3)) % 4]
        out.append("".join(w))
    return out


@pytest.mark.parametrize("k", [12, 31, 40])
def test_gpu_set_hit_lists_are_the_shards_lists_side_by_side(rsb, four_shards, k):
    ss, shards, oixs, reads = four_shards
    rng = np.random.default_rng(k)
    kmers = _kmers_from(reads, rng, 257, k, mutate=0.5) + ["N" * k]
    hits, first = ss.hits_1mm(kmers)
    assert first[0] == 0 and first[-1] == len(hits) and (np.diff(first.astype(np.int64)) >= 0).all()
    for s, (g, oix) in enumerate(zip(shards, oixs)):
        mine = hits[int(first[s]):int(first[s + 1])]
        assert np.array_equal(mine, rsb.hits_1mm_batch(g, kmers))  # the shard's own list
        # ... which is the oracle's exact search of every variant, for a sample of the k-mers
        for qi in range(0, len(kmers) - 1, 16):
            w = kmers[qi]
            want = []
            lo, up = oix.find_interval(w)
            if up >= lo:
                want.append((-1, b"", lo, up))
            for pos in range(k):
                for alt in [c for c in "ACGT" if c != w[pos]]:
                    lo, up = oix.find_interval(w[:pos] + alt + w[pos + 1:])
                    if up >= lo:
                        want.append((pos, alt.encode(), lo, up))
            got = [(int(h["pos"]), bytes(h["base"]), int(h["lower"]), int(h["upper"])) for h in mine[mine["query"] == qi]]
            assert got == want, (s, qi)
    assert len(hits) > 4 * 100
    # a buffer that is too short: nothing written, the sizes reported
    L = rsb.lib()
    a = np.frombuffer("".join(kmers).encode(), np.uint8)
    n = C.c_size_t()
    f2 = np.zeros(5, np.uint64)
    assert L.rsbwt_set_hits_1mm(ss._s, a.ctypes.data, len(kmers), k, k, None, 0, f2.ctypes.data, C.byref(n)) == -7
    assert n.value == len(hits) and np.array_equal(f2, first)


def test_gpu_set_extract_rows_of_several_shards(rsb, four_shards):
    ss, shards, oixs, reads = four_shards
    rng = np.random.default_rng(3)
    sh = rng.integers(0, 4, 3001).astype(np.uint32)
    rows = np.array([rng.integers(0, oixs[s].bwlen()) for s in sh], dtype=np.uint64)
    got, pl = ss.extract(sh, rows, stride=96)
    for i in range(0, len(rows), 7):
        pre, post = oixs[sh[i]].extract(int(rows[i]))
        assert got[i] == pre + post and pl[i] == len(pre)
    assert all(r in set(reads) for r in got[:200])
    with pytest.raises(rsb.RsbwtError):
        ss.extract([9], [0])


def test_gpu_set_query_concatenates_the_partitions_reads(rsb, four_shards):
    """find_reads of a short query in every partition (service.cpp:714-743), lists concatenated per query
    (server.cpp:199-261): every read of the unsharded collection that contains the k-mer, exactly once."""
    ss, shards, oixs, reads = four_shards
    rng = np.random.default_rng(4)
    kmers = _kmers_from(reads, rng, 150, 25) + ["ACGTACGTACGTACGTACGTACGTA", "N" * 25]
    got = ss.query(kmers, read_stride=96)
    for q, w in enumerate(kmers):
        exp = []
        for s, oix in enumerate(oixs):
            lo, up = oix.find_interval(w) if "N" not in w else (1, 0)
            for r in range(lo, up + 1):
                pre, post = oix.extract(r)
                exp.append((s, pre + post))
        assert got[q] == exp, q
        assert sorted(r for _, r in got[q]) == sorted(r for r in reads if w in r)
    assert sum(len(x) for x in got) > 300


def test_gpu_set_device_resident_forms(rsb, four_shards):
    """rsbwt_set_hits_1mm_dev / rsbwt_set_extract_dev (what bench.py --mode 1mm|extract times) against the host forms."""
    import torch
    ss, shards, oixs, reads = four_shards
    L = rsb.lib()
    dev = torch.device("cuda", 0)
    p = lambda t: C.c_void_p(t.data_ptr())
    rng = np.random.default_rng(5)
    k, m, S = 31, 500, 4
    kmers = _kmers_from(reads, rng, m, k, mutate=0.5)
    d_km = torch.from_numpy(np.frombuffer("".join(kmers).encode(), np.uint8).reshape(m, k).copy()).to(dev)
    d_pk = torch.empty(m, dtype=torch.int64, device=dev)
    d_ok = torch.empty(m, dtype=torch.uint8, device=dev)
    assert L.rsbwt_pack_kmers_dev(p(d_km), m, k, k, p(d_pk), p(d_ok), 0, None) == 0
    cap = 8 * m
    d_hits = torch.zeros((S, cap, 4), dtype=torch.int64, device=dev)
    d_tot = torch.zeros(S, dtype=torch.int64, device=dev)
    d_scr = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(ss._s, m, k), dtype=torch.uint8, device=dev)
    assert L.rsbwt_set_hits_1mm_is_fused(ss._s, m, k) == 0  # tables of depth 6 and 8: the shards side by side
    assert L.rsbwt_set_hits_1mm_dev(ss._s, p(d_pk), p(d_ok), m, k, p(d_hits), cap, p(d_tot), p(d_scr), None) == 0
    torch.cuda.synchronize()
    hits, first = ss.hits_1mm(kmers)
    V = 3 * k + 1
    for s in range(S):
        n = int(d_tot[s].item())
        mine = hits[int(first[s]):int(first[s + 1])]
        assert n == len(mine)
        rec = d_hits[s, :n].cpu().numpy().view(np.uint64)
        assert np.array_equal(rec[:, 0], mine["lower"]) and np.array_equal(rec[:, 1], mine["upper"])
        assert np.array_equal(rec[:, 2] // V, mine["query"].astype(np.uint64))
    n = 700
    rows = np.stack([rng.integers(0, oixs[s].bwlen(), n) for s in range(S)]).astype(np.int64)
    d_rows = torch.from_numpy(rows).to(dev)
    d_out = torch.zeros((S, n, 96), dtype=torch.uint8, device=dev)
    d_len = torch.zeros((S, n), dtype=torch.int32, device=dev)
    d_pl = torch.zeros((S, n), dtype=torch.int32, device=dev)
    assert L.rsbwt_set_extract_dev(ss._s, p(d_rows), n, p(d_out), 96, p(d_len), p(d_pl), None) == 0
    torch.cuda.synchronize()
    out, ln, pl = d_out.cpu().numpy(), d_len.cpu().numpy(), d_pl.cpu().numpy()
    want, wpl = ss.extract(np.repeat(np.arange(S), n), rows.reshape(-1), stride=96)
    for s in range(S):
        for i in range(0, n, 5):
            assert out[s, i, :ln[s, i]].tobytes().decode() == want[s * n + i] and pl[s, i] == wpl[s * n + i]


def test_gpu_search_from_prepared_start_records(rsb, four_shards):
    """rsbwt_set_prepare_dev + rsbwt_set_find_interval_pairs_prepared_dev (the start records of a batch computed
    ahead, on another stream) give the pairs of rsbwt_set_find_interval_pairs_dev -- and the oracle's."""
    import torch
    ss, shards, oixs, reads = four_shards
    L = rsb.lib()
    dev = torch.device("cuda", 0)
    p = lambda t: C.c_void_p(t.data_ptr())
    rng = np.random.default_rng(8)
    k, Q, S = 31, 5000, 4
    kmers = _kmers_from(reads, rng, Q - 2, k, mutate=0.3) + ["N" * k, "A" * k]
    d_km = torch.from_numpy(np.frombuffer("".join(kmers).encode(), np.uint8).reshape(Q, k).copy()).to(dev)
    d_pk = torch.empty(Q, dtype=torch.int64, device=dev)
    d_ok = torch.empty(Q, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(side.cuda_stream)
    d_rec = torch.empty(L.rsbwt_set_records_bytes(ss._s, Q), dtype=torch.uint8, device=dev)
    assert d_rec.numel() == S * Q * 16
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        assert L.rsbwt_pack_kmers_dev(p(d_km), Q, k, k, p(d_pk), p(d_ok), 0, sp) == 0
        assert L.rsbwt_set_prepare_dev(ss._s, p(d_pk), p(d_ok), Q, k, p(d_rec), sp) == 0
        done.record(side)
    torch.cuda.current_stream().wait_event(done)
    a = torch.zeros((S, Q, 2), dtype=torch.int64, device=dev)
    b = torch.zeros((S, Q, 2), dtype=torch.int64, device=dev)
    assert L.rsbwt_set_find_interval_pairs_prepared_dev(ss._s, p(d_pk), p(d_ok), p(d_rec), Q, k, p(a), None) == 0
    assert L.rsbwt_set_find_interval_pairs_dev(ss._s, p(d_pk), p(d_ok), Q, k, p(b), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    km = np.frombuffer("".join(kmers).encode(), np.uint8).reshape(Q, k)
    for s in range(S):
        elo, eup = oixs[s].find_intervals(km)
        got = a[s].cpu().numpy().view(np.uint64)
        assert np.array_equal(got[:, 0], elo) and np.array_equal(got[:, 1], eup)


def test_gpu_select_across_stretches_without_the_symbol(rsb, oracle):
    """The select samples name a window exactly while a block of 256 occurrences spreads over at most five
    windows; a stretch of the BWT (nearly) without a symbol makes blocks of that symbol span hundreds of windows,
    where the sample gives a bracket and the walk bisects the window headers.  getOccAt at every occurrence of
    the rare symbol and batched extraction against the oracle."""
    rng = np.random.default_rng(77)
    parts = []
    for blk in range(12):  # alternating: every symbol / no T at all / T once in ~3000 symbols
        R = 40000
        if blk % 3 == 0:
            sym = rng.integers(0, 5, R)
        elif blk % 3 == 1:
            sym = rng.integers(0, 4, R)
        else:
            sym = np.where(rng.random(R) < 0.002, 4, rng.integers(1, 4, R))
        parts.append(((sym.astype(np.uint8)) << 5) | rng.integers(1, 9, R).astype(np.uint8))
    runs = np.concatenate(parts)
    oix = oracle.from_runs(runs)
    n = oix.bwlen()
    for span in (0, 64):
        with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=None) as g:
            tot = g.getOcc("T", n - 1)
            bc = np.arange(1, tot + 1, 3, dtype=np.uint64)
            idx = g.occ_at_batch("T", bc)
            assert (g.char_batch(idx) == ord("T")).all() and np.array_equal(g.occ_batch("T", idx), bc)
            for j in range(0, bc.size, 997):
                assert int(idx[j]) == oix.occ_at("T", int(bc[j]))
            rows = rng.integers(0, n, 20000).astype(np.uint64)
            out = np.zeros((rows.size, 512), np.uint8)
            ln, pl = np.empty(rows.size, np.uint32), np.empty(rows.size, np.uint32)
            assert rsb.lib().rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, 512, ln.ctypes.data,
                                           pl.ctypes.data) == 0
            checked = 0
            for i in range(0, rows.size, 13):
                if ln[i] == 0xFFFFFFFF:
                    continue
                pre, post = oix.extract(int(rows[i]), cap=4096)
                assert out[i, :ln[i]].tobytes().decode() == pre + post and pl[i] == len(pre), (span, i)
                checked += 1
            assert checked > 500


@pytest.fixture
def two_devices(rsb, monkeypatch):
    """Two device numbers a set will treat as two devices: GPUs 0 and 1 where the box has them, else the library's
    test hook (include/rsbwt.h, rsbwt_logical_device: RSBWT_TEST_DEVICE_ALIASES=2 makes 0 and 1 two LOGICAL devices on
    GPU 0), so that the several-device host code of csrc/sets.hip -- a thread, a context pool and a fused launch per
    device group, the merge of the groups' results, the gather onto the first group's device -- runs under -m gpu on
    the one-GPU boxes this repo is tested on.  What the alias cannot reach is RCCL itself (two ranks on one GPU are
    refused): the set then sums on the host and gathers with peer copies, as it does on a box without librccl."""
    L = rsb.lib()
    aliased = L.rsbwt_device_count() < 2
    if aliased:
        monkeypatch.setenv("RSBWT_ENABLE_TEST_HOOKS", "1")
        monkeypatch.setenv("RSBWT_TEST_DEVICE_ALIASES", "2")
    return aliased


def test_gpu_set_split_over_two_devices_equals_one_device(rsb, oracle, tmp_path, two_devices):
    """The cross-device paths of csrc/sets.hip: the same four shards as a set on ONE device and as a
    set split over devices 0 and 1 must give the same intervals, the same summed counts (ncclReduce over the
    per-device communicators where there are two GPUs and librccl; the host-side sum otherwise), the same gathered
    blocks (ncclSend / ncclRecv onto the first device; peer copies otherwise), hit lists, reads and query lists.  On a
    one-GPU box the two devices are two logical devices on GPU 0 (`two_devices`): every line of the several-group host
    code runs, RCCL does not."""
    import torch
    L = rsb.lib()
    aliased = two_devices
    kw = dict(seed=41, genome_len=20000, haplotypes=4, snp_rate=0.004, read_len=60, coverage=3.0)
    paths = []
    for s in range(4):
        p_ = str(tmp_path / f"s{s}.bwt")
        rsb.synth_popbwt(p_, None, shard=s, num_shards=4, **kw)
        paths.append(p_)
    rd = str(tmp_path / "w.reads")
    rsb.synth_popbwt(str(tmp_path / "w.bwt"), rd, **kw)
    reads = open(rd).read().split()
    rng = np.random.default_rng(9)
    kmers = _kmers_from(reads, rng, 3000, 31, mutate=0.3)
    one = [rsb.GpuBWT(p_, device=0) for p_ in paths]
    two = [rsb.GpuBWT(p_, device=(0 if s < 2 else 1)) for s, p_ in enumerate(paths)]
    assert [L.rsbwt_logical_device(g.handle) for g in two] == [0, 0, 1, 1]
    assert [L.rsbwt_device(g.handle) for g in two] == ([0, 0, 0, 0] if aliased else [0, 0, 1, 1])
    s1, s2 = rsb.ShardSet(one), rsb.ShardSet(two)
    assert L.rsbwt_set_devices(s1._s) == 1 and L.rsbwt_set_devices(s2._s) == 2
    # a set whose groups interleave in shard order (0, 1, 0, 1): the per-group blocks are scattered back by shard
    mixed = [rsb.GpuBWT(p_, device=s % 2) for s, p_ in enumerate(paths)]
    s3 = rsb.ShardSet(mixed)
    assert L.rsbwt_set_devices(s3._s) == 2
    lo1, up1 = s1.find_intervals(kmers)
    lo2, up2 = s2.find_intervals(kmers)
    assert np.array_equal(lo1, lo2) and np.array_equal(up1, up2)
    lo3, up3 = s3.find_intervals(kmers)
    assert np.array_equal(lo1, lo3) and np.array_equal(up1, up3)
    # shard 0 against the oracle, so that "equal" is not "equally wrong"
    elo, eup = oracle.load(paths[0]).find_intervals(np.frombuffer("".join(kmers).encode(), np.uint8).reshape(len(kmers), 31))
    assert np.array_equal(lo2[0], elo) and np.array_equal(up2[0], eup)
    c1 = s1.count(kmers)
    assert np.array_equal(c1, s2.count(kmers))  # the RCCL reduce where there are two GPUs and librccl, else the host's sum
    assert np.array_equal(c1, s3.count(kmers))
    assert np.array_equal(c1, np.where(up1 >= lo1, up1 - lo1 + 1, 0).sum(axis=0).astype(np.uint64))
    h1, f1 = s1.hits_1mm(kmers[:300])
    for sx in (s2, s3):
        h2, f2 = sx.hits_1mm(kmers[:300])
        assert np.array_equal(h1, h2) and np.array_equal(f1, f2)
    q1 = s1.query(kmers[:200], read_stride=96)
    assert q1 == s2.query(kmers[:200], read_stride=96) and q1 == s3.query(kmers[:200], read_stride=96)
    sh = rng.integers(0, 4, 2000).astype(np.uint32)
    rows = np.array([rng.integers(0, one[s].getBWLen()) for s in sh], dtype=np.uint64)
    e1 = s1.extract(sh, rows, stride=96)[0]
    assert e1 == s2.extract(sh, rows, stride=96)[0] and e1 == s3.extract(sh, rows, stride=96)[0]
    # concurrent callers of the spanning set: its per-group threads and context pools under load
    import threading
    errs = []
    def caller(i):
        try:
            for _ in range(3):
                lo_, up_ = s2.find_intervals(kmers[i::4])
                assert np.array_equal(lo_, lo1[:, i::4]) and np.array_equal(up_, up1[:, i::4])
                assert np.array_equal(s3.count(kmers[i::4]), c1[i::4])
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))
    th = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    if aliased or L.rsbwt_rccl_available():
        # per-device interval blocks gathered onto device 0: [2][Q] x {lower, upper} from each device
        Q = len(kmers)
        blocks, streams, nbytes = [], [], []
        a = np.frombuffer("".join(kmers).encode(), np.uint8).reshape(Q, 31)
        for dv, ss in ((0, [0, 1]), (0 if aliased else 1, [2, 3])):
            torch.cuda.set_device(dv)
            pr = torch.empty((2, Q, 2), dtype=torch.int64, device=f"cuda:{dv}")
            pr[..., 0] = torch.from_numpy(lo2[ss].astype(np.int64)).to(pr.device)
            pr[..., 1] = torch.from_numpy(up2[ss].astype(np.int64)).to(pr.device)
            blocks.append(pr)
            nbytes.append(pr.numel() * 8)
            streams.append(torch.cuda.Stream(device=dv))
        torch.cuda.set_device(0)
        root = torch.zeros((4, Q, 2), dtype=torch.int64, device="cuda:0")
        streams_keep = streams  # (streams of their own: what was filled on the devices' current streams is waited for)
        for st_, b_ in zip(streams, blocks):
            st_.wait_stream(torch.cuda.current_stream(b_.device))
        streams = [st_.cuda_stream for st_ in streams_keep]
        bp = (C.c_void_p * 2)(*[b.data_ptr() for b in blocks])
        nb = (C.c_size_t * 2)(*nbytes)
        st = (C.c_void_p * 2)(*streams)
        assert L.rsbwt_set_gather_intervals_dev(s2._s, bp, nb, C.c_void_p(root.data_ptr()), st) == 0, L.rsbwt_last_error()
        for dv in ((0,) if aliased else (0, 1)):
            torch.cuda.synchronize(dv)
        got = root.cpu().numpy().view(np.uint64)
        assert np.array_equal(got[..., 0], lo2) and np.array_equal(got[..., 1], up2)
    s1.close(); s2.close(); s3.close()
    for g in one + two + mixed:
        g.close()


@pytest.mark.parametrize("style,span", [("pop", 0), ("pop", 300), ("mixed", 0), ("mixed", 1024), ("dense", 90), ("long", 0)])
def test_gpu_psi_hints_change_nothing_but_the_requests(rsb, oracle, style, span):
    """Window lines with room for it carry a psi hint (where psi takes the rows of the window: csrc/line_format.h),
    written into the resident index by rsbwt_prepare_extraction (the owner's open-time step for a plain-layout shard
    that will serve reads; a first extraction WITHOUT it builds the samples into a side table and leaves the lines
    alone: no hints).  It must (1) leave every other answer
    as it was -- Occ at every position, getChar, getOccAt, findInterval with and without the table, the 1-mismatch
    matrices: the hint sits in piece bytes and behind a header bit no reader may take for data; (2) give the same
    reads as the walk without hints (RSBWT_NO_PSI_HINTS) and as the oracle."""
    import os
    L = rsb.lib()
    rng = np.random.default_rng(len(style) * 31 + span)
    R = 250000
    if style in ("pop", "mixed"):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 4242 | ((1 << 62) if style == "pop" else 0)) == 0
    elif style == "dense":
        runs = (rng.integers(0, 5, R).astype(np.uint8) << 5) | 1
    else:
        runs = (np.where(rng.random(R) < 0.02, 0, rng.integers(1, 5, R)).astype(np.uint8) << 5) | 31
    oix = oracle.from_runs(runs)
    n = oix.bwlen()
    rows = rng.integers(0, n, 30000).astype(np.uint64)
    km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (20000, 14))]
    elo, eup = oix.find_intervals(km)

    def extract(g):
        out = np.zeros((rows.size, 1024), np.uint8)
        ln, pl = np.empty(rows.size, np.uint32), np.empty(rows.size, np.uint32)
        assert L.rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, 1024, ln.ctypes.data, pl.ctypes.data) == 0
        return out, ln, pl

    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=6) as plain:
        o0, l0, p0 = extract(plain)  # (samples into a side table; the lines stay as the open left them)
        assert L.rsbwt_psi_hint_lines(plain.handle) == 0
    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=6) as g:
        before = rsb.find_intervals(g, km)
        assert L.rsbwt_prepare_extraction(g.handle) == 0  # builds the samples and writes the hints
        assert L.rsbwt_prepare_extraction(g.handle) == 0  # (again: nothing to do)
        o1, l1, p1 = extract(g)
        hints = L.rsbwt_psi_hint_lines(g.handle)
        S = g.window_span()
        if S <= 1024 and style != "dense":
            assert hints > 0, (hints, g.num_lines())
            if span == 0:  # the span the builder chooses leaves ~88 pieces per window: about half the lines have room
                assert hints > 0.2 * (g.num_lines() * 16 // 17), (hints, g.num_lines())
        assert np.array_equal(l0, l1) and np.array_equal(p0, p1)
        fit = l1 != 0xFFFFFFFF
        for i in np.nonzero(fit)[0][::1]:
            assert np.array_equal(o0[i, :l0[i]], o1[i, :l1[i]]), i
        for i in np.nonzero(fit)[0][::29]:
            pre, post = oix.extract(int(rows[i]), cap=4096)
            assert o1[i, :l1[i]].tobytes().decode() == pre + post and p1[i] == len(pre)
        # every other reader, after the hints are in the lines
        after = rsb.find_intervals(g, km)
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
        assert np.array_equal(after[0], elo) and np.array_equal(after[1], eup)
        pos = np.arange(0, n, 1, dtype=np.uint64) if n < 3000000 else np.arange(0, n, 3, dtype=np.uint64)
        tot = np.zeros(pos.size, np.uint64)
        for ch in "$ACGT":
            tot += g.occ_batch(ch, pos)
        assert np.array_equal(tot, pos + 1)
        for ch in "ACGT":
            t = oix.occ(ch, n - 1)
            if t:
                bc = rng.integers(1, t + 1, 3000).astype(np.uint64)
                idx = g.occ_at_batch(ch, bc)
                assert (g.char_batch(idx) == ord(ch)).all() and np.array_equal(g.occ_batch(ch, idx), bc)
                assert all(int(idx[j]) == oix.occ_at(ch, int(bc[j])) for j in range(0, 3000, 101))
        samp = pos[::max(1, pos.size // 20000)]
        assert all(int(o) == oix.occ("G", int(p_)) for o, p_ in zip(g.occ_batch("G", samp)[::40], samp[::40]))
        lo1, up1 = rsb.find_intervals_1mm(g, km[:200])
        assert np.array_equal(lo1[:, 0], elo[:200]) and np.array_equal(up1[:, 0], eup[:200])
    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=None) as g2:  # no table: every step through the lines
        extract(g2)
        lo, up = rsb.find_intervals(g2, km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)


@pytest.mark.parametrize("style,span", [("pop", 0), ("pop", 300), ("pop", 40), ("mixed", 0), ("mixed", 1500), ("dense", 90),
                                        ("dense", 70), ("long", 0), ("desert", 0)])
def test_gpu_shard_opened_for_reads(rsb, oracle, style, span):
    """RSBWT_OPEN_READS: every window line is laid out with room for a psi hint (88 own pieces; 84 in a window line
    that ends in a far link), the hints and a sparse select-sample table are built as part of the open.  Held to:
    (1) a hint in every window whose rows have a psi; (2) the reads of a plain handle of the same runs and of the oracle,
    rows of every kind (chunk and far windows, rows a hint only bounds, rows past its reach: the desert stream);
    (3) every other reader exact on this layout -- Occ at every position, getChar, getOccAt through the sparse
    samples, findInterval with and without a table, the 1-mismatch matrices."""
    L = rsb.lib()
    rng = np.random.default_rng(len(style) * 131 + span)
    R = 250000
    if style in ("pop", "mixed"):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 4243 | ((1 << 62) if style == "pop" else 0)) == 0
    elif style == "dense":
        runs = (rng.integers(0, 5, R).astype(np.uint8) << 5) | 1
    elif style == "desert":  # stretches without one symbol: the rows of a window spread over many windows
        sym = np.where((np.arange(R) // 9000) % 2 == 0, rng.integers(0, 5, R), rng.integers(0, 4, R))
        runs = (sym.astype(np.uint8) << 5) | rng.integers(1, 9, R).astype(np.uint8)
    else:
        runs = (np.where(rng.random(R) < 0.02, 0, rng.integers(1, 5, R)).astype(np.uint8) << 5) | 31
    oix = oracle.from_runs(runs)
    n = oix.bwlen()
    rows = np.concatenate([rng.integers(0, n, 30000), np.arange(0, min(n, 4000))]).astype(np.uint64)
    km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (20000, 14))]
    elo, eup = oix.find_intervals(km)

    def extract(g):
        out = np.zeros((rows.size, 1024), np.uint8)
        ln, pl = np.empty(rows.size, np.uint32), np.empty(rows.size, np.uint32)
        assert L.rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, 1024, ln.ctypes.data, pl.ctypes.data) == 0
        return out, ln, pl

    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=6) as plain:
        assert L.rsbwt_opened_for_reads(plain.handle) == 0
        o0, l0, p0 = extract(plain)
        S_plain = plain.window_span()
    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=6, for_reads=True) as g:
        assert L.rsbwt_opened_for_reads(g.handle) == 1
        S = g.window_span()
        assert S == span if span else S < S_plain  # (88 of 96 piece bytes hold pieces: a smaller window for the same fill)
        hints = L.rsbwt_psi_hint_lines(g.handle)  # there before any extraction
        dollars = int((runs & 31)[(runs >> 5) == 0].astype(np.int64).sum())
        assert hints >= 0.97 * ((n - dollars) // S) - 8, (hints, n, dollars, S)
        o1, l1, p1 = extract(g)
        assert L.rsbwt_psi_hint_lines(g.handle) == hints  # the index is immutable once open
        assert np.array_equal(l0, l1) and np.array_equal(p0, p1)
        fit = l1 != 0xFFFFFFFF
        for i in np.nonzero(fit)[0]:
            assert np.array_equal(o0[i, :l0[i]], o1[i, :l1[i]]), i
        for i in np.nonzero(fit)[0][::29]:
            pre, post = oix.extract(int(rows[i]), cap=4096)
            assert o1[i, :l1[i]].tobytes().decode() == pre + post and p1[i] == len(pre)
        lo, up = rsb.find_intervals(g, km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)
        pos = np.arange(0, n, 1, dtype=np.uint64) if n < 3000000 else np.arange(0, n, 3, dtype=np.uint64)
        tot = np.zeros(pos.size, np.uint64)
        for ch in "$ACGT":
            tot += g.occ_batch(ch, pos)
        assert np.array_equal(tot, pos + 1)
        for ch in "ACGT":
            t = oix.occ(ch, n - 1)
            if t:
                bc = rng.integers(1, t + 1, 3000).astype(np.uint64)
                idx = g.occ_at_batch(ch, bc)
                assert (g.char_batch(idx) == ord(ch)).all() and np.array_equal(g.occ_batch(ch, idx), bc)
                assert all(int(idx[j]) == oix.occ_at(ch, int(bc[j])) for j in range(0, 3000, 101))
        samp = pos[::max(1, pos.size // 20000)]
        assert all(int(o) == oix.occ("G", int(p_)) for o, p_ in zip(g.occ_batch("G", samp)[::40], samp[::40]))
        lo1, up1 = rsb.find_intervals_1mm(g, km[:200])
        assert np.array_equal(lo1[:, 0], elo[:200]) and np.array_equal(up1[:, 0], eup[:200])
        from readserver_amd import selfcheck
        res = selfcheck.extraction_vs_mirrors(g, rows[:3000], stride=1024)
        assert res["rows_differing"] == 0, res
    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=None, for_reads=True) as g2:  # no table: every step through the lines
        lo, up = rsb.find_intervals(g2, km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)


def test_gpu_read_packing_kernels_match_the_host_form(rsb):
    """rsbwt_pack_reads_dev / rsbwt_unpack_reads_dev against the torch form of readserver_amd/sharded.py."""
    import torch
    from readserver_amd import sharded
    g = torch.Generator().manual_seed(5)
    for n, stride in ((1, 16), (777, 48), (4096, 256), (33, 512)):
        lens = torch.randint(0, stride + 1, (n,), generator=g, dtype=torch.int32)
        lens[0] = -1
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8)
        reads = lut[torch.randint(0, 4, (n, stride), generator=g)]
        reads[torch.arange(stride)[None, :] >= lens.clamp(min=0)[:, None].long()] = 0x58
        want = sharded.pack_reads(reads, lens)
        got = sharded.pack_reads(reads.cuda(), lens.cuda())
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want), (n, stride)
        back = sharded.unpack_reads(got, lens.cuda())
        torch.cuda.synchronize()
        assert torch.equal(back.cpu(), sharded.unpack_reads(want, lens)), (n, stride)
    L = rsb.lib()
    assert L.rsbwt_pack_reads_dev(C.c_void_p(got.data_ptr()), C.c_void_p(got.data_ptr()), 4, 20, C.c_void_p(got.data_ptr()), 0, None) == -1


def _spelled(km):
    """[m][3k+1][k]: every k-mer, then its single substitutions position by position, alternatives in ACGT order."""
    m, k = km.shape
    out = np.repeat(km[:, None, :], 3 * k + 1, axis=1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for pos in range(k):
        for q in range(m):
            out[q, 1 + 3 * pos:4 + 3 * pos, pos] = [c for c in acgt if c != km[q, pos]][:3]
    return out


@pytest.mark.parametrize("tables", [7, 0, -7, -9])
def test_gpu_set_hits_1mm_all_shards_in_one_launch(rsb, oracle, tables):
    """rsbwt_set_hits_1mm_dev on a one-device set whose shards share a table depth traces the k-mers and resumes their
    variants in ALL shards by one launch each (csrc/sets.hip, set_hits_1mm_fused): every shard's list = the oracle's
    exact search of every spelled-out variant (query.cpp:24-41 per variant), ordered by variant index; a short list
    buffer keeps the count; the set's own counters show the fused launches ran.  tables < 0: the set's interleaved
    tables in their grouped format (rsbwt_set_attach_ktabs_format; at depth 9 most 9-mers of the small shard do not
    occur and are left to the search)."""
    import torch
    L = rsb.lib()
    dev = torch.device("cuda", 0)
    p = lambda t: C.c_void_p(t.data_ptr())
    sizes = [300000, 40000, 900000]
    shards, oixs = [], []
    for i, R in enumerate(sizes):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 2700 + i) == 0
        oixs.append(oracle.from_runs(runs))
        shards.append(rsb.GpuBWT(runs=runs, ktab_depth=None))
    ss = rsb.ShardSet(shards)
    if tables > 0:
        assert L.rsbwt_set_attach_ktabs(ss._s, tables) == 0
    elif tables < 0:
        assert L.rsbwt_set_attach_ktabs_format(ss._s, -tables, 1) == 0
        info = [g.ktab_info() for g in shards]
        assert all(g.ktab_depth() == -tables for g in shards) and all(i[0] == 1 and i[1] == 3 * 4 ** -tables for i in info)
        if tables == -9:  # the smaller the shard, the more T-mers that do not occur
            assert info[1][2] > info[0][2] >= info[2][2] and info[1][2] > 0.2 * 4 ** 9
    S = len(sizes)
    rng = np.random.default_rng(61)
    # (31, 1000): 3 x 1000 x 94 = 282,000 variant searches -- the resumed launch then runs on the one-lane-per-search
    # kernel (search_lines.hip, launch_search: >= 262,144 searches), with the per-shard traces and hit maps there
    for k, m in ((31, 400), (12, 600), (7, 300), (33, 120), (31, 1000)):
        km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(m, k))
        d_half = torch.empty((m // 2, k), dtype=torch.uint8, device=dev)
        assert L.rsbwt_sample_present_kmers_dev(shards[2].handle, m // 2, k, k, 5, p(d_half), None) == 0
        torch.cuda.synchronize()
        km[: m // 2] = d_half.cpu().numpy()
        km[1::5, k // 3] = ord("G")  # one substitution away from a present k-mer, often
        km[7, k // 2] = ord("N")
        V, wpq = 3 * k + 1, (k + 31) // 32
        d_km = torch.from_numpy(km).to(dev)
        d_pk = torch.empty((m, wpq), dtype=torch.int64, device=dev)
        d_ok = torch.empty(m, dtype=torch.uint8, device=dev)
        assert L.rsbwt_pack_kmers_dev(p(d_km), m, k, k, p(d_pk), p(d_ok), 0, None) == 0
        variants = _spelled(km).reshape(m * V, k)
        want, dense = [], []
        for oix in oixs:
            elo, eup = oix.find_intervals(variants, nthreads=8)
            elo[7 * V:8 * V], eup[7 * V:8 * V] = 1, 0  # a k-mer with a foreign symbol is invalid as a whole
            idx = np.nonzero(elo <= eup)[0]
            want.append((idx, elo[idx], eup[idx]))
            dense.append((elo, eup))
        # the dense form (rsbwt_set_find_intervals_1mm_dev: [S][m][3k+1] lower and upper) through the same launches
        d_lo = torch.full((S, m, V), -1, dtype=torch.int64, device=dev)
        d_up = torch.full((S, m, V), -1, dtype=torch.int64, device=dev)
        d_scr1 = torch.empty(L.rsbwt_set_1mm_scratch_bytes(ss._s, m, k), dtype=torch.uint8, device=dev)
        assert L.rsbwt_set_find_intervals_1mm_dev(ss._s, p(d_pk), p(d_ok), m, k, p(d_lo), p(d_up), p(d_scr1), None) == 0
        torch.cuda.synchronize()
        for s in range(S):
            assert np.array_equal(d_lo[s].cpu().numpy().view(np.uint64).reshape(-1), dense[s][0]), (k, s)
            assert np.array_equal(d_up[s].cpu().numpy().view(np.uint64).reshape(-1), dense[s][1]), (k, s)
        assert sum(len(w[0]) for w in want) > m // 2
        d_scr = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(ss._s, m, k), dtype=torch.uint8, device=dev)
        assert L.rsbwt_set_hits_1mm_is_fused(ss._s, m, k) == 1
        for cap in (8 * m, 3):
            d_hits = torch.full((S, cap, 4), -1, dtype=torch.int64, device=dev)
            d_tot = torch.full((S,), -1, dtype=torch.int64, device=dev)
            assert L.rsbwt_set_set_counting(ss._s, 1) == 0
            assert L.rsbwt_set_hits_1mm_dev(ss._s, p(d_pk), p(d_ok), m, k, p(d_hits), cap, p(d_tot), p(d_scr), None) == 0
            torch.cuda.synchronize()
            w = (C.c_uint64 * 16)()
            assert L.rsbwt_set_last_search_counters(ss._s, w) == 0 and L.rsbwt_set_set_counting(ss._s, 0) == 0
            assert int(w[10]) > 0  # WORK_PASSES of the set's own launch
            if tables > 0 and m * V * S >= 262144:
                assert int(w[12]) == 1  # WORK_SOLO: the resumed launch ran on lone lanes
            for s in range(S):
                idx, elo, eup = want[s]
                assert int(d_tot[s].item()) == len(idx), (k, s)
                n = min(cap, len(idx))
                rec = d_hits[s, :n].cpu().numpy().view(np.uint64)
                assert np.array_equal(rec[:, 2], idx[:n].astype(np.uint64)), (k, s)
                assert np.array_equal(rec[:, 0], elo[:n]) and np.array_equal(rec[:, 1], eup[:n]), (k, s)
                assert not rec[:, 3].any()
                assert (d_hits[s, n:] == -1).all()  # nothing past the list's end
    ss.close()
    for g in shards:
        g.close()


@pytest.mark.parametrize("fmt", [0, 1])
def test_gpu_set_hits_1mm_on_bwts_without_terminators(rsb, oracle, fmt):
    """The fused 1-mismatch launches on run streams that are no BWT of reads at all (no '$', a handful of symbols: what
    the fuzz campaign draws, tools/fuzz_parity.py seed 505): there a search that finds nothing at row 0 carries the
    reference's wrapped interval (0, 2^64 - 1) on as a LIVE one (query.cpp:11-15,35: unsigned compare) and every
    further step leaves it as it is, so whole families of variants "hit".  Every shard's list = the oracle's exact
    search of every spelled-out variant."""
    import torch
    L = rsb.lib()
    dev = torch.device("cuda", 0)
    p = lambda t: C.c_void_p(t.data_ptr())
    big = np.empty(40000, np.uint8)
    assert L.rsbwt_synth_runs_host(big.ctypes.data, big.size, 2750) == 0
    streams = [big, np.array([66, 97], np.uint8), np.array([33], np.uint8), np.array([0x9F, 0x9F, 0x41, 0x21], np.uint8)]
    k, T = 31, 5
    rng = np.random.default_rng(77)
    km = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=(40, k))
    for i, c in enumerate(b"ACGT"):
        km[2 * i] = c
        km[2 * i + 1] = c
        km[2 * i + 1, 9 + i] = b"ACGT"[(i + 1) & 3]
    km[8] = np.frombuffer(b"CCG" * 11, np.uint8)[:k]
    km[9, :] = ord("A"); km[9, k - 1] = ord("C")
    km[10, :] = ord("A"); km[10, 0] = ord("G")
    m, V = km.shape[0], 3 * k + 1
    variants = _spelled(km).reshape(m * V, k)
    shards = [rsb.GpuBWT(runs=r, ktab_depth=None) for r in streams]
    oixs = [oracle.from_runs(r) for r in streams]
    ss = rsb.ShardSet(shards)
    assert L.rsbwt_set_attach_ktabs_format(ss._s, T, fmt) == 0
    S = len(streams)
    d_km = torch.from_numpy(km).to(dev)
    d_pk = torch.empty((m, 1), dtype=torch.int64, device=dev)
    d_ok = torch.empty(m, dtype=torch.uint8, device=dev)
    assert L.rsbwt_pack_kmers_dev(p(d_km), m, k, k, p(d_pk), p(d_ok), 0, None) == 0
    assert L.rsbwt_set_hits_1mm_is_fused(ss._s, m, k) == 1
    cap = m * V
    d_hits = torch.full((S, cap, 4), -1, dtype=torch.int64, device=dev)
    d_tot = torch.full((S,), -1, dtype=torch.int64, device=dev)
    d_scr = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(ss._s, m, k), dtype=torch.uint8, device=dev)
    assert L.rsbwt_set_hits_1mm_dev(ss._s, p(d_pk), p(d_ok), m, k, p(d_hits), cap, p(d_tot), p(d_scr), None) == 0
    torch.cuda.synchronize()
    wrapped = 0
    for s_, oix in enumerate(oixs):
        elo, eup = oix.find_intervals(variants, nthreads=8)
        idx = np.nonzero(elo <= eup)[0]
        wrapped += int(np.count_nonzero(eup[idx] == np.uint64(2 ** 64 - 1)))
        n = int(d_tot[s_].item())
        rec = d_hits[s_, :min(n, cap)].cpu().numpy().view(np.uint64)
        got = {int(r[2]): (int(r[0]), int(r[1])) for r in rec}
        exp = {int(i): (int(elo[i]), int(eup[i])) for i in idx}
        miss = sorted(set(exp) - set(got))[:6]
        extra = sorted(set(got) - set(exp))[:6]
        name = lambda i: (i // V, "own" if i % V == 0 else ((i % V - 1) // 3, (i % V - 1) % 3))
        assert not miss and not extra, (s_, n, len(idx), [name(i) for i in miss], [name(i) for i in extra])
        assert got == exp, s_
        # ... and the shard's own list (the host form takes the shards one by one)
        own = rsb.hits_1mm_batch(shards[s_], km)
        assert len(own) == len(idx) and np.array_equal(own["lower"], elo[idx]) and np.array_equal(own["upper"], eup[idx])
    assert wrapped > 100  # the wrapped interval is what this test is about
    ss.close()
    for g in shards:
        g.close()


@pytest.mark.parametrize("tables", [0, 6, -6, 10])
def test_gpu_queries_of_their_own_lengths_in_one_search(rsb, oracle, tables):
    """rsbwt_set_find_intervals_var / rsbwt_set_count_var / rsbwt_set_query_var: a batch whose queries have lengths of
    their own -- a window of the service loop -- in ONE search (start records that say where their search goes on,
    csrc/search_lines.hip search_init_var_kernel).  Every (query, shard) = the oracle's findInterval of that string
    (query.cpp:24-41): lengths 1 .. 70 and the edges (the table's depth and its neighbours, 32 / 33 / 64 / 65 symbols, an
    empty query, foreign symbols, a run stream without terminators where an empty initInterval is stepped on); tables:
    none, plain, grouped, and a depth at which most T-mers of the small shards do not occur.  Then a batch large enough
    for the one-lane kernel."""
    L = rsb.lib()
    streams = []
    for i, R in enumerate([300000, 40000, 900000]):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 3100 + i) == 0
        streams.append(runs)
    streams.append(np.array([66, 97], np.uint8))  # "CCG": no terminator, no A, no T
    shards = [rsb.GpuBWT(runs=r, ktab_depth=None) for r in streams]
    oixs = [oracle.from_runs(r) for r in streams]
    ss = rsb.ShardSet(shards)
    T = abs(tables)
    if tables:
        assert L.rsbwt_set_attach_ktabs_format(ss._s, T, 1 if tables < 0 else 0) == 0
    rng = np.random.default_rng(tables + 100)
    acgt = "ACGT"
    rnd = lambda k: "".join(acgt[x] for x in rng.integers(0, 4, k))
    # strings that occur: read off shard 2's rows by the oracle
    texts = []
    for r in rng.integers(0, oixs[2].bwlen(), 60):
        try:
            pre, post = oixs[2].extract(int(r), cap=2000)
            if len(pre) + len(post) >= 80:
                texts.append(pre + post)
        except AssertionError:
            pass
    assert len(texts) > 10
    qs = ["", "A", "C", "G", "T", "N", "ACGTN", "A" * 40, "C" * 33, "CCG" * 9, "GCC" * 7 + "G", "ACGT" * 16, "ACGT" * 16 + "A"]
    for k in list(range(1, 71)) + [max(T - 1, 1), max(T, 1), T + 1, 31, 32, 33, 63, 64, 65]:
        qs.append(rnd(k))
        t = texts[int(rng.integers(len(texts)))]
        st = int(rng.integers(0, len(t) - k + 1))
        qs.append(t[st:st + k])
        qs.append("A" * (k - 1) + "C")
    def check(qs_):
        lo, up = ss.find_intervals_var(qs_)
        cnt = ss.count_var(qs_)
        want_cnt = np.zeros(len(qs_), np.uint64)
        for s_, oix in enumerate(oixs):
            for q_, w in enumerate(qs_):
                ok_ = len(w) > 0 and all(c in acgt for c in w)
                e = oix.find_interval(w) if ok_ else (1, 0)
                assert (int(lo[s_, q_]), int(up[s_, q_])) == e, (tables, s_, q_, w)
                if e[1] >= e[0]:
                    want_cnt[q_] += np.uint64((e[1] - e[0] + 1) & (2 ** 64 - 1))
        assert np.array_equal(cnt, want_cnt)
    check(qs)
    check(qs[:1])  # a lone empty query
    check(["ACGTAC"])
    # a batch that fills every lane of the launch: >= 262,144 searches per shard set go to the one-lane kernel when the
    # tables are deep enough for it (tables = 10 on these shards)
    big = []
    lens = rng.integers(12, 40, 70000)
    for k in lens[:35000]:
        t = texts[int(rng.integers(len(texts)))]
        st = int(rng.integers(0, len(t) - k + 1))
        big.append(t[st:st + int(k)])
    big += [rnd(int(k)) for k in lens[35000:]]
    lo, up = ss.find_intervals_var(big)
    by_len = {}
    for i, w in enumerate(big):
        by_len.setdefault(len(w), []).append(i)
    for k, idx in by_len.items():
        km = np.frombuffer("".join(big[i] for i in idx).encode(), np.uint8).reshape(len(idx), k)
        for s_, oix in enumerate(oixs):
            elo, eup = oix.find_intervals(km, nthreads=8)
            assert np.array_equal(lo[s_, idx], elo) and np.array_equal(up[s_, idx], eup), (tables, k, s_)
    ss.close()
    for g in shards:
        g.close()


def test_gpu_reads_of_queries_of_their_own_lengths(rsb, four_shards):
    """rsbwt_set_query_var: the reads containing each query of a mixed-length batch, partition by partition -- what the
    fixed-length call (rsbwt_set_query, pinned to the oracle above) gives query by query."""
    ss, shards, oixs, reads = four_shards
    rng = np.random.default_rng(8)
    qs = ["", "ACN", "N" * 12]
    for k in list(range(6, 60)) * 2:
        r = reads[int(rng.integers(len(reads)))]
        st = int(rng.integers(0, len(r) - k + 1))
        qs.append(r[st:st + k])
    qs += ["".join("ACGT"[x] for x in rng.integers(0, 4, int(k))) for k in rng.integers(6, 40, 30)]
    got = ss.query_var(qs, read_stride=96)
    some = 0
    for q_, w in enumerate(qs):
        want = ss.query([w], read_stride=96)[0] if w and "N" not in w else []
        assert got[q_] == want, (q_, w)
        some += len(want)
    assert some > 300


@pytest.mark.parametrize("devices", [1, 2])
def test_gpu_one_process_host_runs_the_benchs_sequence(rsb, oracle, devices, monkeypatch):
    """readserver_amd/onehost.py -- the C++ host's shape, what `bench.py --host cxx` times: per device pack + ONE fused
    launch over its shards + 10-byte records on a stream of its own, the records gathered onto the first device by
    rsbwt_set_gather_intervals_dev (ncclSend / ncclRecv) on a second stream, double-buffered.  Every device's pairs
    against the oracle; with two devices the root's blocks against what the devices searched.  On a one-GPU box the two
    devices are two LOGICAL devices on GPU 0 (include/rsbwt.h, rsbwt_logical_device): the host's whole sequence runs --
    two groups, two stream pairs, double-buffered records -- and the gather goes over peer copies instead of RCCL."""
    import torch
    from readserver_amd import onehost
    L = rsb.lib()
    if torch.cuda.device_count() < devices:
        monkeypatch.setenv("RSBWT_ENABLE_TEST_HOOKS", "1")
        monkeypatch.setenv("RSBWT_TEST_DEVICE_ALIASES", str(devices))
    rng = np.random.default_rng(11)
    Q, k = 30000, 31
    by_dev, oixs = [], []
    for d in range(devices):
        sh = []
        for i in range(3):
            R = [150000, 40000, 260000][i]
            runs = np.empty(R, np.uint8)
            assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, (1 << 62) | (77 + 10 * d + i)) == 0
            sh.append(rsb.GpuBWT(runs=runs, device=d, ktab_depth=None))
            oixs.append(oracle.from_runs(runs))
        by_dev.append(sh)
    host = onehost.OneProcessHost(by_dev, Q, k)
    host.attach_tables(6)
    km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (Q, k))].copy()
    km[::3, :20] = km[0, :20]  # (some shared prefixes)
    d_km = [torch.from_numpy(km).to(torch.device("cuda", host.devices[d])) for d in range(devices)]
    for _ in range(5):  # (past the double buffers)
        host.step(d_km)
    host.synchronize()
    for d in range(devices):
        pr = host.last_pairs(d).cpu().numpy().view(np.uint64)
        for i in range(3):
            elo, eup = oixs[3 * d + i].find_intervals(km, nthreads=4)
            assert np.array_equal(pr[i, :, 0], elo) and np.array_equal(pr[i, :, 1], eup), (d, i)
    ok = host.verify_last_gather()
    assert ok is None if devices == 1 else ok is True
    host.close()
    for sh in by_dev:
        for g in sh:
            g.close()


@pytest.mark.parametrize("grouped", [False, True])
def test_gpu_set_hits_1mm_worklists_keep_the_references_unsigned_carry(rsb, grouped):
    """A BWT without '$' whose rows all begin with one symbol: an empty interval at row 0 is (0, 2^64 - 1) and LIVES by
    the reference's unsigned compare (query.cpp:35, rlebwt.cpp:269) -- every variant of every k-mer then 'occurs'.  The
    set's hit lists (worklists: csrc/mm1_worklist.hip) must report exactly what each shard's own list (round 3's
    launches) reports; found by the fuzz campaign (seed 77) on the first worklist build, which dropped the carry."""
    import torch
    L = rsb.lib()
    k, m = 31, 60
    rng = np.random.default_rng(5)
    shards = [rsb.GpuBWT(runs=np.full(227, (4 << 5) | 31, np.uint8), ktab_depth=4, ktab_grouped=grouped),            # 7,037 x 'T', no '$'
              rsb.GpuBWT(runs=((rng.integers(0, 5, 40000).astype(np.uint8) << 5) | rng.integers(1, 32, 40000).astype(np.uint8)), ktab_depth=4, ktab_grouped=grouped),
              rsb.GpuBWT(runs=np.full(1, (2 << 5) | 31, np.uint8), ktab_depth=4, ktab_grouped=grouped)]               # 31 x 'C'
    ss = rsb.ShardSet(shards)
    assert L.rsbwt_set_hits_1mm_is_fused(ss._s, m, k) == 1
    km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (m, k))].copy()
    km[0] = ord("T")
    km[1] = ord("C")
    p = lambda t: C.c_void_p(t.data_ptr())
    d_km = torch.from_numpy(km).cuda()
    d_pk = torch.empty((m, 1), dtype=torch.int64, device="cuda")
    d_ok = torch.empty(m, dtype=torch.uint8, device="cuda")
    cap = m * (3 * k + 1)
    d_h = torch.zeros((3, cap, 4), dtype=torch.int64, device="cuda")
    d_t = torch.zeros(3, dtype=torch.int64, device="cuda")
    d_s = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(ss._s, m, k), dtype=torch.uint8, device="cuda")
    assert L.rsbwt_pack_kmers_dev(p(d_km), m, k, k, p(d_pk), p(d_ok), 0, None) == 0
    assert L.rsbwt_set_hits_1mm_dev(ss._s, p(d_pk), p(d_ok), m, k, p(d_h), cap, p(d_t), p(d_s), None) == 0
    torch.cuda.synchronize()
    for si, g in enumerate(shards):
        mine = rsb.hits_1mm_batch(g, km)
        rec = d_h[si, :int(d_t[si].item())].cpu().numpy().view(np.uint64)
        assert rec.shape[0] == len(mine), (si, rec.shape[0], len(mine))
        assert np.array_equal(rec[:, 0], mine["lower"]) and np.array_equal(rec[:, 1], mine["upper"])
        assert np.array_equal(rec[:, 2] // (3 * k + 1), mine["query"].astype(np.uint64))
    assert int(d_t[0].item()) == cap  # (every variant 'occurs' in the shard whose rows all begin with 'T', the largest symbol: C[] = 0 throughout)
    ss.close()
    for g in shards:
        g.close()


def test_gpu_set_table_format_auto_and_set_open_flag(rsb, oracle, tmp_path):
    """rsbwt_set_attach_ktabs_format(AUTO) takes the grouped records where the smallest shard's T-mers still have 64 rows
    each and four siblings fit a record (csrc/capi_internal.h, ktab_grouped_sensible), the plain entries elsewhere;
    rsbwt_set_open with RSBWT_OPEN_KTAB_GROUPED sizes the set's tables itself (plain unless grouped is deeper); intervals
    equal the oracle's whatever came out."""
    L = rsb.lib()
    rng = np.random.default_rng(8)
    paths, oixs = [], []
    for i, R in enumerate((3000000, 2500000)):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 5100 + i) == 0
        p = str(tmp_path / f"f{i}.bwt")
        n = int((runs & 31).astype(np.int64).sum())
        with open(p, "wb") as f:  # the 30-byte SGA header + the run bytes (rlebwt_reader.cpp:27-48)
            f.write((0xCACA).to_bytes(2, "little") + (0).to_bytes(8, "little") + n.to_bytes(8, "little") + R.to_bytes(8, "little") + (0).to_bytes(4, "little"))
            f.write(runs.tobytes())
        paths.append(p)
        oixs.append(oracle.from_runs(runs))
    n_min = min(o.bwlen() for o in oixs)
    km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (20000, 31))].copy()
    want = [o.find_intervals(km, nthreads=8) for o in oixs]
    for T in (8, 11):
        shards = [rsb.GpuBWT(p, ktab_depth=None) for p in paths]
        ss = rsb.ShardSet(shards)
        assert L.rsbwt_set_attach_ktabs_format(ss._s, T, 2) == 0
        sensible = 64 <= (n_min >> (2 * T)) and (n_min >> (2 * (T - 1))) <= 2048
        assert (T == 8) == sensible  # (1.4e7 symbols: 220 rows per 8-mer, 3 per 11-mer)
        for g in shards:
            assert g.ktab_depth() == T and g.ktab_info()[0] == (1 if sensible else 0)
        lo, up = ss.find_intervals(km)
        for s in range(2):
            assert np.array_equal(lo[s], want[s][0]) and np.array_equal(up[s], want[s][1]), (T, s)
        ss.close()
        for g in shards:
            g.close()
    # the set opened from files, tables sized by the library
    arr = (C.c_char_p * 2)(*[p.encode() for p in paths])
    h = C.c_void_p()
    assert L.rsbwt_set_open(arr, 2, None, 2, C.byref(h)) == 0  # RSBWT_OPEN_KTAB_GROUPED
    try:
        lo = np.empty((2, 20000), np.uint64)
        up = np.empty((2, 20000), np.uint64)
        assert L.rsbwt_set_find_intervals(h, km.ctypes.data, 20000, 31, 31, lo.ctypes.data, up.ctypes.data) == 0
        for s in range(2):
            assert np.array_equal(lo[s], want[s][0]) and np.array_equal(up[s], want[s][1]), s
        d0 = L.rsbwt_ktab_depth(L.rsbwt_set_shard(h, 0))
        assert d0 >= 2 and d0 == L.rsbwt_ktab_depth(L.rsbwt_set_shard(h, 1))
    finally:
        L.rsbwt_set_close(h)


def test_gpu_first_extraction_beside_searches_writes_nothing_a_search_reads(rsb, oracle, monkeypatch):
    """VERDICT r04 next #8 / ADVICE: a handle of the PLAIN layout used to be mutated by its first extraction -- its host
    view reassigned, the view in HBM re-uploaded, psi hints written into the resident lines -- while searches on other
    threads read all three without a lock.  Now the first extraction (and a set's first fused extraction) builds the
    select samples into a side table and publishes them in an extraction view of its own: the lines are byte for byte
    what rsbwt_open left (rsbwt_debug_peek), rsbwt_psi_hint_lines stays 0, and searches that run WHILE it happens
    answer as they did before.  (rsbwt_prepare_extraction is the explicit, owner-only way to get hints into a plain
    shard: test_gpu_psi_hints_change_nothing_but_the_requests.)"""
    import hashlib
    import threading
    import torch
    monkeypatch.setenv("RSBWT_ENABLE_TEST_HOOKS", "1")
    L = rsb.lib()
    rng = np.random.default_rng(77)
    shards, oixs = [], []
    for i, R in enumerate((400000, 250000)):
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, (1 << 62) | (900 + i)) == 0
        shards.append(rsb.GpuBWT(runs=runs, ktab_depth=8))
        oixs.append(oracle.from_runs(runs))
    ss = rsb.ShardSet(shards)

    def lines_digest(g):
        nb = g.num_lines() * 128
        buf = np.empty(nb, np.uint8)
        assert L.rsbwt_debug_peek(g.handle, 0, 0, buf.ctypes.data, nb) == 0, L.rsbwt_last_error()
        return hashlib.sha256(buf.tobytes()).hexdigest()
    before = [lines_digest(g) for g in shards]
    km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (40000, 31))].copy()
    want = [o.find_intervals(km, nthreads=4) for o in oixs]
    stop = threading.Event()
    errs, rounds = [], [0]

    def searcher(t):
        try:
            while not stop.is_set():
                lo, up = ss.find_intervals(km[t::4])
                for s_ in range(2):
                    assert np.array_equal(lo[s_], want[s_][0][t::4]) and np.array_equal(up[s_], want[s_][1][t::4])
                lo, up = rsb.find_intervals(shards[t % 2], km[t::4])
                assert np.array_equal(lo, want[t % 2][0][t::4]) and np.array_equal(up, want[t % 2][1][t::4])
                rounds[0] += 1
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))
    th = [threading.Thread(target=searcher, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    try:
        # the FIRST extraction of shard 0 (host rows), the first getOccAt of shard 1, the set's first fused extraction
        rows = rng.integers(0, oixs[0].bwlen(), 3000).astype(np.uint64)
        out = np.zeros((rows.size, 1024), np.uint8)
        ln, pl = np.empty(rows.size, np.uint32), np.empty(rows.size, np.uint32)
        assert L.rsbwt_extract(shards[0].handle, rows.ctypes.data, rows.size, out.ctypes.data, 1024, ln.ctypes.data, pl.ctypes.data) == 0
        t_tot = oixs[1].occ("T", oixs[1].bwlen() - 1)
        bc = rng.integers(1, t_tot + 1, 2000).astype(np.uint64)
        idx = shards[1].occ_at_batch("T", bc)
        nmin = min(o.bwlen() for o in oixs)
        rows2 = rng.integers(0, nmin, (2, 2000)).astype(np.int64)
        d_rows = torch.from_numpy(rows2).cuda()
        d_out = torch.zeros((2, 2000, 512), dtype=torch.uint8, device="cuda")
        d_len = torch.empty((2, 2000), dtype=torch.int32, device="cuda")
        d_pl = torch.empty((2, 2000), dtype=torch.int32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        st = torch.cuda.Stream()
        assert L.rsbwt_set_extract_dev(ss._s, p(d_rows), 2000, p(d_out), 512, p(d_len), p(d_pl), C.c_void_p(st.cuda_stream)) == 0
        st.synchronize()
    finally:
        stop.set()
        [t.join() for t in th]
    assert not errs, errs
    assert rounds[0] >= 1
    for i in range(0, rows.size, 7):
        if ln[i] != 0xFFFFFFFF:
            pre, post = oixs[0].extract(int(rows[i]), cap=4096)
            assert out[i, :ln[i]].tobytes().decode() == pre + post and pl[i] == len(pre)
    assert all(int(idx[j]) == oixs[1].occ_at("T", int(bc[j])) for j in range(0, bc.size, 13))
    xl = d_len.cpu().numpy().view(np.uint32)
    xo = d_out.cpu().numpy()
    for s_ in range(2):
        for i in range(0, 2000, 11):
            if xl[s_, i] != 0xFFFFFFFF:
                pre, post = oixs[s_].extract(int(rows2[s_, i]), cap=4096)
                assert xo[s_, i, :xl[s_, i]].tobytes().decode() == pre + post
    # nothing a search reads was written
    assert [lines_digest(g) for g in shards] == before
    assert all(L.rsbwt_psi_hint_lines(g.handle) == 0 and L.rsbwt_opened_for_reads(g.handle) == 0 for g in shards)
    ss.close()
    for g in shards:
        g.close()
