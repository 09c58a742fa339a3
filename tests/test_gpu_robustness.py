"""GPU suite (-m gpu): what the kernels do with values no valid index produces -- positions past n,
a damaged k-mer table, damaged window lines, counts that are wrong (a fault-injection build).

Round 2 has a GPU memory-access fault on record (an A/B build whose rank arithmetic was changed:
DESIGN.md section 4, "the recorded fault").  The faulting access was the fetch of a spill line whose
index had been computed from a window past the last one; these tests hold every fetch index of the
search and walk kernels to staying inside the index whatever the positions are.  Each runs ONCE."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RSBWT_ENABLE_TEST_HOOKS"] = "1"  # rsbwt_debug_poke writes into a resident index: refused without it


def _random_runs(rng, R, with_dollar=True):
    sym = rng.integers(0 if with_dollar else 1, 5, R).astype(np.uint8)
    ln = rng.integers(1, 32, R).astype(np.uint8)
    return (sym << 5) | ln


def _kmers(rng, Q, k):
    return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (Q, k))]


def test_gpu_positions_past_n_have_defined_values(rsb):
    """getOcc / getChar / extraction handed positions the BWT does not have (the reference reads past
    its vSum there, BPTree.h:74-75): Occ(b, p >= n) = Occ(b, n - 1), and a row >= n comes back marked
    (length 0xFFFFFFFF) while the rows beside it are extracted as ever."""
    rng = np.random.default_rng(5)
    runs = _random_runs(rng, 200000)
    oix = ob.load().from_runs(runs)
    n = oix.bwlen()
    with rsb.GpuBWT(runs=runs) as g:
        wild = np.array([n, n + 1, n + 12345, 1 << 32, (1 << 40) - 1, 1 << 40, (1 << 40) + 7, 1 << 63, (1 << 64) - 2],
                        dtype=np.uint64)
        for ch in "ACGT$":
            got = g.occ_batch(ch, wild)
            assert (got == np.uint64(oix.occ(ch, n - 1))).all(), ch
        g.char_batch(wild)  # any symbol, but no fault
        with pytest.raises(rsb.RsbwtError):
            rsb.extract_reads(g, wild, stride=64)  # the Python mirror turns the mark into an error
        import ctypes as C
        rows = np.concatenate([wild, np.arange(0, 3000, 7, dtype=np.uint64)])
        out = np.full((rows.size, 256), 0x7E, np.uint8)
        ln, pl = np.empty(rows.size, np.uint32), np.empty(rows.size, np.uint32)
        assert rsb.lib().rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, 256, ln.ctypes.data,
                                       pl.ctypes.data) == 0
        assert (ln[:wild.size] == 0xFFFFFFFF).all()  # (what the row's buffer holds then is unspecified)
        for i in range(wild.size, rows.size, 37):
            pre, post = oix.extract(int(rows[i]))
            assert out[i, :ln[i]].tobytes().decode() == pre + post and pl[i] == len(pre)
        # the handle still answers
        km = _kmers(rng, 2000, 20)
        lo, up = rsb.find_intervals(g, km)
        elo, eup = oix.find_intervals(km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)


def test_gpu_a_damaged_kmer_table_entry_is_not_believed(rsb):
    """A table entry that is not an interval of this BWT's rows (lower + width > n) is treated as not
    tabulated: the search starts from initInterval (query.cpp:18-21) and ends on the right rows."""
    rng = np.random.default_rng(6)
    runs = _random_runs(rng, 400000)
    oix = ob.load().from_runs(runs)
    n = oix.bwlen()
    km = _kmers(rng, 20000, 31)
    elo, eup = oix.find_intervals(km)
    with rsb.GpuBWT(runs=runs, ktab_depth=8) as g:
        L = rsb.lib()
        ent = np.empty(4 ** 8, np.uint64)
        ent[0::3] = np.uint64(n - 3) | (np.uint64(9) << np.uint64(40))          # runs past the last row
        ent[1::3] = np.uint64((1 << 40) - 1) | (np.uint64(1) << np.uint64(40))  # far outside
        ent[2::3] = np.uint64(n + 1)                                            # empty, but not of this BWT
        assert L.rsbwt_debug_poke(g.handle, 1, 0, ent.ctypes.data, ent.nbytes) == 0
        assert L.rsbwt_debug_poke(g.handle, 1, ent.nbytes - 8, ent.ctypes.data, 16) != 0  # outside the table
        lo, up = rsb.find_intervals(g, km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)
        lo1, up1 = rsb.find_intervals_1mm(g, km[:300])
        assert np.array_equal(lo1[:, 0], elo[:300]) and np.array_equal(up1[:, 0], eup[:300])


def test_gpu_a_damaged_grouped_table_record_is_not_believed(rsb):
    """The same for the grouped table's 12-byte records (rsbwt_attach_ktab_format): a sibling whose rows would run past
    the BWT's last row is not believed, one the record gives no rows is left to the search anyway -- every search then
    starts from initInterval and ends on the right rows; the poke stays inside the table's 3 * 4^T bytes."""
    rng = np.random.default_rng(16)
    runs = _random_runs(rng, 400000)
    oix = ob.load().from_runs(runs)
    n = oix.bwlen()
    km = _kmers(rng, 20000, 31)
    elo, eup = oix.find_intervals(km)
    with rsb.GpuBWT(runs=runs, ktab_depth=8, ktab_grouped=True) as g:
        L = rsb.lib()
        assert g.ktab_info()[:2] == (1, 3 * 4 ** 8)

        def record(base, c):  # 96 bits little-endian: base:40, then four 14-bit running widths (csrc/line_format.h)
            v = base | (c[0] << 40) | (c[1] << 54) | (c[2] << 68) | (c[3] << 82)
            return [v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF, (v >> 64) & 0xFFFFFFFF]
        rec = np.empty((4 ** 7, 3), np.uint32)
        rec[0::3] = record(n - 3, (9, 9, 20, 20))            # siblings 0 and 2 run past the last row, 1 and 3 hold nothing
        rec[1::3] = record((1 << 40) - 1, (1, 2, 3, 4))      # far outside
        rec[2::3] = record(n + 1, (0, 0, 0, 5))              # three hold nothing, the fourth lies past n
        assert L.rsbwt_debug_poke(g.handle, 1, 0, rec.ctypes.data, rec.nbytes) == 0
        assert L.rsbwt_debug_poke(g.handle, 1, rec.nbytes - 8, rec.ctypes.data, 16) != 0  # outside the table
        lo, up = rsb.find_intervals(g, km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)
        lo1, up1 = rsb.find_intervals_1mm(g, km[:300])
        assert np.array_equal(lo1[:, 0], elo[:300]) and np.array_equal(up1[:, 0], eup[:300])


@pytest.mark.parametrize("span", [0, 2944])
def test_gpu_damaged_lines_are_survived(rsb, span):
    """Window lines overwritten with noise -- wild counts (lower / upper far past n), far links to
    anywhere, chunk offsets, piece bytes: no answer can be right, but every wave drains, nothing is read
    outside the index (a fault would kill this process) and a sound handle next to it is not disturbed."""
    rng = np.random.default_rng(7 + span)
    runs = _random_runs(rng, 300000)
    km = _kmers(rng, 50000, 31)
    oix = ob.load().from_runs(runs)
    elo, eup = oix.find_intervals(km)
    L = rsb.lib()
    with rsb.GpuBWT(runs=runs, window_span=span, ktab_depth=6) as bad, rsb.GpuBWT(runs=runs, window_span=span) as good:
        nb = bad.num_lines() * 128
        noise = rng.integers(0, 256, nb, dtype=np.uint8)
        keep = rng.random(bad.num_lines()) < 0.5  # every other line stays, so that searches get going
        noise = noise.reshape(-1, 128)
        for start in range(0, bad.num_lines(), 1 << 16):
            blk = noise[start:start + (1 << 16)]
            sel = ~keep[start:start + (1 << 16)]
            for i in np.nonzero(sel)[0]:
                assert L.rsbwt_debug_poke(bad.handle, 0, (start + int(i)) * 128, blk[i].ctypes.data, 128) == 0
            if start >= (1 << 17):
                break  # ~10^5 damaged lines are plenty
        rsb.find_intervals(bad, km)
        rsb.count_kmers(bad, km)
        rsb.find_intervals_1mm(bad, km[:2000])
        rows = rng.integers(0, bad.getBWLen(), 20000).astype(np.uint64)
        rsb.extract_reads(bad, rows, stride=128)
        bad.occ_batch("A", rows)
        bad.char_batch(rows)
        bad.occ_at_batch(b"C" * 1000, rng.integers(1, 1000, 1000).astype(np.uint64))
        lo, up = rsb.find_intervals(good, km)
        assert np.array_equal(lo, elo) and np.array_equal(up, eup)


@pytest.mark.parametrize("kernel,depth", [("pair", 4), ("solo", 10)])
def test_gpu_wild_counts_do_not_leave_the_index(rsb, kernel, depth):
    """The fault-injection build (tools/build_variant.sh wildocc -DRSB_FAULT_INJECT_WILD_OCC; built by
    __graft_entry__.build()): one Occ in 16 comes back with a wild value added, so lower and upper run
    all over [0, 2^64) -- the class of failure behind the fault recorded in round 2.  Both search kernels
    must drain and stay inside the index.  A child process, so that a fault cannot take the suite down."""
    lib = os.path.join(ROOT, "tools", "bin", "librsbwt_wildocc.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(rsb.lib_path()):
        # build() makes the fault-injection build and it travels with the snapshot; a box that did not get it makes its own
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "wildocc", "-DRSB_FAULT_INJECT_WILD_OCC"])
    env = dict(os.environ, RSBWT_LIB=lib, RSBWT_SEARCH_KERNEL=kernel)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wildocc_probe.py"), str(depth)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "survived" in out.stdout, out.stderr[-2000:]


# ---- the 40-bit range: positions near 2^40, the widest span -------------------------------------------

def test_gpu_fast_window_is_integer_division_up_to_2_pow_40(rsb):
    """w = p / S by an f64 multiply and one fix-up step (csrc/rank_device.h) against integer division over
    the positions a shard can have: p < 2^40 and fewer than 2^32 lines (17 lines per 16 windows), i.e. p <
    min(2^40, S * 4042322160).  Every span 2..2944 on the last 4096 such positions and on multiples of S
    +- 1; seven spans (the widest, the bench streams' own) on the whole last 2^20 positions and 10^6 random
    ones."""
    L = rsb.lib()
    rng = np.random.default_rng(40)

    def top(S):
        return min(1 << 40, S * 4042322160)

    def check(S, p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        w, r = np.empty(p.size, np.uint32), np.empty(p.size, np.uint32)
        assert L.rsbwt_debug_fast_window(p.ctypes.data, p.size, S, w.ctypes.data, r.ctypes.data, 0) == 0
        assert np.array_equal(w.astype(np.uint64), p // np.uint64(S)), S
        assert np.array_equal(r.astype(np.uint64), p % np.uint64(S)), S
    for S in range(2, 2945):
        t = top(S)
        m = rng.integers(1, t // S, 512).astype(np.uint64) * np.uint64(S)
        check(S, np.concatenate([np.arange(t - 4096, t, dtype=np.uint64), m - np.uint64(1), m, m + np.uint64(1),
                                 np.array([0, 1, S - 1, S, (t - 1) // S * S], dtype=np.uint64)]))
    for S in (2, 3, 272, 915, 2233, 2943, 2944):
        check(S, np.arange(top(S) - (1 << 20), top(S), dtype=np.uint64))
        check(S, rng.integers(0, top(S), 1000000).astype(np.uint64))
    one = np.zeros(1, np.uint64)
    assert L.rsbwt_debug_fast_window(one.ctypes.data, 1, 1, one.ctypes.data, one.ctypes.data, 0) == -7


def test_gpu_index_at_the_top_of_the_40_bit_range(rsb):
    """A shard of 2^40 - 1 - ((2^40 - 1) mod 31) symbols (the format's limit is 2^40) with the widest window
    span, 2,944: A^31 C^31 G^31 T^31 repeated, for which Occ has a closed form -- so every answer can be
    checked without a CPU index of 35 GB.  Occ at the last 2^20 positions and at 10^6 random ones, the
    select round trip, and findInterval (with and without the k-mer table) against the same backward search
    done in numpy on the closed form; one symbol more is refused with RSBWT_ERANGE."""
    import ctypes as C
    import torch
    L = rsb.lib()
    R = ((1 << 40) - 1) // 31
    n = 31 * R
    assert (1 << 40) - 31 <= n < (1 << 40)
    d_runs = torch.empty(R + 1, dtype=torch.uint8, device="cuda:0")
    step = 1 << 28
    for i in range(0, R + 1, step):
        j = min(R + 1, i + step)
        d_runs[i:j] = ((((torch.arange(i, j, device="cuda:0") & 3) + 1) << 5) | 31).to(torch.uint8)
    torch.cuda.synchronize()

    def occ(b, p):  # closed form: # of symbol b (1..4) in [0, p]; p = -1 gives 0
        c = p.astype(np.int64) + 1
        m, r = c // 31, c % 31
        return 31 * ((m + 3 - (b - 1)) // 4) + np.where(m % 4 == b - 1, r, 0)

    with pytest.raises(rsb.RsbwtError) as e:  # 31 symbols more: past 2^40
        rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R + 1), window_span=2944, ktab_depth=None)
    assert e.value.code == -7
    g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), window_span=2944, ktab_depth=None)
    del d_runs
    torch.cuda.empty_cache()
    assert g.getBWLen() == n and g.window_span() == 2944
    rng = np.random.default_rng(41)
    pos = np.concatenate([np.arange(n - (1 << 20), n), rng.integers(0, n, 1000000)]).astype(np.uint64)
    tot = {b: int(occ(b, np.array([n - 1]))[0]) for b in (1, 2, 3, 4)}
    Cb = {1: 0, 2: tot[1], 3: tot[1] + tot[2], 4: tot[1] + tot[2] + tot[3]}
    for b, ch in enumerate("ACGT", 1):
        assert g.getPC(ch) == Cb[b]
        assert np.array_equal(g.occ_batch(ch, pos).astype(np.int64), occ(b, pos)), ch
        bc = np.concatenate([rng.integers(1, tot[b] + 1, 20000), [1, tot[b]]]).astype(np.uint64)
        idx = g.occ_at_batch(ch, bc)
        assert (g.char_batch(idx) == ord(ch)).all() and np.array_equal(g.occ_batch(ch, idx), bc)
    assert not g.occ_batch("$", pos).any()
    # findInterval: k-mers that live long on this text (runs of one symbol), random ones, and the corners
    km = _kmers(rng, 30000, 31)
    km[:10000] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (10000, 1))]       # X^31
    sw = rng.integers(1, 31, 5000)
    for i in range(5000):                                                                # X^a Y^(31-a)
        km[10000 + i, sw[i]:] = km[10000 + i, 0] if km[10000 + i, 0] != km[10000 + i, 30] else ord("C")
    code = {65: 1, 67: 2, 71: 3, 84: 4}
    kb = np.vectorize(code.get)(km)
    lo = np.array([Cb[b] for b in kb[:, 30]], dtype=np.int64)
    hi = np.array([Cb[b] + tot[b] - 1 for b in kb[:, 30]], dtype=np.int64)
    alive = np.ones(km.shape[0], bool)
    for j in range(29, -1, -1):  # query.cpp:32-38
        b = kb[:, j]
        nl, nh = lo.copy(), hi.copy()
        for s in (1, 2, 3, 4):
            sel = alive & (b == s)
            nl[sel] = Cb[s] + occ(s, lo[sel] - 1)
            nh[sel] = Cb[s] + occ(s, hi[sel]) - 1
        lo, hi = nl, nh
        alive &= lo <= hi
    glo, gup = rsb.find_intervals(g, km)
    assert np.array_equal(glo.astype(np.int64), lo) and np.array_equal(gup.astype(np.int64), hi)
    assert alive.sum() > 2000 and (hi[alive] > (1 << 39)).any() and (~alive).sum() > 2000  # both kinds are there
    assert L.rsbwt_attach_ktab(g.handle, 12) == 0
    glo, gup = rsb.find_intervals(g, km)
    assert np.array_equal(glo.astype(np.int64), lo) and np.array_equal(gup.astype(np.int64), hi)
    # reads out of the same shard: every step of the walks is an Occ / select near the top of the range
    rows = np.concatenate([rng.integers(n - (1 << 30), n, 2000), rng.integers(0, n, 2000)]).astype(np.uint64)
    out = np.empty((rows.size, 64), np.uint8)
    ln, pl = np.empty(rows.size, np.uint32), np.empty(rows.size, np.uint32)
    assert L.rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, 64, ln.ctypes.data, pl.ctypes.data) == 0
    # no '$' anywhere: every walk runs into the stride (the reference would spin, query.cpp:48) and is marked
    assert (ln == 0xFFFFFFFF).all()
    g.close()
