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
    assert os.path.exists(lib), "build() makes the fault-injection build; it travels with the snapshot"
    env = dict(os.environ, RSBWT_LIB=lib, RSBWT_SEARCH_KERNEL=kernel)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wildocc_probe.py"), str(depth)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "survived" in out.stdout, out.stderr[-2000:]
