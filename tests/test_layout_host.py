"""CPU suite: the window-line layout (readserver_amd/csrc/line_format.h).  The builder's code and
the scalar readers are plain C++ shared by host and kernels; here they are run on the host and held
to naive ranks at every position.  No query path is involved (rsbwt_layout_selftest_host answers
none): all searches run on the GPU only."""
import ctypes as C

import numpy as np
import pytest


def _runs(style, R, rng, L):
    if style == "synth":
        r = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(r.ctypes.data, R, 555) == 0
        return r
    if style == "dense":
        return (rng.integers(1, 5, R).astype(np.uint8) << 5) | 1
    if style == "long":
        return (rng.integers(1, 5, R).astype(np.uint8) << 5) | 31
    if style == "mixed":
        ln = np.where((np.arange(R) // 5000) % 2 == 0, 1, 31).astype(np.uint8)
        return (rng.integers(0, 5, R).astype(np.uint8) << 5) | ln
    return (rng.integers(0, 5, R).astype(np.uint8) << 5) | rng.integers(1, 32, R).astype(np.uint8)


@pytest.mark.parametrize("style,R,span", [("synth", 60000, 0), ("synth", 60000, 256), ("synth", 60000, 1536),
                                          ("synth", 60000, 2944), ("dense", 30000, 0), ("dense", 30000, 2944),
                                          ("dense", 20000, 300), ("long", 30000, 0), ("long", 30000, 256),
                                          ("mixed", 60000, 0), ("mixed", 60000, 768), ("rand", 40000, 0),
                                          ("rand", 1, 0), ("rand", 2, 0), ("rand", 97, 0), ("rand", 17, 5),
                                          ("rand", 3000, 2), ("rand", 3000, 3)])
@pytest.mark.parametrize("room", [False, True])
def test_layout_is_exact_at_every_position(rsb, style, R, span, room):
    """room: the RSBWT_OPEN_READS layout -- every window line keeps its last 8 piece bytes for a psi hint."""
    L = rsb.lib()
    rng = np.random.default_rng(R + span)
    runs = _runs(style, R, rng, L)
    st = (C.c_uint64 * 6)()
    bad = C.c_uint64()
    n = int((runs & 31).astype(np.int64).sum())
    want = span
    if room and span == 0:  # (the GPU builder aims at 88/96 of the plain layout's pieces per window: build_lines.hip)
        want = max(2, min(2944, int(82.0 * n / R + 0.5)))
    rc = L.rsbwt_layout_selftest_host(runs.ctypes.data, R, want | (1 << 31 if room else 0), st, C.byref(bad))
    assert rc == 0, f"first disagreement at position {bad.value}"
    S, nlines, far, chunkw, farw, spilled = list(st)
    assert 2 <= S <= 2944 and (want == 0 or S == want)
    assert nlines == ((n + S - 1) // S + 15) // 16 * 17 + far
    if span == 0 and style != "mixed":  # (the GPU builder also shrinks S when stretches differ: build_lines.hip)
        assert spilled <= 0.03 * n  # ~90 pieces per window: few positions past their line
    if style == "dense" and span:
        assert far > 0 and farw > 0
    if style == "synth" and span == 0 and not room:
        assert chunkw > 0 and nlines * 128 < 1.62 * R  # ~1.55 bytes per run byte


def test_layout_rejects_symbols_above_four(rsb):
    L = rsb.lib()
    runs = np.array([(1 << 5) | 3, (6 << 5) | 2], np.uint8)
    assert L.rsbwt_layout_selftest_host(runs.ctypes.data, 2, 0, None, None) == -3


def test_layout_code_under_address_and_ub_sanitizers(tmp_path):
    """tests/native/fuzz_layout_host.cpp: the layout builder and the scalar readers of line_format.h
    (the code the GPU kernels run) built for the CPU with -fsanitize=address,undefined, on 30 random run
    streams of six shapes at spans 2..2,944, every position held to naive ranks; and the grouped k-mer table's record
    code on 200,000 groups of four siblings, sound ones and arbitrary ones (a sibling that comes back with a width is
    exactly what went in)."""
    import os
    import shutil
    import subprocess
    import pytest
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_layout_host")
    srcs = [os.path.join(root, "tests", "native", "fuzz_layout_host.cpp"),
            os.path.join(root, "readserver_amd", "csrc", "layout_host.cpp")]
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        f"-I{os.path.join(root, 'include')}", *srcs, "-o", exe], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("no sanitizer runtime here")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe, "30"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]


@pytest.mark.parametrize("style,R,span", [("synth", 60000, 0), ("pop", 80000, 0), ("pop", 80000, 300), ("pop", 60000, 1024),
                                          ("dense", 30000, 90), ("long", 30000, 0), ("mixed", 60000, 0), ("rand", 40000, 0),
                                          ("desert", 60000, 0), ("rand", 1, 0), ("rand", 97, 5), ("rand", 3000, 2),
                                          ("pop", 80000, 2000), ("dense", 30000, 2944)])
@pytest.mark.parametrize("room", [False, True])
def test_select_samples_and_psi_hints_are_exact(rsb, style, R, span, room):
    """The select samples name the window of EVERY occurrence (or bound it from below where they say so, and the floor
    search between two samples then ends on it), a psi hint bounds the window psi takes EVERY row of its window to
    (lo..hi, or from lo on where it says so), and the lines that carry hints still answer Occ / getChar / getOccAt like
    the naive BWT at every position -- host run of the code the GPU runs.  room: the RSBWT_OPEN_READS layout (a hint
    slot in every window line, one sample per 4,096 occurrences)."""
    L = rsb.lib()
    rng = np.random.default_rng(R * 7 + span)
    if style == "pop":
        runs = np.empty(R, np.uint8)
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, 77 | (1 << 62)) == 0
    elif style == "desert":  # stretches without one symbol: sample blocks spread over many windows
        sym = np.where((np.arange(R) // 7000) % 2 == 0, rng.integers(0, 5, R), rng.integers(0, 4, R))
        runs = ((sym.astype(np.uint8)) << 5) | rng.integers(1, 9, R).astype(np.uint8)
    else:
        runs = _runs(style, R, rng, L)
    st = (C.c_uint64 * 4)()
    bad = C.c_uint64()
    n = int((runs & 31).astype(np.int64).sum())
    want = span
    if room and span == 0:
        want = max(2, min(2944, int(82.0 * n / R + 0.5)))
    rc = L.rsbwt_layout_selftest_psi_host(runs.ctypes.data, R, want | (1 << 31 if room else 0), st, C.byref(bad))
    assert rc == 0, f"first disagreement at {bad.value}"
    words, inexact, hint_lines, by_hint = list(st)
    assert words >= n // (4096 if room else 256)
    if style == "desert":
        assert inexact > 0
    if style == "pop" and span == 0:
        assert hint_lines > 0.3 * (n // 500) and by_hint > 0.2 * n
    if room and n > 3000:  # every window whose rows have a psi at all carries a hint
        S = want if want else int(90.0 * n / R + 0.5)
        dollars = int((runs & 31)[(runs >> 5) == 0].astype(np.int64).sum())  # ('$' rows end a walk: their windows get none)
        nwin = (n - dollars + S - 1) // S
        assert hint_lines >= 0.9 * nwin - 8, (hint_lines, nwin)
        if style == "pop":
            assert by_hint > 0.8 * n, (by_hint, n)


def test_grouped_ktab_record_code(rsb):
    """The 12-byte record of four sibling T-mers (line_format.h, rsbwt_attach_ktab_format): what it gives back is the
    sibling's interval exactly, or 'left to the search' -- for an empty sibling (the reference's empty interval depends
    on the step the search died at: query.cpp:33-38), a group of 16383 rows or more, siblings that do not tile."""
    L = rsb.lib()
    rng = np.random.default_rng(5)
    G = 20000
    WIDE = 0xFFFFFF
    base = rng.integers(0, (1 << 40) - (1 << 20), G).astype(np.uint64)
    base[:8] = [0, 1, (1 << 40) - 70000, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 255 << 32, (1 << 40) - 16384]
    width = rng.integers(0, 6, (G, 4)).astype(np.uint64) * rng.integers(0, 3000, (G, 4)).astype(np.uint64)
    width[rng.random((G, 4)) < 0.2] = 0
    width[10] = [16382, 0, 0, 0]     # the widest group a record holds
    width[11] = [16383, 0, 0, 0]     # one row too many
    width[12] = [4000, 4000, 4000, 4382]
    width[13] = [0, 0, 0, 0]
    width[14] = [0, 0, 0, 1]
    lower = np.zeros((G, 4), np.uint64)
    upper = np.zeros((G, 4), np.uint64)
    at = base.copy()
    for i in range(4):
        live = width[:, i] > 0
        lower[:, i] = np.where(live, at, rng.integers(1, 1 << 40, G).astype(np.uint64))  # an empty interval sits anywhere
        upper[:, i] = lower[:, i] + width[:, i] - np.uint64(1)
        at = at + width[:, i]
    # groups that do not tile (never for a sound BWT): a gap before the last live sibling
    untiled = np.zeros(G, bool)
    for g in range(100, 200):
        live = np.nonzero(width[g] > 0)[0]
        if len(live) >= 2:
            lower[g, live[-1]] += np.uint64(1)
            upper[g, live[-1]] += np.uint64(1)
            untiled[g] = True
    # the reference's (0, 2^64 - 1) carry (query.cpp:35) is an empty interval too
    lower[15, 0], upper[15, 0], width[15, 0] = 0, np.uint64(0xFFFFFFFFFFFFFFFF), 0
    out = np.zeros((G, 4), np.uint64)
    assert L.rsbwt_ktab_group_selftest_host(lower.ctypes.data, upper.ctypes.data, G, out.ctypes.data) == 0
    total = width.sum(axis=1)
    escaped = (total >= 16383) | untiled
    got_w = (out >> np.uint64(40)).astype(np.int64)
    got_lo = out & np.uint64((1 << 40) - 1)
    for i in range(4):
        said = ~escaped & (width[:, i] > 0)
        assert np.array_equal(got_w[said, i], width[said, i].astype(np.int64))
        assert np.array_equal(got_lo[said, i], lower[said, i])
        assert np.all(got_w[~said, i] == WIDE)
    assert not escaped[10] and escaped[11] and not escaped[12] and np.all(got_w[13] == WIDE) and got_w[14, 3] == 1
    assert (~escaped).sum() > 0.5 * G and escaped.sum() > 1000  # both kinds are well represented


@pytest.mark.parametrize("T", [5, 7, 9])
def test_grouped_ktab_records_of_a_valid_bwt_are_the_oracles_intervals(rsb, oracle, fixture_bwt, T):
    """What the grouped k-mer table rests on, held to the oracle on the golden popBWT (a VALID BWT of '$'-terminated reads,
    the oracle pinned to the compiled reference's vectors): the four T-mers that differ in their LAST symbol tile one
    stretch of rows -- rows that begin with a shorter suffix and '$' sort before all four -- so EVERY group the record
    code gives up on is one of 16,383 rows or more, and every T-mer that occurs in a smaller group comes back from its
    12-byte record as exactly findInterval's answer (query.cpp:24-41).  CPU only: the oracle's intervals through the
    record code the builder kernel and the lookups share (csrc/line_format.h)."""
    L = rsb.lib()
    path, _ = fixture_bwt
    oix = oracle.load(path)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    G = 4 ** (T - 1)
    g = np.arange(G, dtype=np.int64)
    # group g = the first T - 1 symbols (first symbol in the low bits of the code, as the tables index them); sibling = the last
    first = np.stack([(g >> (2 * j)) & 3 for j in range(T - 1)], axis=1)
    km = np.empty((G, 4, T), np.uint8)
    km[:, :, :T - 1] = acgt[first][:, None, :]
    km[:, :, T - 1] = acgt[None, :]
    lo, up = oix.find_intervals(km.reshape(-1, T), nthreads=8)
    lo, up = lo.reshape(G, 4).copy(), up.reshape(G, 4).copy()
    out = np.zeros((G, 4), np.uint64)
    assert L.rsbwt_ktab_group_selftest_host(lo.ctypes.data, up.ctypes.data, G, out.ctypes.data) == 0
    live = up >= lo  # (no interval of this BWT ends at 2^64 - 1: its first row is a '$' row)
    width = np.where(live, up - lo + np.uint64(1), 0).astype(np.int64)
    total = width.sum(axis=1)
    got_w = (out >> np.uint64(40)).astype(np.int64)
    got_lo = (out & np.uint64((1 << 40) - 1))
    small = total < 16383
    escaped = (got_w == 0xFFFFFF).all(axis=1)
    assert not (escaped & small & (total > 0)).any()            # no group of a valid BWT fails to tile
    assert (escaped | small).all()                               # a group too wide for its record is given up whole
    said = live & small[:, None]
    assert np.array_equal(got_w[said], width[said]) and np.array_equal(got_lo[said], lo[said])
    assert (got_w[~said] == 0xFFFFFF).all()                      # a T-mer that does not occur is left to the search
    if T >= 7:  # (555 rows per 7-mer of this BWT: nearly every group fits its record; at T = 5 none does)
        assert said.sum() > 0.9 * live.sum()
    else:
        assert escaped.all()
    # the siblings of a group are neighbours in row order: each begins where the one before it ended
    for i in range(1, 4):
        prev_end = np.where(live[:, i - 1], up[:, i - 1] + np.uint64(1), 0)
        both = live[:, i] & live[:, i - 1]
        assert np.array_equal(lo[both, i], prev_end[both])
