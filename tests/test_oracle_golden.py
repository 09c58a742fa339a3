"""CPU suite: pins oracle/rlebwt_oracle.c to the reference's golden vectors and to naive rank.

The golden vectors (tests/golden/popbwt_v1.npz) were produced by ReadServer's own src/bwt code
compiled from /root/reference (tests/golden/make_golden.py); the fixture BWT is re-synthesised
here from the committed parameters and checked against its committed SHA-256.
"""
import os

import numpy as np
import pytest

import oracle_binding as ob


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "popbwt_v1.npz"))


@pytest.fixture(scope="module")
def oix(oracle, fixture_bwt):
    path, meta = fixture_bwt
    ix = oracle.load(path)
    assert ix.bwlen() == meta["num_symbols"]
    assert ix.num_runs() == meta["num_runs"]
    return ix


def test_oracle_pc_and_len(oix, golden):
    assert [oix.pc(c) for c in "$ACGT"] == golden["pc"].tolist()


def test_oracle_find_interval_31mers(oix, golden):
    lo, up, steps = oix.find_intervals(golden["kmers31"], nthreads=4, want_steps=True)
    assert np.array_equal(lo, golden["lower31"])
    assert np.array_equal(up, golden["upper31"])
    assert steps.max() == 30 and steps.min() >= 1


def test_oracle_find_interval_ladder(oix, golden, fixture_bwt):
    for kk in fixture_bwt[1]["ladder"]:
        lo, up = oix.find_intervals(golden[f"kmers{kk}"])
        assert np.array_equal(lo, golden[f"lower{kk}"]), kk
        assert np.array_equal(up, golden[f"upper{kk}"]), kk


def test_oracle_occ_table(oix, golden):
    pos = golden["occ_pos"]
    for c, ch in enumerate("$ACGT"):
        got = np.array([oix.occ(ch, int(p)) for p in pos[::7]], np.uint64)
        assert np.array_equal(got, golden["occ_tab"][c][::7]), ch


def test_oracle_occ_at(oix, golden):
    for c, bc, ix in zip(golden["sel_sym"], golden["sel_bc"], golden["sel_idx"]):
        assert oix.occ_at("$ACGT"[c], int(bc)) == int(ix)


def test_oracle_extract(oix, golden):
    for r, e, n, pl in zip(golden["rows"], golden["ext"], golden["ext_len"], golden["ext_prefix_len"]):
        pre, post = oix.extract(int(r))
        assert len(pre) == pl
        assert (pre + post).encode() == e[:n].tobytes()


def test_oracle_single_vs_batch(oix, golden):
    for w, lo, up in list(zip(golden["kmers31"], golden["lower31"], golden["upper31"]))[:200]:
        assert oix.find_interval(w.tobytes()) == (int(lo), int(up))
    assert oix.find_interval("ACGNACG") == (1, 0)
    assert oix.find_interval("") == (1, 0)


# ---- the oracle against naive rank where the reference itself is unsound (D1 / D2 regimes) ----

def _random_runs(rng, R, with_dollar=True):
    sym = rng.integers(0 if with_dollar else 1, 5, R).astype(np.uint8)
    ln = rng.integers(1, 32, R).astype(np.uint8)
    return (sym << 5) | ln


@pytest.mark.parametrize("R", [1, 2, 63, 64, 65, 1023, 1024, 1025, 5000, 70000])
def test_oracle_vs_naive_all_positions(oracle, R):
    rng = np.random.default_rng(R)
    runs = _random_runs(rng, R)
    ix = oracle.from_runs(runs)
    nv = ob.NaiveIndex(runs)
    assert ix.bwlen() == nv.n
    pos = np.arange(nv.n) if nv.n <= 40000 else np.unique(
        np.concatenate([rng.integers(0, nv.n, 20000), np.arange(nv.n - 3000, nv.n), np.arange(3000)]))
    for c, ch in enumerate("$ACGT"):
        assert ix.pc(ch) == nv.C[c]
        for p in pos[:: max(1, len(pos) // 4000)]:
            assert ix.occ(ch, int(p)) == nv.occ(c, int(p)), (R, ch, p)
    for p in pos[:: max(1, len(pos) // 4000)]:
        assert ix.char(int(p)) == "$ACGT"[nv.bwt[p]]
    assert ix.occ("A", -1) == 0


def test_oracle_d2_exact_multiple_of_65536(oracle):
    # n an exact multiple of 65,536 with the threshold crossed by the last run (defect D2)
    runs = np.full(4228, (1 << 5) | 31, np.uint8)  # 4228 * 31 = 131068
    runs = np.concatenate([runs, np.array([(2 << 5) | 4], np.uint8)])  # +4 = 131072
    ix = oracle.from_runs(runs)
    assert ix.bwlen() == 131072
    assert ix.occ("A", 131071) == 131068
    assert ix.occ("C", 131071) == 4
    assert ix.occ("C", 131067) == 0


@pytest.mark.parametrize("R", [100, 3000, 200000])
def test_oracle_find_interval_vs_naive(oracle, R):
    rng = np.random.default_rng(1000 + R)
    runs = _random_runs(rng, R, with_dollar=(R != 3000))  # R=3000: no '$', exercises lower == 0
    ix = oracle.from_runs(runs)
    nv = ob.NaiveIndex(runs)
    for k in (1, 2, 5, 12, 31):
        for _ in range(300):
            w = "".join("ACGT"[x] for x in rng.integers(0, 4, k))
            assert ix.find_interval(w) == nv.find_interval(w), (R, w)


def test_oracle_select_vs_naive(oracle):
    rng = np.random.default_rng(5)
    runs = _random_runs(rng, 30000)
    ix = oracle.from_runs(runs)
    nv = ob.NaiveIndex(runs)
    for c, ch in enumerate("$ACGT"):
        where = np.nonzero(nv.bwt == c)[0]
        for bc in rng.integers(1, where.size + 1, 400):
            assert ix.occ_at(ch, int(bc)) == where[bc - 1]
