#!/usr/bin/env python3
"""Generates tests/golden/popbwt_v1.{json,npz} from the REAL reference.

Runs only in the build container (needs /root/reference): compiles the reference's own
src/bwt sources into oracle/_ref/libref_bwt.so (oracle/Makefile `ref`), synthesises the fixture
popBWT with this repo's deterministic builder, verifies that the reference is SOUND on it
(SURVEY 8c defects D1/D2: getOcc must equal naive rank at every position for every symbol,
getChar must equal the BWT), and records the reference's answers:

  * findInterval on 31-mers (half drawn from the indexed reads, half random), edge k-mers and a
    ladder of other lengths
  * extractPrefix + extractPostfix for a sample of rows
  * C[] (getPC), getBWLen, a strided getOcc table, a getOccAt sample

Only outputs (inputs + expected values) are stored, never reference source.
"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding  # noqa: E402
import readserver_amd as rsb  # noqa: E402

SYNTH = dict(seed=20261003, genome_len=400000, haplotypes=8, snp_rate=0.002, read_len=70,
             coverage=3.0, shard=-1, num_shards=1)


def load_ref():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bwt.so"))
    vp = C.c_void_p
    L.ref_open.restype = vp
    L.ref_open.argtypes = [C.c_char_p]
    L.ref_close.argtypes = [vp]
    L.ref_bwlen.restype = C.c_uint64
    L.ref_bwlen.argtypes = [vp]
    L.ref_pc.restype = C.c_uint64
    L.ref_pc.argtypes = [vp, C.c_char]
    L.ref_occ.restype = C.c_uint64
    L.ref_occ.argtypes = [vp, C.c_char, C.c_uint64]
    L.ref_occ_at.restype = C.c_uint64
    L.ref_occ_at.argtypes = [vp, C.c_char, C.c_uint64]
    L.ref_char.restype = C.c_char
    L.ref_char.argtypes = [vp, C.c_uint64]
    L.ref_occ_table.argtypes = [vp, C.c_char, vp, C.c_size_t, vp]
    L.ref_find_interval.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ref_extract.restype = C.c_size_t
    L.ref_extract.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    return L


def main():
    rsb.build()
    L = load_ref()
    tmp = os.environ.get("TMPDIR", "/tmp")
    bwt_path = os.path.join(tmp, "popbwt_v1.bwt")
    reads_path = os.path.join(tmp, "popbwt_v1.reads")
    rsb.synth_popbwt(bwt_path, reads_path, **SYNTH)
    sha = hashlib.sha256(open(bwt_path, "rb").read()).hexdigest()
    nstr, nsym, runs = oracle_binding.read_bwt_file(bwt_path)
    reads = open(reads_path).read().split()
    print(f"fixture: {nstr} reads, n={nsym}, R={runs.size}, sha256={sha[:16]}")
    assert runs.size >= (1 << 20), "fixture must have >= 2^20 runs (reference sound regime)"
    assert nsym % 65536 != 0

    bpi = bwt_path + ".bpi2"
    if os.path.exists(bpi):
        os.remove(bpi)
    h = L.ref_open(bwt_path.encode())
    assert L.ref_bwlen(h) == nsym

    # ---- soundness of the reference on this fixture: every position, every symbol
    naive = oracle_binding.NaiveIndex(runs)
    allpos = np.arange(nsym, dtype=np.uint64)
    for c, ch in enumerate("$ACGT"):
        out = np.empty(nsym, np.uint64)
        L.ref_occ_table(h, ch.encode(), allpos.ctypes.data, nsym, out.ctypes.data)
        bad = np.nonzero(out.astype(np.int64) != naive.cum[c, 1:])[0]
        assert bad.size == 0, f"reference unsound for {ch} from position {bad[0]}"
    step = 97
    for i in range(0, nsym, step):
        assert L.ref_char(h, i) == b"$ACGT"[naive.bwt[i]:naive.bwt[i] + 1]
    print("reference getOcc == naive rank at all positions; getChar sample ok")

    rng = np.random.default_rng(7)

    def ref_interval(w):
        lo, up = C.c_uint64(), C.c_uint64()
        L.ref_find_interval(h, w.encode(), len(w), C.byref(lo), C.byref(up))
        return lo.value, up.value

    # ---- 31-mers: half present, half random, plus edge cases
    k = 31
    kmers = []
    for _ in range(5000):
        r = reads[rng.integers(len(reads))]
        s = rng.integers(0, len(r) - k + 1)
        kmers.append(r[s:s + k])
    for _ in range(5000):
        kmers.append("".join("ACGT"[x] for x in rng.integers(0, 4, k)))
    kmers += ["A" * k, "C" * k, "G" * k, "T" * k, "AC" * 15 + "A", "T" * 30 + "A", "A" * 30 + "T"]
    kmers += [reads[0][:k], reads[-1][-k:], reads[len(reads) // 2][20:20 + k]]
    lo31 = np.empty(len(kmers), np.uint64)
    up31 = np.empty(len(kmers), np.uint64)
    for i, w in enumerate(kmers):
        lo31[i], up31[i] = ref_interval(w)
        assert (lo31[i], up31[i]) == naive.find_interval(w), w
    present = int(np.sum(up31 >= lo31))
    print(f"31-mers: {len(kmers)} queries, {present} non-empty")

    # ---- other lengths
    ladder = {}
    for kk in (1, 2, 3, 8, 15, 16, 17, 30, 32, 33, 48, 63, 64, 65, 70):
        ws = []
        for _ in range(150):
            r = reads[rng.integers(len(reads))]
            s = rng.integers(0, len(r) - kk + 1)
            ws.append(r[s:s + kk])
        for _ in range(50):
            ws.append("".join("ACGT"[x] for x in rng.integers(0, 4, kk)))
        lo = np.empty(len(ws), np.uint64)
        up = np.empty(len(ws), np.uint64)
        for i, w in enumerate(ws):
            lo[i], up[i] = ref_interval(w)
        ladder[kk] = (np.frombuffer("".join(ws).encode(), np.uint8).reshape(len(ws), kk), lo, up)

    # ---- extraction
    rows = np.sort(rng.choice(nsym, 1200, replace=False)).astype(np.uint64)
    ext = []
    pre_len = np.empty(rows.size, np.uint32)
    buf = C.create_string_buffer(4096)
    for i, r in enumerate(rows):
        pl = C.c_size_t()
        n = L.ref_extract(h, int(r), buf, 4096, C.byref(pl))
        ext.append(buf.raw[:n])
        pre_len[i] = pl.value
    width = max(len(e) for e in ext)
    ext_arr = np.zeros((rows.size, width), np.uint8)
    ext_len = np.empty(rows.size, np.uint32)
    for i, e in enumerate(ext):
        ext_arr[i, :len(e)] = np.frombuffer(e, np.uint8)
        ext_len[i] = len(e)

    # ---- C[], strided Occ, OccAt
    pc = np.array([L.ref_pc(h, ch.encode()) for ch in "$ACGT"], np.uint64)
    occ_pos = np.arange(0, nsym, 997, dtype=np.uint64)
    occ_tab = np.empty((5, occ_pos.size), np.uint64)
    for c, ch in enumerate("$ACGT"):
        L.ref_occ_table(h, ch.encode(), occ_pos.ctypes.data, occ_pos.size, occ_tab[c].ctypes.data)
    sel_sym = rng.integers(0, 5, 2000).astype(np.uint8)
    totals = naive.cum[:, -1]
    sel_bc = np.array([rng.integers(1, totals[c] + 1) for c in sel_sym], np.uint64)
    sel_idx = np.array([L.ref_occ_at(h, b"$ACGT"[c:c + 1], int(bc)) for c, bc in zip(sel_sym, sel_bc)], np.uint64)
    for c, bc, ix in zip(sel_sym, sel_bc, sel_idx):
        assert naive.bwt[ix] == c and naive.cum[c, ix + 1] == bc

    L.ref_close(h)

    out = dict(
        kmers31=np.frombuffer("".join(kmers).encode(), np.uint8).reshape(len(kmers), k),
        lower31=lo31, upper31=up31,
        rows=rows, ext=ext_arr, ext_len=ext_len, ext_prefix_len=pre_len,
        pc=pc, occ_pos=occ_pos, occ_tab=occ_tab,
        sel_sym=sel_sym, sel_bc=sel_bc, sel_idx=sel_idx,
    )
    for kk, (a, lo, up) in ladder.items():
        out[f"kmers{kk}"] = a
        out[f"lower{kk}"] = lo
        out[f"upper{kk}"] = up
    np.savez_compressed(os.path.join(HERE, "popbwt_v1.npz"), **out)
    meta = dict(
        what="golden vectors produced by ReadServer's own src/bwt compiled from /root/reference",
        generator="tests/golden/make_golden.py",
        synth=SYNTH, bwt_sha256=sha, num_strings=int(nstr), num_symbols=int(nsym),
        num_runs=int(runs.size), ladder=sorted(ladder), n31=len(kmers), n31_nonempty=present,
        reference_sound="getOcc == naive rank at all positions, all five symbols",
    )
    json.dump(meta, open(os.path.join(HERE, "popbwt_v1.json"), "w"), indent=1)
    print("wrote popbwt_v1.npz / popbwt_v1.json")


if __name__ == "__main__":
    main()
