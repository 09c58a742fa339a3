"""Host-side mirror of ReadServer's ``BWT`` / ``query.h`` interface over librsbwt.so.

Names and argument meaning follow the reference so parity tests read like its call sites:
``class BWT`` (include/bwt/bwt.h:6-15), ``RLEBWT(filename)`` (include/bwt/rlebwt.h:17),
``findInterval / extractPrefix / extractPostfix / query / query_exactmatch``
(include/bwt/query.h:18-32).  Every query runs in the HIP library; numpy is used only to hand
buffers across the C-ABI.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _native
from ._native import RsbwtError, check, lib

BWTInterval = namedtuple("BWTInterval", ["lower", "upper"])  # include/bwt/query.h:8-11


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _kmer_matrix(kmers):
    """list of equal-length str/bytes, or a (Q, k) uint8 array -> (contiguous uint8 (Q, k), k)."""
    if isinstance(kmers, np.ndarray):
        a = np.ascontiguousarray(kmers, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("k-mer array must be (Q, k) uint8")
        return a, a.shape[1]
    ks = [s.encode() if isinstance(s, str) else bytes(s) for s in kmers]
    if not ks:
        return np.zeros((0, 0), np.uint8), 0
    k = len(ks[0])
    if any(len(s) != k for s in ks):
        raise ValueError("all k-mers of one batch must have the same length")
    return np.frombuffer(b"".join(ks), dtype=np.uint8).reshape(len(ks), k).copy(), k


class GpuBWT:
    """One BWT shard resident in HBM: ``RLEBWT`` (include/bwt/rlebwt.h:15-61) on the GPU.

    GpuBWT(filename)                      # SGA .bwt, as RLEBWT(filename)
    GpuBWT(runs=uint8 array, num_strings=) # RLUnit bytes in host memory
    GpuBWT(device_runs=(ptr, n), ...)      # RLUnit bytes already in HBM (e.g. a torch tensor)
    """

    def __init__(self, filename=None, device=0, *, runs=None, device_runs=None, num_strings=0,
                 ktab_depth=0, window_span=0, for_reads=False, ktab_grouped=False):
        """ktab_depth: depth of the k-mer table (0 = auto, None = no table); ktab_grouped: RSBWT_OPEN_KTAB_GROUPED
        -- its 3-bytes-per-T-mer format (include/rsbwt.h).  window_span: symbols
        per window of the HBM layout (0 = from the data: ~88 run pieces per 128-byte line).
        for_reads: RSBWT_OPEN_READS -- a psi hint in every window line, built with the index
        (the layout for a shard that serves read extraction)."""
        self._h = C.c_void_p()
        L = lib()
        flags = (31 if ktab_depth is None else int(ktab_depth) & 0x1F) << 5
        flags |= (int(window_span) & 0xFFF) << 12
        flags |= 1 if for_reads else 0
        flags |= 2 if ktab_grouped else 0
        if filename is not None:
            check(L.rsbwt_open(str(filename).encode(), device, flags, C.byref(self._h)))
        elif runs is not None:
            r = np.ascontiguousarray(runs, dtype=np.uint8)
            check(L.rsbwt_open_runs(_ptr(r), r.size, num_strings, device, flags, C.byref(self._h)))
        elif device_runs is not None:
            ptr, n = device_runs
            check(L.rsbwt_open_device_runs(C.c_void_p(ptr), n, num_strings, device, flags,
                                           C.byref(self._h)))
        else:
            raise ValueError("one of filename, runs, device_runs is required")

    # -- lifetime
    def close(self):
        if self._h:
            lib().rsbwt_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    # -- class BWT (include/bwt/bwt.h:6-15)
    def getBWLen(self):
        return lib().rsbwt_bwlen(self._h)

    def getPC(self, b):
        return lib().rsbwt_pc(self._h, b.encode() if isinstance(b, str) else b)

    def getF(self, index):
        return lib().rsbwt_f(self._h, index).decode()

    def getOcc(self, b, index):
        out = C.c_uint64()
        check(lib().rsbwt_occ(self._h, b.encode() if isinstance(b, str) else b,
                              index & 0xFFFFFFFFFFFFFFFF, C.byref(out)))
        return out.value

    def getChar(self, index):
        out = C.create_string_buffer(1)
        check(lib().rsbwt_char(self._h, index, out))
        return out.raw.decode()

    def getOccAt(self, b, bc):
        out = C.c_uint64()
        check(lib().rsbwt_occ_at(self._h, b.encode() if isinstance(b, str) else b, bc, C.byref(out)))
        return out.value

    # -- batched forms
    def occ_batch(self, syms, index):
        s = np.ascontiguousarray(np.frombuffer(syms.encode() if isinstance(syms, str) else bytes(syms),
                                               dtype=np.uint8))
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        if s.size == 1 and idx.size > 1:
            s = np.repeat(s, idx.size)
        out = np.empty(idx.size, np.uint64)
        check(lib().rsbwt_occ_batch(self._h, _ptr(s), _ptr(idx), idx.size, _ptr(out)))
        return out

    def char_batch(self, index):
        idx = np.ascontiguousarray(index, dtype=np.uint64)
        out = np.empty(idx.size, np.uint8)
        check(lib().rsbwt_char_batch(self._h, _ptr(idx), idx.size, _ptr(out)))
        return out

    def occ_at_batch(self, syms, bc):
        s = np.ascontiguousarray(np.frombuffer(syms.encode() if isinstance(syms, str) else bytes(syms),
                                               dtype=np.uint8))
        c = np.ascontiguousarray(bc, dtype=np.uint64)
        if s.size == 1 and c.size > 1:
            s = np.repeat(s, c.size)
        out = np.empty(c.size, np.uint64)
        check(lib().rsbwt_occ_at_batch(self._h, _ptr(s), _ptr(c), c.size, _ptr(out)))
        return out

    # -- shape
    def num_runs(self):
        return lib().rsbwt_num_runs(self._h)

    def num_strings(self):
        return lib().rsbwt_num_strings(self._h)

    def num_lines(self):
        return lib().rsbwt_num_lines(self._h)

    def ktab_depth(self):
        return lib().rsbwt_ktab_depth(self._h)

    def ktab_info(self):
        """(format, bytes, untabulated): 0 = plain / 1 = grouped, the table's HBM bytes, the T-mers a grouped table
        leaves to the search (rsbwt_ktab_info)."""
        f, b, u = C.c_uint32(), C.c_uint64(), C.c_uint64()
        check(lib().rsbwt_ktab_info(self._h, C.byref(f), C.byref(b), C.byref(u)))
        return f.value, b.value, u.value

    def window_span(self):
        return lib().rsbwt_window_span(self._h)

    def far_lines(self):
        return lib().rsbwt_far_lines(self._h)

    def spilled_symbols(self):
        return lib().rsbwt_spilled_symbols(self._h)

    def hbm_bytes(self):
        return lib().rsbwt_hbm_bytes(self._h)


# ---- query.h (src/bwt/query.cpp) ------------------------------------------------------------

def find_intervals(pBWT, kmers):
    """Batched findInterval (query.cpp:24-41): returns (lower, upper) uint64 arrays."""
    a, k = _kmer_matrix(kmers)
    Q = a.shape[0]
    lower = np.empty(Q, np.uint64)
    upper = np.empty(Q, np.uint64)
    check(lib().rsbwt_find_intervals(pBWT.handle, _ptr(a), Q, k, max(k, 1), _ptr(lower), _ptr(upper)))
    return lower, upper


def count_kmers(pBWT, kmers):
    """Batched count of count_reads (src/service/service.cpp:303-304)."""
    a, k = _kmer_matrix(kmers)
    Q = a.shape[0]
    out = np.empty(Q, np.uint64)
    check(lib().rsbwt_count(pBWT.handle, _ptr(a), Q, k, max(k, 1), _ptr(out)))
    return out


def find_intervals_1mm(pBWT, kmers):
    """1-mismatch search by composition: (lower, upper) of shape (Q, 3k+1); column 0 is the k-mer
    itself, column 1 + 3i + d position i with the d-th base of ACGT minus the original."""
    a, k = _kmer_matrix(kmers)
    Q = a.shape[0]
    lower = np.empty((Q, 3 * k + 1), np.uint64)
    upper = np.empty((Q, 3 * k + 1), np.uint64)
    check(lib().rsbwt_find_intervals_1mm(pBWT.handle, _ptr(a), Q, k, max(k, 1), _ptr(lower), _ptr(upper)))
    return lower, upper


HIT_1MM = np.dtype([("lower", "<u8"), ("upper", "<u8"), ("query", "<u4"), ("pos", "<i2"), ("base", "S1"),
                    ("reserved", "u1")])  # = rsbwt_hit_1mm


def hits_1mm_batch(pBWT, kmers, cap=None):
    """1-mismatch search with SURVEY 8 f3's output: a structured array (HIT_1MM) of the variants
    that occur, sorted by (query, pos, base); pos = -1 / base = b'' for the k-mer itself."""
    a, k = _kmer_matrix(kmers)
    Q = a.shape[0]
    cap = int(cap) if cap is not None else max(1024, 4 * Q)
    while True:
        out = np.zeros(cap, HIT_1MM)
        n = C.c_size_t()
        rc = lib().rsbwt_hits_1mm(pBWT.handle, _ptr(a), Q, k, max(k, 1), _ptr(out), cap, C.byref(n))
        if rc == -7 and n.value > cap:  # RSBWT_ERANGE: the list is longer than the buffer
            cap = n.value
            continue
        check(rc)
        return out[:n.value]


def hits_1mm(kmer, lower_row, upper_row):
    """The sorted list of (pos, base, lower, upper) of the non-empty variants of one k-mer
    (pos = -1 for the exact hit), from one row of find_intervals_1mm."""
    out = []
    if upper_row[0] >= lower_row[0]:
        out.append((-1, "", int(lower_row[0]), int(upper_row[0])))
    for v in range(1, len(lower_row)):
        if upper_row[v] >= lower_row[v]:
            pos, d = (v - 1) // 3, (v - 1) % 3
            alt = [c for c in "ACGT" if c != kmer[pos]][d]
            out.append((pos, alt, int(lower_row[v]), int(upper_row[v])))
    return out


def findInterval(pBWT, w):
    """BWTInterval findInterval(const BWT*, const std::string& w) (query.cpp:24-41)."""
    lo, up = find_intervals(pBWT, [w])
    return BWTInterval(int(lo[0]), int(up[0]))


def extractPrefix(pBWT, index, limit=1 << 16):
    """query.cpp:43-63: LF-walk left from row `index` until '$'."""
    out = []
    idx = index
    while True:
        b = pBWT.getChar(idx)
        if b == "$":
            break
        if len(out) >= limit:
            raise RsbwtError(-1, "extractPrefix did not meet '$'")
        idx = pBWT.getPC(b) + pBWT.getOcc(b, idx - 1)
        out.append(b)
    return "".join(reversed(out))


def extractPostfix(pBWT, index, limit=1 << 16):
    """query.cpp:65-85: F/select walk right from row `index` until '$'."""
    out = []
    idx = index
    while True:
        f = pBWT.getF(idx)
        if f == "$":
            break
        if len(out) >= limit:
            raise RsbwtError(-1, "extractPostfix did not meet '$'")
        fc = idx - pBWT.getPC(f) + 1
        idx = pBWT.getOccAt(f, fc)
        out.append(f)
    return "".join(out)


def extract_reads(pBWT, rows, stride=512):
    """Batched extractPrefix(row) + extractPostfix(row) (query.cpp:43-85) on the GPU: returns
    (list of read strings, prefix lengths)."""
    r = np.ascontiguousarray(rows, dtype=np.uint64)
    out = np.zeros((r.size, stride), np.uint8)
    ln = np.empty(r.size, np.uint32)
    pl = np.empty(r.size, np.uint32)
    check(lib().rsbwt_extract(pBWT.handle, _ptr(r), r.size, _ptr(out), stride, _ptr(ln), _ptr(pl)))
    if (ln == 0xFFFFFFFF).any():
        raise RsbwtError(-1, "a read does not fit the stride / a row is out of range")
    return [out[i, :ln[i]].tobytes().decode() for i in range(r.size)], pl


def query_batch(pBWT, kmers, read_stride=256):
    """Batched query (query.cpp:87-100) through rsbwt_query: a list, per k-mer, of the reads that
    contain it, in SA-row order."""
    a, k = _kmer_matrix(kmers)
    Q = a.shape[0]
    first = np.zeros(Q + 1, np.uint64)
    n = C.c_size_t()
    rc = lib().rsbwt_query(pBWT.handle, _ptr(a), Q, k, max(k, 1), _ptr(first), None, read_stride, None, 0, C.byref(n))
    if rc not in (0, -7):
        check(rc)
    total = n.value
    reads = np.zeros((max(total, 1), read_stride), np.uint8)
    ln = np.zeros(max(total, 1), np.uint32)
    if total:
        check(lib().rsbwt_query(pBWT.handle, _ptr(a), Q, k, max(k, 1), _ptr(first), _ptr(reads), read_stride, _ptr(ln),
                                total, C.byref(n)))
        if (ln[:total] == 0xFFFFFFFF).any():
            raise RsbwtError(-1, "a read does not fit read_stride")
    out = []
    for q in range(Q):
        out.append([reads[r, :ln[r]].tobytes().decode() for r in range(int(first[q]), int(first[q + 1]))])
    return out


def query_exactmatch_batch(pBWT, kmers):
    """Batched query_exactmatch (query.cpp:102-120) through rsbwt_query_exactmatch: bool array."""
    a, k = _kmer_matrix(kmers)
    Q = a.shape[0]
    found = np.zeros(Q, np.uint8)
    check(lib().rsbwt_query_exactmatch(pBWT.handle, _ptr(a), Q, k, max(k, 1), _ptr(found)))
    return found.astype(bool)


def query(pBWT, w):
    """vector<string> query(const BWT*, const string& w) (query.cpp:87-100): every read containing w."""
    if len(w) == 0:
        return []
    return query_batch(pBWT, [w])[0]


def query_exactmatch(pBWT, w):
    """bool query_exactmatch(const BWT*, const string& w) (query.cpp:102-120): is w itself a read."""
    if len(w) == 0:
        return False
    return bool(query_exactmatch_batch(pBWT, [w])[0])


# ---- shard sets -----------------------------------------------------------------

class ShardSet:
    """The shards held by one process; every query is searched in all of them (the reference
    broadcasts each request to all partitions: src/service/server.cpp:124,578)."""

    def __init__(self, shards):
        self.shards = list(shards)
        arr = (C.c_void_p * len(self.shards))(*[s.handle for s in self.shards])
        self._s = C.c_void_p()
        check(lib().rsbwt_set_from_handles(arr, len(self.shards), C.byref(self._s)))

    def close(self):
        if self._s:
            lib().rsbwt_set_close(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def find_intervals(self, kmers):
        a, k = _kmer_matrix(kmers)
        Q, S = a.shape[0], len(self.shards)
        lower = np.empty((S, Q), np.uint64)
        upper = np.empty((S, Q), np.uint64)
        check(lib().rsbwt_set_find_intervals(self._s, _ptr(a), Q, k, max(k, 1), _ptr(lower), _ptr(upper)))
        return lower, upper

    def count(self, kmers):
        a, k = _kmer_matrix(kmers)
        out = np.empty(a.shape[0], np.uint64)
        check(lib().rsbwt_set_count(self._s, _ptr(a), a.shape[0], k, max(k, 1), _ptr(out)))
        return out

    # -- queries of lengths of their own in one call (a window of the service loop): rsbwt_set_*_var
    @staticmethod
    def _var_text(queries):
        bs = [q if isinstance(q, (bytes, bytearray)) else str(q).encode() for q in queries]
        off = np.zeros(len(bs) + 1, np.uint64)
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        text = np.frombuffer(b"".join(bs) + b"\0", np.uint8).copy()
        return text, off

    def find_intervals_var(self, queries):
        text, off = self._var_text(queries)
        Q, S = len(queries), len(self.shards)
        lower = np.empty((S, Q), np.uint64)
        upper = np.empty((S, Q), np.uint64)
        check(lib().rsbwt_set_find_intervals_var(self._s, _ptr(text), _ptr(off), Q, _ptr(lower), _ptr(upper)))
        return lower, upper

    def count_var(self, queries):
        text, off = self._var_text(queries)
        out = np.empty(len(queries), np.uint64)
        check(lib().rsbwt_set_count_var(self._s, _ptr(text), _ptr(off), len(queries), _ptr(out)))
        return out

    def query_var(self, queries, read_stride=256):
        """per query: [(shard, read)] of every read containing it, shard 0's first (rsbwt_set_query_var)"""
        text, off = self._var_text(queries)
        Q = len(queries)
        first = np.zeros(Q + 1, np.uint64)
        n = C.c_size_t()
        rc = lib().rsbwt_set_query_var(self._s, _ptr(text), _ptr(off), Q, _ptr(first), None, None, read_stride, None, 0, C.byref(n))
        if rc not in (0, -7):
            check(rc)
        total = n.value
        reads = np.zeros((max(total, 1), read_stride), np.uint8)
        ln = np.zeros(max(total, 1), np.uint32)
        sh = np.zeros(max(total, 1), np.uint32)
        if total:
            check(lib().rsbwt_set_query_var(self._s, _ptr(text), _ptr(off), Q, _ptr(first), _ptr(sh), _ptr(reads), read_stride, _ptr(ln), total, C.byref(n)))
            if (ln[:total] == 0xFFFFFFFF).any():
                raise RsbwtError(-1, "a read does not fit read_stride")
        return [[(int(sh[r]), reads[r, :ln[r]].tobytes().decode()) for r in range(int(first[q]), int(first[q + 1]))] for q in range(Q)]

    # -- BASELINE configs[3] / configs[4] over the set: per-shard results side by side, the way the front-end
    # concatenates its partitions' replies (src/service/server.cpp:199-261)
    def hits_1mm(self, kmers):
        """Every shard's 1-mismatch hit list: (HIT_1MM array, first) with shard i's hits at first[i]:first[i+1]."""
        a, k = _kmer_matrix(kmers)
        Q, S = a.shape[0], len(self.shards)
        cap = max(1024, 4 * Q * S)
        first = np.zeros(S + 1, np.uint64)
        while True:
            out = np.zeros(cap, HIT_1MM)
            n = C.c_size_t()
            rc = lib().rsbwt_set_hits_1mm(self._s, _ptr(a), Q, k, max(k, 1), _ptr(out), cap, _ptr(first), C.byref(n))
            if rc == -7 and n.value > cap:
                cap = n.value
                continue
            check(rc)
            return out[:n.value], first

    def extract(self, shard_of, rows, stride=512):
        """Reads at (shard, row) pairs: (list of strings, prefix lengths)."""
        sh = np.ascontiguousarray(shard_of, dtype=np.uint32)
        r = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros((r.size, stride), np.uint8)
        ln, pl = np.empty(r.size, np.uint32), np.empty(r.size, np.uint32)
        check(lib().rsbwt_set_extract(self._s, _ptr(sh), _ptr(r), r.size, _ptr(out), stride, _ptr(ln), _ptr(pl)))
        if (ln == 0xFFFFFFFF).any():
            raise RsbwtError(-1, "a read does not fit the stride / a row is out of range")
        return [out[i, :ln[i]].tobytes().decode() for i in range(r.size)], pl

    def query(self, kmers, read_stride=256):
        """query() in every shard (query.cpp:87-100): per k-mer a list of (shard, read), shard 0's reads first."""
        a, k = _kmer_matrix(kmers)
        Q = a.shape[0]
        first = np.zeros(Q + 1, np.uint64)
        n = C.c_size_t()
        rc = lib().rsbwt_set_query(self._s, _ptr(a), Q, k, max(k, 1), _ptr(first), None, None, read_stride, None, 0, C.byref(n))
        if rc not in (0, -7):
            check(rc)
        total = n.value
        reads = np.zeros((max(total, 1), read_stride), np.uint8)
        ln = np.zeros(max(total, 1), np.uint32)
        sh = np.zeros(max(total, 1), np.uint32)
        if total:
            check(lib().rsbwt_set_query(self._s, _ptr(a), Q, k, max(k, 1), _ptr(first), _ptr(sh), _ptr(reads), read_stride,
                                        _ptr(ln), total, C.byref(n)))
            if (ln[:total] == 0xFFFFFFFF).any():
                raise RsbwtError(-1, "a read does not fit read_stride")
        return [[(int(sh[r]), reads[r, :ln[r]].tobytes().decode()) for r in range(int(first[q]), int(first[q + 1]))]
                for q in range(Q)]


def write_bpi2(bwt_path, bpi2_path=None):
    """src/util/index_rlebwt.cpp:19-22: writes the reference's FM-index file for a .bwt (default
    "<bwt>.bpi2"), byte-identical to RLEBWT::serialiseFMIndex.  Host only."""
    out = str(bpi2_path) if bpi2_path else str(bwt_path) + ".bpi2"
    check(lib().rsbwt_bpi2_write(str(bwt_path).encode(), out.encode()))
    return out


def check_bpi2(pBWT, bpi2_path, max_samples=1 << 20):
    """Validates a prebuilt .bpi2 against the resident index on the GPU: (buckets checked,
    mismatches, first difference or '')."""
    n, bad = C.c_uint64(), C.c_uint64()
    check(lib().rsbwt_bpi2_check(pBWT.handle, str(bpi2_path).encode(), max_samples, C.byref(n), C.byref(bad)))
    return n.value, bad.value, (lib().rsbwt_last_error().decode() if bad.value else "")


def synth_popbwt(bwt_path, reads_path=None, *, seed, genome_len, haplotypes, snp_rate, read_len,
                 coverage, shard=-1, num_shards=1):
    """Deterministic synthetic population BWT written as an SGA .bwt (host only; csrc/synth.cpp)."""
    check(lib().rsbwt_synth_popbwt(str(bwt_path).encode(),
                                   str(reads_path).encode() if reads_path else None,
                                   seed, genome_len, haplotypes, snp_rate, read_len, coverage,
                                   shard, num_shards))
