"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE (the parity checker)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
_vp = C.c_void_p


def load():
    so = os.path.join(ODIR, "liboracle.so")
    src = os.path.join(ODIR, "rlebwt_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ODIR, "liboracle.so"])
    L = C.CDLL(so)
    L.rso_load.restype = _vp
    L.rso_load.argtypes = [C.c_char_p]
    L.rso_from_runs.restype = _vp
    L.rso_from_runs.argtypes = [_vp, C.c_uint64, C.c_uint64, C.c_int]
    L.rso_free.argtypes = [_vp]
    for f in ("rso_num_runs", "rso_num_strings", "rso_index_bytes", "rso_bwlen"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [_vp]
    L.rso_pc.restype = C.c_uint64
    L.rso_pc.argtypes = [_vp, C.c_char]
    L.rso_occ.restype = C.c_uint64
    L.rso_occ.argtypes = [_vp, C.c_char, C.c_uint64]
    L.rso_occ_at.restype = C.c_uint64
    L.rso_occ_at.argtypes = [_vp, C.c_char, C.c_uint64]
    L.rso_char.restype = C.c_char
    L.rso_char.argtypes = [_vp, C.c_uint64]
    L.rso_f.restype = C.c_char
    L.rso_f.argtypes = [_vp, C.c_uint64]
    L.rso_find_interval.argtypes = [_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rso_extract_prefix.restype = C.c_size_t
    L.rso_extract_prefix.argtypes = [_vp, C.c_uint64, C.c_char_p, C.c_size_t]
    L.rso_extract_postfix.restype = C.c_size_t
    L.rso_extract_postfix.argtypes = [_vp, C.c_uint64, C.c_char_p, C.c_size_t]
    L.rso_find_intervals.argtypes = [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, _vp, _vp, C.c_int]
    L.rso_extract_batch.restype = None
    L.rso_extract_batch.argtypes = [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, C.c_int]
    return Oracle(L)


class OracleIndex:
    def __init__(self, L, h, keep=None):
        self.L, self.h, self._keep = L, h, keep
        if not h:
            raise RuntimeError("oracle: could not build index")

    def close(self):
        if self.h:
            self.L.rso_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def bwlen(self):
        return self.L.rso_bwlen(self.h)

    def num_runs(self):
        return self.L.rso_num_runs(self.h)

    def pc(self, b):
        return self.L.rso_pc(self.h, b.encode())

    def occ(self, b, i):
        return self.L.rso_occ(self.h, b.encode(), i & 0xFFFFFFFFFFFFFFFF)

    def occ_at(self, b, bc):
        return self.L.rso_occ_at(self.h, b.encode(), bc)

    def char(self, i):
        return self.L.rso_char(self.h, i).decode()

    def f(self, i):
        return self.L.rso_f(self.h, i).decode()

    def find_interval(self, w):
        lo, up = C.c_uint64(), C.c_uint64()
        wb = w.encode() if isinstance(w, str) else bytes(w)
        self.L.rso_find_interval(self.h, wb, len(wb), C.byref(lo), C.byref(up))
        return lo.value, up.value

    def find_intervals(self, kmers, nthreads=1, want_steps=False):
        a = np.ascontiguousarray(kmers, dtype=np.uint8)
        Q, k = a.shape
        lo = np.empty(Q, np.uint64)
        up = np.empty(Q, np.uint64)
        st = np.zeros(Q, np.uint8)
        self.L.rso_find_intervals(self.h, a.ctypes.data, Q, k, k, lo.ctypes.data, up.ctypes.data,
                                  st.ctypes.data if want_steps else None, nthreads)
        return (lo, up, st) if want_steps else (lo, up)

    def extract_batch(self, rows, stride=512, nthreads=1):
        """extractPrefix + extractPostfix of every row over `nthreads` threads: ([n][stride] bytes, lengths
        (UINT32_MAX: longer than stride), prefix lengths)"""
        r = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros((r.size, stride), np.uint8)
        ln = np.empty(r.size, np.uint32)
        pl = np.empty(r.size, np.uint32)
        self.L.rso_extract_batch(self.h, r.ctypes.data, r.size, out.ctypes.data, stride, ln.ctypes.data, pl.ctypes.data, nthreads)
        return out, ln, pl

    def extract(self, row, cap=4096):
        buf = C.create_string_buffer(cap)
        n = self.L.rso_extract_prefix(self.h, row, buf, cap)
        assert n != C.c_size_t(-1).value, "prefix walk did not terminate"
        pre = buf.raw[:n].decode()
        n = self.L.rso_extract_postfix(self.h, row, buf, cap)
        assert n != C.c_size_t(-1).value, "postfix walk did not terminate"
        return pre, buf.raw[:n].decode()


class Oracle:
    def __init__(self, L):
        self.L = L

    def load(self, path):
        return OracleIndex(self.L, self.L.rso_load(str(path).encode()))

    def from_runs(self, runs, num_strings=0):
        r = np.ascontiguousarray(runs, dtype=np.uint8)
        return OracleIndex(self.L, self.L.rso_from_runs(r.ctypes.data, r.size, num_strings, 1), keep=r)


# ---- independent naive helpers (numpy), used to validate the oracle itself on small inputs ----

def read_bwt_file(path):
    raw = open(path, "rb").read()
    magic = int.from_bytes(raw[0:2], "little")
    assert magic == 0xCACA
    nstr = int.from_bytes(raw[2:10], "little")
    nsym = int.from_bytes(raw[10:18], "little")
    nruns = int.from_bytes(raw[18:26], "little")
    runs = np.frombuffer(raw, dtype=np.uint8, count=nruns, offset=30)
    return nstr, nsym, runs


def expand_runs(runs):
    """run bytes -> array of symbol ranks 0..4, one per BWT position."""
    runs = np.asarray(runs, dtype=np.uint8)
    return np.repeat((runs >> 5).astype(np.uint8), (runs & 31).astype(np.int64))


class NaiveIndex:
    """Plain cumulative counts over the expanded BWT: the ground truth for rank/select."""

    def __init__(self, runs):
        self.bwt = expand_runs(runs)
        self.n = self.bwt.size
        self.cum = np.zeros((5, self.n + 1), np.int64)
        for c in range(5):
            np.cumsum(self.bwt == c, out=self.cum[c, 1:])
        tot = self.cum[:, -1]
        self.C = np.concatenate([[0], np.cumsum(tot)[:-1]])

    def occ(self, c, i):  # number of c in bwt[0..i]
        return int(self.cum[c, i + 1]) if i >= 0 else 0

    def find_interval(self, w):
        rk = {"A": 1, "C": 2, "G": 3, "T": 4}
        if not w or any(ch not in rk for ch in w):
            return 1, 0
        c = rk[w[-1]]
        lo = int(self.C[c])
        up = lo + self.occ(c, self.n - 1) - 1
        for ch in reversed(w[:-1]):
            c = rk[ch]
            lo, up = int(self.C[c]) + self.occ(c, lo - 1), int(self.C[c]) + self.occ(c, up) - 1
            if lo > up:
                break
        return lo & 0xFFFFFFFFFFFFFFFF, up & 0xFFFFFFFFFFFFFFFF
