"""Self-checks that need no CPU oracle: one GPU path of the engine held against another.

extraction_vs_mirrors: the wave-cooperative extraction kernels (csrc/extract_lines.hip) against the
same walks replayed step by step with the class-BWT mirrors (rsbwt_char_batch / rsbwt_occ_batch /
rsbwt_occ_at_batch: thread-per-item kernels over line_format.h's scalar readers, which the tests
hold to the oracle).  Usable at sizes the oracle cannot follow (tools/check_extract_at_scale.py: a
20 GB shard) -- extractPrefix / extractPostfix, src/bwt/query.cpp:43-85.
"""
import numpy as np

from ._native import lib

_ACGT = np.frombuffer(b"ACGT", np.uint8)


def _walk(g, rows, stride, lf):
    cv = np.array([g.getPC(c) for c in "ACGT"], np.uint64)
    idx = rows.copy()
    nr = rows.size
    alive = np.ones(nr, bool)
    buf = np.zeros((nr, stride + 1), np.uint8)
    cnt = np.zeros(nr, np.int64)
    steps = 0
    while alive.any():
        ids = np.nonzero(alive)[0]
        if lf:  # query.cpp:49-57: getChar, then C[b] + Occ(b, i) - 1
            ch = g.char_batch(idx[ids])
            end = ~np.isin(ch, _ACGT)
        else:  # query.cpp:72-80: getF, then getOccAt
            f = np.searchsorted(cv, idx[ids], side="right")  # 0: '$'
            end = f == 0
            ch = _ACGT[np.maximum(f, 1) - 1]
        alive[ids[end]] = False
        ids, ch = ids[~end], ch[~end]
        if ids.size == 0:
            break
        rank = np.searchsorted(_ACGT, ch)
        if lf:
            idx[ids] = cv[rank] + g.occ_batch(ch, idx[ids]) - np.uint64(1)
        else:
            idx[ids] = g.occ_at_batch(ch, idx[ids] - cv[rank] + np.uint64(1))
        buf[ids, cnt[ids]] = ch
        cnt[ids] += 1
        alive[ids[cnt[ids] > stride]] = False  # does not fit: the kernels must say so too
        steps += ids.size
    return buf, cnt, steps


def extraction_vs_mirrors(g, rows, stride=1024):
    """Extracts `rows` with rsbwt_extract and replays every walk with the mirrors.  Returns a dict:
    rows, mirror_steps, rows_differing, first_difference."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    nr = rows.size
    out = np.zeros((nr, stride), np.uint8)
    ln = np.empty(nr, np.uint32)
    plen = np.empty(nr, np.uint32)
    if lib().rsbwt_extract(g.handle, rows.ctypes.data, nr, out.ctypes.data, stride, ln.ctypes.data, plen.ctypes.data) != 0:
        raise RuntimeError(lib().rsbwt_last_error().decode())
    pre, npre, s1 = _walk(g, rows, stride, True)
    post, npost, s2 = _walk(g, rows, stride, False)
    fits = (npre <= stride) & (npost <= stride) & (npre + npost <= stride)
    said_fits = ln != 0xFFFFFFFF
    wrong = fits != said_fits
    both = fits & said_fits
    wrong[both] |= (ln[both].astype(np.int64) != (npre + npost)[both]) | (plen[both].astype(np.int64) != npre[both])
    col = np.arange(stride)[None, :]
    ok_rows = np.nonzero(both & ~wrong)[0]
    for lo in range(0, ok_rows.size, 100000):  # character by character, a block of rows at a time
        ids = ok_rows[lo:lo + 100000]
        p, q = npre[ids][:, None], npost[ids][:, None]
        want = np.zeros((ids.size, stride), np.uint8)
        src = np.take_along_axis(pre[ids], np.clip(p - 1 - col, 0, stride), axis=1)  # the prefix is produced right to left
        want = np.where(col < p, src, want)
        src = np.take_along_axis(post[ids], np.clip(col - p, 0, stride), axis=1)
        want = np.where((col >= p) & (col < p + q), src, want)
        got = np.where(col < p + q, out[ids], 0)
        wrong[ids] |= (got != want).any(axis=1)
    bad = int(wrong.sum())
    first = None
    if bad:
        i = int(np.nonzero(wrong)[0][0])
        first = {"row": int(rows[i]), "want_len": int(npre[i] + npost[i]), "got_len": int(ln[i]),
                 "prefix_len": [int(npre[i]), int(plen[i])]}
    return {"rows": int(nr), "mirror_steps": int(s1 + s2), "rows_differing": bad, "first_difference": first}
