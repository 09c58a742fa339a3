import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import readserver_amd as rsb
import oracle_binding as ob
R, shift = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(R + shift)
sym = rng.integers(0, 5, R).astype(np.uint8); ln = rng.integers(1, 32, R).astype(np.uint8)
runs = (sym << 5) | ln
nv = ob.NaiveIndex(runs)
g = rsb.GpuBWT(runs=runs, dir_shift=shift, ktab_depth=None)
n = g.getBWLen(); pos = np.arange(n, dtype=np.uint64)
for c, ch in enumerate("$ACGT"):
    got = g.occ_batch(ch, pos); exp = nv.cum[c, 1:].astype(np.uint64)
    bad = np.nonzero(got != exp)[0]
    print(ch, "bad", bad.size, "first", bad[:10], "got", got[bad[:5]], "exp", exp[bad[:5]])
    if bad.size:
        # block boundaries near first bad
        ends = np.cumsum((runs & 31).astype(np.int64)); P0 = np.concatenate([[0], ends[95::96]])
        b0 = bad[0]; j = np.searchsorted(P0, b0, side="right") - 1
        print("  first bad pos", b0, "block", j, "P0", P0[j], "next", P0[j+1] if j+1 < len(P0) else None, "window", b0 >> shift, "pin", b0 & ((1<<shift)-1))
