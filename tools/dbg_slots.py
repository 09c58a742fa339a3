import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import readserver_amd as rsb
import oracle_binding as ob
log = open("/root/repo/gpurun_out/dbg_slots.log", "w")
def P(*a):
    print(*a, file=log, flush=True)
rng = np.random.default_rng(4256)
R = 300000
runs = (rng.integers(1, 5, R).astype(np.uint8) << 5) | 31
orc = ob.load(); oix = orc.from_runs(runs)
for span in (0,):
    g = rsb.GpuBWT(runs=runs, slots=True, slot_span=span, ktab_depth=None)
    P("span", span, "S", g.slot_span(), "ovf", g.slot_overflow_blocks(), "n", g.getBWLen(), "dirshift", g.dir_shift())
    for k in (31,):
        for Q in (20000,):
            km = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (Q, k))]
            P("  k", k, "Q", Q, "...")
            lo, up = rsb.find_intervals(g, km)
            elo, eup = oix.find_intervals(km)
            P("  k", k, "Q", Q, "match", np.array_equal(lo, elo) and np.array_equal(up, eup))
            bad = np.nonzero((lo != elo) | (up != eup))[0]
            P("bad", bad.size, bad[:20])
            _, _, st = oix.find_intervals(km, want_steps=True)
            for i in bad[:6]:
                P(i, km[i].tobytes(), "got", int(lo[i]), int(up[i]), "exp", int(elo[i]), int(eup[i]), "steps", int(st[i]))
            P("steps hist", np.bincount(st)[:32])
    g.close()
