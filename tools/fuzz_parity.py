#!/usr/bin/env python3
"""Randomised differential campaign on the GPU box (not part of the suite: it runs for as long as it is
told to): random run streams x layouts (window span, k-mer table depth) x query lengths, the HIP path
against the oracle -- intervals, counts, the 1-mismatch hit list against the dense matrices, getOccAt, read
extraction row by row, and (two-shard sets) the set-level hit lists and query lists.  Test infrastructure: the oracle is the checker, as in tests/.
Prints a progress line per configuration and one JSON line at the end; exit code 1 on any difference.
usage: tools/fuzz_parity.py [seconds=300] [seed=1]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding  # noqa: E402
import readserver_amd as rsb  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = rsb.lib()
orc = oracle_binding.load()
rng = np.random.default_rng(SEED)
acgt = np.frombuffer(b"ACGT", np.uint8)
t_end = time.time() + SECONDS
done, failures, sets, fused_sets, two_group_sets = 0, [], 0, 0, 0


def make_runs(R, shape):
    sym = rng.integers(0, 5, R)
    ln = rng.integers(1, 32, R)
    if shape == 1:
        ln[:] = 31
    elif shape == 2:
        ln = rng.integers(1, 3, R)
    elif shape == 3:
        sym[rng.random(R) < 0.3] = 0
    elif shape == 4:
        sym = np.where(rng.random(R) < 0.995, 1 + (np.arange(R) // 5000) % 4, sym)
    elif shape == 5:  # the library's own generators
        runs = np.empty(R, np.uint8)
        style = [0, 1 << 63, 1 << 62][int(rng.integers(0, 3))]  # mixed, long-run, population stream
        assert L.rsbwt_synth_runs_host(runs.ctypes.data, R, int(rng.integers(1, 1 << 30)) | style) == 0
        return runs
    return ((sym << 5) | ln).astype(np.uint8)


while time.time() < t_end:
    R = int(2 ** rng.uniform(0, 21.5))
    shape = int(rng.integers(0, 6))
    runs = make_runs(R, shape)
    span = 0 if rng.random() < 0.4 else int(rng.integers(2, 2945))
    T = [None, 0, 0, int(rng.integers(2, 13))][int(rng.integers(0, 4))]
    k = int(rng.integers(1, 71)) if rng.random() < 0.5 else 31
    oix = orc.from_runs(runs)
    n = oix.bwlen()
    reads_layout = bool(rng.random() < 0.4)  # RSBWT_OPEN_READS: a psi hint slot in every window line, sparse select samples
    grouped = bool(rng.random() < 0.5)       # RSBWT_OPEN_KTAB_GROUPED: the k-mer table's 12-byte records of four siblings (any depth: groups too
                                             # wide for their record and T-mers that do not occur are left to the search)
    cfg = {"runs": R, "shape": shape, "span": span, "T": T, "k": k, "n": int(n), "for_reads": reads_layout, "ktab_grouped": grouped}
    try:
        with rsb.GpuBWT(runs=runs, ktab_depth=T, window_span=span, for_reads=reads_layout, ktab_grouped=grouped) as g:
            Q = 300000 if rng.random() < 0.1 else 20000  # the larger batch takes the one-lane-per-search kernel on a deep table
            cfg["Q"] = Q
            cfg["ktab"] = (g.ktab_depth(),) + g.ktab_info()  # (depth, format, bytes, T-mers left to the search)
            km = acgt[rng.integers(0, 4, (Q, k))].copy()
            # k-mers that occur: spelled from the BWT's own rows by the oracle's extraction
            rows = rng.integers(0, n, 400, dtype=np.uint64)
            texts = []
            for r in rows[:200]:
                try:
                    pre, post = oix.extract(int(r), cap=2000)
                    texts.append((pre + post).encode())
                except AssertionError:  # a walk longer than the oracle's buffer (streams with hardly any '$')
                    texts.append(None)
            j = 0
            for tx in texts:
                if tx is None:
                    continue
                for a in range(0, max(0, len(tx) - k + 1), max(1, k // 2)):
                    if j < Q // 2:
                        km[2 * j] = np.frombuffer(tx[a:a + k], np.uint8)
                        j += 1
            elo, eup = oix.find_intervals(km, nthreads=8)
            lo, up = rsb.find_intervals(g, km)
            assert np.array_equal(lo, elo) and np.array_equal(up, eup), "intervals"
            assert np.array_equal(rsb.count_kmers(g, km), np.where(eup >= elo, eup - elo + 1, 0).astype(np.uint64)), "counts"
            if rng.random() < 0.15:  # a shard set: this shard next to a second one, one fused launch
                runs2 = make_runs(int(2 ** rng.uniform(0, 19)), int(rng.integers(0, 6)))
                oix2 = orc.from_runs(runs2)
                # (a second shard with the first one's table depth: the set then searches both in one traced and one
                # resumed launch, csrc/sets.hip set_hits_1mm_fused)
                # now and then the second shard sits on another LOGICAL device (the library's test hook, include/rsbwt.h,
                # rsbwt_logical_device): the set then spans two device groups -- a thread and a fused launch per group, the
                # groups' results merged on the host -- and only its host entry points apply
                two_dev = bool(rng.random() < 0.3)
                if two_dev:
                    os.environ["RSBWT_ENABLE_TEST_HOOKS"], os.environ["RSBWT_TEST_DEVICE_ALIASES"] = "1", "2"
                g2_depth, g2_reads, g2_grouped = [None, 0, 5, T, T][int(rng.integers(0, 5))], bool(rng.random() < 0.5), bool(rng.random() < 0.5)
                cfg["second_shard"] = {"runs": int(runs2.size), "ktab_depth": g2_depth, "for_reads": g2_reads, "ktab_grouped": g2_grouped}
                try:
                    g2cm = rsb.GpuBWT(runs=runs2, device=1 if two_dev else 0, ktab_depth=g2_depth, for_reads=g2_reads, ktab_grouped=g2_grouped)
                finally:
                    if two_dev:
                        del os.environ["RSBWT_TEST_DEVICE_ALIASES"]
                with g2cm as g2:
                    ss = rsb.ShardSet([g, g2])
                    assert L.rsbwt_set_devices(ss._s) == (2 if two_dev else 1), "device groups of the set"
                    cfg["set_devices"] = 2 if two_dev else 1
                    if two_dev:  # the summed counts of the two groups (host-side sum: RCCL refuses two ranks on one GPU)
                        e2c = oix2.find_intervals(km[:5000], nthreads=8)
                        wantc = np.where(eup[:5000] >= elo[:5000], eup[:5000] - elo[:5000] + 1, 0) + np.where(e2c[1] >= e2c[0], e2c[1] - e2c[0] + 1, 0)
                        assert np.array_equal(ss.count(km[:5000]), wantc.astype(np.uint64)), "set counts over two device groups"
                    slo, sup = ss.find_intervals(km[:5000])
                    e2lo, e2up = oix2.find_intervals(km[:5000], nthreads=8)
                    ok2 = (np.array_equal(slo[0], elo[:5000]) and np.array_equal(sup[0], eup[:5000]) and
                           np.array_equal(slo[1], e2lo) and np.array_equal(sup[1], e2up))
                    # queries of lengths of their own in one search (rsbwt_set_find_intervals_var): prefixes of the k-mers
                    vq = [bytes(km[i, :int(rng.integers(1, k + 1))]) for i in range(0, 600)]
                    vlo, vup = ss.find_intervals_var(vq)
                    by_l = {}
                    for i, w_ in enumerate(vq):
                        by_l.setdefault(len(w_), []).append(i)
                    ok_v = True
                    for L_, idx in by_l.items():
                        arr = np.frombuffer(b"".join(vq[i] for i in idx), np.uint8).reshape(len(idx), L_)
                        for si_, ox_ in enumerate((oix, oix2)):
                            a_lo, a_up = ox_.find_intervals(arr, nthreads=8)
                            ok_v = ok_v and np.array_equal(vlo[si_, idx], a_lo) and np.array_equal(vup[si_, idx], a_up)
                    ok2 = ok2 and ok_v
                    if Q > 20000:  # the whole batch over the set: both shards behind deep tables = the headline's launch (one lane per search)
                        flo, fup = ss.find_intervals(km)
                        f2lo, f2up = oix2.find_intervals(km, nthreads=8)
                        ok2 = ok2 and (np.array_equal(flo[0], elo) and np.array_equal(fup[0], eup) and np.array_equal(flo[1], f2lo) and np.array_equal(fup[1], f2up))
                        cfg["set_full_batch"] = True
                    why = [] if ok2 else ["set intervals" if ok_v else "set find_intervals_var (mixed lengths)"]
                    # configs[3] / configs[4] over the set: every shard's own list / reads, side by side
                    if k <= 40:
                        sh, first = ss.hits_1mm(km[:60])
                        ok_h = np.array_equal(sh[int(first[0]):int(first[1])], rsb.hits_1mm_batch(g, km[:60])) and np.array_equal(sh[int(first[1]):int(first[2])], rsb.hits_1mm_batch(g2, km[:60]))
                        if not ok_h:
                            why.append("set hits_1mm (host) vs the shards' own lists")
                        ok2 = ok2 and ok_h
                    if k <= 40 and not two_dev:
                        # the device-resident form (fused launches when both tables have one depth) leaves the same lists
                        import torch
                        p_ = lambda t: C.c_void_p(t.data_ptr())
                        d_km = torch.from_numpy(km[:60].copy()).cuda()
                        d_pk = torch.empty((60, (k + 31) // 32), dtype=torch.int64, device="cuda")
                        d_ok = torch.empty(60, dtype=torch.uint8, device="cuda")
                        cap1 = max(1, len(sh))
                        d_h = torch.zeros((2, cap1, 4), dtype=torch.int64, device="cuda")
                        d_t = torch.zeros(2, dtype=torch.int64, device="cuda")
                        d_s = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(ss._s, 60, k), dtype=torch.uint8, device="cuda")
                        assert L.rsbwt_pack_kmers_dev(p_(d_km), 60, k, k, p_(d_pk), p_(d_ok), 0, None) == 0
                        assert L.rsbwt_set_hits_1mm_dev(ss._s, p_(d_pk), p_(d_ok), 60, k, p_(d_h), cap1, p_(d_t), p_(d_s), None) == 0
                        torch.cuda.synchronize()
                        cfg["set_1mm_fused"] = int(L.rsbwt_set_hits_1mm_is_fused(ss._s, 60, k))
                        for si in range(2):
                            mine = sh[int(first[si]):int(first[si + 1])]
                            rec = d_h[si, :int(d_t[si].item())].cpu().numpy().view(np.uint64)
                            ok_d = rec.shape[0] == len(mine) and np.array_equal(rec[:, 0], mine["lower"]) and np.array_equal(rec[:, 1], mine["upper"])
                            ok_d = ok_d and np.array_equal(rec[:, 2] // (3 * k + 1), mine["query"].astype(np.uint64))
                            if not ok_d:
                                why.append(f"set hits_1mm_dev shard {si}: {rec.shape[0]} records against {len(mine)}")
                                # what the same call leaves when made again: as it was, on zeroed scratch, on scratch full of
                                # 0xA5 (a dependence on what the caller's scratch held shows as a difference between these)
                                again = {}
                                for name_, fill_ in (("again", None), ("zeroed", 0), ("a5", 0xA5)):
                                    if fill_ is not None:
                                        d_s.fill_(fill_)
                                    d_t.zero_()
                                    assert L.rsbwt_set_hits_1mm_dev(ss._s, p_(d_pk), p_(d_ok), 60, k, p_(d_h), cap1, p_(d_t), p_(d_s), None) == 0
                                    torch.cuda.synchronize()
                                    again[name_] = [int(x) for x in d_t.cpu().numpy()]
                                why.append(f"totals when called again: {again}")
                                os.makedirs("gpurun_out", exist_ok=True)
                                np.savez_compressed(f"gpurun_out/fuzz_fail_seed{SEED}_{done}.npz", runs=runs, runs2=runs2, km=km[:60], k=k,
                                                    views=np.array([L.rsbwt_ktab_depth(g.handle), L.rsbwt_ktab_depth(g2.handle)]),
                                                    want=np.array([len(sh[int(first[i_]):int(first[i_ + 1])]) for i_ in range(2)]))
                            ok2 = ok2 and ok_d
                    # read extraction over the set, device-resident: ONE launch sequence walks the rows of both shards
                    # (csrc/extract_lines.hip), against each shard's own host call
                    import torch
                    p_ = lambda t: C.c_void_p(t.data_ptr())
                    nr_ = 0 if two_dev else 300
                    n2 = oix2.bwlen()
                    rws = np.stack([rng.integers(0, n + 3, nr_), rng.integers(0, n2 + 3, nr_)]).astype(np.uint64)  # (a few past the end)
                    d_rw = torch.from_numpy(rws.view(np.int64)).cuda()
                    d_o = torch.zeros((2, nr_, 512), dtype=torch.uint8, device="cuda")
                    d_l = torch.empty((2, nr_), dtype=torch.int32, device="cuda")
                    d_p = torch.empty((2, nr_), dtype=torch.int32, device="cuda")
                    if nr_:
                        assert L.rsbwt_set_extract_dev(ss._s, p_(d_rw), nr_, p_(d_o), 512, p_(d_l), p_(d_p), None) == 0
                    else:  # two device groups: the host form, rows addressed as (shard, row), against the oracle
                        shx = rng.integers(0, 2, 200).astype(np.uint32)
                        rwx = np.array([rng.integers(0, (n, n2)[int(s_)]) for s_ in shx], dtype=np.uint64)
                        try:  # (streams with hardly any '$' have reads longer than any buffer: the call then says so)
                            got_x = ss.extract(shx, rwx, stride=2048)[0]
                        except rsb.RsbwtError:
                            got_x = []
                        for s_, r_, t_ in list(zip(shx, rwx, got_x))[::7]:
                            try:
                                pre_, post_ = (oix, oix2)[int(s_)].extract(int(r_), cap=2000)
                            except AssertionError:
                                continue
                            if len(pre_) + len(post_) <= 2048 and t_ != pre_ + post_:
                                why.append("set extract over two device groups")
                                ok2 = False
                    torch.cuda.synchronize()
                    for si, gg in enumerate((g, g2) if nr_ else ()):
                        o1 = np.zeros((nr_, 512), np.uint8)
                        l1, p1 = np.empty(nr_, np.uint32), np.empty(nr_, np.uint32)
                        assert L.rsbwt_extract(gg.handle, rws[si].ctypes.data, nr_, o1.ctypes.data, 512, l1.ctypes.data, p1.ctypes.data) == 0
                        gl, gp, go = d_l[si].cpu().numpy().view(np.uint32), d_p[si].cpu().numpy().view(np.uint32), d_o[si].cpu().numpy()
                        ok_x = np.array_equal(gl, l1) and np.array_equal(gp[l1 != 0xFFFFFFFF], p1[l1 != 0xFFFFFFFF])
                        ok_x = ok_x and all(np.array_equal(go[i, :l1[i]], o1[i, :l1[i]]) for i in range(nr_) if l1[i] != 0xFFFFFFFF)
                        if not ok_x:
                            why.append(f"set extract_dev shard {si}")
                        ok2 = ok2 and ok_x
                    # query() in every shard, lists concatenated per k-mer -- for k-mers whose intervals are narrow in
                    # both shards (every row of an interval is extracted into a 2 KB buffer: a 1-mer's would be gigabytes)
                    w1 = np.where(eup[:5000] >= elo[:5000], eup[:5000] - elo[:5000] + 1, 0)
                    w2 = np.where(e2up >= e2lo, e2up - e2lo + 1, 0)
                    narrow = np.nonzero((w1 <= 32) & (w2 <= 32))[0][:150]
                    try:  # (streams with hardly any '$' have reads longer than any buffer)
                        qs = ss.query(km[narrow], read_stride=2048) if narrow.size else None
                    except rsb.RsbwtError:
                        qs = None
                    if qs is not None:
                        for qi, lst in zip(narrow, qs):
                            exp = []
                            for si, (ox, lo_, up_) in enumerate(((oix, elo, eup), (oix2, e2lo, e2up))):
                                if exp is not None and up_[qi] >= lo_[qi]:
                                    try:
                                        exp += [(si, "".join(ox.extract(int(r), cap=2000))) for r in range(int(lo_[qi]), int(up_[qi]) + 1)]
                                    except AssertionError:
                                        exp = None
                            if exp is not None and all(len(t) <= 2048 for _, t in exp):
                                if lst != exp:
                                    why.append("set query lists")
                                ok2 = ok2 and lst == exp
                    ss.close()
                oix2.close()
                assert ok2, "shard set: " + "; ".join(sorted(set(why)))
                cfg["set"] = True
            if k <= 40:
                # (now and then enough variant searches for the resumed launch to take the one-lane-per-search kernel)
                m = 3000 if (Q >= 3000 and rng.random() < 0.08) else 150
                cfg["m_1mm"] = m
                dlo, dup = rsb.find_intervals_1mm(g, km[:m])
                want = []
                for qi in range(m):
                    want += [(qi,) + h for h in rsb.hits_1mm(km[qi].tobytes().decode(), dlo[qi], dup[qi])]
                hl = rsb.hits_1mm_batch(g, km[:m])
                got = [(int(r["query"]), int(r["pos"]), r["base"].decode(), int(r["lower"]), int(r["upper"])) for r in hl]
                assert got == want, "1-mismatch hit list"
                vlo, vup = oix.find_intervals(np.array([list(km[0])], np.uint8))
                assert (int(dlo[0, 0]), int(dup[0, 0])) == (int(vlo[0]), int(vup[0])), "1-mismatch column 0"
            # getOccAt (select) at random occurrences of every symbol: the select samples name the window
            for ch in "ACGT$":
                tot = oix.occ(ch, n - 1)
                if tot:
                    bcs = rng.integers(1, tot + 1, 300).astype(np.uint64)
                    gi = g.occ_at_batch(ch, bcs)
                    assert all(int(gi[i]) == oix.occ_at(ch, int(bcs[i])) for i in range(0, 300, 7)), "getOccAt"
                    assert np.array_equal(g.occ_batch(ch, gi), bcs), "getOcc(getOccAt)"
            stride = 2048
            out = np.zeros((rows.size, stride), np.uint8)
            ln = np.empty(rows.size, np.uint32)
            pl = np.empty(rows.size, np.uint32)
            assert L.rsbwt_extract(g.handle, rows.ctypes.data, rows.size, out.ctypes.data, stride, ln.ctypes.data, pl.ctypes.data) == 0
            for i in range(200):
                tx = texts[i]
                if tx is not None and len(tx) <= stride:
                    assert ln[i] != 0xFFFFFFFF and out[i, :ln[i]].tobytes() == tx, f"extraction of row {int(rows[i])}"
        done += 1
        sets += 1 if cfg.get("set") else 0
        two_group_sets += 1 if cfg.get("set_devices") == 2 else 0
        fused_sets += cfg.get("set_1mm_fused", 0)
        print(f"ok {done}: {cfg}", file=sys.stderr, flush=True)
    except AssertionError as e:
        failures.append({"config": cfg, "what": str(e)})
        print(f"FAILED: {cfg}: {e}", file=sys.stderr, flush=True)
        if len(failures) >= 5:
            break
    oix.close()
print(json.dumps({"seconds": SECONDS, "seed": SEED, "search_kernel": os.environ.get("RSBWT_SEARCH_KERNEL", "auto"), "configurations": done, "as_two_shard_sets": sets, "sets_over_two_device_groups": two_group_sets,
                  "sets_searched_by_the_fused_1mm_launches": fused_sets, "failures": failures}))
sys.exit(1 if failures else 0)
