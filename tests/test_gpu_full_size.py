"""GPU suite (-m gpu): BASELINE configs[2]'s per-GPU load at FULL size -- 8 shards of 2e10 run bytes resident in one
MI355X -- held to the oracle and to itself, in the three modes bench.py times ("missing" #3).

The 8 shards are ONE stream (the bench's population mix), so every shard must answer every question exactly as
shard 0 does -- a free check of every shard's lines, tables, samples and hints, bit for bit -- and shard 0 is held to
the oracle (oracle/rlebwt_oracle.c over the same 2e10 run bytes, built on the host while the GPU builds its shards).
Needs a whole MI355X (skipped below 280 GB of free HBM) and ~25 GB of host memory; about two minutes."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 20_000_000_000
S = 8
SEED = (1 << 62) | 7000077  # the population stream (csrc/synth_runs.h)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _spelled(km):
    m, k = km.shape
    out = np.repeat(km[:, None, :], 3 * k + 1, axis=1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for pos in range(k):
        for q in range(m):
            out[q, 1 + 3 * pos:4 + 3 * pos, pos] = [c for c in acgt if c != km[q, pos]][:3]
    return out


def _build(rsb, torch, L, for_reads, keep_host):
    shards, host = [], None
    for s in range(S):
        d_runs = torch.empty(R, dtype=torch.uint8, device="cuda:0")
        assert L.rsbwt_synth_runs_dev(_p(d_runs), R, SEED, 0, None) == 0
        torch.cuda.synchronize()
        if s == 0 and keep_host:
            host = d_runs.cpu().numpy()
        shards.append(rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), ktab_depth=None, for_reads=for_reads))
        del d_runs
        torch.cuda.empty_cache()
    return shards, host


def test_gpu_eight_20gb_shards_exact_1mm_extract_at_full_size(rsb, oracle):
    import torch
    L = rsb.lib()
    if torch.cuda.mem_get_info(0)[0] < 280e9:
        pytest.skip("needs the 288 GB of an MI355X")
    k = 31
    # ---------------- the plain layout with the bench's tables: exact search, 1-mismatch, class-BWT properties
    shards, host_runs = _build(rsb, torch, L, False, True)
    box = {}
    th = threading.Thread(target=lambda: box.setdefault("oix", oracle.from_runs(host_runs)))  # (~25 s, beside the GPU's work)
    th.start()
    sset = rsb.ShardSet(shards)
    # (the bench's tables: grouped 15-mer tables, 3.2 GB per shard in the HBM plain 14-mer tables would take 2.1 of)
    assert L.rsbwt_set_attach_ktabs_format(sset._s, 15, 1) == 0
    assert all(g.ktab_depth() == 15 and g.ktab_info()[:2] == (1, 3 * 4 ** 15) and g.ktab_info()[2] < 1e-3 * 4 ** 15 for g in shards)
    n = int(shards[0].getBWLen())
    assert n > (1 << 36) and all(int(g.getBWLen()) == n and g.window_span() == shards[0].window_span() for g in shards)
    Q = 1_000_000
    g = torch.Generator(device="cuda:0").manual_seed(5)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda:0")
    d_km = lut[torch.randint(0, 4, (Q, k), generator=g, device="cuda:0", dtype=torch.uint8).long()]
    d_pres = torch.empty((Q // 2, k), dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_sample_present_kmers_dev(shards[0].handle, Q // 2, k, k, 99, _p(d_pres), None) == 0
    torch.cuda.synchronize()
    d_km[::2] = d_pres
    d_pk = torch.empty((Q, 1), dtype=torch.int64, device="cuda:0")
    d_ok = torch.empty(Q, dtype=torch.uint8, device="cuda:0")
    d_pr = torch.empty((S, Q, 2), dtype=torch.int64, device="cuda:0")
    assert L.rsbwt_pack_kmers_dev(_p(d_km), Q, k, k, _p(d_pk), _p(d_ok), 0, None) == 0
    assert L.rsbwt_set_find_interval_pairs_dev(sset._s, _p(d_pk), _p(d_ok), Q, k, _p(d_pr), None) == 0
    torch.cuda.synchronize()
    for s in range(1, S):
        assert torch.equal(d_pr[s], d_pr[0]), f"shard {s} answers differently from shard 0 (same run bytes)"
    pr0 = d_pr[0].cpu().numpy().view(np.uint64)
    assert (pr0[::2, 1] >= pr0[::2, 0]).all(), "a 31-mer drawn from the index must occur"
    # class BWT on a shard other than 0, at positions past 2^36: the five Occ add up to p + 1, select undoes rank
    rng = np.random.default_rng(3)
    pos = rng.integers(1 << 36, n, 200_000).astype(np.uint64)
    pos[:3] = [n - 1, n - 2, (1 << 36) + 1]
    gx = shards[5]
    tot = np.zeros(pos.size, np.uint64)
    for ch in "$ACGT":
        tot += gx.occ_batch(ch, pos)
    assert np.array_equal(tot, pos + 1)
    ch_at = gx.char_batch(pos)
    for ch in "ACGT":
        sel = pos[ch_at == ord(ch)][:20000]
        assert sel.size and np.array_equal(gx.occ_at_batch(ch, gx.occ_batch(ch, sel)), sel)
    # 1-mismatch hit lists (configs[3]): 20,000 of the k-mers, every shard's ordered list
    m = 20_000
    cap = 8 * m
    d_hits = torch.zeros((S, cap, 4), dtype=torch.int64, device="cuda:0")
    d_tot = torch.zeros(S, dtype=torch.int64, device="cuda:0")
    d_scr = torch.empty(L.rsbwt_set_hits_1mm_scratch_bytes(sset._s, m, k), dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_set_hits_1mm_dev(sset._s, _p(d_pk), _p(d_ok), m, k, _p(d_hits), cap, _p(d_tot), _p(d_scr), None) == 0
    torch.cuda.synchronize()
    totals = d_tot.cpu().numpy()
    assert (totals == totals[0]).all() and 0 < totals[0] <= cap
    for s in range(1, S):
        assert torch.equal(d_hits[s, :totals[0]], d_hits[0, :totals[0]]), f"shard {s}'s hit list differs from shard 0's"
    hits0 = d_hits[0, :totals[0]].cpu().numpy().view(np.uint64)
    km_host = d_km[:m].cpu().numpy()
    # ---------------- the oracle on shard 0: 100,000 of the k-mers, the hit list of 150 k-mers, reads below
    th.join()
    oix = box["oix"]
    assert oix.bwlen() == n
    elo, eup = oix.find_intervals(d_km[::10].cpu().numpy(), nthreads=16)
    assert np.array_equal(pr0[::10, 0], elo) and np.array_equal(pr0[::10, 1], eup), "GPU intervals differ from the oracle at full size"
    V = 3 * k + 1
    sp = _spelled(km_host[:150]).reshape(-1, k)
    vlo, vup = oix.find_intervals(sp, nthreads=16)
    want = [(int(vlo[i]), int(vup[i]), i) for i in range(sp.shape[0]) if vup[i] >= vlo[i] and vup[i] < n]
    got = [(int(h[0]), int(h[1]), int(h[2])) for h in hits0 if h[2] < 150 * V]
    assert got == want, "1-mismatch hit list differs from the oracle's exact search of every variant"
    sset.close()
    for gq in shards:
        gq.close()
    del d_pr, d_hits, d_scr, shards, sset
    torch.cuda.empty_cache()
    # ---------------- the layout for reads: locate + extract (configs[4]), the same rows of every shard
    shards, _ = _build(rsb, torch, L, True, False)
    sset = rsb.ShardSet(shards)
    assert all(L.rsbwt_opened_for_reads(gq.handle) == 1 and L.rsbwt_psi_hint_lines(gq.handle) > 0 for gq in shards)
    NR, stride = 400_000, 256
    starts = torch.randint(0, n - 8, (NR // 8,), generator=g, device="cuda:0", dtype=torch.int64)
    rows1 = (starts[:, None] + torch.arange(8, device="cuda:0")[None, :]).reshape(-1)
    rows1[:4] = torch.tensor([0, 1, n - 1, n - 2], device="cuda:0")
    rows = rows1[None, :].repeat(S, 1).contiguous()
    d_out = torch.zeros((S, NR, stride), dtype=torch.uint8, device="cuda:0")
    d_len = torch.empty((S, NR), dtype=torch.int32, device="cuda:0")
    d_pl = torch.empty((S, NR), dtype=torch.int32, device="cuda:0")
    assert L.rsbwt_set_extract_dev(sset._s, _p(rows), NR, _p(d_out), stride, _p(d_len), _p(d_pl), None) == 0
    torch.cuda.synchronize()
    ln = d_len[0].long().clamp(min=0)
    keep = torch.arange(stride, device="cuda:0")[None, :] < ln[:, None]  # (bytes past a read's end are not part of the answer)
    for s in range(1, S):
        assert torch.equal(d_len[s], d_len[0]) and torch.equal(d_pl[s], d_pl[0])
        assert torch.equal(d_out[s] * keep, d_out[0] * keep), f"shard {s}'s reads differ from shard 0's"
    out0, len0, pl0, r0 = d_out[0].cpu().numpy(), d_len[0].cpu().numpy().view(np.uint32), d_pl[0].cpu().numpy().view(np.uint32), rows1.cpu().numpy()
    fits = 0
    for i in list(range(0, 64)) + list(range(64, NR, NR // 1500)):
        pre, post = oix.extract(int(r0[i]), cap=4096)
        if len(pre) + len(post) <= stride:
            fits += 1
            assert len0[i] == len(pre) + len(post) and pl0[i] == len(pre) and out0[i, :len0[i]].tobytes().decode() == pre + post, i
        else:
            assert len0[i] == 0xFFFFFFFF, i
    assert fits > 500
    sset.close()
    for gq in shards:
        gq.close()
