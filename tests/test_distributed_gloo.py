"""CPU suite: the N > 1 path (readserver_amd/sharded.py) with world_size 2 over gloo.

Each rank owns two of four suffix shards of a synthetic popBWT.  The per-shard search itself needs
the GPU, so here the oracle stands in for it (tests may use the oracle as checker/stand-in); what is
under test is the sharding, the broadcast of the batch, the gather of intervals and the reduction
of counts -- against the unsharded index."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_binding
        from readserver_amd import sharded
        orc = oracle_binding.load()
        S = 4
        mine = sharded.local_shards(rank, S, world)
        assert mine == [2 * rank, 2 * rank + 1]
        assert all(sharded.shard_owner(s, S, world) == rank for s in mine)
        idx = [orc.load(os.path.join(tmp, f"s{s}.bwt")) for s in mine]
        Q, k = 500, 31
        km = torch.zeros((Q, k), dtype=torch.uint8)
        if rank == 0:
            km = torch.from_numpy(np.load(os.path.join(tmp, "kmers.npy")))
        km = sharded.broadcast_queries(km)
        a = km.numpy()
        lo = np.empty((len(idx), Q), np.uint64)
        up = np.empty((len(idx), Q), np.uint64)
        for i, ix in enumerate(idx):
            lo[i], up[i] = ix.find_intervals(a)
        tl = torch.from_numpy(lo.view(np.int64))
        tu = torch.from_numpy(up.view(np.int64))
        g = sharded.gather_intervals(tl, tu)
        cnt = torch.from_numpy(np.where(up >= lo, up - lo + 1, 0).sum(0).astype(np.int64))
        tot = sharded.reduce_counts(cnt)
        if rank == 0:
            assert g.shape == (world, 2, 2, Q)
            whole = orc.load(os.path.join(tmp, "whole.bwt"))
            wlo, wup = whole.find_intervals(a)
            exp = np.where(wup >= wlo, wup - wlo + 1, 0)
            assert np.array_equal(tot.numpy().astype(np.uint64), exp.astype(np.uint64))
            # rank r's slice is shards 2r, 2r+1 in order
            for r in range(world):
                for j in range(2):
                    ix = orc.load(os.path.join(tmp, f"s{2 * r + j}.bwt"))
                    elo, eup = ix.find_intervals(a)
                    assert np.array_equal(g[r, 0, j].numpy().view(np.uint64), elo)
                    assert np.array_equal(g[r, 1, j].numpy().view(np.uint64), eup)
            q.put("ok")
        else:
            assert g is None and tot is None
    finally:
        dist.destroy_process_group()


def test_world2_gather_and_reduce(rsb, tmp_path):
    kw = dict(seed=17, genome_len=8000, haplotypes=4, snp_rate=0.004, read_len=50, coverage=3.0)
    for s in range(4):
        rsb.synth_popbwt(str(tmp_path / f"s{s}.bwt"), None, shard=s, num_shards=4, **kw)
    rd = str(tmp_path / "whole.reads")
    rsb.synth_popbwt(str(tmp_path / "whole.bwt"), rd, **kw)
    reads = open(rd).read().split()
    rng = np.random.default_rng(0)
    km = np.array([np.frombuffer(reads[i][j:j + 31].encode(), np.uint8)
                   for i, j in zip(rng.integers(0, len(reads), 500), rng.integers(0, 19, 500))])
    km[::5] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, km[::5].shape)]
    np.save(str(tmp_path / "kmers.npy"), km)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get() == "ok"


def _pipeline_worker(rank, world, port, q, interleaved=False, packed=False, out_depth=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from readserver_amd import sharded
        S, Q, steps = 2, 1000, 7
        g = sharded.IntervalGatherer(S, Q, torch.device("cpu"), depth=2, interleaved=interleaved, packed=packed, out_depth=out_depth)
        lo_of = (lambda t: t[..., 0]) if interleaved else (lambda t: t[0])  # {lower, upper} pairs / two arrays
        up_of = (lambda t: t[..., 1]) if interleaved else (lambda t: t[1])
        seen = {}
        for i in range(steps):
            buf = g.acquire(i)  # waits for batch i - 2's gather before the buffer is rewritten
            assert buf.shape == ((S, Q, 2) if interleaved else (2, S, Q))
            lo_of(buf)[:] = 1000 * i + 10 * rank + torch.arange(S * Q, dtype=torch.int64).reshape(S, Q)
            up_of(buf)[:] = lo_of(buf) + 7
            if packed:  # the corners the 10-byte form must carry: empty, invalid (1, 0), the (0, 2^64-1) wrap, 2^40-1
                up_of(buf)[0, :4] = lo_of(buf)[0, :4] - 1
                buf[1, 0] = torch.tensor([1, 0])
                buf[1, 1] = torch.tensor([0, -1])
                buf[1, 2] = torch.tensor([(1 << 40) - 1, (1 << 40) - 2])
                buf[1, 3] = torch.tensor([1, (1 << 40) - 1])
            if out_depth == 1 and rank == 0 and i >= 1:
                # one batch's blocks on rank 0: batch i's gather lands where batch i - 1's did, so batch i - 1 is read
                # before batch i is submitted (submit would wait for it too)
                g._work[(i - 1) % 2].wait()
                seen[i - 1] = [t.clone() for t in g.result(i - 1)]
            g.submit(i)
            if out_depth != 1 and rank == 0 and i >= 1:
                # batch i - 1 is complete once its handle has been waited for; acquire(i + 1) does
                # that for its buffer, so read it only after an explicit wait here
                g._work[(i - 1) % 2].wait()
                seen[i - 1] = [t.clone() for t in g.result(i - 1)]
        g.drain()
        if rank == 0 and packed:  # what arrived is the 10-byte form: 5/8 of the pairs' bytes
            assert all(t.dtype == torch.uint8 and t.numel() == sharded.packed_pairs_bytes(S * Q) for t in g.result(0))
            seen[steps - 1] = [t.clone() for t in g.result(steps - 1)]
            seen = {i: [g.unpack_block(t) for t in v] for i, v in seen.items()}
            for i in range(steps):
                for r in range(world):
                    b = seen[i][r]
                    assert torch.equal(b[1, 0], torch.tensor([1, 0])) and torch.equal(b[1, 1], torch.tensor([0, -1]))
                    assert torch.equal(b[1, 2], torch.tensor([(1 << 40) - 1, (1 << 40) - 2]))
                    assert torch.equal(b[1, 3], torch.tensor([1, (1 << 40) - 1]))
                    assert torch.equal(b[0, :4, 1], b[0, :4, 0] - 1)
                    assert torch.equal(b[0, 4:, 0], (1000 * i + 10 * r + torch.arange(S * Q, dtype=torch.int64).reshape(S, Q))[0, 4:])
                    assert torch.equal(b[0, 4:, 1], b[0, 4:, 0] + 7)
            q.put("ok")
        elif rank == 0:
            seen[steps - 1] = [t.clone() for t in g.result(steps - 1)]
            base = torch.arange(S * Q, dtype=torch.int64).reshape(S, Q)
            for i in range(steps):
                for r in range(world):
                    assert torch.equal(lo_of(seen[i][r]), 1000 * i + 10 * r + base), (i, r)
                    assert torch.equal(up_of(seen[i][r]), 1000 * i + 10 * r + base + 7), (i, r)
            q.put("ok")
        else:
            assert g.result(0) is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("interleaved,out_depth", [(False, None), (True, None), ("packed", None), ("packed", 1), (True, 1)])
def test_world2_pipelined_interval_gather(interleaved, out_depth):
    """bench.py's N > 1 data path: searches write into one of two resident buffers while the other one's
    gather to rank 0 is in flight ({lower, upper} pairs as rsbwt_set_find_interval_pairs_dev writes them,
    or two arrays).  out_depth = 1: rank 0 keeps one batch's gathered blocks (bench.py from 4 ranks on)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + 3 + [False, True, "packed"].index(interleaved) + 5 * (out_depth or 0)) % 2000
    packed = interleaved == "packed"
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q, bool(interleaved), packed, out_depth)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == "ok"


def test_single_process_gatherer_is_a_no_op():
    from readserver_amd import sharded
    g = sharded.IntervalGatherer(1, 10, torch.device("cpu"))
    b = g.acquire(0)
    b.fill_(3)
    g.submit(0)
    g.drain()
    assert g.result(0)[0] is g.pair(0) and int(g.pair(0).sum()) == 60


def test_pack_pairs_check_refuses_what_the_record_cannot_carry():
    """{lower:40, width:40} carries every interval findInterval leaves; an arbitrary pair may not fit, and
    check=True says so instead of truncating it on the wire."""
    import torch
    from readserver_amd import sharded
    ok = torch.tensor([[0, -1], [1, 0], [5, 9], [(1 << 40) - 1, (1 << 40) - 2], [7, 7 + (1 << 40) - 2]], dtype=torch.int64)
    assert torch.equal(sharded.unpack_pairs(sharded.pack_pairs(ok, check=True), ok.shape[0]), ok)
    for bad in ([[1 << 40, (1 << 40) + 3]], [[0, 1 << 40]]):
        t = torch.tensor(bad, dtype=torch.int64)
        with pytest.raises(ValueError):
            sharded.pack_pairs(t, check=True)
        sharded.pack_pairs(t)  # unchecked: truncates, as documented


def _concat_worker(rank, world, port, q, depth=2):
    """configs[3] / configs[4] at N > 1: what every rank's shards give is gathered on rank 0 (fixed-size blocks,
    pipelined like the interval gather) and laid side by side in global shard order."""
    import torch
    import torch.distributed as dist
    from readserver_amd import sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, cap, n, stride = 3, 50, 7, 16
        gh = sharded.BlockGatherer((S, cap, 4), torch.int64, torch.device("cpu"), depth=depth)
        gt = sharded.BlockGatherer((S,), torch.int64, torch.device("cpu"), depth=depth)
        go = sharded.BlockGatherer((S, n, stride), torch.uint8, torch.device("cpu"), depth=depth)
        gl = sharded.BlockGatherer((S, n), torch.int32, torch.device("cpu"), depth=depth)
        ok = True
        for i in range(5):  # more batches than buffers: acquire waits for the gather that used the buffer
            hb, tb, ob, lb = gh.acquire(i), gt.acquire(i), go.acquire(i), gl.acquire(i)
            hb.zero_()
            for s in range(S):
                cnt = (7 * rank + 3 * s + i) % cap
                tb[s] = cnt
                # record j of global shard g in batch i: {lower, upper, index, 0}
                g = rank * S + s
                hb[s, :cnt, 0] = 1000 * g + torch.arange(cnt) + i
                hb[s, :cnt, 1] = hb[s, :cnt, 0] + 5
                hb[s, :cnt, 2] = torch.arange(cnt)
            ob.copy_(((torch.arange(S * n * stride).reshape(S, n, stride) + 31 * rank + i) % 251).to(torch.uint8))
            lb.copy_(((torch.arange(S * n).reshape(S, n) + rank + i) % stride).to(torch.int32))
            for gg in (gh, gt, go, gl):
                gg.submit(i)
            if i >= 1 and rank == 0:  # batch i - 1 has been waited for by nobody yet: drain, then look at it
                pass
        for gg in (gh, gt, go, gl):
            gg.drain()
        if rank == 0:
            i = 4
            rec, first = sharded.concat_hit_lists(gh.result(i), gt.result(i))
            assert first.numel() == world * S + 1 and int(first[-1]) == rec.shape[0]
            for r in range(world):
                for s in range(S):
                    g = r * S + s
                    cnt = (7 * r + 3 * s + i) % cap
                    blk = rec[int(first[g]):int(first[g + 1])]
                    ok &= blk.shape[0] == cnt and bool((blk[:, 0] == 1000 * g + torch.arange(cnt) + i).all())
                    ok &= bool((blk[:, 1] == blk[:, 0] + 5).all()) and bool((blk[:, 2] == torch.arange(cnt)).all())
            reads, lens = sharded.concat_reads(go.result(i), gl.result(i))
            ok &= tuple(reads.shape) == (world * S, n, stride) and tuple(lens.shape) == (world * S, n)
            for r in range(world):
                want = ((torch.arange(S * n * stride).reshape(S, n, stride) + 31 * r + i) % 251).to(torch.uint8)
                ok &= bool(torch.equal(reads[r * S:(r + 1) * S], want))
                ok &= bool(torch.equal(lens[r * S:(r + 1) * S], ((torch.arange(S * n).reshape(S, n) + r + i) % stride).to(torch.int32)))
            try:
                bad = [t.clone() for t in gt.result(i)]
                bad[1][0] = cap + 1
                sharded.concat_hit_lists(gh.result(i), bad)
                ok = False
            except ValueError:
                pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_world2_hit_lists_and_reads_are_concatenated_in_shard_order():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + 11) % 2000
    procs = [ctx.Process(target=_concat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_world3_hit_lists_and_reads_with_one_gather_in_flight():
    """bench.py's rows modes from 3 ranks on: ONE batch's blocks per rank (BlockGatherer depth 1: a batch is written once
    the gather of the one before it is complete), three ranks, global shard order."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + 911) % 2000
    procs = [ctx.Process(target=_concat_worker, args=(r, 3, port, q, 1)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True), (2, True)]


def test_read_packing_round_trips_on_the_host():
    """2 bits per base for the wire (what a rank sends of an extraction batch): pack -> unpack gives the ASCII bytes up
    to each read's length, NUL beyond; a read marked as not fitting (length -1 = UINT32_MAX) travels as nothing."""
    import torch
    from readserver_amd import sharded
    g = torch.Generator().manual_seed(3)
    n, stride = 500, 64
    lens = torch.randint(0, stride + 1, (n,), generator=g, dtype=torch.int32)
    lens[7] = -1
    lens[8] = 0
    lens[9] = stride
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8)
    reads = lut[torch.randint(0, 4, (n, stride), generator=g)]
    reads[torch.arange(stride)[None, :] >= lens.clamp(min=0)[:, None].long()] = 0x7E  # garbage past the end must not travel
    packed = sharded.pack_reads(reads, lens)
    assert packed.shape == (n, stride // 4) and packed.dtype == torch.uint8
    back = sharded.unpack_reads(packed, lens)
    want = torch.where(torch.arange(stride)[None, :] < lens.clamp(min=0)[:, None].long(), reads, torch.zeros_like(reads))
    assert torch.equal(back, want)
    assert not packed[7].any() and not packed[8].any()
    # shapes with leading dimensions ([S][n][stride], as bench.py holds them)
    p3 = sharded.pack_reads(reads.reshape(5, 100, stride), lens.reshape(5, 100))
    assert p3.shape == (5, 100, stride // 4) and torch.equal(p3.reshape(n, -1), packed)


def test_world3_pipelined_interval_gather_with_one_batch_of_blocks_on_the_root():
    """The same pipeline with three ranks and out_depth = 1 (bench.py's choice from 4 ranks on: rank 0 keeps ONE batch's
    gathered blocks): every rank's packed pairs of every batch arrive, rank by rank, in the blocks the batch before used."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + 1777) % 2000
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 3, port, q, True, True, 1)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == "ok"
