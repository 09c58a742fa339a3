"""f1, the service slice: the proto2 codec against the Python protobuf runtime (CPU), and the
batched count_reads against the oracle (GPU)."""
import ctypes as C

import numpy as np
import pytest

import proto_schema


@pytest.fixture(scope="module")
def pb():
    return proto_schema.build()


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def test_codec_matches_protobuf_runtime(rsb, pb):
    Request, Reply = pb
    L = rsb.lib()
    rng = np.random.default_rng(0)
    for _ in range(300):
        r = Request()
        r.t = int(rng.integers(1, 5))
        r.rt = int(rng.integers(1, 5))
        r.q = "".join("ACGTN"[x] for x in rng.integers(0, 5, int(rng.integers(0, 320))))
        if rng.random() < 0.5:
            r.k, r.s, r.p = 31, int(rng.integers(0, 9)), int(rng.integers(-5, 500))
            r.a, r.isalt = "ACGT", 1
        wire = r.SerializeToString()
        t, rt, q, ql = C.c_int(), C.c_int(), C.c_char_p(), C.c_size_t()
        buf = (C.c_uint8 * len(wire)).from_buffer_copy(wire)
        assert L.rsbwt_proto_decode_request(buf, len(wire), C.byref(t), C.byref(rt), C.byref(q), C.byref(ql)) == 0
        got_q = C.string_at(C.cast(q, C.c_void_p).value, ql.value).decode() if ql.value else ""
        assert (t.value, rt.value, got_q) == (r.t, r.rt, r.q)
        for revcomp in (0, 1):
            for c in (0, 1, 127, 128, 300000, 2 ** 31 - 1, -1, -(2 ** 31)):
                rep = Reply()
                rep.rt, rep.t, rep.q = r.t, 1, r.q
                (rep.c.revcomp_matches if revcomp else rep.c.forward_matches).c = c
                exp = rep.SerializeToString()
                out = (C.c_uint8 * (len(exp) + 16))()
                n = L.rsbwt_proto_encode_count_reply(out, len(out), r.t, r.q.encode(), len(r.q), revcomp, c)
                assert bytes(out[:n]) == exp
    # a message missing a required field, and a truncated one, are rejected
    bad = Request(); bad.t = 1; bad.rt = 1
    wire = bad.SerializePartialToString()
    buf = (C.c_uint8 * max(len(wire), 1)).from_buffer_copy(wire or b"\0")
    assert L.rsbwt_proto_decode_request(buf, len(wire), None, None, None, None) == -3
    ok = Request(); ok.t, ok.rt, ok.q = 1, 1, "ACGT"
    wire = ok.SerializeToString()[:-2]
    buf = (C.c_uint8 * len(wire)).from_buffer_copy(wire)
    assert L.rsbwt_proto_decode_request(buf, len(wire), None, None, None, None) == -3


@pytest.mark.gpu
def test_gpu_service_counts_batch(rsb, oracle, pb, tmp_path):
    Request, Reply = pb
    L = rsb.lib()
    kw = dict(seed=23, genome_len=20000, haplotypes=4, snp_rate=0.004, read_len=60, coverage=3.0)
    shards, oixs = [], []
    for s in range(2):
        p = str(tmp_path / f"s{s}.bwt")
        rsb.synth_popbwt(p, None, shard=s, num_shards=2, **kw)
        shards.append(rsb.GpuBWT(p))
        oixs.append(oracle.load(p))
    rd = str(tmp_path / "w.reads")
    rsb.synth_popbwt(str(tmp_path / "w.bwt"), rd, **kw)
    reads = open(rd).read().split()
    rng = np.random.default_rng(3)
    reqs = []
    for i in range(400):
        r = Request()
        kind = i % 5
        r.t, r.rt = (1, 1) if kind < 2 else (2, 1) if kind == 2 else (2, 2) if kind == 3 else (3, 1)
        rr = reads[rng.integers(len(reads))]
        k = int(rng.choice([20, 31, 31, 37, 60]))
        st = int(rng.integers(0, len(rr) - k + 1))
        q = rr[st:st + k]
        if i % 7 == 0:
            q = _rc(q)
        if i % 31 == 0:
            q = q[:5] + "N" + q[6:]
        if i == 13:
            q = ""
        r.q = q
        reqs.append(r)
    wires = [r.SerializeToString() for r in reqs]
    off = np.concatenate([[0], np.cumsum([len(w) for w in wires])]).astype(np.uint64)
    blob = np.frombuffer(b"".join(wires), np.uint8).copy()
    ss = rsb.ShardSet(shards)
    rep_off = np.zeros(2 * len(reqs) + 1, np.uint64)
    need = C.c_size_t()
    rc = L.rsbwt_service_counts(ss._s, blob.ctypes.data, blob.size, off.ctypes.data, len(reqs), None, 0,
                                rep_off.ctypes.data, C.byref(need))
    assert rc == -7 and need.value > 0
    out = np.zeros(need.value, np.uint8)
    assert L.rsbwt_service_counts(ss._s, blob.ctypes.data, blob.size, off.ctypes.data, len(reqs), out.ctypes.data,
                                  out.size, rep_off.ctypes.data, C.byref(need)) == 0
    # offsets that are not an ascending range inside the buffer are refused, with a message of their own
    bad_off = off.copy()
    bad_off[3], bad_off[4] = bad_off[4], bad_off[3]
    assert L.rsbwt_service_counts(ss._s, blob.ctypes.data, blob.size, bad_off.ctypes.data, len(reqs), out.ctypes.data,
                                  out.size, rep_off.ctypes.data, C.byref(need)) == -1
    assert b"ascending" in L.rsbwt_last_error()
    assert L.rsbwt_service_counts(ss._s, blob.ctypes.data, blob.size - 1, off.ctypes.data, len(reqs), out.ctypes.data,
                                  out.size, rep_off.ctypes.data, C.byref(need)) == -1
    for i, r in enumerate(reqs):
        f = out[int(rep_off[2 * i]):int(rep_off[2 * i + 1])].tobytes()
        v = out[int(rep_off[2 * i + 1]):int(rep_off[2 * i + 2])].tobytes()
        if not (r.t == 1 or (r.t == 2 and r.rt == 1)):
            assert f == b"" and v == b""
            continue
        fm, vm = Reply(), Reply()
        fm.ParseFromString(f); vm.ParseFromString(v)
        assert (fm.rt, fm.t, fm.q) == (r.t, 1, r.q) and (vm.rt, vm.t, vm.q) == (r.t, 1, r.q)
        assert fm.c.HasField("forward_matches") and not fm.c.HasField("revcomp_matches")
        assert vm.c.HasField("revcomp_matches") and not vm.c.HasField("forward_matches")

        def count(w):  # count_reads, service.cpp:299-304, summed over partitions
            if not w or any(ch not in "ACGT" for ch in w):
                return 0
            tot = 0
            for ox in oixs:
                lo, up = ox.find_interval(w)
                tot += up - lo + 1 if up >= lo else 0
            return tot
        assert fm.c.forward_matches.c == count(r.q)
        assert vm.c.revcomp_matches.c == count(_rc(r.q))
    ss.close()
    for g in shards:
        g.close()
