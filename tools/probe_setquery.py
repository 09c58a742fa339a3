import ctypes as C, os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
import readserver_amd as rsb
L = rsb.lib()
P=4
with tempfile.TemporaryDirectory() as td:
    kw = dict(seed=77, genome_len=300000, haplotypes=8, snp_rate=0.002, read_len=100, coverage=8.0)
    shards, reads = [], []
    for s in range(P):
        p, rd = os.path.join(td, f"s{s}.bwt"), os.path.join(td, f"s{s}.reads")
        rsb.synth_popbwt(p, rd, shard=s, num_shards=P, **kw)
        shards.append(rsb.GpuBWT(p, for_reads=True))
        reads += open(rd).read().split()
    ss = rsb.ShardSet(shards)
    rng = np.random.default_rng(1)
    for k in (40, 80):
        m = 55
        qs = []
        for _ in range(m):
            r = reads[int(rng.integers(len(reads)))]; st = int(rng.integers(0, len(r)-k+1)); qs.append(r[st:st+k])
        def t(f, n=30):
            f(); t0=time.time()
            for _ in range(n): f()
            return (time.time()-t0)/n*1e3
        print(k, "set.find_intervals ms", round(t(lambda: ss.find_intervals(qs)),3))
        print(k, "shard0 find_intervals ms", round(t(lambda: rsb.find_intervals(shards[0], qs)),3))
        print(k, "set.query ms", round(t(lambda: ss.query(qs, read_stride=256)),3), "reads", sum(len(x) for x in ss.query(qs, read_stride=256)))
        lo, up = rsb.find_intervals(shards[0], qs)
        rows = np.concatenate([np.arange(l, u+1, dtype=np.uint64) for l,u in zip(lo,up) if u>=l])
        print(k, "shard0 extract", len(rows), "rows ms", round(t(lambda: rsb.extract_reads(shards[0], rows, stride=256)) ,3))
