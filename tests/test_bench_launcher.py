"""`python bench.py --gpus N` launches its own N ranks (the driver calls it exactly like that).

The parent may not have made any GPU call before it starts the ranks -- a process that has
initialised HIP must not be forked or replaced on this pool -- so these tests hold it to importing
neither torch nor the library, to the command it builds, and to its exit code."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_launch_command_shape():
    b = _bench()
    cmd = b.self_launch_cmd(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]  # the ranks get the same arguments


def test_parent_touches_no_gpu_code_and_relays_the_exit_code(tmp_path):
    """A stand-in for torch.distributed.run records what it was started with and how: the parent
    reached it without torch or the library in sys.modules, and passes its exit code on."""
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import sys, json, os\n"
        "sys.path.insert(0, %r)\n"
        "import bench, subprocess\n"
        "seen = {}\n"
        "class P:\n"
        "    def __init__(self, cmd, env=None):\n"
        "        seen['cmd'] = cmd; seen['mods'] = [m for m in ('torch', 'readserver_amd', 'numpy.core._multiarray_umath') if m in sys.modules]\n"
        "        seen['ipc'] = env.get('HSA_ENABLE_IPC_MODE_LEGACY')\n"
        "    def wait(self):\n"
        "        return 7\n"
        "subprocess.Popen = P\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--rehearse-on-one-gpu']\n"
        "os.environ.pop('WORLD_SIZE', None)\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    seen['rc'] = e.code\n"
        "print(json.dumps(seen))\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(probe)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    import json
    seen = json.loads(out.stdout.strip().splitlines()[-1])
    assert seen["rc"] == 7
    assert "torch" not in seen["mods"] and "readserver_amd" not in seen["mods"]
    assert seen["ipc"] == "0"
    assert seen["cmd"][-3:] == ["--gpus", "2", "--rehearse-on-one-gpu"]
    assert "--nproc-per-node=2" in seen["cmd"]


def test_inside_a_launcher_it_does_not_launch_again():
    """With WORLD_SIZE set (a rank under torch.distributed.run) the script goes straight on; here that
    ends at the GPU check (no GPU in the build container) or at the world-size check."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr and "launching" not in out.stderr


def test_cxx_host_is_one_process_and_the_second_leg_of_an_n_gpu_run(monkeypatch):
    """`--host cxx` is the C++ host's shape: ONE process over all --gpus devices, so it must NOT start ranks; and an
    N > 1 run of the default host re-runs the same workload under it as a second leg (config.cxx_host), started by
    rank 0 as a child with the run's own sizes."""
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--host", "cxx", "--gpus", "4", "--runs", "3e8", "--queries", "2e5", "--steps", "3"])
    a = b.parse()
    assert a.host == "cxx" and a.gpus == 4
    launched, ran = [], []
    monkeypatch.setattr(b, "self_launch", lambda *x: launched.append(x))
    monkeypatch.setattr(b, "run_exact_cxx", lambda a_: ran.append(a_) or {"ok": True})
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    b.main()
    assert ran and not launched  # no torch.distributed.run, no ranks
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2", "--runs", "2e10", "--queries", "1e7"])
    cmd = b.cxx_leg_cmd(b.parse())
    assert cmd[:2] == [sys.executable, os.path.join(ROOT, "bench.py")]
    for flag, val in (("--host", "cxx"), ("--gpus", "8"), ("--steps", "7"), ("--warmup", "2"), ("--shards-per-gpu", "8"), ("--mix", "population")):
        assert cmd[cmd.index(flag) + 1] == val
    assert float(cmd[cmd.index("--runs") + 1]) == 2e10 and float(cmd[cmd.index("--queries") + 1]) == 1e7


def test_bench_picks_the_tables_the_hbm_left_over_allows():
    """bench.py pick_tables: one depth and format for the job out of the HBM that is free once every buffer exists, less
    8 GB -- the grouped format (3 B per T-mer) where it is a level deeper than the plain one (8 B) and a T-mer still has
    64 rows; and bench.total sums byte tensors without an int64 copy of them."""
    import argparse
    import torch
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(ktab_format="auto")
    n = 117219747342  # symbols of a 2e10-run-byte `pop` shard
    assert bench.pick_tables(a, int(42e9), 8, n, 13) == (15, 1)       # the headline: 8 shards, 42 GB left
    assert bench.pick_tables(a, int(30e9), 8, n, 13) == (14, 0)       # rank 0 with two batches' gathered blocks: plain 14-mers
    assert bench.pick_tables(a, int(270e9), 1, n, 13) == (16, 0)      # one shard: a 17-mer has 7 rows -- plain 16-mers
    assert bench.pick_tables(a, int(200e9), 8, 2_100_000_000, 12) == (15, 0)  # the valid popBWT's shards: too few rows per T-mer
    assert bench.pick_tables(argparse.Namespace(ktab_format="plain"), int(42e9), 8, n, 13) == (14, 0)
    assert bench.pick_tables(a, int(42e9), 8, n, 0) == (0, 0)
    t = torch.arange(300, dtype=torch.int32).to(torch.uint8)
    assert bench.total(t) == int(t.to(torch.int64).sum()) and bench.total(torch.empty(0, dtype=torch.uint8)) == 0
    assert bench.total(torch.tensor([-1, 5, -1], dtype=torch.int32)) == 3


def test_the_headline_line_is_out_before_the_second_leg_and_the_leg_lives_on_what_is_left(monkeypatch):
    """VERDICT r04 next #1a.  Rank 0 of an N > 1 run prints its line BEFORE it starts the one-process leg and again with
    the leg's record after it; the leg gets what is left of --total-budget (<= 540 s: the driver allows 600), never its
    fixed 420 s; a leg that times out, fails, or is not worth starting leaves both lines valid."""
    import json
    import subprocess
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    a = b.parse()
    assert a.total_budget <= 540.0
    # the budget rule
    assert b.cxx_leg_budget(a, 100.0) == (420.0, None)             # early: the leg's own cap
    lim, why = b.cxx_leg_budget(a, 300.0)
    assert lim == 540.0 - 300.0 - 20.0 and why is None            # a per-rank leg of 300 s leaves 220 s, not 420
    lim, why = b.cxx_leg_budget(a, 460.0)
    assert lim is None and "not started" in why                    # < 90 s: the shards alone take 25 s per GPU
    head = {"metric": "m", "value": 1.0, "config": {"workload": "w"}}

    def run(elapsed, runner):
        printed, calls = [], []
        monkeypatch.setattr(b, "T_START", 1000.0)
        def wrapped(cmd, timeout):
            calls.append((cmd, timeout, len(printed)))
            return runner(cmd, timeout)
        b.second_leg(a, json.loads(json.dumps(head)), emit=printed.append, runner=wrapped, clock=lambda: 1000.0 + elapsed)
        return [json.loads(x) for x in printed], calls

    class R:
        def __init__(self, rc, out, err=""):
            self.returncode, self.stdout, self.stderr = rc, out, err
    # a leg that answers
    lines, calls = run(120.0, lambda cmd, t: R(0, 'noise\n{"value": 7.0, "n_gpus": 8}\n'))
    assert len(lines) == 2 and calls[0][2] == 1                     # one line was out when the child started
    assert "pending" in lines[0]["config"]["cxx_host"] and lines[0]["value"] == 1.0
    assert lines[1]["config"]["cxx_host"]["value"] == 7.0 and lines[1]["value"] == 1.0
    assert calls[0][1] == 400.0 and calls[0][0][calls[0][0].index("--host") + 1] == "cxx"
    # a leg that runs out of time: the exception is the record, both lines parse
    def slow(cmd, t):
        raise subprocess.TimeoutExpired(cmd, t)
    lines, calls = run(330.0, slow)
    assert len(lines) == 2 and "TimeoutExpired" in lines[1]["config"]["cxx_host"]["error"] and calls[0][1] == 190.0
    # a leg that fails
    lines, _ = run(50.0, lambda cmd, t: R(1, "", "hipErrorOutOfMemory"))
    assert "hipErrorOutOfMemory" in lines[1]["config"]["cxx_host"]["error"]
    # no time left: not started, said so
    lines, calls = run(500.0, lambda cmd, t: R(0, "{}"))
    assert not calls and "not started" in lines[1]["config"]["cxx_host"]["skipped"]


def test_rank0_hbm_plan_at_2_4_8_gpus():
    """VERDICT r04 next #1c: what a rank holds, item by item, as plain arithmetic (readserver_amd/sharded.py, hbm_plan) --
    bench.py checks it against torch.cuda.mem_get_info before the first shard is built.  configs[2]'s load: 8 shards of
    2e10 run bytes (32.7 GB of lines each), 1e7 31-mers per batch, grouped 15-mer tables (3.2 GB per shard)."""
    from readserver_amd import sharded
    lines, runs, q, tab = int(1.633 * 2e10), int(2e10), 10**7, 3 * 4**15
    hbm = 309_220_868_096  # what an MI355X reports as total (288 GiB)
    plans = {w: sharded.hbm_plan(w, 0, 8, q, 31, lines, runs, ktab_bytes_per_shard=tab) for w in (1, 2, 4, 8)}
    for w, p in plans.items():
        assert p["shards"] == 8 * lines and p["tables"] == 8 * tab
        assert p["steady"] == sum(p[x] for x in ("shards", "batch", "wire", "gathered", "scratch", "tables", "reserve"))
        assert p["batch"] == q * 31 + q * 8 + q + 2 * 8 * q * 16
        sharded.check_hbm_plan(p, hbm - (1 << 30))  # fits, with the 8 GiB reserve inside the plan
    assert plans[1]["wire"] == 0 and plans[1]["gathered"] == 0
    blk = sharded.packed_pairs_bytes(8 * q)                         # a rank's batch as 10-byte records: 0.8 GB
    assert plans[2]["gathered"] == 2 * 2 * blk and plans[4]["gathered"] == 4 * blk and plans[8]["gathered"] == 8 * blk
    assert plans[8]["out_depth"] == 1 and plans[2]["out_depth"] == 2
    # the other ranks gather nothing
    assert sharded.hbm_plan(8, 3, 8, q, 31, lines, runs, ktab_bytes_per_shard=tab)["gathered"] == 0
    # what does NOT fit says so, with the table: two batches' blocks on rank 0 of 8 (round 4's first layout), unpacked pairs
    big = sharded.hbm_plan(8, 0, 8, q, 31, lines, runs, ktab_bytes_per_shard=tab, out_depth=2, wire_packed=False)
    with pytest.raises(MemoryError) as e:
        sharded.check_hbm_plan(big, hbm - (1 << 30))
    msg = str(e.value)
    assert "rank 0 of 8" in msg and "gathered" in msg and "tables" in msg and "GB" in msg
    # ... and the same job with plain 14-mer tables does (what pick_tables falls back to)
    sharded.check_hbm_plan(sharded.hbm_plan(8, 0, 8, q, 31, lines, runs, ktab_bytes_per_shard=8 * 4**14, out_depth=2), hbm - (1 << 30))
    # the last shard's build (its run bytes and prefix arrays beside the lines it writes) is the peak before the tables exist
    assert plans[1]["build_peak"] == 7 * lines + runs + int(runs * 0.09) + lines
