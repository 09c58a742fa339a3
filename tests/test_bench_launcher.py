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
