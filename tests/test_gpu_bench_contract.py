"""GPU suite (-m gpu): bench.py keeps its contract with the driver -- ONE JSON line on stdout with the keys the
round's records are made from -- on a small instance of every mode, started the way the driver starts it (a plain
`python bench.py ...`), including the self-launched two-rank rehearsal on one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--runs", "3e8", "--shards-per-gpu", "2", "--steps", "2", "--warmup", "1"]


def _run(args, timeout=600, more_env=None, lines_expected=1):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(more_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == lines_expected, out.stdout[-2000:]
    if lines_expected == 2:
        # an N > 1 run of the exact search: the headline line BEFORE the one-process leg, and again with the leg's record
        # (VERDICT r04 next #1a) -- the same line but for config.cxx_host
        first, last = json.loads(lines[0]), json.loads(lines[1])
        assert "pending" in first["config"]["cxx_host"] and "pending" not in last["config"]["cxx_host"]
        assert first["value"] == last["value"] and first["roofline"] == last["roofline"]
        a, b = dict(first["config"]), dict(last["config"])
        a.pop("cxx_host"), b.pop("cxx_host")
        assert a == b
    return json.loads(lines[-1])


def _common(d, n_gpus):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_gpu_bench_exact_line_and_its_mixes():
    d = _run(SMALL + ["--queries", "4e5", "--cpu-sample", "2e4"])
    _common(d, 1)
    assert d["metric"].startswith("31-mer backward-search queries/sec") and d["unit"] == "searches/s"
    assert abs(d["queries_per_s"] - d["value"] / 2) < 1e-6 * d["value"]
    mixes = d["config"]["mixes"]
    assert set(mixes) == {"population", "disjoint"}  # (the valid-popBWT leg belongs to full-size runs)
    assert mixes["population"]["mean_lf_steps_per_search"] > mixes["disjoint"]["mean_lf_steps_per_search"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["gpu_matches_oracle_on_sample"] is True and c["cores"] >= 1
    ref = c["reference_beside_port"]
    assert ref is None or "error" not in ref
    if ref:  # the compiled reference travelled: same answers as the port
        assert all(v["same_answers"] for k, v in ref.items() if k.endswith("_threads"))


@pytest.mark.parametrize("mode,extra", [("1mm", ["--kmers", "2e4"]), ("extract", ["--rows", "1e5"])])
def test_gpu_bench_rows_modes(mode, extra):
    d = _run(SMALL + ["--mode", mode] + extra)
    _common(d, 1)
    assert ("configs[3]" if mode == "1mm" else "configs[4]") in d["config"]["workload"]
    # the mode's own CPU baseline: the oracle's composition / walk of the sample the GPU's output is held to
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["gpu_matches_oracle_on_sample"] is True
    assert d["config"]["hit_lists_verified" if mode == "1mm" else "reads_verified"] is True


def test_gpu_bench_1mm_by_the_traced_launch_and_the_branch_kernel():
    """The A/B path of a set's 1-mismatch hit lists (RSBWT_SET_1MM_NO_WALK: a traced launch of the pair kernel,
    wl_branch_kernel, wl_own_kernel instead of the one walk of the k-mers) leaves the same lists: shard 0's against the
    oracle's, as in the default run."""
    d = _run(SMALL + ["--mode", "1mm", "--kmers", "2e4"], more_env={"RSBWT_SET_1MM_NO_WALK": "1"})
    assert d["config"]["hit_lists_verified"] is True and d["cpu_baseline"]["gpu_matches_oracle_on_sample"] is True
    assert "traced" in d["roofline"]["kernel"]


def test_gpu_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (here: the two ranks share the one GPU and gather over gloo)."""
    d = _run(SMALL + ["--queries", "2e5", "--gpus", "2", "--rehearse-on-one-gpu"], lines_expected=2)
    _common(d, 2)
    assert d["config"]["gather_verified"] is True and "REHEARSAL" in d["config"]["multi_gpu"]
    assert d["config"]["shards"] == 4
    leg = d["config"]["cxx_host"]  # (the second leg ran -- on one device here -- inside what was left of the run's budget)
    assert "error" not in leg and leg["value"] > 0 and leg["leg_seconds"] <= leg["leg_limit_seconds"] <= 420.0
    p = d["config"]["hbm_plan"]   # (rank 0's plan: its two batches of gathered blocks at N = 2)
    assert p["world"] == 2 and p["rank"] == 0 and p["gathered"] > 0 and p["unit"] == "GB"


@pytest.mark.parametrize("mode,extra,key", [("1mm", ["--kmers", "2e4"], "hit_lists_verified"), ("extract", ["--rows", "1e5"], "reads_verified")])
def test_gpu_bench_rows_modes_over_two_ranks(mode, extra, key):
    """configs[3] / configs[4] at N = 2 from the plain command (two ranks on the one GPU, gathered over gloo): rank 0
    holds every rank's lists / reads in global shard order, checked against the ranks' checksums."""
    d = _run(SMALL + ["--mode", mode, "--gpus", "2", "--rehearse-on-one-gpu"] + extra)
    _common(d, 2)
    assert d["config"][key] is True and "REHEARSAL" in d["config"]["multi_gpu"]
