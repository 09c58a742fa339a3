"""GPU suite (-m gpu): a 30-second slice each of the randomised differential campaign and of the thread-safety soak
the builder runs for much longer (tools/fuzz_parity.py, tools/soak.py; profiles/r0*_fuzz_parity.json, r0*_soak.json),
with fixed seeds, so that the driver's record -- not only the builder's -- says the concurrent paths and the odd
layouts hold ("weak" #10)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args, timeout=400):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *map(str, args)], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-4000:]
    return r.returncode, json.loads(lines[-1]), r.stderr


def test_gpu_fuzz_campaign_30_second_slice():
    """Random run streams x window span x layout (plain / opened for reads) x table depth x k against the oracle:
    intervals, counts, the 1-mismatch hit list, getOccAt, extraction, and two-shard sets (set-level lists, the fused
    1-mismatch launches, the fused extraction launch)."""
    rc, out, err = _run("fuzz_parity.py", 30, 4242)
    assert rc == 0 and out["failures"] == [], out["failures"] or err[-3000:]
    assert out["configurations"] >= 100 and out["as_two_shard_sets"] >= 5, out


def test_gpu_thread_safety_soak_30_second_slice():
    """8 host threads on one handle and one set (host entry points and the device-resident fused launches from a
    stream per thread), every answer equal to the single-threaded one."""
    rc, out, err = _run("soak.py", 30, 8, 5e7)
    assert rc == 0, json.dumps(out)[:2000] + err[-3000:]
    assert out["errors"] == [] and sum(v["calls"] for v in out["calls"].values()) > 1000, out
    assert "set_extract_dev" in out["calls"] and "set_hits_1mm_dev" in out["calls"], out
