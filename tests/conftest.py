import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """oracle/liboracle.so (CPU restatement of the reference algorithm) -- the checker."""
    import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def rsb():
    """The product package; builds librsbwt.so if it is missing or stale (hipcc, no GPU needed)."""
    import readserver_amd
    readserver_amd.build()
    readserver_amd.lib()
    return readserver_amd


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def fixture_bwt(rsb, tmp_path_factory, golden_dir):
    """The golden popBWT fixture, re-synthesised from its committed parameters and checked
    against the committed SHA-256, so the golden vectors and the BWT cannot drift apart."""
    import hashlib
    import json
    meta = json.load(open(os.path.join(golden_dir, "popbwt_v1.json")))
    d = tmp_path_factory.mktemp("fixture")
    path = str(d / "popbwt_v1.bwt")
    rsb.synth_popbwt(path, None, **meta["synth"])
    sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
    assert sha == meta["bwt_sha256"], "synthesiser no longer reproduces the golden fixture"
    return path, meta
