"""CPU suite: the C-ABI library loads and exports exactly what include/rsbwt.h declares, and
fails loudly (RSBWT_ENODEV, never a CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "rsbwt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsbwt_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound(rsb):
    from readserver_amd import _native
    names = _declared()
    assert len(names) >= 40
    L = C.CDLL(rsb.lib_path())
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rsbwt.h but not exported"
    assert sorted(_native.SIGNATURES) == names, "ctypes binding and header disagree"


def test_version_and_errors(rsb):
    L = rsb.lib()
    assert b"rsbwt" in L.rsbwt_version()
    assert L.rsbwt_strerror(-5) == b"no usable HIP device"
    assert L.rsbwt_strerror(-3) == b"not an SGA run-length BWT file"


def test_no_gpu_means_loud_failure(rsb, tmp_path):
    L = rsb.lib()
    if L.rsbwt_device_count() > 0:
        pytest.skip("a GPU is present")
    runs = np.array([(1 << 5) | 3, (2 << 5) | 1], np.uint8)
    with pytest.raises(rsb.RsbwtError) as e:
        rsb.GpuBWT(runs=runs, num_strings=0)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_bad_files_are_reported_not_fatal(rsb, tmp_path):
    # the reference exits the process on a bad magic (src/bwt/rlebwt_reader.cpp:31-34)
    with pytest.raises(rsb.RsbwtError) as e:
        rsb.GpuBWT(str(tmp_path / "missing.bwt"))
    assert e.value.code == -2
    bad = tmp_path / "bad.bwt"
    bad.write_bytes(b"\xEF\xEF" + b"\0" * 40)
    with pytest.raises(rsb.RsbwtError) as e:
        rsb.GpuBWT(str(bad))
    assert e.value.code == -3
    trunc = tmp_path / "trunc.bwt"
    trunc.write_bytes(b"\xCA\xCA" + (1).to_bytes(8, "little") + (100).to_bytes(8, "little")
                      + (50).to_bytes(8, "little") + b"\0\0\0\0" + b"\x21" * 10)
    with pytest.raises(rsb.RsbwtError) as e:
        rsb.GpuBWT(str(trunc))
    assert e.value.code == -3


def test_synth_runs_host_matches_documented_mix(rsb):
    L = rsb.lib()
    a = np.empty(200000, np.uint8)
    b = np.empty(200000, np.uint8)
    assert L.rsbwt_synth_runs_host(a.ctypes.data, a.size, 42) == 0
    assert L.rsbwt_synth_runs_host(b.ctypes.data, b.size, 42) == 0
    assert np.array_equal(a, b)
    ln = a & 31
    assert ln.min() >= 1 and 9.5 < ln.mean() < 11.5
    assert (a >> 5).max() <= 4 and 0.005 < np.mean((a >> 5) == 0) < 0.02


def test_header_is_plain_c(tmp_path):
    """include/rsbwt.h is the drop-in boundary: it must compile as C99 with nothing but libc headers
    (plain pointers and sizes, no C++ or torch types), and again from C++."""
    import os
    import shutil
    import subprocess
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "rsbwt.h"\nint main(void) { return (int)sizeof(rsbwt_hit_1mm) - 24; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        r = subprocess.run([cc, std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-x", "c" if cc == "gcc" else "c++",
                            f"-I{os.path.join(root, 'include')}", "-fsyntax-only", str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_context_pool_does_not_hang_when_streams_cannot_be_made(tmp_path):
    """readserver_amd/csrc/ctx_pool.h (the per-call contexts of a handle: a stream pair each) with a maker that fails,
    on the CPU under -fsanitize=thread: every caller is refused or served, nobody waits for a release that cannot
    come (two callers failing at once; a release that happens while a caller is inside its failing maker), and the
    pool never holds more than its bound (tests/native/ctx_pool_test.cpp)."""
    import os
    import shutil
    import subprocess
    import pytest
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ctx_pool_test")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", os.path.join(root, "tests", "native", "ctx_pool_test.cpp"),
                        "-lpthread", "-o", exe], capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("no sanitizer runtime here")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr and "ctx_pool ok" in r.stdout, r.stdout + r.stderr[-3000:]
