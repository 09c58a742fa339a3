"""CPU suite: include/rsbwt_gpubwt.hpp compiles against the reference's own headers and links with
the reference's unmodified src/bwt/query.cpp + librsbwt.so -- the drop-in claim of INTEGRATION.md.
Needs /root/reference (build container); skipped on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

PROG = r"""
#include <iostream>
#include <memory>
#include "rsbwt_gpubwt.hpp"
int main(int argc, char** argv) {
  try {
    GpuBWT* g = new GpuBWT(argv[1]);
    const BWT* pbwt = g;                              // the reference's abstract interface
    BWTInterval itv = findInterval(pbwt, argv[2]);    // the reference's own query.cpp
    std::vector<BWTInterval> b = findIntervals(g, {argv[2], argv[2]});
    std::cout << itv.lower << " " << itv.upper << " " << b[1].lower << " " << b[1].upper << "\n";
    delete g;
  } catch (const std::exception& e) {
    std::cout << "error: " << e.what() << "\n";
    return 3;
  }
  return 0;
}
"""


@pytest.mark.skipif(not os.path.isdir(REF + "/src/bwt"), reason="reference tree not present")
def test_shim_builds_against_reference_and_fails_loudly_without_gpu(rsb, tmp_path):
    src = tmp_path / "shim_main.cpp"
    src.write_text(PROG)
    exe = tmp_path / "shim_main"
    libdir = os.path.dirname(rsb.lib_path())
    subprocess.check_call([
        "g++", "-std=c++11", "-O1", "-w", f"-I{REF}/include/bwt", f"-I{ROOT}/include", str(src),
        f"{REF}/src/bwt/query.cpp", "-o", str(exe), f"-L{libdir}", "-lrsbwt", f"-Wl,-rpath,{libdir}",
        "-Wl,-rpath,/opt/rocm/lib"])
    bwt = tmp_path / "t.bwt"
    rsb.synth_popbwt(str(bwt), None, seed=1, genome_len=500, haplotypes=2, snp_rate=0.01, read_len=30,
                     coverage=2.0)
    p = subprocess.run([str(exe), str(bwt), "ACGT"], capture_output=True, text=True)
    if rsb.lib().rsbwt_device_count() == 0:
        assert p.returncode == 3 and "no CPU fallback" in p.stdout
    else:
        a = p.stdout.split()
        assert p.returncode == 0 and a[0] == a[2] and a[1] == a[3]
