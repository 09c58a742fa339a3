// shim_demo.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The drop-in claim of INTEGRATION.md run for real: the reference's UNMODIFIED src/bwt/query.cpp
// (findInterval, extractPrefix, extractPostfix, query, query_exactmatch) compiled from where it lies
// under /root/reference, driving `class GpuBWT : public BWT` (include/rsbwt_gpubwt.hpp), i.e. every
// virtual of include/bwt/bwt.h:6-15 answered by the GPU.  Built by oracle/Makefile into
// oracle/_ref/shim_demo in the build container only (the GPU box has no reference tree) and run there
// by tests/test_gpu_parity.py::test_gpu_reference_query_cpp_runs_on_the_shim.
//
//   shim_demo <file.bwt> <k-mer> ...   prints per k-mer: lower upper exactmatch nreads read...
#include <iostream>
#include <string>
#include <vector>

#include "rsbwt_gpubwt.hpp"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    try {
        GpuBWT *g = new GpuBWT(argv[1]);
        const BWT *pbwt = g;  // the reference's abstract interface
        for (int i = 2; i < argc; ++i) {
            const std::string w(argv[i]);
            const BWTInterval itv = findInterval(pbwt, w);  // the reference's own query.cpp
            std::cout << itv.lower << " " << itv.upper << " " << (query_exactmatch(pbwt, w) ? 1 : 0);
            const std::vector<std::string> reads = query(pbwt, w);
            std::cout << " " << reads.size();
            for (const std::string &r : reads) std::cout << " " << r;
            std::cout << "\n";
        }
        delete g;
    } catch (const std::exception &e) {
        std::cout << "error: " << e.what() << "\n";
        return 3;
    }
    return 0;
}
