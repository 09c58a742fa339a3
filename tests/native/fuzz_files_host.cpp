// fuzz_files_host.cpp -- AddressSanitizer / UBSan run of the host-side file readers (csrc/bpi2.cpp:
// bpi2_load, bpi2_from_bwt; csrc/bwt_file.cpp: bwt_open_read) on mutated and truncated files.  A valid
// .bwt and its .bpi2 are written first (bpi2_builder over seeded runs), then every iteration flips a few
// bytes -- the size fields among them -- and/or cuts the file short.  The readers may refuse a file; they
// may not crash, over-allocate from a forged length (sizes are held against the file size first) or
// read past a buffer.  CPU only.  Built and run by tests/test_bpi2.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../readserver_amd/csrc/bpi2.h"
#include "../../readserver_amd/csrc/bwt_file.h"

static uint64_t s = 0x2545F4914F6CDD1Dull;
static uint64_t rnd() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return s;
}
static std::vector<uint8_t> slurp(const std::string &p) {
    std::vector<uint8_t> b;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return b;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}
static void spit(const std::string &p, const std::vector<uint8_t> &b, size_t n) {
    FILE *f = fopen(p.c_str(), "wb");
    if (f) { fwrite(b.data(), 1, n, f); fclose(f); }
}

int main(int argc, char **argv) {
    const size_t iters = argc > 1 ? (size_t)atof(argv[1]) : 2000;
    const std::string dir = argc > 2 ? argv[2] : "/tmp";
    const std::string bwt = dir + "/fz.bwt", bpi = dir + "/fz.bwt.bpi2", mb = dir + "/fzm.bwt", mp = dir + "/fzm.bpi2";
    const size_t R = 70000;  // three levels would need 2^20 runs; two are enough to cover the level loop
    std::vector<uint8_t> runs(R);
    uint64_t nsym = 0;
    for (auto &r : runs) {
        const uint32_t sym = (uint32_t)(rnd() % 5), len = 1 + (uint32_t)(rnd() % 31);
        r = (uint8_t)(sym << 5 | len);
        nsym += len;
    }
    rsb::bwt_header h = {123, nsym, R, 0};
    if (rsb::bwt_write(bwt.c_str(), h, runs.data())) return 2;
    std::string err;
    rsb::bpi2_index ix;
    if (rsb::bpi2_from_bwt(bwt.c_str(), &ix, &err) || rsb::bpi2_save(ix, bpi.c_str(), &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
    const std::vector<uint8_t> good_bwt = slurp(bwt), good_bpi = slurp(bpi);
    size_t ok_bwt = 0, ok_bpi = 0;
    for (size_t i = 0; i < iters; ++i) {
        std::vector<uint8_t> b = good_bpi;
        for (int m = 1 + (int)(rnd() % 4); m > 0; --m) {
            // half of the flips land in the first 200 bytes: depth, widths, lengths, blocks, buckets
            const size_t at = (rnd() & 1) ? rnd() % 200 : rnd() % b.size();
            b[at] = (uint8_t)rnd();
        }
        spit(mp, b, (rnd() % 4) ? b.size() : rnd() % (b.size() + 1));
        rsb::bpi2_index got;
        if (rsb::bpi2_load(mp.c_str(), &got, &err) == 0) ++ok_bpi;
        std::vector<uint8_t> w = good_bwt;
        for (int m = 1 + (int)(rnd() % 3); m > 0; --m) w[rnd() % 64] = (uint8_t)rnd();
        spit(mb, w, (rnd() % 4) ? w.size() : rnd() % (w.size() + 1));
        FILE *f = nullptr;
        rsb::bwt_header hh;
        if (rsb::bwt_open_read(mb.c_str(), &f, &hh) == 0) {
            ++ok_bwt;
            fclose(f);
            rsb::bpi2_index again;  // a header that passed: the builder must cope with what follows it
            (void)rsb::bpi2_from_bwt(mb.c_str(), &again, &err);
        }
    }
    for (const std::string &p : {bwt, bpi, mb, mp}) remove(p.c_str());
    printf("%zu mutated files of each kind: %zu .bpi2 and %zu .bwt accepted\n", iters, ok_bpi, ok_bwt);
    return 0;
}
