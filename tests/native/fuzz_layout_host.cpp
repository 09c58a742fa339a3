// fuzz_layout_host.cpp -- AddressSanitizer / UBSan run of the window-line layout code the GPU builder
// and the scalar readers share with the host (csrc/line_format.h through csrc/layout_host.cpp's test
// hook): random run streams of several shapes (short runs, 31-symbol units, '$'-heavy, zero-length
// bytes, single-symbol deserts) laid out at window spans from 2 to 2,944 and held to naive ranks at
// every position.  CPU only.  Built and run by tests/test_layout_host.py.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "rsbwt.h"

extern "C" int rsbwt_layout_selftest_host(const uint8_t *runs, uint64_t num_runs, uint32_t window_span,
                                          uint64_t *stats6, uint64_t *first_bad);
extern "C" int rsbwt_layout_selftest_psi_host(const uint8_t *runs, uint64_t num_runs, uint32_t window_span,
                                              uint64_t *stats4, uint64_t *first_bad);
extern "C" int rsbwt_ktab_group_selftest_host(const uint64_t *lower, const uint64_t *upper, size_t groups, uint64_t *entries);

static uint64_t s = 0xD1B54A32D192ED03ull;
static uint64_t rnd() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return s;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 60;
    const uint32_t spans[] = {0, 2, 3, 17, 64, 300, 915, 2233, 2944};
    uint64_t hint_lines = 0, hinted_rows = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t R = 1 + rnd() % 30000;
        const int shape = it % 6;
        std::vector<uint8_t> runs(R);
        for (auto &r : runs) {
            uint32_t sym = (uint32_t)(rnd() % 5), len = 1 + (uint32_t)(rnd() % 31);
            if (shape == 1) len = 31;                                   // long runs split into full units
            if (shape == 2) len = 1 + (uint32_t)(rnd() % 2);            // many pieces per window
            if (shape == 3 && rnd() % 3 == 0) sym = 0;                  // '$'-heavy
            if (shape == 4 && rnd() % 4 == 0) len = 0;                  // zero-length bytes
            if (shape == 5) sym = (rnd() % 200) ? 1u : (uint32_t)(rnd() % 5);  // deserts of one symbol
            r = (uint8_t)(sym << 5 | len);
        }
        const uint32_t span = spans[rnd() % (sizeof spans / sizeof spans[0])];
        uint64_t st[6] = {0, 0, 0, 0, 0, 0}, bad = 0;
        const int rc = rsbwt_layout_selftest_host(runs.data(), R, span, st, &bad);
        if (rc != RSBWT_OK) {
            fprintf(stderr, "iteration %d: shape %d, %llu runs, span %u: rc %d, first bad position %llu\n", it, shape,
                    (unsigned long long)R, span, rc, (unsigned long long)bad);
            return 1;
        }
        // the select samples and psi hints over the same stream: every occurrence, every row, every reader again
        uint64_t st4[4] = {0, 0, 0, 0};
        const int rc2 = rsbwt_layout_selftest_psi_host(runs.data(), R, span, st4, &bad);
        if (rc2 != RSBWT_OK) {
            fprintf(stderr, "iteration %d (samples / hints): shape %d, %llu runs, span %u: rc %d, first bad %llu\n", it, shape,
                    (unsigned long long)R, span, rc2, (unsigned long long)bad);
            return 1;
        }
        hint_lines += st4[2];
        hinted_rows += st4[3];
    }
    printf("%d run streams laid out and checked at every position; %llu lines with a psi hint answered %llu rows\n", iters,
           (unsigned long long)hint_lines, (unsigned long long)hinted_rows);
    // the grouped k-mer table's record code (line_format.h: ktab_group_encode / ktab_group_entry) on four siblings that
    // tile a stretch of rows and on four arbitrary (lower, upper) pairs: whatever goes in, an entry that comes out with a
    // width is EXACTLY its sibling's interval; everything else is left to the search
    {
        const uint64_t WIDE = 0xFFFFFFull;
        const size_t G = 200000;
        std::vector<uint64_t> lo(4 * G), up(4 * G), out(4 * G);
        for (size_t g = 0; g < G; ++g) {
            const bool sound = g % 2 == 0;
            uint64_t at = rnd() % ((1ull << 40) - (1ull << 18));
            for (int i = 0; i < 4; ++i) {
                if (sound) {
                    const uint64_t w = rnd() % 4 == 0 ? 0 : rnd() % (g % 16 == 0 ? 9000 : 300);
                    lo[4 * g + i] = w ? at : 1 + rnd() % (1ull << 40);  // (an empty interval sits anywhere)
                    up[4 * g + i] = lo[4 * g + i] + w - 1;
                    at += w;
                } else {  // anything at all: wrapped bounds, rows past 2^40, overlaps
                    lo[4 * g + i] = rnd() >> (rnd() % 64);
                    up[4 * g + i] = rnd() % 3 == 0 ? lo[4 * g + i] + rnd() % 5000 : rnd() >> (rnd() % 64);
                }
            }
        }
        if (rsbwt_ktab_group_selftest_host(lo.data(), up.data(), G, out.data()) != RSBWT_OK) return 1;
        size_t said = 0;
        for (size_t i = 0; i < 4 * G; ++i) {
            const uint64_t w = out[i] >> 40, l = out[i] & ((1ull << 40) - 1);
            if (w == WIDE) continue;
            ++said;
            if (w == 0 || l != lo[i] || l + w - 1 != up[i]) {
                fprintf(stderr, "grouped record: sibling %zu of group %zu (%llu, %llu) came back as (%llu, width %llu)\n", i % 4, i / 4,
                        (unsigned long long)lo[i], (unsigned long long)up[i], (unsigned long long)l, (unsigned long long)w);
                return 1;
            }
        }
        if (said < G / 2) {
            fprintf(stderr, "grouped record: only %zu of %zu siblings answered\n", said, 4 * G);
            return 1;
        }
        printf("%zu groups of four siblings through the grouped table's record code: %zu answered exactly, the rest left to the search\n", G, said);
    }
    return 0;
}
