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
    return 0;
}
