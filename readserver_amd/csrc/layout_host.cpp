// layout_host.cpp -- host-side check of the window-line layout (line_format.h).
//
// TEST HOOK, not a query path: lays a run stream out with the very code the GPU builder kernels
// run (build_group) and holds the scalar readers the secondary kernels run (view_occ, view_char,
// view_occ_at) to naive ranks at every position.  It answers no query and nothing in the engine
// calls it; all searches, mirrors and extractions run on the GPU only.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/rsbwt.h"
#include "line_format.h"

using namespace rsb;

extern "C" int rsbwt_layout_selftest_host(const uint8_t *runs, uint64_t num_runs, uint32_t window_span,
                                          uint64_t *stats6, uint64_t *first_bad) {
    if (!runs && num_runs) return RSBWT_EINVAL;
    // naive expansion
    std::vector<uint8_t> bwt;
    for (uint64_t r = 0; r < num_runs; ++r) {
        if ((runs[r] >> 5) > 4) return RSBWT_EFORMAT;
        bwt.insert(bwt.end(), runs[r] & 31u, (uint8_t)(runs[r] >> 5));
    }
    const uint64_t n = bwt.size();
    if (first_bad) *first_bad = ~0ull;
    if (n == 0) return RSBWT_OK;
    uint32_t S = window_span;
    if (!S) S = (uint32_t)(90.0 * (double)n / (double)num_runs + 0.5);
    const span_params sp = make_span(S);
    const uint64_t nwin = (n + sp.S - 1) / sp.S, ngroups = (nwin + GROUP - 1) / GROUP;
    // pass 1: far lines per group
    std::vector<uint64_t> far_base(ngroups + 1, 0);
    group_stats tot = {0, 0, 0, 0};
    {
        const uint64_t zero[4] = {0, 0, 0, 0};
        run_reader rd;
        rd.start(runs, num_runs, 0, zero);
        for (uint64_t g = 0; g < ngroups; ++g) {
            const group_stats st = build_group<false>(sp, n, nwin, g, rd, nullptr, 0);
            far_base[g + 1] = far_base[g] + st.far_lines;
            tot.far_lines += st.far_lines;
            tot.chunk_windows += st.chunk_windows;
            tot.far_windows += st.far_windows;
            tot.spilled_symbols += st.spilled_symbols;
        }
    }
    const uint64_t first_far = ngroups * (GROUP + 1), nlines = first_far + far_base[ngroups];
    std::vector<uint32_t> lines(nlines * LINE_DWORDS, 0);
    // pass 2: every group from a reader re-seated at its first symbol (as the GPU threads are)
    {
        uint64_t cnt[4] = {0, 0, 0, 0};
        uint64_t r = 0, at = 0;
        for (uint64_t g = 0; g < ngroups; ++g) {
            const uint64_t gstart = g * GROUP * (uint64_t)sp.S;
            // runs wholly before the group's first symbol
            while (r < num_runs && at + (runs[r] & 31u) <= gstart) {
                const uint32_t sy = runs[r] >> 5;
                if (sy >= 1 && sy <= 4) cnt[sy - 1] += runs[r] & 31u;
                at += runs[r] & 31u;
                ++r;
            }
            run_reader rd;
            rd.start(runs, num_runs, r, cnt);
            rd.skip_symbols(gstart - at);
            build_group<true>(sp, n, nwin, g, rd, lines.data(), first_far + far_base[g]);
        }
    }
    shard_view v;
    memset(&v, 0, sizeof v);
    v.lines = lines.data();
    v.n = n;
    v.nwin = nwin;
    v.nlines = nlines;
    v.first_far = first_far;
    v.sp = sp;
    for (uint64_t p = 0; p < n; ++p) v.total[bwt[p]]++;
    for (int c = 1; c < 5; ++c) v.C[c] = v.C[c - 1] + v.total[c - 1];
    if (stats6) {
        stats6[0] = sp.S;
        stats6[1] = nlines;
        stats6[2] = tot.far_lines;
        stats6[3] = tot.chunk_windows;
        stats6[4] = tot.far_windows;
        stats6[5] = tot.spilled_symbols;
    }
    // every position: Occ of all five symbols, the symbol itself, and select of that occurrence
    uint64_t occ[5] = {0, 0, 0, 0, 0};
    for (uint64_t p = 0; p < n; ++p) {
        const uint32_t c = bwt[p];
        occ[c]++;
        bool ok = view_char(v, p) == c;
        for (uint32_t b = 0; b < 5 && ok; ++b) ok = view_occ(v, b, p) == occ[b];
        uint64_t oc = 0;
        ok = ok && view_char_occ(v, p, &oc) == c && oc == occ[c];
        ok = ok && view_occ_at(v, c, occ[c], 0, nwin - 1) == p;
        if (!ok) {
            if (first_bad) *first_bad = p;
            return RSBWT_EFORMAT;
        }
    }
    return RSBWT_OK;
}
