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

// (bit 31 of window_span, both hooks: the RSBWT_OPEN_READS layout -- room for a psi hint in every window line)
static constexpr uint32_t SELFTEST_ROOM = 1u << 31;

extern "C" int rsbwt_layout_selftest_host(const uint8_t *runs, uint64_t num_runs, uint32_t window_span,
                                          uint64_t *stats6, uint64_t *first_bad) {
    if (!runs && num_runs) return RSBWT_EINVAL;
    const bool room = (window_span & SELFTEST_ROOM) != 0u;
    window_span &= ~SELFTEST_ROOM;
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
            const group_stats st = build_group<false>(sp, n, nwin, g, rd, nullptr, 0, room);
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
            build_group<true>(sp, n, nwin, g, rd, lines.data(), first_far + far_base[g], room);
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
    v.sel_shift = room ? SEL_SHIFT_SPARSE : SEL_SHIFT_DENSE;
    v.hint_room = room ? 1u : 0u;
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


// TEST HOOK, host only: the select samples and psi hints (line_format.h: window_samples, sample_window,
// window_psi_hint, hint_window -- the code the builder kernels and the walk kernels run) built over a host-side layout
// of `runs` and held to the naive answer: for EVERY occurrence of every symbol the sample names the window that holds
// it (or, where it says "not exact", a window at or before it); for EVERY row whose window's line carries a hint that
// claims to be exact, the hint names the window psi takes the row to; and with the hints written into the lines every
// scalar reader still gives the naive answer at every position.  stats4 = {sample words, occurrences whose sample is
// only a bound, lines with a hint, rows answered by a hint}.
extern "C" int rsbwt_layout_selftest_psi_host(const uint8_t *runs, uint64_t num_runs, uint32_t window_span, uint64_t *stats4,
                                              uint64_t *first_bad) {
    if (!runs && num_runs) return RSBWT_EINVAL;
    const bool room = (window_span & SELFTEST_ROOM) != 0u;
    window_span &= ~SELFTEST_ROOM;
    std::vector<uint8_t> bwt;
    for (uint64_t r = 0; r < num_runs; ++r) {
        if ((runs[r] >> 5) > 4) return RSBWT_EFORMAT;
        bwt.insert(bwt.end(), runs[r] & 31u, (uint8_t)(runs[r] >> 5));
    }
    const uint64_t n = bwt.size();
    if (first_bad) *first_bad = ~0ull;
    if (stats4) stats4[0] = stats4[1] = stats4[2] = stats4[3] = 0;
    if (n == 0) return RSBWT_OK;
    uint32_t S = window_span;
    if (!S) S = (uint32_t)(90.0 * (double)n / (double)num_runs + 0.5);
    const span_params sp = make_span(S);
    const uint64_t nwin = (n + sp.S - 1) / sp.S, ngroups = (nwin + GROUP - 1) / GROUP;
    std::vector<uint64_t> far_base(ngroups + 1, 0);
    {
        const uint64_t zero[4] = {0, 0, 0, 0};
        run_reader rd;
        rd.start(runs, num_runs, 0, zero);
        for (uint64_t g = 0; g < ngroups; ++g) far_base[g + 1] = far_base[g] + build_group<false>(sp, n, nwin, g, rd, nullptr, 0, room).far_lines;
    }
    const uint64_t first_far = ngroups * (GROUP + 1), nlines = first_far + far_base[ngroups];
    std::vector<uint32_t> lines(nlines * LINE_DWORDS, 0);
    {
        const uint64_t zero[4] = {0, 0, 0, 0};
        run_reader rd;
        rd.start(runs, num_runs, 0, zero);
        for (uint64_t g = 0; g < ngroups; ++g) build_group<true>(sp, n, nwin, g, rd, lines.data(), first_far + far_base[g], room);
    }
    shard_view v;
    memset(&v, 0, sizeof v);
    v.lines = lines.data();
    v.n = n;
    v.nwin = nwin;
    v.nlines = nlines;
    v.first_far = first_far;
    v.sp = sp;
    v.sel_shift = room ? SEL_SHIFT_SPARSE : SEL_SHIFT_DENSE;
    v.hint_room = room ? 1u : 0u;
    for (uint64_t p = 0; p < n; ++p) v.total[bwt[p]]++;
    for (int c = 1; c < 5; ++c) v.C[c] = v.C[c - 1] + v.total[c - 1];
    // positions of the occurrences of every symbol (the naive select)
    std::vector<uint64_t> where[5];
    for (uint64_t p = 0; p < n; ++p) where[bwt[p]].push_back(p);
    // ---- samples
    const uint64_t stride = select_stride(v);
    std::vector<uint64_t> sel(5 * stride, 0);
    uint64_t words = 0, inexact = 0;
    for (uint64_t w = 0; w < nwin; ++w)
        for (uint32_t c = 0; c <= 4; ++c)
            window_samples(v, w, c, [&](uint64_t m, uint64_t word) {
                sel[c * stride + m] = word;
                ++words;
            });
    for (uint32_t c = 0; c <= 4; ++c)
        for (uint64_t bc = 1; bc <= v.total[c]; ++bc) {
            bool exact;
            const uint32_t w = sample_window(sel[c * stride + ((bc - 1) >> v.sel_shift)], bc, v.sel_shift, &exact);
            const uint64_t truth = where[c][bc - 1] / sp.S;
            if (!exact) ++inexact;
            // exact or a bound -- and, either way, the floor search between the samples ends on the window
            if ((exact ? w != truth : w > truth) || select_window(v, sel.data(), stride, c, bc) != truth) {
                if (first_bad) *first_bad = where[c][bc - 1];
                return RSBWT_EFORMAT;
            }
        }
    // ---- hints, written into the lines as the kernel writes them
    uint64_t hint_lines = 0, by_hint = 0;
    for (uint64_t w = 0; w < nwin; ++w) {
        uint32_t w0, kk;
        if (!window_psi_hint(v, sel.data(), stride, w, &w0, &kk)) continue;
        uint32_t *Ln = lines.data() + line_of_window(w) * LINE_DWORDS;
        const uint32_t hd = hint_dword(parse_line(Ln).kind);
        Ln[hd] = w0;
        Ln[hd + 1u] = kk;
        Ln[1] |= 1u << (8u + HINT_META0_BIT);
        ++hint_lines;
    }
    const uint32_t hs = hint_shift(sp.S);
    for (uint64_t i = 0; i < n; ++i) {  // row i: psi(i) = select_f(i - C[f] + 1), f = F(i)
        uint32_t f = 0;
        while (f < 4u && v.C[f + 1] <= i) ++f;
        if (f == 0u) continue;
        const uint64_t w = i / sp.S, r0 = w * (uint64_t)sp.S;
        const uint32_t *Ln = lines.data() + line_of_window(w) * LINE_DWORDS;
        const line_meta lm = parse_line(Ln);
        if (!lm.hint || Ln[hint_dword(lm.kind)] == HINT_NONE || r0 < v.C[f]) continue;  // (the walk kernel's own conditions)
        const hint_range hr = hint_windows(Ln[hint_dword(lm.kind)], Ln[hint_dword(lm.kind) + 1u], (uint32_t)(i - r0), hs);
        const uint64_t truth = where[f][i - v.C[f]] / sp.S;
        // the row's window lies in lo..hi -- or, an open range, anywhere from lo on
        if (truth < hr.lo || (!hr.open && truth > hr.hi)) {
            if (first_bad) *first_bad = i;
            return RSBWT_EFORMAT;
        }
        if (hr.lo == hr.hi && !hr.open) ++by_hint;
    }
    if (stats4) {
        stats4[0] = words;
        stats4[1] = inexact;
        stats4[2] = hint_lines;
        stats4[3] = by_hint;
    }
    // ---- every scalar reader again, over the lines that now carry hints
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

// TEST HOOK: the grouped k-mer table's record code (line_format.h: what ktab_group_encode_kernel writes and what
// ktab_entry reads) on the host: `groups` x 4 (lower, upper) pairs in, the 4 entries each record gives back out, in
// the plain table's form {lower:40 | width:24}, width RSBWT_KTAB_WIDE = left to the search.
extern "C" int rsbwt_ktab_group_selftest_host(const uint64_t *lower, const uint64_t *upper, size_t groups, uint64_t *entries) {
    if ((!lower || !upper || !entries) && groups) return RSBWT_EINVAL;
    for (size_t g = 0; g < groups; ++g) {
        uint32_t rec[3];
        ktab_group_encode(lower + 4 * g, upper + 4 * g, rec);
        for (uint32_t i = 0; i < 4u; ++i) entries[4 * g + i] = ktab_group_entry(rec[0], rec[1], rec[2], i);
    }
    return RSBWT_OK;
}
