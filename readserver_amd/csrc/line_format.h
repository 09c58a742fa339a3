// line_format.h -- the HBM layout of one popBWT shard: "window lines" (DESIGN.md section 3).
//
// The BWT is cut into WINDOWS of S symbols.  Window w is one 128-byte line at a computable
// address, so an Occ lookup (RLEBWT::getOcc, src/bwt/rlebwt.cpp:268-301) is ONE HBM request:
//
//     w = p / S                 (any S in 2..2944; kernels divide by an f64 multiply + one fix-up step)
//     line(w) = w + (w >> 4)    (every 16 window lines are followed by their group's SPILL line)
//
// The run bytes are ReadServer's RLUnit bytes verbatim (include/bwt/rlunit.h:8-11:
// rank(sym) << 5 | len, len 1..31, rank $=0 A=1 C=2 G=3 T=4; alphabet.h:8-9), split where a run
// crosses a window border ("pieces").  S is chosen from the data so that a window holds ~90 pieces
// on average; the 1.5 % or so of positions whose piece does not fit their window's line are found
// one more request away (spill chunk / far line), so memory is ~1.5 bytes per run byte, whatever
// the local run lengths, against 4/3 for blocks cut by run count -- which need a second,
// dependent request (a directory) to be found.
//
// WINDOW LINE = 32 dwords:
//   dwords 0..7    four header words (u64, little endian), word t for symbol t+1 (A, C, G, T):
//        bits  0..39  # of that symbol in BWT[0, w*S)                       (absolute)
//        bits 40..63  meta_t:
//            meta_0 = s1 | s2 << 10              quarters are pieces [0,24) [24,48) [48,72) [72,96);
//            meta_1 = d3 | d4 << 10 | kind << 20   s_t = symbols held by quarters 0..t-1,
//                                                  d3 = s3 - s2, d4 = span - s3,
//                                                  span = symbols held by this line's own pieces
//            meta_2 = halfA | halfC << 11 | (cdw/2 & 3) << 22     half_x = # of x in quarters 0 and 1
//            meta_3 = halfG | halfT << 11 | (cdw/2 >> 2) << 22
//        kind = 0: the line holds its whole window (span = symbols of the window)
//        kind = 1: the window has 97..120 pieces: the excess (<= 24 pieces) is a CHUNK at dword
//                  cdw (even) of the group's spill line
//        kind = 2: the line holds 92 pieces and its last dword (31) the index of a FAR line that
//                  continues the window (more than 120 pieces, or the spill line was full)
//   dwords 8..31   96 piece bytes (unused = 0; a valid piece has len >= 1)
//   $ before the window = w*S - (A + C + G + T).
//   PSI HINT (bit 21 of meta_0 set): the line's last two piece dwords do not hold pieces but where psi takes the
//   ROWS w*S .. w*S + S - 1 (read extraction's select, src/bwt/query.cpp:72-80): a WHOLE or CHUNK line then holds
//   at most 88 pieces of its own and the hint in dwords 30 and 31; a FAR window line 84 pieces, the hint in dwords
//   29 and 30 and its link in dword 31 as ever.  Those rows are consecutive occurrences of one symbol f (the F
//   symbol of row w*S), which lie in a few consecutive windows of the BWT: first dword = the window w0 of the
//   first of them (HINT_NONE: the slot is not filled), second dword = four bytes b0..b3, b_j = (K_j - 1) >> s with
//   K_j = how many of them lie in windows <= w0 + j (255 when that is all of them) and s = hint_shift(S), the
//   smallest shift that brings a row offset below 256.  Row w*S + r goes to window w0 + #{j : r >= K_j}: at least
//   w0 + #{j : (r >> s) > b_j}, at most that plus #{j : (r >> s) == b_j} -- and past b3 the upper bound is only a
//   first guess (hint_windows).  Two ways a line gets one: a shard opened for reads (RSBWT_OPEN_READS) is laid out
//   with the room in EVERY window line (88 / 84 own pieces, the slot marked HINT_NONE by the builder and filled by
//   the hint pass that follows the build); any other shard gets hints on its first extraction, in the WHOLE lines
//   that happen to hold at most 88 pieces.  No search ever looks at a hint: a lookup never reads past the symbols
//   its line's own pieces hold (span, in the header), and the flag is a header bit every reader masks off.
//
// SPILL LINE (line 17g + 16 of group g) = chunks at even dwords, each
//   dword 0   totA | totC << 12 | (csym & 0xFF) << 24      tot_x = # of x in the 96 own pieces of
//   dword 1   totG | totT << 12 | (csym >> 8) << 24        the window's line; csym = symbols held
//   dwords 2.. the excess pieces, zero-padded to an even number of dwords (at most 6)
//
// FAR LINE (index >= first_far) = a window line whose counts are absolute at ITS first piece; it
//   may itself continue in another far line (kind = 2).
//
// K-MER TABLE (8 B per T-mer, 4^T entries): the interval findInterval returns for every string of
// T symbols over ACGT: bits 0..39 lower, bits 40..63 width = upper - lower + 1 (0 = empty, upper =
// lower - 1; RSBWT_KTAB_WIDE = not tabulated).  Code of a T-mer = its 2-bit packing.  The tables of
// the shards a set holds on one GPU are INTERLEAVED (entry of shard s for code c at [c * S + s]): a
// query's start records for all S shards then come out of one 8S-byte stretch -- one request
// instead of S random ones.
//
// GROUPED K-MER TABLE (KTAB_GROUPED: 12 B per four T-mers = 3 B per T-mer, which buys one level more -- one LF step
// less of every search -- out of the same HBM): the four T-mers that share their first T - 1 symbols and differ in the
// LAST one (the symbol backward search takes first, the high two bits of the code) are neighbours in the BWT's row
// order, and their intervals tile one stretch of rows: one record holds the stretch's first row and the four running
// widths.  Record of group g = code & (4^(T-1) - 1), 96 bits little-endian: bits 0..39 base (lower of the first
// non-empty sibling), then c_0..c_3 of 14 bits each, c_i = rows of siblings 0..i.  Sibling i = code >> (2T - 2) holds
// rows base + c_(i-1) .. base + c_i - 1.  What the record cannot say is left to the search itself (it starts from
// initInterval, query.cpp:18-21, like an untabulated k-mer, and ends on the same rows): an EMPTY sibling (the
// reference's empty interval depends on the step it died at), a group of 16383 rows or more, siblings that do not
// tile (c_3 = 0x3FFF marks the whole record so).  Interleaved like the plain table: record of shard s at
// [(g * S + s) * 12].
//
// Everything below is plain C++ usable on the host and in kernels: the layout logic (what the
// builder writes, what a scalar reader finds) is one piece of code for both, so tests can hold it
// to naive ranks on the CPU while all queries run on the GPU only.
#ifndef RSBWT_LINE_FORMAT_H
#define RSBWT_LINE_FORMAT_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RSB_HD __host__ __device__ inline
#else
#define RSB_HD inline
#endif

namespace rsb {

constexpr uint32_t LINE_BYTES = 128;
constexpr uint32_t LINE_DWORDS = 32;
constexpr uint32_t HDR_DWORDS = 8;
constexpr uint32_t LINE_PIECES = 96;
constexpr uint32_t QUARTER_PIECES = 24;
constexpr uint32_t FAR_PIECES = 92;    // own pieces of a line whose last dword is a far link
constexpr uint32_t CHUNK_MAX_PIECES = 24;
constexpr uint32_t GROUP_SHIFT = 4;    // 16 window lines + 1 spill line
constexpr uint32_t GROUP = 1u << GROUP_SHIFT;
constexpr uint32_t KIND_WHOLE = 0, KIND_CHUNK = 1, KIND_FAR = 2;
constexpr uint32_t HINT_PIECES = 88;       // own pieces of a WHOLE / CHUNK line whose last two dwords are a psi hint
constexpr uint32_t FAR_HINT_PIECES = 84;   // own pieces of a FAR window line that carries one (dwords 29, 30; 31 = link)
constexpr uint32_t HINT_META0_BIT = 21;    // bit of meta_0 (= bit 29 of dword 1) saying so
constexpr uint32_t HINT_NONE = 0xFFFFFFFFu;  // first hint dword of a slot that is not filled
constexpr uint32_t COUNT_BITS = 40;
constexpr uint64_t COUNT_MASK = (1ull << COUNT_BITS) - 1;
constexpr uint64_t MAX_SYMBOLS = 1ull << 40;  // per shard; counts are 40-bit
constexpr uint32_t MAX_SPAN = 2944;           // S <= 92 * 32: a line of full units never needs more
constexpr uint32_t KTAB_WIDE = 0xFFFFFFu;
constexpr uint32_t KTAB_PLAIN = 0, KTAB_GROUPED = 1;  // shard_view::ktab_fmt
constexpr uint32_t KTAB_GROUP_BYTES = 12;
constexpr uint32_t KTAB_CUM_BITS = 14, KTAB_CUM_ESCAPE = (1u << KTAB_CUM_BITS) - 1u;

// w = p / S.  Kernels compute it as (uint32)((double)p * inv) plus one fix-up step: inv is 1/S
// rounded down by 2^-50 relative, so for p < 2^40 the product's floor is w or w - 1, never above.
struct span_params {
    uint32_t S, reserved;
    double inv;
};

// What a kernel needs to search one shard.  Plain pointers into HBM.
struct shard_view {
    const uint32_t *lines;  // nlines * 32 dwords
    uint64_t n;             // symbols (getBWLen)
    uint64_t nwin;          // windows = ceil(n / S)
    uint64_t nlines;        // ngroups * 17 + far lines
    uint64_t first_far;     // = ngroups * 17
    span_params sp;
    uint32_t ktab_depth;    // T (0 = no table)
    uint32_t ktab_stride;   // entries between consecutive T-mers: 1, or the number of shards whose tables are
                            // interleaved (a shard set's shards on one GPU: entry(code) = ktab[code * stride])
    const uint64_t *ktab;
    uint32_t ktab_fmt;      // KTAB_PLAIN (8-byte entries) / KTAB_GROUPED (12-byte records of four siblings)
    uint32_t reserved0;
    uint64_t C[5];          // C[c] = # symbols with rank < c   (getPC)
    uint64_t total[5];      // occurrences of each symbol in the whole BWT
    uint32_t sel_shift;     // one select sample per 2^sel_shift occurrences of a symbol (SEL_SHIFT_DENSE / _SPARSE)
    uint32_t hint_room;     // 1: every window line was laid out with room for a psi hint (RSBWT_OPEN_READS)
    const uint64_t *sel;    // the select samples, 5 x sel_stride words (nullptr until built: read extraction, getOccAt)
    uint64_t sel_stride;    // entries per symbol (select_stride)
};

// bytes of one shard's k-mer table of depth T
RSB_HD uint64_t ktab_bytes(uint32_t fmt, uint32_t T) {
    return fmt == KTAB_GROUPED ? (uint64_t)KTAB_GROUP_BYTES << (2u * (T - 1u)) : 8ull << (2u * T);
}
// The grouped record of four siblings from their (lower, upper) -- live[i] = lower <= upper -- as three dwords.
RSB_HD void ktab_group_encode(const uint64_t lo[4], const uint64_t up[4], uint32_t rec[3]) {
    uint64_t base = 0, next = 0, cum[4];
    bool any = false, ok = true;
    for (int i = 0; i < 4; ++i) {
        if (lo[i] <= up[i] && up[i] != ~0ull) {
            if (!any) { base = lo[i]; next = lo[i]; any = true; }
            if (lo[i] != next) ok = false;  // the siblings do not tile: never for a sound BWT
            next = up[i] + 1ull;
        }
        cum[i] = next - base;
    }
    if (!ok || cum[3] >= KTAB_CUM_ESCAPE || base > COUNT_MASK) {
        base = 0;
        cum[0] = cum[1] = cum[2] = cum[3] = KTAB_CUM_ESCAPE;
    }
    const uint64_t hi = (base >> 32) | (cum[0] << 8) | (cum[1] << 22) | (cum[2] << 36) | (cum[3] << 50);
    rec[0] = (uint32_t)base;
    rec[1] = (uint32_t)hi;
    rec[2] = (uint32_t)(hi >> 32);
}
// Sibling `slot` of a grouped record, in the plain table's 8-byte form (lower | width << 40; KTAB_WIDE = ask the search)
RSB_HD uint64_t ktab_group_entry(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t slot) {
    const uint64_t hi = ((uint64_t)d2 << 32) | d1;
    const uint64_t base = ((hi & 0xFFull) << 32) | d0;
    const uint32_t c3 = (uint32_t)(hi >> 50) & KTAB_CUM_ESCAPE;
    const uint32_t chi = (uint32_t)(hi >> (8u + KTAB_CUM_BITS * slot)) & KTAB_CUM_ESCAPE;
    const uint32_t clo = slot ? (uint32_t)(hi >> (KTAB_CUM_BITS * slot - 6u)) & KTAB_CUM_ESCAPE : 0u;
    // (base + chi past 2^40: never written by the builder -- a damaged record must not wrap into a row number that is believed)
    if (c3 == KTAB_CUM_ESCAPE || chi <= clo || base + chi > COUNT_MASK + 1ull) return (uint64_t)KTAB_WIDE << COUNT_BITS;
    return (base + clo) | ((uint64_t)(chi - clo) << COUNT_BITS);
}

RSB_HD uint64_t window_of(const span_params &sp, uint64_t p) { return p / sp.S; }
RSB_HD uint64_t line_of_window(uint64_t w) { return w + (w >> GROUP_SHIFT); }
RSB_HD uint64_t spill_line_of_window(uint64_t w) { return ((w >> GROUP_SHIFT) * (GROUP + 1)) + GROUP; }
// pieces a line holds itself, and where its psi hint sits when it has one (above, PSI HINT)
RSB_HD uint32_t own_pieces(uint32_t kind, uint32_t hint) {
    return kind == KIND_FAR ? (hint ? FAR_HINT_PIECES : FAR_PIECES) : (hint ? HINT_PIECES : LINE_PIECES);
}
RSB_HD uint32_t hint_dword(uint32_t kind) { return kind == KIND_FAR ? LINE_DWORDS - 3u : LINE_DWORDS - 2u; }
// the shift that brings a row offset r < S below 256 (the hint's boundaries are bytes)
RSB_HD uint32_t hint_shift(uint32_t S) {
    uint32_t s = 0;
    while (((S - 1u) >> s) > 255u) ++s;
    return s;
}

struct line_meta {
    uint64_t cnt[4];
    uint32_t s1, s2, s3, span, kind, cdw, hint;
    uint32_t half[4];
};

RSB_HD line_meta parse_line(const uint32_t *L) {
    line_meta m;
    uint32_t meta[4];
    for (int t = 0; t < 4; ++t) {
        m.cnt[t] = ((uint64_t)(L[2 * t + 1] & 0xFFu) << 32) | L[2 * t];
        meta[t] = L[2 * t + 1] >> 8;
    }
    m.s1 = meta[0] & 0x3FFu;
    m.s2 = (meta[0] >> 10) & 0x7FFu;
    m.hint = (meta[0] >> HINT_META0_BIT) & 1u;
    m.s3 = m.s2 + (meta[1] & 0x3FFu);
    m.span = m.s3 + ((meta[1] >> 10) & 0x3FFu);
    m.kind = (meta[1] >> 20) & 3u;
    m.half[0] = meta[2] & 0x7FFu;
    m.half[1] = (meta[2] >> 11) & 0x7FFu;
    m.half[2] = meta[3] & 0x7FFu;
    m.half[3] = (meta[3] >> 11) & 0x7FFu;
    m.cdw = 2u * (((meta[2] >> 22) & 3u) | (((meta[3] >> 22) & 3u) << 2));
    return m;
}

RSB_HD uint32_t dword_piece(const uint32_t *D, uint32_t i) { return (D[i >> 2] >> (8u * (i & 3u))) & 0xFFu; }

// Visits the pieces of window w in order -- own pieces, then the spill chunk or the far lines --
// until f(sym, len) returns true.
template <class F>
RSB_HD void walk_window(const shard_view &v, uint64_t w, F &&f) {
    uint64_t line = line_of_window(w);
    for (uint32_t guard = 0; guard < 64; ++guard) {  // a window has at most S <= 2944 pieces = 33 lines
        const uint32_t *L = v.lines + line * LINE_DWORDS;
        const line_meta m = parse_line(L);
        const uint32_t own = own_pieces(m.kind, m.hint);
        for (uint32_t i = 0; i < own; ++i) {
            const uint32_t u = dword_piece(L + HDR_DWORDS, i);
            if ((u & 31u) == 0u) break;
            if (f(u >> 5, u & 31u)) return;
        }
        if (m.kind == KIND_CHUNK) {
            const uint32_t *Cn = v.lines + spill_line_of_window(w) * LINE_DWORDS + m.cdw;
            const uint32_t csym = (Cn[0] >> 24) | ((Cn[1] >> 24) << 8);
            uint32_t seen = 0;
            // (the dword bound matters for a damaged header only: a built chunk ends inside its spill line)
            for (uint32_t i = 0; i < CHUNK_MAX_PIECES && seen < csym && m.cdw + 2u + (i >> 2) < LINE_DWORDS; ++i) {
                const uint32_t u = dword_piece(Cn + 2, i);
                if ((u & 31u) == 0u) break;
                seen += u & 31u;
                if (f(u >> 5, u & 31u)) return;
            }
            return;
        }
        if (m.kind != KIND_FAR) return;
        line = L[LINE_DWORDS - 1];
        if (line < v.first_far || line >= v.nlines) return;  // corrupt link: never for a built index
    }
}

// # of symbol b (rank 0..4) in BWT[0, w*S)
RSB_HD uint64_t count_before_window(const shard_view &v, uint64_t w, uint32_t b) {
    if (w >= v.nwin) return v.total[b];
    const uint32_t *L = v.lines + line_of_window(w) * LINE_DWORDS;
    if (b != 0u) return (((uint64_t)(L[2 * (b - 1) + 1] & 0xFFu)) << 32) | L[2 * (b - 1)];
    uint64_t s = 0;
    for (int t = 0; t < 4; ++t) s += (((uint64_t)(L[2 * t + 1] & 0xFFu)) << 32) | L[2 * t];
    return w * v.sp.S - s;
}

// psi hint (above): where psi takes the r-th row of the hinted window (r = 0 .. S-1): a window in lo..hi; `open`:
// the row lies past the hint's last boundary and hi is only the first window to try -- it lies there or further on
// (measured on the population stream: 15 % of the rows are past the fourth boundary -- a window holds 114 +- 44 of
// a block's 467 rows -- and four in five of those lie in the very next window; extrapolating from the windows the
// hint knows does no better than that).  lo == hi and not open: the hint settles the row.  (The boundaries are
// non-decreasing, so the comparisons count.)
struct hint_range {
    uint32_t lo, hi;
    bool open;
};
RSB_HD hint_range hint_windows(uint32_t w0, uint32_t kk, uint32_t r, uint32_t shift) {
    const uint32_t rq = r >> shift;
    const uint32_t b0 = kk & 0xFFu, b1 = (kk >> 8) & 0xFFu, b2 = (kk >> 16) & 0xFFu, b3 = kk >> 24;
    const uint32_t gt = (rq > b0 ? 1u : 0u) + (rq > b1 ? 1u : 0u) + (rq > b2 ? 1u : 0u) + (rq > b3 ? 1u : 0u);
    const uint32_t eq = (rq == b0 ? 1u : 0u) + (rq == b1 ? 1u : 0u) + (rq == b2 ? 1u : 0u) + (rq == b3 ? 1u : 0u);
    hint_range h;
    h.lo = w0 + gt;
    h.hi = h.lo + eq;
    h.open = rq >= b3;
    return h;
}

// RLEBWT::getOcc (src/bwt/rlebwt.cpp:268-301): # of symbol b in BWT[0..p], p < n.
RSB_HD uint64_t view_occ(const shard_view &v, uint32_t b, uint64_t p) {
    const uint64_t w = window_of(v.sp, p);
    uint32_t rem = (uint32_t)(p - w * v.sp.S) + 1u;
    uint64_t occ = count_before_window(v, w, b);
    walk_window(v, w, [&](uint32_t sym, uint32_t len) {
        const uint32_t take = len < rem ? len : rem;
        if (sym == b) occ += take;
        rem -= take;
        return rem == 0u;
    });
    return occ;
}

// RLEBWT::getChar (src/bwt/rlebwt.cpp:202-227): rank of the symbol at position p < n.
RSB_HD uint32_t view_char(const shard_view &v, uint64_t p) {
    const uint64_t w = window_of(v.sp, p);
    uint32_t rem = (uint32_t)(p - w * v.sp.S) + 1u, c = 0;
    walk_window(v, w, [&](uint32_t sym, uint32_t len) {
        c = sym;
        if (len >= rem) return true;
        rem -= len;
        return false;
    });
    return c;
}

// getChar and getOcc of that symbol at once: the LF step of extractPrefix (query.cpp:49-57).
RSB_HD uint32_t view_char_occ(const shard_view &v, uint64_t p, uint64_t *occ_of_char) {
    const uint64_t w = window_of(v.sp, p);
    uint32_t rem = (uint32_t)(p - w * v.sp.S) + 1u, c = 0;
    uint32_t in[5] = {0, 0, 0, 0, 0};
    walk_window(v, w, [&](uint32_t sym, uint32_t len) {
        c = sym;
        const uint32_t take = len < rem ? len : rem;
        if (sym < 5u) in[sym] += take;
        rem -= take;
        return rem == 0u;
    });
    *occ_of_char = count_before_window(v, w, c) + in[c];
    return c;
}

// RLEBWT::getOccAt (src/bwt/rlebwt.cpp:233-266): position of the bc-th b (1 <= bc <= total[b]).
// The window is found by a floor search over the window headers between lo and hi (BPTree::select's
// role, include/bwt/BPTree.h:50-67); callers narrow [lo, hi] with a sample table.
RSB_HD uint64_t view_occ_at(const shard_view &v, uint32_t b, uint64_t bc, uint64_t lo, uint64_t hi) {
    while (hi > lo) {  // largest w in [lo, hi] with count_before(w) < bc
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (count_before_window(v, mid, b) >= bc) hi = mid - 1;
        else lo = mid;
    }
    uint64_t left = bc - count_before_window(v, lo, b);
    uint64_t pos = lo * v.sp.S;
    bool found = false;
    walk_window(v, lo, [&](uint32_t sym, uint32_t len) {
        if (sym == b) {
            if (left <= len) {
                pos += left - 1;
                found = true;
                return true;
            }
            left -= len;
        }
        pos += len;
        return false;
    });
    return found ? pos : v.n;
}

// ---------------------------------------------------------------------------------------------
// Select samples and psi hints (read extraction's getOccAt, src/bwt/rlebwt.cpp:233-266): what the builder
// kernels write (kernels.hip) and the host-side layout test checks, one piece of code for both.
// ---------------------------------------------------------------------------------------------
// One select sample per 2^sel_shift occurrences of every symbol (shard_view::sel_shift): DENSE for a shard whose
// extraction leans on them (8 bytes per 256 occurrences = n / 32 bytes: the sample names the window of ANY occurrence
// of its block, see below), SPARSE for a shard laid out with a psi hint in every window line (RSBWT_OPEN_READS:
// n / 512 bytes), where a sample is only the way out for the few rows a hint does not settle.
constexpr uint32_t SEL_SHIFT_DENSE = 8, SEL_SHIFT_SPARSE = 12;

// entries per symbol of the sample table
RSB_HD uint64_t select_stride(const shard_view &ix) {
    uint64_t mx = 0;
    for (int c = 0; c <= 4; ++c) mx = ix.total[c] > mx ? ix.total[c] : mx;
    return (mx >> ix.sel_shift) + 2;
}

// Sample word of the block of 2^shift occurrences of a symbol that starts with occurrence (m << shift) + 1:
// bits 0..31 = the window w0 holding that first occurrence, then four bytes k0..k3 with k_j + 1 = how many of
// the block's occurrences lie in windows <= w0 + j (capped at 256).  The window of the block's r-th occurrence
// (r = 0 .. 2^shift - 1) is w0 + [r > k0] + [r > k1] + [r > k2] + [r > k3] -- EXACT while r <= k3, a lower bound
// beyond (the block's first 256 occurrences spread over more than five windows, or r >= 256 in a sparse table --
// where a capped count byte, 255, no longer counts).
RSB_HD uint32_t sample_window(uint64_t word, uint64_t bc, uint32_t shift, bool *exact) {
    const uint32_t r = (uint32_t)((bc - 1) & ((1ull << shift) - 1ull));
    const uint32_t k0 = (uint32_t)(word >> 32) & 0xFFu, k1 = (uint32_t)(word >> 40) & 0xFFu;
    const uint32_t k2 = (uint32_t)(word >> 48) & 0xFFu, k3 = (uint32_t)(word >> 56);
    *exact = r <= k3;
    // (a count byte of 255 is capped: it says "256 or more", and nothing about an r beyond 255 -- sparse tables)
    return (uint32_t)word + (r > k0 && k0 != 255u ? 1u : 0u) + (r > k1 && k1 != 255u ? 1u : 0u) +
           (r > k2 && k2 != 255u ? 1u : 0u) + (r > k3 && k3 != 255u ? 1u : 0u);
}

// The sample words of the blocks of symbol c whose first occurrence lies in window w: emit(m, word).
template <class E>
RSB_HD void window_samples(const shard_view &ix, uint64_t w, uint32_t c, E &&emit) {
    // (held to the symbol's total: the sample index is then inside the table whatever the lines say -- a
    // damaged count word would otherwise send the caller writing far outside it)
    const uint32_t shift = ix.sel_shift;
    const uint64_t tc = ix.total[c];
    const uint64_t cb = count_before_window(ix, w, c);
    uint64_t ce = count_before_window(ix, w + 1, c);
    ce = ce < tc ? ce : tc;
    if (ce <= cb) return;
    // occurrences cb+1 .. ce live here; sample m is occurrence (m << shift) + 1
    uint64_t m = (cb + (1ull << shift) - 1) >> shift;
    if ((m << shift) >= ce) return;
    uint64_t upto[4] = {ce, 0, 0, 0};
    for (int j = 1; j < 4; ++j) {
        const uint64_t x = count_before_window(ix, w + 1 + j, c);
        upto[j] = x < tc ? x : tc;
        if (upto[j] < upto[j - 1]) upto[j] = upto[j - 1];  // monotone whatever the lines say
    }
    for (; (m << shift) < ce; ++m) {
        const uint64_t before = m << shift;  // occurrences before the block
        uint64_t word = w & 0xFFFFFFFFull;
        for (int j = 0; j < 4; ++j) {
            uint64_t kj = upto[j] - before;  // >= 1 for j = 0
            kj = kj > 256 ? 256 : kj;
            word |= (kj - 1) << (32 + 8 * j);
        }
        emit(m, word);
    }
}

// The window that holds the bc-th b (1 <= bc <= total[b]): the sample names it, or bounds it together with the next
// sample -- then the floor search over the window headers in between (BPTree::select's role, BPTree.h:50-67).
RSB_HD uint64_t select_window(const shard_view &ix, const uint64_t *sel, uint64_t stride_m, uint32_t b, uint64_t bc) {
    const uint64_t m = (bc - 1) >> ix.sel_shift;
    bool exact;
    uint64_t lo = sample_window(sel[b * stride_m + m], bc, ix.sel_shift, &exact);
    if (lo >= ix.nwin) lo = ix.nwin - 1;
    if (exact) return lo;
    uint64_t hi = ((m + 1) << ix.sel_shift) < ix.total[b] ? (uint32_t)sel[b * stride_m + m + 1] : ix.nwin - 1;
    if (hi >= ix.nwin) hi = ix.nwin - 1;
    if (hi < lo) hi = lo;
    while (hi > lo) {  // largest w in [lo, hi] with count_before(w) < bc
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (count_before_window(ix, mid, b) >= bc) hi = mid - 1;
        else lo = mid;
    }
    return lo;
}

// The psi hint of window w (above, WINDOW LINE): true when the line may carry one -- it was laid out with the room
// (its flag is set, its slot still says HINT_NONE) or it is a WHOLE line of at most 88 pieces -- with its two dwords.
RSB_HD bool window_psi_hint(const shard_view &ix, const uint64_t *sel, uint64_t stride_m, uint64_t w, uint32_t *w0_out,
                            uint32_t *kk_out) {
    const uint32_t S = ix.sp.S;
    const uint32_t *Ln = ix.lines + line_of_window(w) * LINE_DWORDS;
    const line_meta m = parse_line(Ln);
    if (m.hint) {
        if (Ln[hint_dword(m.kind)] != HINT_NONE) return false;  // filled already
    } else {
        if (m.kind != KIND_WHOLE) return false;
        uint32_t np = 0;  // pieces the line holds: a hint needs the last 8 piece bytes free
        for (uint32_t i = 0; i < LINE_PIECES; ++i) {
            if ((dword_piece(Ln + HDR_DWORDS, i) & 31u) == 0u) break;
            ++np;
        }
        if (np > HINT_PIECES) return false;
    }
    const uint64_t r0 = w * (uint64_t)S;
    uint32_t f = 0;
    while (f < 4u && ix.C[f + 1] <= r0) ++f;  // F symbol of row r0 (getF, rlebwt.cpp:307-314)
    if (f == 0u) return false;                 // '$' rows end a walk: nobody takes psi of them
    const uint64_t tot = ix.total[f];
    if (r0 < ix.C[f]) return false;
    const uint64_t bc0 = r0 - ix.C[f] + 1ull;  // row r0 is the bc0-th f
    if (bc0 > tot) return false;
    uint64_t seff = tot - bc0 + 1ull;          // rows of this window that belong to f's block
    if (seff > S) seff = S;
    if (r0 + seff > ix.n) seff = ix.n - r0;
    const uint64_t w0 = select_window(ix, sel, stride_m, f, bc0);
    // the window must be right: count(w0) < bc0 <= count(w0 + 1)
    if (!(count_before_window(ix, w0, f) < bc0 && bc0 <= count_before_window(ix, w0 + 1, f))) return false;
    const uint32_t shift = hint_shift(S);
    uint32_t kk = 0;
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t c = count_before_window(ix, w0 + 1 + j, f);
        uint64_t upto = c < bc0 - 1ull ? 0ull : c - (bc0 - 1ull);  // of the block's occurrences, those in windows <= w0 + j
        if (upto > seff) upto = seff;
        // b_j = (K_j - 1) >> shift; 255 = "all of them" (then every row of the window is at or before window w0 + j)
        const uint32_t bj = upto >= seff ? 255u : (uint32_t)(((upto ? upto : 1ull) - 1ull) >> shift);
        kk |= bj << (8u * j);
    }
    *w0_out = (uint32_t)w0;
    *kk_out = kk;
    return true;
}

// ---------------------------------------------------------------------------------------------
// The builder's core, shared by the GPU kernels (build_lines.hip) and the host-side layout test.
// ---------------------------------------------------------------------------------------------

// Sequential reader of the run bytes from a known position on.
struct run_reader {
    const uint8_t *runs;
    uint64_t R;
    uint64_t r;       // next run byte to read
    uint32_t left;    // symbols of the current run not yet handed out
    uint32_t sym;
    uint64_t cnt[4];  // A,C,G,T handed out so far (absolute)

    // starts at run index r0, `skip` symbols into it, with the counts of everything before
    RSB_HD void start(const uint8_t *runs_, uint64_t R_, uint64_t r0, const uint64_t cnt0[4]) {
        runs = runs_;
        R = R_;
        r = r0;
        left = 0;
        sym = 0;
        for (int c = 0; c < 4; ++c) cnt[c] = cnt0[c];
    }
    // next piece of at most `want` symbols; 0 at the end of the stream
    RSB_HD uint32_t take(uint32_t want, uint32_t *piece_sym) {
        while (left == 0u) {
            if (r >= R) return 0u;
            const uint8_t u = runs[r++];
            left = u & 31u;  // zero-length units hold no symbol: skipped
            sym = u >> 5;
        }
        const uint32_t len = left < want ? left : want;
        left -= len;
        *piece_sym = sym;
        if (sym >= 1u && sym <= 4u) cnt[sym - 1u] += len;
        return len;
    }
    RSB_HD void skip_symbols(uint64_t k) {
        uint32_t s;
        while (k) {
            const uint32_t len = take(k > 31u ? 31u : (uint32_t)k, &s);
            if (!len) return;
            k -= len;
        }
    }
};

struct group_stats {
    uint32_t far_lines;       // far lines this group needs
    uint32_t chunk_windows;   // windows continued in the spill line
    uint32_t far_windows;     // windows continued in far lines
    uint64_t spilled_symbols; // symbols not held by their window's own line
};

// Packs one window / far line: counts at its first piece, its np pieces, how it continues.  hint_slot: the line is
// laid out with room for a psi hint (np <= own_pieces(kind, 1)): flag set, slot marked HINT_NONE.
RSB_HD void emit_line(uint32_t *L, const uint64_t cnt0[4], const uint8_t *pieces, uint32_t np, uint32_t kind,
                      uint32_t cdw, uint32_t far_link, bool hint_slot) {
    uint32_t start[5] = {0, 0, 0, 0, 0};
    uint32_t half[4] = {0, 0, 0, 0};
    uint32_t sofar = 0;
    for (uint32_t q = 0; q < 4; ++q) {
        start[q] = sofar;
        for (uint32_t i = q * QUARTER_PIECES; i < (q + 1) * QUARTER_PIECES && i < np; ++i) {
            const uint32_t len = pieces[i] & 31u, sym = pieces[i] >> 5;
            sofar += len;
            if (q < 2 && sym >= 1u && sym <= 4u) half[sym - 1u] += len;
        }
    }
    start[4] = sofar;
    const uint32_t meta[4] = {start[1] | (start[2] << 10) | ((hint_slot ? 1u : 0u) << HINT_META0_BIT),
                              (start[3] - start[2]) | ((start[4] - start[3]) << 10) | (kind << 20),
                              half[0] | (half[1] << 11) | (((cdw >> 1) & 3u) << 22),
                              half[2] | (half[3] << 11) | (((cdw >> 3) & 3u) << 22)};
    for (uint32_t t = 0; t < 4; ++t) {
        const uint64_t word = (cnt0[t] & COUNT_MASK) | ((uint64_t)meta[t] << COUNT_BITS);
        L[2 * t] = (uint32_t)word;
        L[2 * t + 1] = (uint32_t)(word >> 32);
    }
    for (uint32_t d = 0; d < LINE_DWORDS - HDR_DWORDS; ++d) {
        uint32_t x = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = 4 * d + k;
            if (i < np) x |= (uint32_t)pieces[i] << (8 * k);
        }
        L[HDR_DWORDS + d] = x;
    }
    if (hint_slot) {
        L[hint_dword(kind)] = HINT_NONE;
        L[hint_dword(kind) + 1u] = 0u;
    }
    if (kind == KIND_FAR) L[LINE_DWORDS - 1] = far_link;
}

// Lays out group g (windows 16g .. 16g+15).  `rd` stands at symbol 16*g*S.  With WRITE the 17 lines
// of the group and its far lines (from absolute line far_base on) are written; without, only the
// statistics are gathered -- by the same decisions, so that a counting pass sizes the far region.
// `room`: every window line keeps its last 8 piece bytes for a psi hint (RSBWT_OPEN_READS).
template <bool WRITE>
RSB_HD group_stats build_group(const span_params &sp, uint64_t n, uint64_t nwin, uint64_t g, run_reader &rd,
                               uint32_t *lines, uint64_t far_base, bool room = false) {
    group_stats st = {0, 0, 0, 0};
    uint32_t spill[LINE_DWORDS];
    for (uint32_t d = 0; d < LINE_DWORDS; ++d) spill[d] = 0;
    uint32_t sdw = 0;  // next free dword of the spill line (even)
    constexpr uint32_t BUF = LINE_PIECES + CHUNK_MAX_PIECES;
    uint8_t buf[BUF];
    for (uint32_t wi = 0; wi < GROUP; ++wi) {
        const uint64_t w = g * GROUP + wi;
        if (w >= nwin) break;
        const uint64_t wstart = w * (uint64_t)sp.S;
        uint32_t remaining = (uint32_t)((n - wstart) < sp.S ? (n - wstart) : sp.S);
        const uint32_t wsyms = remaining;
        uint64_t cnt_line[4] = {rd.cnt[0], rd.cnt[1], rd.cnt[2], rd.cnt[3]};
        uint64_t line = line_of_window(w);
        uint32_t nb = 0;
        bool first = true;
        for (;;) {
            while (remaining && nb < BUF) {
                uint32_t s;
                const uint32_t len = rd.take(remaining, &s);
                if (!len) { remaining = 0; break; }
                buf[nb++] = (uint8_t)((s << 5) | len);
                remaining -= len;
            }
            const bool ends = remaining == 0u;
            const bool slot = room && first;  // only a window's first line is ever looked at for a hint
            const uint32_t cap = own_pieces(KIND_WHOLE, slot ? 1u : 0u);
            uint32_t keep, kind = KIND_WHOLE, cdw = 0, link = 0;
            if (ends && nb <= cap) {
                keep = nb;
            } else {
                const uint32_t extra = nb > cap ? nb - cap : 0u;
                const uint32_t need = 2u + 2u * ((extra + 7u) / 8u);  // header + pieces, in dwords, even
                if (first && ends && extra <= CHUNK_MAX_PIECES && sdw + need <= LINE_DWORDS) {
                    keep = cap;
                    kind = KIND_CHUNK;
                    cdw = sdw;
                    uint32_t tot[4] = {0, 0, 0, 0}, csym = 0;
                    for (uint32_t i = 0; i < cap; ++i) {
                        const uint32_t sy = buf[i] >> 5;
                        if (sy >= 1u && sy <= 4u) tot[sy - 1u] += buf[i] & 31u;
                    }
                    for (uint32_t i = cap; i < nb; ++i) {
                        csym += buf[i] & 31u;
                        spill[sdw + 2u + ((i - cap) >> 2)] |= (uint32_t)buf[i] << (8u * ((i - cap) & 3u));
                    }
                    spill[sdw] = tot[0] | (tot[1] << 12) | ((csym & 0xFFu) << 24);
                    spill[sdw + 1] = tot[2] | (tot[3] << 12) | ((csym >> 8) << 24);
                    sdw += need;
                    st.chunk_windows += 1;
                    st.spilled_symbols += csym;
                } else {
                    keep = own_pieces(KIND_FAR, slot ? 1u : 0u);
                    kind = KIND_FAR;
                    link = (uint32_t)(far_base + st.far_lines);
                    st.far_lines += 1;
                    if (first) st.far_windows += 1;
                }
            }
            uint32_t held = 0;
            for (uint32_t i = 0; i < keep; ++i) held += buf[i] & 31u;
            if (first && kind == KIND_FAR) st.spilled_symbols += wsyms - held;
            if (WRITE) emit_line(lines + line * LINE_DWORDS, cnt_line, buf, keep, kind, cdw, link, slot);
            if (kind != KIND_FAR) break;
            // the window goes on in the far line: counts at its first piece, pieces shifted down
            for (uint32_t i = 0; i < keep; ++i) {
                const uint32_t sy = buf[i] >> 5;
                if (sy >= 1u && sy <= 4u) cnt_line[sy - 1u] += buf[i] & 31u;
            }
            for (uint32_t i = keep; i < nb; ++i) buf[i - keep] = buf[i];
            nb -= keep;
            line = link;
            first = false;
        }
    }
    if (WRITE && sdw) {
        uint32_t *Sp = lines + (g * (GROUP + 1) + GROUP) * LINE_DWORDS;
        for (uint32_t d = 0; d < LINE_DWORDS; ++d) Sp[d] = spill[d];
    }
    return st;
}

inline span_params make_span(uint32_t S) {
    span_params sp;
    sp.S = S < 2u ? 2u : (S > MAX_SPAN ? MAX_SPAN : S);
    sp.reserved = 0;
    sp.inv = (1.0 / (double)sp.S) * (1.0 - 0x1p-50);
    return sp;
}

}  // namespace rsb
#endif
