// extract_lines.hip -- batched read extraction over window lines, wave-cooperative (gfx950).
//
// extractPrefix + extractPostfix, src/bwt/query.cpp:43-85, for a batch of SA rows: the read whose
// suffix is row i = the LF walk left until '$' (getChar + getOcc per step, query.cpp:49-57) followed by
// the psi walk right until '$' (getF + getOccAt per step, query.cpp:72-80).
//   * every lane walks one row; rows are handed out dynamically (one counter, ballot + popcount), so a
//     wave is not held by its longest walk;
//   * a step's window line is fetched the way the search kernel fetches its lines (wave_lines.h: a full
//     128-byte line per octet of lanes, direct to LDS) and parsed lane-privately: the header names the
//     quarter, 24 pieces are scanned; a position past its line's own pieces continues lazily (spill
//     chunk / far line) in the lane's next pass;
//   * LF step (prefix): one line gives the symbol at the position AND its rank (usually one pass);
//   * psi step (postfix): sampled select (one sample per 256 occurrences) bounds the window; the
//     window is found by probing count words (8-byte loads that fly with the other lanes' line
//     fetches), then selected in: count words of three quarter boundaries from the header and two
//     v_dot4 sums, one quarter scanned.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "line_format.h"
#include "rank_device.h"
#include "wave_lines.h"

namespace rsb {


// the header fields a lane-private parse needs
struct line_head {
    uint32_t s1, s2, s3, span, kind;
};
__device__ __forceinline__ line_head read_head(const staged_line &L) {
    const uint4 h0 = L.u4(0);
    const uint32_t m0 = h0.y >> 8, m1 = h0.w >> 8;
    line_head h;
    h.s1 = m0 & 0x3FFu;
    h.s2 = (m0 >> 10) & 0x7FFu;
    h.s3 = h.s2 + (m1 & 0x3FFu);
    h.span = h.s3 + ((m1 >> 10) & 0x3FFu);
    h.kind = (m1 >> 20) & 3u;
    return h;
}
__device__ __forceinline__ uint64_t read_count(const staged_line &L, uint32_t b) {  // b = 1..4
    const uint2 cw = L.u2(2u * (b - 1u));
    return ((uint64_t)(cw.y & 0xFFu) << 32) | cw.x;
}
__device__ __forceinline__ uint32_t read_half(const staged_line &L, uint32_t b) {
    const uint32_t hm = L.dword(5u + 2u * ((b - 1u) >> 1)) >> 8;
    return (hm >> (11u * ((b - 1u) & 1u))) & 0x7FFu;
}
__device__ __forceinline__ uint32_t read_chunk_dword(const staged_line &L) {
    const uint32_t m2 = L.dword(5) >> 8, m3 = L.dword(7) >> 8;
    return 2u * (((m2 >> 22) & 3u) | (((m3 >> 22) & 3u) << 2));
}

// rank (0..4) of the symbol of the piece holding the rem-th symbol (rem >= 1) of the 24 pieces r6
__device__ __forceinline__ uint32_t char_at24(const uint32_t r6[6], uint32_t rem) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const uint32_t u = __builtin_amdgcn_ubfe(r6[i >> 2], 8 * (i & 3), 8), len = u & 31u;
        const bool hit = (rem - 1u) < len;  // rem == 0 (already found) wraps to "no"
        c = hit ? (u >> 5) : c;
        rem = rem > len ? rem - len : 0u;
    }
    return c;
}

// position (symbols from the first of the 24 pieces r6) of the t-th b (t >= 1) among them; *left =
// what remains of t when they hold fewer (0: found)
__device__ __forceinline__ uint32_t select24(const uint32_t r6[6], uint32_t b, uint32_t t, uint32_t *left) {
    uint32_t pos = 0, prefix = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const uint32_t u = __builtin_amdgcn_ubfe(r6[i >> 2], 8 * (i & 3), 8), len = u & 31u;
        const bool act = (u >> 5) == b && t != 0u;  // t == 0: found already
        const bool hit = act && t <= len;
        pos = hit ? prefix + t - 1u : pos;
        t = hit ? 0u : (act ? t - len : t);
        prefix += len;
    }
    *left = t;
    return pos;
}

// Hands rows to the lanes that have none.  A wave draws chunks of ROW_CHUNK consecutive rows from the
// global counter (one atomic per chunk, not per pass: the atomic's round trip would otherwise sit in
// front of every pass's line fetch) and gives the next one to whichever lane is free.
constexpr uint32_t ROW_CHUNK = 256;
struct row_pool {
    uint64_t next = 0, end = 0;  // wave-uniform
    bool drained = false;
};
__device__ __forceinline__ bool draw_row(bool want, unsigned long long *pool, size_t n, uint32_t lane, row_pool &rp,
                                         size_t *row) {
    const uint64_t mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0ull) return false;
    if (rp.next >= rp.end && !rp.drained) {
        unsigned long long c = 0;
        if (lane == 0u) c = atomicAdd(pool, (unsigned long long)ROW_CHUNK);
        c = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
            __builtin_amdgcn_readfirstlane((uint32_t)c);
        rp.next = c;
        rp.end = c + ROW_CHUNK < n ? c + ROW_CHUNK : n;
        if (c >= n) {
            rp.drained = true;
            rp.next = rp.end = 0;
        }
    }
    const uint64_t mine = rp.next + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    const bool got = want && mine < rp.end;
    if (got) *row = (size_t)mine;
    const uint64_t taken = rp.next + __builtin_popcountll(mask);
    rp.next = taken < rp.end ? taken : rp.end;
    return got;
}

// ---------------------------------------------------------------------------------------------------
// extractPrefix (query.cpp:43-63): LF walk left until '$'.  The characters are produced right to
// left, so they are written downwards from the end of the row's buffer; plen = their number
// (UINT32_MAX: the walk does not fit `stride`, or the row is out of range).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * WG_WAVES)
extract_prefix_wave_kernel(const shard_view ix, const uint64_t *__restrict__ rows, size_t n, uint8_t *__restrict__ out,
                           uint32_t stride, uint32_t *__restrict__ plen, unsigned long long *__restrict__ pool) {
    __shared__ uint4 s_stage[WG_WAVES][64 * SLOT_U4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *stage = s_stage[wave];
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    const staged_line L = {own_stage_row(stage, lane), lane & 7u};
    const char *lines_bytes = reinterpret_cast<const char *>(ix.lines);
    const uint32_t S = ix.sp.S, nlines = (uint32_t)ix.nlines;
    const double inv = ix.sp.inv;
    uint32_t ctab_lo, ctab_hi;  // C[1..4] in lanes 0..3, read with ds_bpermute
    {
        const uint32_t l3 = lane & 3u;
        const uint64_t cv = l3 == 0u ? ix.C[1] : l3 == 1u ? ix.C[2] : l3 == 2u ? ix.C[3] : ix.C[4];
        ctab_lo = (uint32_t)cv;
        ctab_hi = (uint32_t)(cv >> 32);
    }
    bool have = false;
    row_pool rp;
    size_t r = 0;
    uint64_t idx = 0, acc = 0;
    uint32_t len = 0, c = 0, phase = 0;  // phase 0: symbol at idx; 1: rank of symbol c at idx
    uint32_t cont = 0, cblk = 0, cdw = 0, co = 0, tries = 0, w = 0;
    // characters are produced right to left: four at a time go out as one aligned dword when the row
    // buffers allow it (a byte store per character is a request per character)
    const bool packed_out = (stride & 3u) == 0u && ((uintptr_t)out & 3u) == 0u;
    uint32_t chars = 0;
    for (;;) {
        size_t nr = 0;
        if (draw_row(!have, pool, n, lane, rp, &nr)) {
            r = nr;
            idx = rows[r];
            len = 0;
            phase = 0;
            cont = 0;
            chars = 0;
            have = true;
            if (idx >= ix.n) {
                plen[r] = 0xFFFFFFFFu;
                have = false;
            }
        }
        if (__builtin_amdgcn_ballot_w64(have) == 0ull) {
            if (rp.drained) break;
            continue;
        }
        // ---- this lane's line
        uint32_t line = 0, o = 0;
        if (have && cont == 0u) {
            uint32_t pin;
            w = fast_window(idx, S, inv, pin);
            line = w + (w >> GROUP_SHIFT);
            o = pin + 1u;
            if (line >= nlines) line = 0;
            tries = 0;
        }
        const uint32_t want = have ? (cont ? cblk : line) : ~0u;
        glds_fetch(lines_bytes, want, lane, stage_lds);
        glds_wait();
        // ---- phase 0: the symbol at the position (RLEBWT::getChar, rlebwt.cpp:202-227)
        bool got_char = false, same_line = false;
        if (have && phase == 0u) {
            uint32_t dw = HDR_DWORDS, rem = 0;
            bool scan = false;
            if (cont != KIND_CHUNK) {
                const line_head h = read_head(L);
                const uint32_t oe = cont ? co : o;
                if (oe <= h.span) {
                    const uint32_t cq = (oe > h.s1 ? 1u : 0u) + (oe > h.s2 ? 1u : 0u) + (oe > h.s3 ? 1u : 0u);
                    const uint32_t start = cq == 0u ? 0u : cq == 1u ? h.s1 : cq == 2u ? h.s2 : h.s3;
                    dw = HDR_DWORDS + 6u * cq;
                    rem = oe - start;
                    scan = true;
                    same_line = cont == 0u;
                } else if (h.kind == KIND_FAR) {
                    cblk = L.dword(LINE_DWORDS - 1u);
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_FAR;
                    co = oe - h.span;
                } else if (h.kind == KIND_CHUNK && cont == 0u) {
                    cdw = read_chunk_dword(L);
                    cblk = (w >> GROUP_SHIFT) * (GROUP + 1u) + GROUP;
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_CHUNK;
                    co = oe - h.span;
                } else {
                    scan = true;  // beyond what the index holds: never for idx < n; ends the walk as '$'
                }
            } else {
                dw = cdw + 2u;
                rem = co;
                scan = true;
            }
            if (!scan && ++tries > 72u) scan = true;
            if (scan) {
                uint32_t r6[6];
                load24(L, dw, r6);
                c = rem ? char_at24(r6, rem) : 0u;
                got_char = true;
            }
        }
        if (got_char) {
            if (c == 0u || c > 4u) {  // '$': the read starts here (query.cpp:52)
                if (packed_out)  // the characters not yet written: the (len & 3) most recent ones
                    for (uint32_t t = 0; t < (len & 3u); ++t)
                        out[r * (size_t)stride + (stride - len + t)] = (uint8_t)(chars >> (8u * t));
                plen[r] = len;
                have = false;
            } else if (len == stride) {  // the reference would spin (query.cpp:48)
                plen[r] = 0xFFFFFFFFu;
                have = false;
            } else {
                phase = 1;
                if (!same_line) cont = 0;  // the rank starts over at the window's line, next pass
            }
        }
        // ---- phase 1: Occ(c, idx) (RLEBWT::getOcc, rlebwt.cpp:268-301), as in search_lines.hip
        const bool ranking = have && phase == 1u && (same_line || !got_char);
        bool done = false;
        uint64_t occ = 0;
        if (ranking) {
            bool scan = false;
            uint64_t base = 0;
            uint32_t dw = HDR_DWORDS, rem = 0;
            if (cont != KIND_CHUNK) {
                const line_head h = read_head(L);
                const uint32_t oe = cont ? co : o;
                const uint64_t cnt = read_count(L, c);
                if (oe <= h.span) {
                    const uint32_t cq = (oe > h.s1 ? 1u : 0u) + (oe > h.s2 ? 1u : 0u) + (oe > h.s3 ? 1u : 0u);
                    const uint32_t start = cq == 0u ? 0u : cq == 1u ? h.s1 : cq == 2u ? h.s2 : h.s3;
                    const uint32_t hb = read_half(L, c);
                    const uint32_t m = matched24(L, HDR_DWORDS + 6u * (cq & 2u), c);
                    base = cnt + (cq >= 2u ? hb : 0u) + ((cq & 1u) ? m : 0u);
                    dw = HDR_DWORDS + 6u * cq;
                    rem = oe - start;
                    scan = true;
                } else if (h.kind == KIND_FAR) {
                    cblk = L.dword(LINE_DWORDS - 1u);
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_FAR;
                    co = oe - h.span;
                } else if (h.kind == KIND_CHUNK && cont == 0u) {
                    acc = cnt;
                    cdw = read_chunk_dword(L);
                    cblk = (w >> GROUP_SHIFT) * (GROUP + 1u) + GROUP;
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_CHUNK;
                    co = oe - h.span;
                } else {
                    base = cnt;
                    scan = true;
                }
            } else {
                const uint2 hd = L.u2(cdw);
                const uint32_t hw = (c <= 2u) ? hd.x : hd.y;
                base = acc + ((hw >> (12u * ((c - 1u) & 1u))) & 0xFFFu);
                dw = cdw + 2u;
                rem = co;
                scan = true;
            }
            if (!scan && ++tries > 72u) scan = true;
            if (scan) {
                uint32_t r6[6];
                load24(L, dw, r6);
                occ = base + runs_scan<6>(r6, c, rem);
                done = true;
            }
        }
        // C[c], with every lane active (a ds_bpermute returns 0 from a masked-off source lane)
        const uint32_t ci = (c - 1u) & 3u;
        const uint64_t pc = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(ci << 2), (int)ctab_hi) << 32) |
                            (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ci << 2), (int)ctab_lo);
        if (done) {
            const uint32_t ch = (0x54474341u >> (8u * (c - 1u))) & 0xFFu;  // "ACGT"[c-1]
            if (packed_out) {
                chars = (chars << 8) | ch;  // most recent character in the low byte = lowest address
                if ((len & 3u) == 3u) {
                    // chars = [c(len-3) c(len-2) c(len-1) c(len)] high to low; memory order is the reverse of
                    // production order: address stride-1-len holds c(len)
                    *reinterpret_cast<uint32_t *>(out + r * (size_t)stride + (stride - 1u - len)) = chars;
                }
            } else {
                out[r * (size_t)stride + (stride - 1u - len)] = (uint8_t)ch;
            }
            ++len;
            idx = pc + occ - 1ull;  // C[b] + Occ(b, idx-1) of the reference = this row's LF target
            phase = 0;
            cont = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// extractPostfix (query.cpp:65-85): F / select walk right until '$', appended after the prefix.
// tlen = length of the whole read (UINT32_MAX: it does not fit, or the prefix did not).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * WG_WAVES)
extract_postfix_wave_kernel(const shard_view ix, const uint32_t *__restrict__ sel, uint64_t stride_m,
                            const uint64_t *__restrict__ rows, size_t n, uint8_t *__restrict__ out, uint32_t stride,
                            const uint32_t *__restrict__ plen, uint32_t *__restrict__ tlen,
                            unsigned long long *__restrict__ pool) {
    __shared__ uint4 s_stage[WG_WAVES][64 * SLOT_U4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *stage = s_stage[wave];
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    const staged_line L = {own_stage_row(stage, lane), lane & 7u};
    const char *lines_bytes = reinterpret_cast<const char *>(ix.lines);
    const uint32_t S = ix.sp.S, nlines = (uint32_t)ix.nlines;
    const uint64_t C1 = ix.C[1], C2 = ix.C[2], C3 = ix.C[3], C4 = ix.C[4];
    const uint64_t T1 = ix.total[1], T2 = ix.total[2], T3 = ix.total[3], T4 = ix.total[4];
    bool have = false;
    row_pool rp;
    size_t r = 0;
    uint64_t idx = 0, bc = 0, posbase = 0;
    uint32_t len = 0, f = 0;
    // phase 0: F symbol + select samples; 1: the window search between lo and hi; 2: select in window wlo
    uint32_t phase = 0, wlo = 0, whi = 0, probe = 0, tries = 0;
    uint32_t samp_lo = 0, samp_hi = 0;
    uint2 probe_word = {0, 0};
    uint32_t cont = 0, cblk = 0, cdw = 0;
    uint64_t t = 0;  // occurrences of f still to pass (select's running argument)
    // characters go out four at a time as aligned dwords when the row buffers allow it; `word` holds the
    // bytes of the dword being filled (low byte = lowest address), seeded with the prefix's last bytes
    const bool packed_out = (stride & 3u) == 0u && ((uintptr_t)out & 3u) == 0u;
    uint32_t word = 0;
    for (;;) {
        size_t nr = 0;
        if (draw_row(!have, pool, n, lane, rp, &nr)) {
            r = nr;
            idx = rows[r];
            const uint32_t pl = plen[r];
            have = true;
            phase = 0;
            if (pl == 0xFFFFFFFFu || idx >= ix.n) {
                tlen[r] = 0xFFFFFFFFu;
                have = false;
            } else {
                // (the prefix was moved to the head of the row's buffer by move_prefix_kernel)
                const uint8_t *buf = out + r * (size_t)stride;
                word = 0;
                if (packed_out && (pl & 3u)) word = *reinterpret_cast<const uint32_t *>(buf + (pl & ~3u)) & ((1u << (8u * (pl & 3u))) - 1u);
                len = pl;
            }
        }
        if (__builtin_amdgcn_ballot_w64(have) == 0ull) {
            if (rp.drained) break;
            continue;
        }
        // ---- phase 1 (second half): the probe issued in the last pass has landed
        if (have && phase == 1u && probe) {
            const uint64_t cnt = ((uint64_t)(probe_word.y & 0xFFu) << 32) | probe_word.x;
            if (cnt >= bc) whi = probe - 1u;  // floor search: largest window with count-before < bc
            else wlo = probe;
            probe = 0;
        }
        if (have && phase == 3u) {
            wlo = samp_lo;
            whi = samp_hi < samp_lo ? samp_lo : samp_hi;
            phase = 1;
        }
        // ---- phase 1 (first half): probe the count word of the middle window, or go and select
        if (have && phase == 1u) {
            if (whi > wlo) {
                probe = wlo + ((whi - wlo + 1u) >> 1);
                const uint32_t pl = probe + (probe >> GROUP_SHIFT);
                probe_word = *reinterpret_cast<const uint2 *>(lines_bytes + (uint64_t)pl * LINE_BYTES + 8u * (f - 1u));
            } else {
                phase = 2;
                cont = 0;
                tries = 0;
            }
        }
        // ---- phase 2: the window's line (or its continuation)
        const bool selecting = have && phase == 2u;
        uint32_t line = 0;
        if (selecting && cont == 0u) {
            line = wlo + (wlo >> GROUP_SHIFT);
            if (line >= nlines) line = 0;
        }
        const uint32_t want = selecting ? (cont ? cblk : line) : ~0u;
        glds_fetch(lines_bytes, want, lane, stage_lds);
        glds_wait();
        if (selecting) {
            bool found = false;
            uint64_t pos = 0;
            if (cont != KIND_CHUNK) {
                // a window line, or the far line that continues it: both carry absolute counts at
                // their first piece, so the select argument is bc minus the line's count word
                const line_head h = read_head(L);
                const uint64_t cnt = read_count(L, f);
                if (cont == 0u) posbase = (uint64_t)wlo * S;
                t = bc - cnt;
                const uint32_t c1 = matched24(L, HDR_DWORDS, f), c2 = read_half(L, f);
                const uint32_t c3 = c2 + matched24(L, HDR_DWORDS + 12u, f);
                const uint32_t tt = t > 4095ull ? 4095u : (uint32_t)t;
                const uint32_t cq = (tt > c1 ? 1u : 0u) + (tt > c2 ? 1u : 0u) + (tt > c3 ? 1u : 0u);
                const uint32_t before = cq == 0u ? 0u : cq == 1u ? c1 : cq == 2u ? c2 : c3;
                const uint32_t start = cq == 0u ? 0u : cq == 1u ? h.s1 : cq == 2u ? h.s2 : h.s3;
                uint32_t r6[6], left = 0;
                load24(L, HDR_DWORDS + 6u * cq, r6);
                const uint32_t p = select24(r6, f, tt - before, &left);
                if (left == 0u) {
                    found = true;
                    pos = posbase + start + p;
                } else if (h.kind == KIND_FAR) {
                    cblk = L.dword(LINE_DWORDS - 1u);
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_FAR;
                    posbase += h.span;
                } else if (h.kind == KIND_CHUNK && cont == 0u) {
                    cdw = read_chunk_dword(L);
                    cblk = (wlo >> GROUP_SHIFT) * (GROUP + 1u) + GROUP;
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_CHUNK;
                    posbase += h.span;
                    t = left;  // what the chunk's pieces still have to provide
                } else {
                    found = true;  // never for a valid bc: end the walk
                    pos = ix.n;
                }
            } else {
                uint32_t r6[6], left = 0;
                load24(L, cdw + 2u, r6);
                const uint32_t p = select24(r6, f, (uint32_t)t, &left);
                found = true;
                pos = left == 0u ? posbase + p : ix.n;
            }
            if (!found && ++tries > 72u) {
                found = true;
                pos = ix.n;
            }
            if (found) {
                if (pos >= ix.n) {  // a corrupt index: end the read instead of walking off
                    tlen[r] = len;
                    have = false;
                } else {
                    const uint32_t ch = (0x54474341u >> (8u * (f - 1u))) & 0xFFu;  // "ACGT"[f-1]
                    if (packed_out) {
                        word |= ch << (8u * (len & 3u));
                        if ((len & 3u) == 3u) {
                            *reinterpret_cast<uint32_t *>(out + r * (size_t)stride + (len - 3u)) = word;
                            word = 0;
                        }
                    } else {
                        out[r * (size_t)stride + len] = (uint8_t)ch;
                    }
                    ++len;
                    idx = pos;
                    phase = 0;
                    cont = 0;
                }
            }
        }
        // ---- phase 0 (rows that just arrived or just stepped): getF (rlebwt.cpp:307-314) and the select
        // samples around the bc-th f; the sample loads fly with the next pass's fetches
        if (have && phase == 0u) {
            f = (idx >= C1 ? 1u : 0u) + (idx >= C2 ? 1u : 0u) + (idx >= C3 ? 1u : 0u) + (idx >= C4 ? 1u : 0u);
            if (f == 0u) {  // '$': the read ends here (query.cpp:76)
                if (packed_out)  // the bytes of the dword still being filled
                    for (uint32_t k = len & ~3u; k < len; ++k) out[r * (size_t)stride + k] = (uint8_t)(word >> (8u * (k & 3u)));
                tlen[r] = len;
                have = false;
            } else if (len == stride) {
                tlen[r] = 0xFFFFFFFFu;
                have = false;
            } else {
                const uint64_t cf = f == 1u ? C1 : f == 2u ? C2 : f == 3u ? C3 : C4;
                const uint64_t tf = f == 1u ? T1 : f == 2u ? T2 : f == 3u ? T3 : T4;
                bc = idx - cf + 1ull;
                const uint64_t m = (bc - 1ull) >> SEL_SHIFT;
                samp_lo = sel[f * stride_m + m];
                samp_hi = ((m + 1ull) << SEL_SHIFT) < tf ? sel[f * stride_m + m + 1ull] : (uint32_t)(ix.nwin - 1ull);
                phase = 3;  // the samples are used from the next pass on
            }
        }
    }
}

// The prefix, written right to left from the end of the row's buffer, moved to its head: one wave per
// row, 64 bytes per step, low addresses first (source >= destination, so a step never overwrites a later
// step's source, and within a step every lane has loaded before any lane stores).
__global__ void __launch_bounds__(256)
move_prefix_kernel(uint8_t *__restrict__ out, uint32_t stride, const uint32_t *__restrict__ plen, size_t n) {
    const size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (r >= n) return;
    const uint32_t pl = plen[r];
    if (pl == 0xFFFFFFFFu || pl == 0u || pl >= stride) return;
    uint8_t *buf = out + r * (size_t)stride;
    const uint32_t off = stride - pl;
    for (uint32_t k0 = 0; k0 < pl; k0 += 64u) {
        const uint32_t k = k0 + lane;
        uint8_t ch = 0;
        if (k < pl) ch = buf[off + k];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): every lane's byte is in before any store of this step
        __builtin_amdgcn_wave_barrier();
        if (k < pl) buf[k] = ch;
    }
}

hipError_t launch_extract_wave(scratch_cache &scratch, const shard_view &ix, const uint32_t *d_sel, const void *d_rows,
                               size_t n, void *d_out, uint32_t stride, void *d_plen, void *d_len, int num_cus,
                               hipStream_t stream) {
    if (n == 0) return hipSuccess;
    scratch_cache::lease mem;
    hipError_t e = scratch.take(2 * sizeof(unsigned long long), stream, &mem);
    if (e != hipSuccess) return e;
    unsigned long long *pool = (unsigned long long *)mem.p;
    e = hipMemsetAsync(pool, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) {
        scratch.give(mem, stream);
        return e;
    }
    size_t g = (n + 64 * WG_WAVES - 1) / (64 * WG_WAVES);
    const size_t cap = (size_t)num_cus * 4;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(extract_prefix_wave_kernel, dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, ix,
                       (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (uint32_t *)d_plen, pool);
    hipLaunchKernelGGL(move_prefix_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (uint8_t *)d_out, stride,
                       (const uint32_t *)d_plen, n);
    hipLaunchKernelGGL(extract_postfix_wave_kernel, dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, ix, d_sel,
                       select_sample_stride(ix), (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride,
                       (const uint32_t *)d_plen, (uint32_t *)d_len, pool + 1);
    e = hipGetLastError();
    scratch.give(mem, stream);
    return e;
}

}  // namespace rsb
