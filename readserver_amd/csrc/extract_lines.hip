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
//   * LF step (prefix): one line gives the symbol at the position AND its rank in one pass, off one
//     look at the quarter's 24 pieces (rank_device.h, char_rank24: dword totals by v_dot4, only the
//     dword holding the position is taken apart); a spilled position takes one more pass, the window
//     line's four counts travelling with the lane;
//   * psi step (postfix): one 8-byte select sample per 256 occurrences NAMES the window of the wanted
//     occurrence (its window and how the block's occurrences spread over the following windows:
//     kernels.h, sample_window), so the first line fetched is the right one -- round 2 interpolated between
//     two bare window numbers and was wrong 31 % of the time, a second fetch and pass each; then selected
//     in: three quarter boundaries from the header and two v_dot4 sums, one quarter taken apart
//     (rank_device.h, select_in24).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
// COUNT_WORK (both walk kernels): work[] receives, summed over the waves, 0 passes, 1 lanes holding a
// row over those passes, 2 steps completed, 3 lanes on a continuation line, 4 lanes that fetched a line,
// 5 cycles in all, 6 cycles from issuing the fetches until they have landed, 7 (postfix) steps whose window
// came from the psi hint of the line the previous step landed in (no sample read).
enum { XW_PASSES = 0, XW_ACTIVE = 1, XW_STEPS = 2, XW_CONT = 3, XW_FETCHED = 4, XW_CYCLES = 5, XW_WAIT = 6, XW_PROBES = 7, XW_WORDS = 8 };

template <bool COUNT_WORK>
__global__ void __launch_bounds__(64 * WG_WAVES)
extract_prefix_wave_kernel(const shard_view ix, const uint64_t *__restrict__ rows, size_t n, uint8_t *__restrict__ out,
                           uint32_t stride, uint32_t *__restrict__ plen, unsigned long long *__restrict__ pool,
                           unsigned long long *__restrict__ work) {
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
    uint64_t idx = 0;
    uint32_t len = 0;
    uint32_t cont = 0, cblk = 0, cdw = 0, co = 0, tries = 0, w = 0;
    uint32_t acc_lo[4] = {0, 0, 0, 0}, acc_hi = 0;  // the window line's four counts, kept for its spill chunk
    // characters are produced right to left: four at a time go out as one aligned dword when the row
    // buffers allow it (a byte store per character is a request per character)
    const bool packed_out = (stride & 3u) == 0u && ((uintptr_t)out & 3u) == 0u;
    // ... and SIXTEEN at a time as one aligned 16-byte store when they allow that: every store is a request of its
    // own (lanes write into rows of their own), and at a dword per four steps the stores were a fifth of this
    // kernel's requests.  q0..q3 = the 16 most recent characters, the most recent in q0's low byte (lowest address).
    const bool out16 = (stride & 15u) == 0u && ((uintptr_t)out & 15u) == 0u;
    uint32_t chars = 0, q1 = 0, q2 = 0, q3 = 0;  // (chars doubles as q0)
    unsigned long long xw[XW_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = COUNT_WORK ? __builtin_amdgcn_s_memtime() : 0ull;
    for (;;) {
        size_t nr = 0;
        if (draw_row(!have, pool, n, lane, rp, &nr)) {
            r = nr;
            idx = rows[r];
            len = 0;
            cont = 0;
            chars = 0;
            q1 = q2 = q3 = 0;
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
        if (COUNT_WORK) {
            ++xw[XW_PASSES];
            xw[XW_ACTIVE] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(have));
            xw[XW_CONT] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(have && cont != 0u));
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
        unsigned long long t_fetch = 0;
        if (COUNT_WORK) {
            xw[XW_FETCHED] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(want != ~0u));
            __builtin_amdgcn_sched_barrier(0);
            t_fetch = __builtin_amdgcn_s_memtime();
        }
        glds_fetch(lines_bytes, want, lane, stage_lds);
        glds_wait();
        if (COUNT_WORK) {
            __builtin_amdgcn_sched_barrier(0);
            xw[XW_WAIT] += __builtin_amdgcn_s_memtime() - t_fetch;
        }
        // ---- where in this line the position lies (or where its continuation is)
        const bool in_chunk = cont == KIND_CHUNK;
        bool scan = false;
        uint32_t dw = HDR_DWORDS, rem = 0, cq = 0;
        if (have) {
            if (!in_chunk) {  // a window line, or the far line that continues one (same header)
                const line_head h = read_head(L);
                const uint32_t oe = cont ? co : o;
                if (oe <= h.span) {
                    cq = (oe > h.s1 ? 1u : 0u) + (oe > h.s2 ? 1u : 0u) + (oe > h.s3 ? 1u : 0u);
                    const uint32_t start = cq == 0u ? 0u : cq == 1u ? h.s1 : cq == 2u ? h.s2 : h.s3;
                    dw = HDR_DWORDS + 6u * cq;
                    rem = oe - start;
                    scan = true;
                } else if (h.kind == KIND_FAR) {
                    cblk = L.dword(LINE_DWORDS - 1u);
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_FAR;
                    co = oe - h.span;
                } else if (h.kind == KIND_CHUNK && cont == 0u) {
                    const uint4 h0 = L.u4(0), h1 = L.u4(4);  // the four count words (dwords 0..7)
                    acc_lo[0] = h0.x; acc_lo[1] = h0.z; acc_lo[2] = h1.x; acc_lo[3] = h1.z;
                    acc_hi = (h0.y & 0xFFu) | ((h0.w & 0xFFu) << 8) | ((h1.y & 0xFFu) << 16) | (h1.w << 24);
                    cdw = read_chunk_dword(L);
                    cblk = (w >> GROUP_SHIFT) * (GROUP + 1u) + GROUP;
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_CHUNK;
                    co = oe - h.span;
                } else {
                    scan = true;  // beyond what the index holds: never for idx < n; rem = 0 ends the walk as '$'
                }
            } else {
                dw = cdw + 2u;
                rem = co;
                scan = true;
            }
            if (!scan && ++tries > 72u) scan = true;  // a corrupt chain: rem = 0 ends the walk
        }
        // ---- the symbol at the position and its rank, off the same 24 pieces (RLEBWT::getChar,
        // rlebwt.cpp:202-227, and RLEBWT::getOcc, rlebwt.cpp:268-301: one LF step, query.cpp:49-57)
        uint32_t r6[6];
        load24(L, dw, r6);
        const char_rank cr = char_rank24(r6, scan ? rem : 0u, 0u);
        const uint32_t c = cr.c;
        const uint32_t ci = (c - 1u) & 3u;
        // what the line says about c before those pieces
        uint64_t base;
        if (in_chunk) {
            const uint2 hd = L.u2(cdw);
            const uint32_t hw = ci < 2u ? hd.x : hd.y;
            const uint32_t alo = ci == 0u ? acc_lo[0] : ci == 1u ? acc_lo[1] : ci == 2u ? acc_lo[2] : acc_lo[3];
            base = (((uint64_t)((acc_hi >> (8u * ci)) & 0xFFu) << 32) | alo) + ((hw >> (12u * (ci & 1u))) & 0xFFFu);
        } else {
            const uint32_t hb = read_half(L, ci + 1u);
            const uint32_t m = matched24(L, HDR_DWORDS + 6u * (cq & 2u), ci + 1u);
            base = read_count(L, ci + 1u) + (cq >= 2u ? hb : 0u) + ((cq & 1u) ? m : 0u);
        }
        // C[c], with every lane active (a ds_bpermute returns 0 from a masked-off source lane)
        const uint64_t pc = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(ci << 2), (int)ctab_hi) << 32) |
                            (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ci << 2), (int)ctab_lo);
        bool done = false;
        if (scan) {
            if (c == 0u || c > 4u) {  // '$': the read starts here (query.cpp:52)
                if (out16) {
                    // the (len & 15) characters not yet written sit in q0..q3 from the low byte up, the oldest next to
                    // the 16-byte boundary B: whole dwords go out aligned to B (a funnel shift lines them up), then
                    // the (len & 3) most recent bytes
                    const uint32_t rem = len & 15u, nd = rem >> 2, sh = rem & 3u;
                    uint8_t *row = out + r * (size_t)stride;
                    const uint32_t B = stride - (len - rem);
                    for (uint32_t jj = 0; jj < nd; ++jj) {
                        const uint32_t di = nd - 1u - jj;
                        const uint32_t lo_d = di == 0u ? chars : di == 1u ? q1 : q2;
                        const uint32_t hi_d = di == 0u ? q1 : di == 1u ? q2 : q3;
                        const uint32_t v = sh == 0u ? lo_d : sh == 1u ? __builtin_amdgcn_alignbyte(hi_d, lo_d, 1u)
                                                   : sh == 2u ? __builtin_amdgcn_alignbyte(hi_d, lo_d, 2u)
                                                              : __builtin_amdgcn_alignbyte(hi_d, lo_d, 3u);
                        *reinterpret_cast<uint32_t *>(row + B - 4u * (jj + 1u)) = v;
                    }
                    for (uint32_t t = 0; t < sh; ++t) row[stride - len + t] = (uint8_t)(chars >> (8u * t));
                } else if (packed_out) {  // the characters not yet written: the (len & 3) most recent ones
                    for (uint32_t t = 0; t < (len & 3u); ++t)
                        out[r * (size_t)stride + (stride - len + t)] = (uint8_t)(chars >> (8u * t));
                }
                plen[r] = len;
                have = false;
            } else if (len == stride) {  // the reference would spin (query.cpp:48)
                plen[r] = 0xFFFFFFFFu;
                have = false;
            } else {
                const uint32_t ch = (0x54474341u >> (8u * ci)) & 0xFFu;  // "ACGT"[c-1]
                if (out16) {
                    q3 = __builtin_amdgcn_alignbyte(q3, q2, 3u);  // (q3:q2:q1:chars) <<= 8
                    q2 = __builtin_amdgcn_alignbyte(q2, q1, 3u);
                    q1 = __builtin_amdgcn_alignbyte(q1, chars, 3u);
                    chars = (chars << 8) | ch;
                    if ((len & 15u) == 15u)  // address stride-1-len is 16-aligned: the chunk's 16 characters at once
                        *reinterpret_cast<uint4 *>(out + r * (size_t)stride + (stride - 1u - len)) = make_uint4(chars, q1, q2, q3);
                } else if (packed_out) {
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
                idx = pc + base + cr.occ - 1ull;  // C[b] + Occ(b, idx) - 1 = this row's LF target (query.cpp:55-56)
                cont = 0;
                done = true;
            }
        }
        if (COUNT_WORK) xw[XW_STEPS] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(done));
    }
    if (COUNT_WORK && lane == 0u) {
        xw[XW_CYCLES] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int i = 0; i < XW_WORDS; ++i) atomicAdd(&work[i], xw[i]);
    }
}

// ---------------------------------------------------------------------------------------------------
// extractPostfix (query.cpp:65-85): F / select walk right until '$', appended after the prefix.
// tlen = length of the whole read (UINT32_MAX: it does not fit, or the prefix did not).
// ---------------------------------------------------------------------------------------------------
template <bool COUNT_WORK>
__global__ void __launch_bounds__(64 * WG_WAVES)
extract_postfix_wave_kernel(const shard_view ix, const uint64_t *__restrict__ sel, uint64_t stride_m,
                            const uint64_t *__restrict__ rows, size_t n, uint8_t *__restrict__ out, uint32_t stride,
                            const uint32_t *__restrict__ plen, uint32_t *__restrict__ tlen,
                            unsigned long long *__restrict__ pool, unsigned long long *__restrict__ work) {
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
    // phase 0: F symbol + select samples; 3: the samples are in flight; 2: select in window wcur, which
    // lies between the samples' windows wlo and whi
    uint32_t phase = 0, wlo = 0, whi = 0, wcur = 0, tries = 0;
    uint64_t samp = 0;
    uint32_t cont = 0, cblk = 0, cdw = 0;
    const uint32_t nwin = (uint32_t)ix.nwin;
    uint64_t t = 0;  // occurrences of f still to pass (select's running argument)
    // characters go out four at a time as aligned dwords when the row buffers allow it; `word` holds the
    // bytes of the dword being filled (low byte = lowest address), seeded with the prefix's last bytes
    const bool packed_out = (stride & 3u) == 0u && ((uintptr_t)out & 3u) == 0u;
    // ... sixteen at a time where the buffers allow it (see the prefix kernel): word, w1, w2, w3 = the 16-byte chunk
    // being filled, byte len & 15 next
    const bool out16 = (stride & 15u) == 0u && ((uintptr_t)out & 15u) == 0u;
    uint32_t word = 0, w1 = 0, w2 = 0, w3 = 0;
    unsigned long long xw[XW_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = COUNT_WORK ? __builtin_amdgcn_s_memtime() : 0ull;
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
                word = w1 = w2 = w3 = 0;
                if (out16) {
                    if (pl & 15u) {  // the chunk the prefix ends in: its bytes below pl & 15 are the prefix's
                        const uint4 e = *reinterpret_cast<const uint4 *>(buf + (pl & ~15u));
                        const uint32_t kb = pl & 15u;  // bytes to keep
                        const uint32_t part = (1u << (8u * (kb & 3u))) - 1u;  // of the dword the boundary falls in
                        word = kb >= 4u ? e.x : e.x & part;
                        w1 = kb >= 8u ? e.y : kb > 4u ? e.y & part : 0u;
                        w2 = kb >= 12u ? e.z : kb > 8u ? e.z & part : 0u;
                        w3 = kb > 12u ? e.w & part : 0u;
                    }
                } else if (packed_out && (pl & 3u)) {
                    word = *reinterpret_cast<const uint32_t *>(buf + (pl & ~3u)) & ((1u << (8u * (pl & 3u))) - 1u);
                }
                len = pl;
            }
        }
        if (__builtin_amdgcn_ballot_w64(have) == 0ull) {
            if (rp.drained) break;
            continue;
        }
        if (COUNT_WORK) {
            ++xw[XW_PASSES];
            xw[XW_ACTIVE] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(have));
        }
        // ---- the sample issued at the end of the last pass has landed: it names the window of the bc-th f
        // outright (kernels.h, sample_window: exact unless the block's 256 occurrences spread over more than
        // five windows, then a lower bound the search below walks up from)
        if (have && phase == 3u) {
            bool exact;
            wcur = sample_window(samp, bc, &exact);
            if (wcur >= nwin) wcur = nwin - 1u;  // never for a sample this index built
            wlo = whi = wcur;
            if (!exact) {
                // the block's occurrences spread over more than five windows (a stretch of the BWT nearly without
                // f): the bc-th lies between w0 + 4 and the window the next block starts in -- a dependent load,
                // taken by the few lanes that need it, and the bisection of the window headers below
                const uint64_t m = (bc - 1ull) >> SEL_SHIFT;
                const uint64_t tf = f == 1u ? T1 : f == 2u ? T2 : f == 3u ? T3 : T4;
                const uint32_t nxt = ((m + 1ull) << SEL_SHIFT) < tf ? (uint32_t)sel[f * stride_m + m + 1ull] : nwin - 1u;
                whi = nxt > wlo ? (nxt < nwin ? nxt : nwin - 1u) : wlo;
                wcur = wlo + ((whi - wlo) >> 1);
            }
            phase = 2;
            cont = 0;
            tries = 0;
        }
        // ---- phase 2: the window's line (or its continuation).  The bc-th f is in wcur iff
        // count(wcur) < bc <= count(wcur + 1): the line's own count word settles the first half, the
        // second shows when the window's pieces run out before the select argument does.
        const bool selecting = have && phase == 2u;
        uint32_t line = 0;
        if (selecting && cont == 0u) {
            line = wcur + (wcur >> GROUP_SHIFT);
            if (line >= nlines) line = 0;
        }
        const uint32_t want = selecting ? (cont ? cblk : line) : ~0u;
        unsigned long long t_fetch = 0;
        if (COUNT_WORK) {
            xw[XW_FETCHED] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(selecting));
            xw[XW_CONT] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(selecting && cont != 0u));
            __builtin_amdgcn_sched_barrier(0);
            t_fetch = __builtin_amdgcn_s_memtime();
        }
        glds_fetch(lines_bytes, want, lane, stage_lds);
        glds_wait();
        if (COUNT_WORK) {
            __builtin_amdgcn_sched_barrier(0);
            xw[XW_WAIT] += __builtin_amdgcn_s_memtime() - t_fetch;
        }
        bool stepped = false, moved = false;
        // the psi hint of the window the step lands in (line_format.h), taken while its line is staged
        bool hint_here = false, hinted_now = false;
        uint32_t hint_w0 = 0, hint_kk = 0, hint_win = 0;
        if (selecting) {
            bool found = false;
            uint64_t pos = 0;
            int move = 0;  // -1 / +1: the bc-th f is in an earlier / a later window than wcur
            if (cont != KIND_CHUNK) {
                // a window line, or the far line that continues it: both carry absolute counts at
                // their first piece, so the select argument is bc minus the line's count word
                const line_head h = read_head(L);
                const uint64_t cnt = read_count(L, f);
                if (cont == 0u) {
                    posbase = (uint64_t)wcur * S;
                    if (cnt >= bc) move = -1;
                }
                t = bc - cnt;
                const uint32_t c1 = matched24(L, HDR_DWORDS, f), c2 = read_half(L, f);
                const uint32_t c3 = c2 + matched24(L, HDR_DWORDS + 12u, f);
                const uint32_t tt = t > 4095ull ? 4095u : (uint32_t)t;
                const uint32_t cq = (tt > c1 ? 1u : 0u) + (tt > c2 ? 1u : 0u) + (tt > c3 ? 1u : 0u);
                const uint32_t before = cq == 0u ? 0u : cq == 1u ? c1 : cq == 2u ? c2 : c3;
                const uint32_t start = cq == 0u ? 0u : cq == 1u ? h.s1 : cq == 2u ? h.s2 : h.s3;
                uint32_t r6[6], left = 0;
                load24(L, HDR_DWORDS + 6u * cq, r6);
                const uint32_t p = select_in24(r6, f, tt - before, &left);
                // a hit lies within the symbols the line's own pieces hold: when the window tried is an
                // earlier one than the bc-th f's, the argument outlasts the pieces and may "find" its f in
                // the bytes after them (a far line's link)
                if (move != 0) {
                    // (nothing of this line is of use)
                } else if (left == 0u && start + p < h.span) {
                    found = true;
                    pos = posbase + start + p;
                    if (cont == 0u && ((L.dword(1) >> (8u + HINT_META0_BIT)) & 1u) != 0u) {
                        hint_here = true;
                        hint_w0 = L.dword(LINE_DWORDS - 2u);
                        hint_kk = L.dword(LINE_DWORDS - 1u);
                        hint_win = wcur;
                    }
                } else if (h.kind == KIND_FAR) {
                    cblk = L.dword(LINE_DWORDS - 1u);
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_FAR;
                    posbase += h.span;
                } else if (h.kind == KIND_CHUNK && cont == 0u) {
                    cdw = read_chunk_dword(L);
                    cblk = (wcur >> GROUP_SHIFT) * (GROUP + 1u) + GROUP;
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_CHUNK;
                    posbase += h.span;
                    t = left;  // what the chunk's pieces still have to provide
                } else {
                    move = 1;  // the window's pieces ran out first
                }
            } else {
                uint32_t r6[6], left = 0;
                load24(L, cdw + 2u, r6);
                const uint32_t p = select_in24(r6, f, (uint32_t)t, &left);
                const uint2 hd = L.u2(cdw);
                const uint32_t csym = (hd.x >> 24) | ((hd.y >> 24) << 8);  // symbols the chunk holds; the next chunk follows
                found = left == 0u && p < csym;
                pos = posbase + p;
                if (!found) move = 1;
            }
            if (move < 0) {
                if (wcur == 0u) {  // count(0) = 0 < bc: only a corrupt index gets here
                    found = true;
                    pos = ix.n;
                }
                whi = wcur - 1u;
                wlo = wlo < whi ? wlo : whi;
            } else if (move > 0) {
                wlo = wcur + 1u;
                whi = whi > wlo ? whi : wlo;
                if (wlo >= nwin) {  // past the last window: a corrupt index
                    found = true;
                    pos = ix.n;
                }
            }
            if (move != 0) {
                wcur = wlo + ((whi - wlo) >> 1);
                cont = 0;
                moved = true;
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
                    if (out16) {
                        const uint32_t sb = ch << (8u * (len & 3u)), di = (len >> 2) & 3u;
                        word |= di == 0u ? sb : 0u;
                        w1 |= di == 1u ? sb : 0u;
                        w2 |= di == 2u ? sb : 0u;
                        w3 |= di == 3u ? sb : 0u;
                        if ((len & 15u) == 15u) {
                            *reinterpret_cast<uint4 *>(out + r * (size_t)stride + (len - 15u)) = make_uint4(word, w1, w2, w3);
                            word = w1 = w2 = w3 = 0;
                        }
                    } else if (packed_out) {
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
                    stepped = true;
                }
            }
        }
        if (COUNT_WORK) {
            xw[XW_STEPS] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(stepped));
            (void)moved;
        }
        // ---- phase 0 (rows that just arrived or just stepped): getF (rlebwt.cpp:307-314) and the select
        // samples around the bc-th f; the sample loads fly with the next pass's fetches
        if (have && phase == 0u) {
            f = (idx >= C1 ? 1u : 0u) + (idx >= C2 ? 1u : 0u) + (idx >= C3 ? 1u : 0u) + (idx >= C4 ? 1u : 0u);
            if (f == 0u) {  // '$': the read ends here (query.cpp:76)
                if (out16) {  // the chunk still being filled: its whole dwords, then the last bytes
                    uint8_t *row = out + r * (size_t)stride;
                    const uint32_t base = len & ~15u, rem = len & 15u, nd = rem >> 2;
                    for (uint32_t jj = 0; jj < nd; ++jj)
                        *reinterpret_cast<uint32_t *>(row + base + 4u * jj) = jj == 0u ? word : jj == 1u ? w1 : w2;
                    const uint32_t lastw = nd == 0u ? word : nd == 1u ? w1 : nd == 2u ? w2 : w3;
                    for (uint32_t t = 0; t < (rem & 3u); ++t) row[base + 4u * nd + t] = (uint8_t)(lastw >> (8u * t));
                } else if (packed_out)  // the bytes of the dword still being filled
                    for (uint32_t k = len & ~3u; k < len; ++k) out[r * (size_t)stride + k] = (uint8_t)(word >> (8u * (k & 3u)));
                tlen[r] = len;
                have = false;
            } else if (len == stride) {
                tlen[r] = 0xFFFFFFFFu;
                have = false;
            } else {
                const uint64_t cf = f == 1u ? C1 : f == 2u ? C2 : f == 3u ? C3 : C4;
                bc = idx - cf + 1ull;
                // The line this row was found in may say where psi takes the rows of its window: then the next
                // window is known now, with no sample read and no round trip for it.  The hint speaks of the F
                // symbol of the window's first row: it holds for this row when that row lies in f's block too.
                bool via_hint = false;
                if (hint_here) {
                    const uint64_t r0 = (uint64_t)hint_win * S;
                    if (r0 >= cf && idx >= r0) {
                        bool exact;
                        const uint32_t wn = hint_window(hint_w0, hint_kk, (uint32_t)(idx - r0), &exact);
                        if (exact && wn < nwin) {
                            wcur = wlo = whi = wn;
                            phase = 2;
                            cont = 0;
                            tries = 0;
                            via_hint = true;
                        }
                    }
                }
                hinted_now = via_hint;
                if (!via_hint) {
                    const uint64_t m = (bc - 1ull) >> SEL_SHIFT;
                    samp = sel[f * stride_m + m];
                    phase = 3;  // the sample is used from the next pass on
                }
            }
        }
        if (COUNT_WORK) xw[XW_PROBES] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hinted_now));
    }
    if (COUNT_WORK && lane == 0u) {
        xw[XW_CYCLES] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int i = 0; i < XW_WORDS; ++i) atomicAdd(&work[i], xw[i]);
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

hipError_t launch_extract_wave(scratch_cache &scratch, const shard_view &ix, const uint64_t *d_sel, const void *d_rows,
                               size_t n, void *d_out, uint32_t stride, void *d_plen, void *d_len, int num_cus,
                               hipStream_t stream, unsigned long long *d_work) {
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
    // workgroups per CU: the walk kernels' 88-95 VGPRs and 32 KB of LDS per workgroup admit 5; RSBWT_EXTRACT_WGS_PER_CU
    // overrides (A/B knob, tools/README.md)
    static const size_t wgs_per_cu = [] {
        const char *e = getenv("RSBWT_EXTRACT_WGS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return (size_t)(v > 0 && v <= 8 ? v : 4);
    }();
    const size_t cap = (size_t)num_cus * wgs_per_cu;
    if (g > cap) g = cap;
    if (d_work)
        hipLaunchKernelGGL(extract_prefix_wave_kernel<true>, dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, ix,
                           (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (uint32_t *)d_plen, pool, d_work);
    else
        hipLaunchKernelGGL(extract_prefix_wave_kernel<false>, dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, ix,
                           (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (uint32_t *)d_plen, pool, d_work);
    hipLaunchKernelGGL(move_prefix_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (uint8_t *)d_out, stride,
                       (const uint32_t *)d_plen, n);
    if (d_work)
        hipLaunchKernelGGL(extract_postfix_wave_kernel<true>, dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, ix, d_sel,
                           select_sample_stride(ix), (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride,
                           (const uint32_t *)d_plen, (uint32_t *)d_len, pool + 1, d_work + XW_WORDS);
    else
        hipLaunchKernelGGL(extract_postfix_wave_kernel<false>, dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, ix, d_sel,
                           select_sample_stride(ix), (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride,
                           (const uint32_t *)d_plen, (uint32_t *)d_len, pool + 1, d_work + XW_WORDS);
    e = hipGetLastError();
    scratch.give(mem, stream);
    return e;
}

}  // namespace rsb
