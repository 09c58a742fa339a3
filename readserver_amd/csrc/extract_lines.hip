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
//   * psi step (postfix): the window of the wanted occurrence comes from the PSI HINT of the line the walk stands
//     in (line_format.h: where psi takes the rows of that window -- the line the previous step landed in, or, for a
//     row that has just been handed out, its own window's line when the shard was laid out with a hint in every
//     line): no other read.  A row the hint bounds without settling (within a boundary's granule, or past its last
//     boundary) tries the lower candidate first -- the line's own count word says "earlier", its pieces running out
//     says "later" -- and only a guess that fails past the hint's reach, or a line without a hint, reads an 8-byte
//     select sample (kernels.h, sample_window: it names the window, or bounds it with the next sample for a floor
//     search over the window headers).  Then selected in: three quarter boundaries from the header and two
//     v_dot4 sums, one quarter taken apart (rank_device.h, select_in24).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "line_format.h"
#include "rank_device.h"
#include "wave_lines.h"

namespace rsb {

// Waves per workgroup of the walk kernels and the workgroups per CU they are compiled for (tuning knobs,
// tools/build_variant.sh): a wave's 8 KB stage is all a workgroup shares, so smaller workgroups only change how many
// waves a CU's LDS and registers admit.
#ifndef RSB_WALK_WG_WAVES
#define RSB_WALK_WG_WAVES 4
#endif
#ifndef RSB_WALK_MIN_WGS
#define RSB_WALK_MIN_WGS 4
#endif
constexpr int XWG_WAVES = RSB_WALK_WG_WAVES;


// Hands rows to the lanes that have none.  A wave draws chunks of `chunk` (at most ROW_CHUNK) consecutive rows from the
// global counter (one atomic per chunk, not per pass: the atomic's round trip would otherwise sit in
// front of every pass's line fetch) and gives the next one to whichever lane is free.
constexpr uint32_t ROW_CHUNK = 256;
struct row_pool {
    uint64_t next = 0, end = 0;  // wave-uniform
    bool drained = false;
};
__device__ __forceinline__ bool draw_row(bool want, unsigned long long *pool, size_t n, uint32_t lane, row_pool &rp,
                                         size_t *row, uint32_t chunk) {
    const uint64_t mask = __builtin_amdgcn_ballot_w64(want);
    if (mask == 0ull) return false;
    if (rp.next >= rp.end && !rp.drained) {
        unsigned long long c = 0;
        if (lane == 0u) c = atomicAdd(pool, (unsigned long long)chunk);
        c = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
            __builtin_amdgcn_readfirstlane((uint32_t)c);
        rp.next = c;
        rp.end = c + chunk < n ? c + chunk : n;
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
__global__ void __launch_bounds__(64 * XWG_WAVES, RSB_WALK_MIN_WGS)
extract_prefix_wave_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ rows_all,
                           size_t n, uint8_t *__restrict__ out_all, uint32_t stride, uint32_t *__restrict__ plen_all,
                           unsigned long long *__restrict__ pools, unsigned long long *__restrict__ work, uint32_t row_chunk) {
    __shared__ uint4 s_stage[XWG_WAVES][64 * SLOT_U4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *stage = s_stage[wave];
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    const staged_line L = {own_stage_row(stage, lane), lane & 7u};
    unsigned long long xw[XW_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = COUNT_WORK ? __builtin_amdgcn_s_memtime() : 0ull;
    // A wave walks the rows of ONE shard at a time (rows, buffers and lengths of shard s: block s of n): it starts on
    // shard blockIdx % nshards and draws from that shard's pool until it is empty and its own walks have ended, then
    // moves on to the next shard that still has rows -- the shards of a set walked by one launch (sets.hip): every
    // lane walks nshards x n / lanes rows before the launch's one tail, and everything shard-specific sits in scalar
    // registers.
    uint32_t sid = blockIdx.x % nshards;
    for (uint32_t visited = 0; visited < nshards; ++visited, sid = (sid + 1u == nshards) ? 0u : sid + 1u) {
    const shard_view *sv = shards + sid;
    const char *lines_bytes = reinterpret_cast<const char *>(sv->lines);
    const uint32_t S = sv->sp.S, nlines = (uint32_t)sv->nlines;
    const double inv = sv->sp.inv;
    const uint64_t ix_n = sv->n;
    const uint64_t *__restrict__ rows = rows_all + (size_t)sid * n;
    uint8_t *__restrict__ out = out_all + (size_t)sid * n * stride;
    uint32_t *__restrict__ plen = plen_all + (size_t)sid * n;
    unsigned long long *pool = pools + (size_t)sid * POOL_STRIDE;  // (a line group apart: kernels.h)
    uint32_t ctab_lo, ctab_hi;  // C[1..4] in lanes 0..3, read with ds_bpermute
    {
        const uint64_t c1 = sv->C[1], c2 = sv->C[2], c3 = sv->C[3], c4 = sv->C[4];
        const uint32_t l3 = lane & 3u;
        const uint64_t cv = l3 == 0u ? c1 : l3 == 1u ? c2 : l3 == 2u ? c3 : c4;
        ctab_lo = (uint32_t)cv;
        ctab_hi = (uint32_t)(cv >> 32);
    }
    bool have = false;
    row_pool rp;
    uint32_t r = 0;
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
    for (;;) {
        size_t nr = 0;
        if (draw_row(!have, pool, n, lane, rp, &nr, row_chunk)) {
            r = (uint32_t)nr;
            idx = rows[r];
            len = 0;
            cont = 0;
            chars = 0;
            q1 = q2 = q3 = 0;
            have = true;
            if (idx >= ix_n) {
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
            const uint32_t m = matched24(L, HDR_DWORDS + 6u * (cq & 2u), cr.tab);  // (c = ci + 1 wherever the step is taken)
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
    }  // (the next shard)
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
__global__ void __launch_bounds__(64 * XWG_WAVES, RSB_WALK_MIN_WGS)
extract_postfix_wave_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ rows_all,
                            size_t n, uint8_t *__restrict__ out_all, uint32_t stride, const uint32_t *__restrict__ plen_all,
                            uint32_t *__restrict__ tlen_all, unsigned long long *__restrict__ pools,
                            unsigned long long *__restrict__ work, uint32_t row_chunk) {
    __shared__ uint4 s_stage[XWG_WAVES][64 * SLOT_U4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *stage = s_stage[wave];
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    const staged_line L = {own_stage_row(stage, lane), lane & 7u};
    unsigned long long xw[XW_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = COUNT_WORK ? __builtin_amdgcn_s_memtime() : 0ull;
    // (one shard at a time, as the prefix kernel)
    uint32_t sid = blockIdx.x % nshards;
    for (uint32_t visited = 0; visited < nshards; ++visited, sid = (sid + 1u == nshards) ? 0u : sid + 1u) {
    const shard_view *sv = shards + sid;
    const char *lines_bytes = reinterpret_cast<const char *>(sv->lines);
    const uint32_t S = sv->sp.S, nlines = (uint32_t)sv->nlines;
    const uint64_t C1 = sv->C[1], C2 = sv->C[2], C3 = sv->C[3], C4 = sv->C[4];
    const uint64_t T1 = sv->total[1], T2 = sv->total[2], T3 = sv->total[3], T4 = sv->total[4];
    const uint64_t ix_n = sv->n;
    const uint64_t *__restrict__ sel = sv->sel;
    const uint64_t stride_m = sv->sel_stride;
    const uint64_t *__restrict__ rows = rows_all + (size_t)sid * n;
    uint8_t *__restrict__ out = out_all + (size_t)sid * n * stride;
    const uint32_t *__restrict__ plen = plen_all + (size_t)sid * n;
    uint32_t *__restrict__ tlen = tlen_all + (size_t)sid * n;
    unsigned long long *pool = pools + (size_t)sid * POOL_STRIDE;  // (a line group apart: kernels.h)
    bool have = false;
    row_pool rp;
    uint32_t r = 0;
    uint64_t idx = 0, bc = 0;
    uint32_t poff = 0;  // symbols of window wcur held by the lines before the one staged (a continuation's offset)
    uint32_t len = 0, f = 0;
    // phase 0: F symbol, then where the bc-th f lies -- from the hint in hand, or: 4: the row's own window line is
    // being fetched for its hint (a row just handed out, on a shard with a hint in every line); 3: a select sample is
    // in flight; 2: select in window wcur, one of wlo..whi (guess_open: whi is only the hint's first guess)
    uint32_t phase = 0, wlo = 0, whi = 0, wcur = 0, tries = 0;
    bool guess_open = false, fresh = false;
    uint32_t wmin = 0;  // windows below this one were tried and lie before the bc-th f (a hint's guesses that failed)
    uint32_t otry = 0;  // windows tried past an open hint's reach
    // the psi hint of the window line last parsed (the line the walk now stands in)
    bool hv = false;
    uint32_t hw0 = 0, hkk = 0, roff = 0;  // (roff: where in its window the walk stands -- the row's offset among the hint's rows)
    uint64_t samp = 0;
    uint32_t cont = 0, cblk = 0, cdw = 0;
    const uint32_t nwin = (uint32_t)sv->nwin;
    const uint32_t sel_shift = sv->sel_shift, hshift = hint_shift(S);
    const bool rich = sv->hint_room != 0u;
    const double inv = sv->sp.inv;
    uint32_t t = 0;  // occurrences of f still to pass (select's running argument; held to 32 bits: past 4,095 the window is not this one)
    // characters go out four at a time as aligned dwords when the row buffers allow it; `word` holds the
    // bytes of the dword being filled (low byte = lowest address), seeded with the prefix's last bytes
    const bool packed_out = (stride & 3u) == 0u && ((uintptr_t)out & 3u) == 0u;
    // ... sixteen at a time where the buffers allow it (see the prefix kernel): word, w1, w2, w3 = the 16-byte chunk
    // being filled, byte len & 15 next
    const bool out16 = (stride & 15u) == 0u && ((uintptr_t)out & 15u) == 0u;
    uint32_t word = 0, w1 = 0, w2 = 0, w3 = 0;
    for (;;) {
        size_t nr = 0;
        if (draw_row(!have, pool, n, lane, rp, &nr, row_chunk)) {
            r = (uint32_t)nr;
            idx = rows[r];
            const uint32_t pl = plen[r];
            have = true;
            phase = 0;
            fresh = true;
            hv = false;
            if (pl == 0xFFFFFFFFu || idx >= ix_n) {
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
        // outright (line_format.h, sample_window: exact within the first occurrences of its block unless they
        // spread over more than five windows; else a lower bound the search below walks up from)
        if (have && phase == 3u) {
            bool exact;
            wcur = sample_window(samp, bc, sel_shift, &exact);
            if (wcur >= nwin) wcur = nwin - 1u;  // never for a sample this index built
            wlo = whi = wcur;
            if (!exact) {
                // the bc-th f lies between the sample's bound and the window the next block starts in -- a dependent
                // load, taken by the few lanes that need it, and the bisection of the window headers below
                const uint64_t m = (bc - 1ull) >> sel_shift;
                const uint64_t tf = f == 1u ? T1 : f == 2u ? T2 : f == 3u ? T3 : T4;
                const uint32_t nxt = ((m + 1ull) << sel_shift) < tf ? (uint32_t)sel[f * stride_m + m + 1ull] : nwin - 1u;
                whi = nxt > wlo ? (nxt < nwin ? nxt : nwin - 1u) : wlo;
                wlo = wlo > wmin ? wlo : wmin;
                whi = whi > wlo ? whi : wlo;
                wcur = wlo + ((whi - wlo) >> 1);
            }
            phase = 2;
            cont = 0;
            tries = 0;
            guess_open = false;
        }
        // ---- phase 2: the window's line (or its continuation).  The bc-th f is in wcur iff
        // count(wcur) < bc <= count(wcur + 1): the line's own count word settles the first half, the
        // second shows when the window's pieces run out before the select argument does.
        // ---- phase 4: the row's own window line, for its hint
        const bool selecting = have && phase == 2u;
        const bool asking = have && phase == 4u;
        uint32_t line = 0;
        if (selecting && cont == 0u) {
            line = wcur + (wcur >> GROUP_SHIFT);
            if (line >= nlines) line = 0;
        }
        if (asking) {
            uint32_t pin;
            const uint32_t hwin = fast_window(idx, S, inv, pin);
            roff = pin;
            line = hwin + (hwin >> GROUP_SHIFT);
            if (line >= nlines) line = 0;
        }
        const uint32_t want = selecting ? (cont ? cblk : line) : asking ? line : ~0u;
        unsigned long long t_fetch = 0;
        if (COUNT_WORK) {
            xw[XW_FETCHED] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(selecting || asking));
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
        bool stepped = false, hinted_now = false;
        // the psi hint of a window line, taken while the line is staged (line_format.h: dwords 30, 31; 29, 30 of a
        // line that ends in a far link)
        if (asking || (selecting && cont == 0u)) {
            const uint32_t d1 = L.dword(1), d3 = L.dword(3);
            const uint32_t hd = ((d3 >> 28) & 3u) == KIND_FAR ? LINE_DWORDS - 3u : LINE_DWORDS - 2u;
            hv = ((d1 >> (8u + HINT_META0_BIT)) & 1u) != 0u;
            hw0 = L.dword(hd);
            hkk = L.dword(hd + 1u);
            if (asking) phase = 0;  // the block below takes it from here
        }
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
                    poff = 0;
                    if (cnt >= bc) move = -1;
                }
                t = bc - cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(bc - cnt);
                // which quarter holds the t-th f: the header says what the first half holds (c2); ONE whole-quarter sum
                // -- of quarter 0 or of quarter 2, whichever half the argument falls in -- settles the rest
                const uint32_t c2 = read_half(L, f);
                const uint32_t tt = t > 4095u ? 4095u : t;
                const bool late = tt > c2;
                const uint32_t mq = matched24(L, HDR_DWORDS + (late ? 12u : 0u), f);
                const uint32_t cm = late ? c2 + mq : mq;  // f's in quarters 0..2 (late) / in quarter 0
                const uint32_t cq = (late ? 2u : 0u) + (tt > cm ? 1u : 0u);
                const uint32_t before = cq == 0u ? 0u : cq == 1u ? cm : cq == 2u ? c2 : cm;
                const uint32_t start = cq == 0u ? 0u : cq == 1u ? h.s1 : cq == 2u ? h.s2 : h.s3;
                uint32_t r6[6], left = 0;
                load24(L, HDR_DWORDS + 6u * cq, r6);
                const uint32_t p = select_in24(r6, f, tt - before, &left);
                // a hit lies within the symbols the line's own pieces hold: when the window tried is an
                // earlier one than the bc-th f's, the argument outlasts the pieces and may "find" its f in
                // the bytes after them (a psi hint, a far line's link)
                if (move != 0) {
                    // (nothing of this line is of use)
                } else if (left == 0u && start + p < h.span) {
                    found = true;
                    roff = poff + start + p;
                    pos = (uint64_t)wcur * S + roff;
                } else if (h.kind == KIND_FAR) {
                    cblk = L.dword(LINE_DWORDS - 1u);
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_FAR;
                    poff += h.span;
                } else if (h.kind == KIND_CHUNK && cont == 0u) {
                    cdw = read_chunk_dword(L);
                    cblk = (wcur >> GROUP_SHIFT) * (GROUP + 1u) + GROUP;
                    if (cblk >= nlines) cblk = 0;
                    cont = KIND_CHUNK;
                    poff += h.span;
                    // (t stays bc minus the window's count word: the chunk's header says what the line's own pieces
                    // hold of f -- `left` may not be used for that: the scan above ran over the whole quarter, and in
                    // a line with a psi hint the quarter's last bytes are the hint, not pieces)
                } else {
                    move = 1;  // the window's pieces ran out first
                }
            } else {
                uint32_t r6[6], left = 0;
                load24(L, cdw + 2u, r6);
                const uint2 hd = L.u2(cdw);
                const uint32_t own = ((f < 3u ? hd.x : hd.y) >> (12u * ((f - 1u) & 1u))) & 0xFFFu;  // f's in the line's own pieces
                const uint32_t tc = t > own ? t - own : 0u;  // (0: never for a sound index -- select_in24 then reports position 0, refused below)
                const uint32_t p = select_in24(r6, f, tc > 4095u ? 4095u : tc, &left);
                const uint32_t csym = (hd.x >> 24) | ((hd.y >> 24) << 8);  // symbols the chunk holds; the next chunk follows
                found = tc != 0u && left == 0u && p < csym;
                roff = poff + p;
                pos = (uint64_t)wcur * S + roff;
                if (!found) move = 1;
            }
            bool to_sample = false;
            if (move < 0) {
                if (wcur == 0u) {  // count(0) = 0 < bc: only a corrupt index gets here
                    found = true;
                    pos = ix_n;
                }
                whi = wcur - 1u;
                wlo = wlo < whi ? wlo : whi;
            } else if (move > 0) {
                // past an open hint's reach: the next window, and the one after (four in five such rows lie in the
                // first window past the reach); then a sample bounds what is left
                if (guess_open && wcur >= whi) to_sample = ++otry > 2u;
                wlo = wcur + 1u;
                whi = whi > wlo ? whi : wlo;
                if (wlo >= nwin) {  // past the last window: a corrupt index
                    found = true;
                    pos = ix_n;
                    to_sample = false;
                }
            }
            if (move != 0) {
                wcur = wlo + ((whi - wlo) >> 1);
                cont = 0;
            }
            if (!found && ++tries > 72u) {
                found = true;
                pos = ix_n;
                to_sample = false;
            }
            if (to_sample) {
                wmin = wlo < nwin ? wlo : nwin - 1u;
                samp = sel[f * stride_m + ((bc - 1ull) >> sel_shift)];
                phase = 3;
            }
            if (found) {
                if (pos >= ix_n) {  // a corrupt index: end the read instead of walking off
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
        if (COUNT_WORK) xw[XW_STEPS] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(stepped));
        // ---- phase 0 (rows that just arrived, just stepped, or just got their own line): getF (rlebwt.cpp:307-314),
        // then the window of the bc-th f
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
                // The line the walk stands in may say where psi takes the rows of its window: then the next window is
                // known now (or bounded: the lower candidate is tried first), with no sample read and no round trip
                // for it.  The hint speaks of the F symbol of the window's first row: it holds for this row when that
                // row lies in f's block too.
                bool via_hint = false;
                if (hv && hw0 != HINT_NONE) {
                    const uint64_t r0 = idx - roff;  // the first row of the window the walk stands in
                    if (r0 >= cf && roff < S) {
                        const hint_range hr = hint_windows(hw0, hkk, roff, hshift);
                        if (hr.hi < nwin && hr.lo <= hr.hi) {
                            wlo = hr.lo;
                            whi = hr.hi;
                            wcur = wlo + ((whi - wlo) >> 1);
                            guess_open = hr.open;
                            otry = 0;
                            phase = 2;
                            cont = 0;
                            tries = 0;
                            via_hint = true;
                        }
                    }
                }
                hinted_now = via_hint;
                if (!via_hint) {
                    if (fresh && rich) {
                        phase = 4;  // this row's own window line carries the hint
                    } else {
                        wmin = 0;
                        samp = sel[f * stride_m + ((bc - 1ull) >> sel_shift)];
                        phase = 3;  // the sample is used from the next pass on
                    }
                }
                fresh = false;
                hv = false;
            }
        }
        if (COUNT_WORK) xw[XW_PROBES] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hinted_now));
    }
    }  // (the next shard)
    if (COUNT_WORK && lane == 0u) {
        xw[XW_CYCLES] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int i = 0; i < XW_WORDS; ++i) atomicAdd(&work[i], xw[i]);
    }
}

// The prefix, written right to left from the end of the row's buffer, moved to its head.  Sixteen lanes per row, 16
// bytes per lane and step (two aligned 16-byte loads and a funnel shift: the prefix starts wherever stride - plen
// falls), low addresses first: source >= destination, so a step never overwrites a later step's source, and within a
// step every lane has loaded before any lane stores.  The bytes of the last chunk past the prefix's end are whatever
// followed it: the postfix walk overwrites them (it keeps the bytes below plen & 15 of that chunk, no others).
__global__ void __launch_bounds__(256)
move_prefix16_kernel(uint8_t *__restrict__ out, uint32_t stride, const uint32_t *__restrict__ plen, size_t n) {
    const size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15u;
    uint32_t pl = r < n ? plen[r] : 0u;
    if (pl == 0xFFFFFFFFu || pl >= stride) pl = 0u;
    uint8_t *buf = out + (r < n ? r : 0) * (size_t)stride;
    const uint32_t off = stride - pl;
    for (uint32_t k0 = 0; __builtin_amdgcn_ballot_w64(k0 < pl) != 0ull; k0 += 256u) {  // (every lane of the wave reaches the barrier)
        const uint32_t k = k0 + 16u * sub;
        const bool active = k < pl;
        uint4 A = make_uint4(0, 0, 0, 0), B = make_uint4(0, 0, 0, 0);
        uint32_t sh = 0;
        if (active) {
            const uint32_t a = off + k, a0 = a & ~15u;
            sh = a & 15u;
            A = *reinterpret_cast<const uint4 *>(buf + a0);
            if (a0 + 16u < stride) B = *reinterpret_cast<const uint4 *>(buf + a0 + 16u);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): every lane's bytes are in before any store of this step
        __builtin_amdgcn_wave_barrier();
        if (active) {
            const uint32_t ds = sh >> 2, bs = sh & 3u;
            const uint32_t e0 = ds == 0u ? A.x : ds == 1u ? A.y : ds == 2u ? A.z : A.w;
            const uint32_t e1 = ds == 0u ? A.y : ds == 1u ? A.z : ds == 2u ? A.w : B.x;
            const uint32_t e2 = ds == 0u ? A.z : ds == 1u ? A.w : ds == 2u ? B.x : B.y;
            const uint32_t e3 = ds == 0u ? A.w : ds == 1u ? B.x : ds == 2u ? B.y : B.z;
            const uint32_t e4 = ds == 0u ? B.x : ds == 1u ? B.y : ds == 2u ? B.z : B.w;
            *reinterpret_cast<uint4 *>(buf + k) =
                make_uint4(__builtin_amdgcn_alignbyte(e1, e0, bs), __builtin_amdgcn_alignbyte(e2, e1, bs),
                           __builtin_amdgcn_alignbyte(e3, e2, bs), __builtin_amdgcn_alignbyte(e4, e3, bs));
        }
    }
}

// ... and for row buffers that are not 16-byte aligned: one wave per row, a byte per lane, 64 bytes per step.
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

hipError_t launch_extract_wave(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_rows,
                               size_t n, void *d_out, uint32_t stride, void *d_plen, void *d_len, int num_cus,
                               hipStream_t stream, unsigned long long *d_work) {
    if (n == 0 || nshards == 0) return hipSuccess;
    constexpr size_t MAX_ROWS = 1ull << 31;  // (the walk kernels number a shard's rows in 32 bits)
    if (n > MAX_ROWS) {
        if (nshards != 1) return hipErrorInvalidValue;  // (a set's caller cuts its batches)
        for (size_t i0 = 0; i0 < n; i0 += MAX_ROWS) {
            const size_t m = n - i0 < MAX_ROWS ? n - i0 : MAX_ROWS;
            const hipError_t e2 = launch_extract_wave(scratch, d_shards, 1, (const uint64_t *)d_rows + i0, m, (uint8_t *)d_out + i0 * (size_t)stride,
                                                      stride, (uint32_t *)d_plen + i0, (uint32_t *)d_len + i0, num_cus, stream, d_work);
            if (e2 != hipSuccess) return e2;
        }
        return hipSuccess;
    }
    scratch_cache::lease mem;
    // (the shards' row counters a line group apart, as the search kernels' query pools are: adjacent counters are one
    // line of one L2 channel that every wave of a small launch, and of any launch's tail, hits with atomics)
    const size_t pool_bytes = 2 * (size_t)nshards * POOL_STRIDE * sizeof(unsigned long long);
    hipError_t e = scratch.take(pool_bytes, stream, &mem);
    if (e != hipSuccess) return e;
    unsigned long long *pool = (unsigned long long *)mem.p;
    e = hipMemsetAsync(pool, 0, pool_bytes, stream);
    if (e != hipSuccess) {
        scratch.give(mem, stream);
        return e;
    }
    const size_t total = n * (size_t)nshards;
    size_t g = (total + 64 * XWG_WAVES - 1) / (64 * XWG_WAVES);
    // Workgroups: what is resident at once (4 per CU: 99 of 128 VGPRs, 32 KB of LDS each) and no more.  A walk kernel
    // ends in a tail as long as its longest walk (a few hundred passes, most lanes idle), so the more rows each lane
    // walks before that tail the better: the shards of a set are walked by ONE launch (a launch per shard, side by
    // side on streams of the set -- round 3 -- queued the shards' grids behind one another: every lane walked 8 rows
    // instead of 60 and a third of the lane-passes were idle).  RSBWT_EXTRACT_WGS_PER_CU overrides the 4 (A/B knob,
    // tools/README.md)
    static const size_t wgs_per_cu = [] {
        const char *e = getenv("RSBWT_EXTRACT_WGS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return (size_t)(v > 0 && v <= 20 ? v : RSB_WALK_MIN_WGS);
    }();
    const size_t cap = (size_t)num_cus * wgs_per_cu;
    if (g > cap) g = cap;
    if (g >= nshards) g -= g % nshards;  // (every shard starts with as many workgroups as any other)
    // rows per draw from a shard's counter: ROW_CHUNK for a launch that fills the chip, fewer for a small one so that
    // every wave launched draws twice or more (until round 5 a call of a few hundred rows -- the rows of a service
    // window's intervals -- gave all of them to the first wave that asked: 242 rows took four walks one after the other,
    // 1.24 ms, tools/probe_setquery.py)
    uint32_t row_chunk = ROW_CHUNK;
    while (row_chunk > 1u && (size_t)row_chunk * g * XWG_WAVES * 2u > total) row_chunk >>= 1;
    if (d_work)
        hipLaunchKernelGGL(extract_prefix_wave_kernel<true>, dim3((unsigned)g), dim3(64 * XWG_WAVES), 0, stream, d_shards, nshards,
                           (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (uint32_t *)d_plen, pool, d_work, row_chunk);
    else
        hipLaunchKernelGGL(extract_prefix_wave_kernel<false>, dim3((unsigned)g), dim3(64 * XWG_WAVES), 0, stream, d_shards, nshards,
                           (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (uint32_t *)d_plen, pool, d_work, row_chunk);
    // (the row buffers of the shards lie back to back: one launch moves every prefix)
    if ((stride & 15u) == 0u && ((uintptr_t)d_out & 15u) == 0u)
        hipLaunchKernelGGL(move_prefix16_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, (uint8_t *)d_out, stride,
                           (const uint32_t *)d_plen, total);
    else
        hipLaunchKernelGGL(move_prefix_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, (uint8_t *)d_out, stride,
                           (const uint32_t *)d_plen, total);
    if (d_work)
        hipLaunchKernelGGL(extract_postfix_wave_kernel<true>, dim3((unsigned)g), dim3(64 * XWG_WAVES), 0, stream, d_shards, nshards,
                           (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (const uint32_t *)d_plen, (uint32_t *)d_len,
                           pool + (size_t)nshards * POOL_STRIDE, d_work + XW_WORDS, row_chunk);
    else
        hipLaunchKernelGGL(extract_postfix_wave_kernel<false>, dim3((unsigned)g), dim3(64 * XWG_WAVES), 0, stream, d_shards, nshards,
                           (const uint64_t *)d_rows, n, (uint8_t *)d_out, stride, (const uint32_t *)d_plen, (uint32_t *)d_len,
                           pool + (size_t)nshards * POOL_STRIDE, d_work + XW_WORDS, row_chunk);
    e = hipGetLastError();
    scratch.give(mem, stream);
    return e;
}

}  // namespace rsb
