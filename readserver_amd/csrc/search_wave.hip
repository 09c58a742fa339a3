// search_wave.hip -- batched findInterval, wave-cooperative form (gfx950).
//
// Same algorithm and layout as kernels.hip's octet kernel (findInterval, src/bwt/query.cpp:24-41;
// one directory entry + one 128-B block per Occ lookup), different work split.  The octet kernel
// spends ~20 VALU instructions per lookup, most of them per-quad overhead replicated 16 times per
// wave, and is VALU-issue bound.  Here
//   * a wavefront carries 32 queries: lane i resolves Occ(b, lower-1) of query i, lane i+32
//     Occ(b, upper); each lane does its own slot arithmetic / directory lookup;
//   * blocks are still fetched the way the memory system likes them (tools/gather_bench.hip): a
//     full 128-B line per group of adjacent lanes, by LDS-DMA straight into the wave's LDS stage
//     (glds_fetch below), each distinct block of a query once;
//   * every lane then ranks ITS block out of LDS: the header names the quarter holding the
//     position and (slots) what the quarters before it hold of every symbol, so only that one
//     quarter's 16 runs are scanned run by run (SDWA, 5 VALU per run); classic blocks (24-run
//     quarters, no such counts) add the earlier quarters up 4 bytes at a time (v_dot4_u32_u8
//     against a 0/1 match mask).
// Overhead is shared by 64 lookups instead of 16 and the scan is 16 bytes instead of 96 per lookup.
// Requires slots or the exact (s = 8) directory; other indexes use the octet kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "bwt_device.h"
#include "kernels.h"

namespace rsb {

// LDS stage: 128 B per lane, 8 KB per wave, 32 KB per 4-wave workgroup, so exactly 5 workgroups
// (20 waves) fit a CU's 160 KB.  Where a lane's row sits and how its 16-byte chunks are swizzled
// follows from the fetch (glds_fetch).
constexpr int SLOT_U4 = 8;
constexpr int WG_WAVES = 4; // waves per workgroup (one-wave groups would pack 17 per CU but measured 1.4x slower)

// The stage is written as uint4 and parsed as dwords / 8- / 16-byte pieces: the read types may
// alias anything, or type-based alias analysis lets hipcc reuse values read before a re-fetch.
typedef uint32_t __attribute__((may_alias)) lds_u32;
typedef uint2 __attribute__((may_alias)) lds_u2;
typedef uint4 __attribute__((may_alias)) lds_u4;

// Fetch of up to 64 blocks into the wave's LDS stage, direct to LDS (global_load_lds_dwordx4,
// gfx950): no register round trip, no ds_write pass.  One instruction writes 1 KB of LDS in lane
// order, so the work is split the way that makes this the stage layout itself: instruction k
// (0..7) serves lanes T = 8o + k, the eight lanes of octet o each bringing 16 B of the block
// lane T wants (a full 128-B line per octet: the request shape tools/gather_bench.hip measures
// fastest).  Lane T's block then sits at k * 1 KB + o * 128 B, chunk c at position c ^ k -- the
// swizzle is applied on the SOURCE side (lane l of the octet loads chunk (l & 7) ^ k) -- so a wave
// reading one chunk of every row is bank-conflict free.  8 KB per wave, 32 KB per 4-wave
// workgroup: exactly 5 workgroups (20 waves) fit a CU's 160 KB.
// want == ~0u: that lane needs nothing (its octet's lanes are masked off for that instruction and
// the row keeps what it held).
typedef __attribute__((address_space(3))) void *lds_void_ptr;
typedef const __attribute__((address_space(1))) void *global_void_ptr;

__device__ __forceinline__ void glds_fetch(const char *blocks, uint32_t want, uint32_t lane, uint32_t stage_lds) {
    uint32_t tb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        tb[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & ~7u) + k) << 2), (int)want);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (tb[k] != ~0u) {
            const char *src = blocks + (uint64_t)tb[k] * 128u + (((lane & 7u) ^ (uint32_t)k) << 4);
            __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)(uintptr_t)(stage_lds + k * 1024u), 16, 0, 0);
        }
    }
}
// the blocks are in LDS once every outstanding load has returned
// (the builtin, not inline asm: hipcc's wait-count bookkeeping then knows nothing is outstanding and
// does not add its own vmcnt(0) at the head of the next pass, in front of that pass's loads, where
// it would sit out the result stores of the queries that just ended)
__device__ __forceinline__ void glds_wait() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt left alone (gfx9 encoding)
    asm volatile("" ::: "memory");
}

// Start state of every query, computed ahead of the search so that a query entering the wave
// costs one independent 16-byte load instead of a chain (validity byte + packed word -> k-mer
// table entry) in front of every pass.  Record = { lower | flags, upper }.
constexpr uint64_t INIT_INVALID = 1ull << 63;   // symbol outside ACGT: result (1, 0)
constexpr uint64_t INIT_FALLBACK = 1ull << 62;  // not from the k-mer table: continue at symbol k-2
constexpr uint64_t INIT_EXPLICIT = 1ull << 61;  // continue at the symbol named in bits 40..55 (1-mismatch variants)

template <bool KTAB>
__device__ __forceinline__ ulonglong2 start_record(const rsbwt_view &ix, const uint64_t *pq, uint32_t k) {
    ulonglong2 rec;
    const uint64_t last = pq[(k - 1u) >> 5];
    if (KTAB) {
        const uint32_t T = ix.ktab_depth;
        const uint32_t off = 2u * (k - T);
        const uint32_t w0 = off >> 6, sh = off & 63u;
        uint64_t bits = (w0 == ((k - 1u) >> 5) ? last : pq[w0]) >> sh;
        if (sh + 2u * T > 64u) bits |= last << (64u - sh);
        const uint64_t e = ix.ktab[bits & ((1ull << (2u * T)) - 1ull)];
        const uint32_t width = (uint32_t)(e >> RSBWT_COUNT_BITS);
        if (width != RSBWT_KTAB_WIDE) {
            rec.x = e & RSBWT_COUNT_MASK;
            rec.y = rec.x + width - 1ull;
            return rec;
        }
    }
    // initInterval, query.cpp:18-21
    const uint32_t b = (uint32_t)((last >> (2u * ((k - 1u) & 31u))) & 3u) + 1u;
    rec.x = ix.C[b] | INIT_FALLBACK;
    rec.y = ix.C[b] + ix.total[b] - 1ull;
    return rec;
}

template <bool KTAB>
__global__ void __launch_bounds__(256)
search_init_kernel(const rsbwt_view ix, const uint64_t *__restrict__ packed,
                   const uint8_t *__restrict__ valid, size_t Q, uint32_t k, uint32_t wpq,
                   ulonglong2 *__restrict__ init) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    ulonglong2 rec;
    if (valid[q] == 0) {
        rec.x = INIT_INVALID;
        rec.y = 0;
    } else {
        rec = start_record<KTAB>(ix, packed + q * wpq, k);
    }
    init[q] = rec;
}

// Start records of the 3k+1 variants of m k-mers (1-mismatch search, variants_kernel's order).  A
// variant whose substituted position is left of the k-mer table's reach shares its whole suffix
// with the k-mer itself: it starts from the interval the k-mer's own (traced) search had when it
// was about to take that position -- trace[q][pos] -- and takes the substituted symbol first.
template <bool KTAB>
__global__ void __launch_bounds__(256)
search_init_1mm_kernel(const rsbwt_view ix, const uint64_t *__restrict__ vpacked,
                       const uint8_t *__restrict__ vvalid, size_t mv, uint32_t k, uint32_t wpq, uint32_t V,
                       const ulonglong2 *__restrict__ trace, uint32_t trace_n,
                       ulonglong2 *__restrict__ init) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mv) return;
    ulonglong2 rec;
    if (vvalid[i] == 0) {
        rec.x = INIT_INVALID;
        rec.y = 0;
    } else {
        const size_t q = i / V;
        const uint32_t v = (uint32_t)(i - q * V);
        const uint32_t pos = v ? (v - 1u) / 3u : ~0u;
        if (pos < trace_n) {
            const ulonglong2 t = trace[q * trace_n + pos];
            rec.x = (t.x & RSBWT_COUNT_MASK) | ((uint64_t)pos << RSBWT_COUNT_BITS) | INIT_EXPLICIT;
            rec.y = t.y;
        } else {
            rec = start_record<KTAB>(ix, vpacked + i * wpq, k);
        }
    }
    init[i] = rec;
}

// LONGK: k > 32, i.e. a query spans several packed words.  A template parameter because with the
// reload on the path -- however it is guarded at run time -- hipcc waits for vmcnt(0) at the top of
// every pass, which also waits for the start-up loads just issued by entering lanes.
template <bool COUNT_WORK, bool COUNTS_ONLY, bool KTAB, bool SLOTS, bool LONGK>
__global__ void __launch_bounds__(64 * WG_WAVES)
search_wave_kernel(const rsbwt_view ix, const slot_view sv, const uint64_t *__restrict__ packed,
                   const ulonglong2 *__restrict__ init, unsigned long long *__restrict__ next_query,
                   size_t Q, uint32_t k, uint32_t wpq,
                   uint64_t *__restrict__ out_lower, uint64_t *__restrict__ out_upper,
                   unsigned long long *__restrict__ work,
                   ulonglong2 *__restrict__ trace, uint32_t trace_n, uint32_t qchunk) {
    __shared__ uint4 s_stage[WG_WAVES][64 * SLOT_U4];
    // C[b]: lanes 0..3 of every wave keep C[1..4] and a lane picks its symbol's entry with two
    // ds_bpermute reads (the LDS crossbar idles while the VALU is the busier pipe; selects out
    // of scalar registers cost 15 VALU instructions a pass.  Left as ix.C[b], hipcc turns the
    // lookup into a dependent global load from the kernel-argument segment in every pass; an LDS
    // table would cost the 128 bytes that keep a fifth workgroup off the CU.)
    uint32_t ctab_lo, ctab_hi;
    {
        const uint32_t l3 = threadIdx.x & 3u;
        const uint64_t cv = l3 == 0u ? ix.C[1] : l3 == 1u ? ix.C[2] : l3 == 2u ? ix.C[3] : ix.C[4];
        ctab_lo = (uint32_t)cv;
        ctab_hi = (uint32_t)(cv >> 32);
    }

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t side = lane >> 5;  // 0: lower-1 side, 1: upper side
    uint4 *stage = s_stage[wave];
    const char *blocks_bytes = reinterpret_cast<const char *>(SLOTS ? sv.slots : ix.blocks);
    // LDS byte address of this wave's stage, in a scalar register (it goes to M0)
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    const uint32_t nblk_total = SLOTS ? (uint32_t)(sv.p.nslots + sv.noverflow) : 0u;

    // Queries are handed out dynamically: a wave draws chunks of QCHUNK consecutive queries from
    // one global counter (one atomic per chunk) and gives the next one to whichever lane pair
    // has finished (ballot + popcount, no further atomics).  A static q += stride schedule makes
    // every wave as slow as its unluckiest pair: 36 % idle lane-passes on the bench batch.
    const uint32_t QCHUNK = qchunk;  // chosen by the launcher: 1024 for big batches, less so that small ones still spread over the GPU
    uint64_t pool_next = 0, pool_end = 0;  // wave-uniform
    bool drained = false;                  // the global counter ran past Q
    size_t q = 0;          // the query this lane pair is stepping
    bool has_q = false;
    size_t nq = 0;         // the one it runs next, start record already prefetched
    bool has_n = false;
    ulonglong2 nrec = {0, 0};
    uint64_t nword = 0;
    int j = 0;
    uint64_t word = 0, lo = 0, hi = 0;
    unsigned long long w_steps = 0, w_occ = 0, w_blocks = 0, w_ktab = 0;
    // counting mode also stamps where a pass spends its cycles (shares only: the stamps fence)
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, stamp = 0, passes = 0;
#define STAMP(i)                                                        \
    if (COUNT_WORK) {                                                   \
        __builtin_amdgcn_sched_barrier(0);                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
        ph[i] += now_ - stamp;                                          \
        stamp = now_;                                                   \
        __builtin_amdgcn_sched_barrier(0);                              \
    }
    if (COUNT_WORK) stamp = __builtin_amdgcn_s_memtime();

    // symbol a table-started query continues with, and the packed word holding it
    const int j_table = KTAB ? (int)(k - ix.ktab_depth) - 1 : (int)k - 2;
    const uint32_t w_table = j_table > 0 ? (uint32_t)j_table >> 5 : 0u;

    for (;;) {
        // ---- a pair whose query ended in the last pass takes up the one it had prefetched: it
        // steps in this very pass (a start record fetched on demand instead would cost every query
        // one idle pass, ~10 % of all lane-passes on the bench batch)
        bool done = false;
        if (!has_q && has_n) {
            has_q = true;
            has_n = false;
            q = nq;
            if (nrec.x & INIT_INVALID) {
                lo = 1;
                hi = 0;
                j = -1;
                done = true;
            } else if (nrec.x & INIT_EXPLICIT) {  // a 1-mismatch variant resuming its k-mer's search
                lo = nrec.x & RSBWT_COUNT_MASK;
                hi = nrec.y;
                j = (int)((nrec.x >> RSBWT_COUNT_BITS) & 0xFFFFull);
                word = nword;
                done = lo > hi;  // the shared suffix was already absent (query.cpp:35-37)
                if (LONGK) {
                    if (!done && ((uint32_t)j >> 5) != w_table) word = packed[q * wpq + ((uint32_t)j >> 5)];
                }
            } else {
                const bool fallback = !KTAB || (nrec.x & INIT_FALLBACK) != 0ull;
                lo = nrec.x & RSBWT_COUNT_MASK;
                hi = nrec.y;
                j = fallback ? (int)k - 2 : j_table;
                word = nword;
                if (COUNT_WORK && !fallback) w_ktab += 1;
                // a tabulated suffix that is already empty ends the search (query.cpp:35-37)
                done = (j < 0) || (!fallback && lo > hi);
                if (LONGK) {
                    if (!done && ((uint32_t)j >> 5) != w_table) word = packed[q * wpq + ((uint32_t)j >> 5)];
                }
            }
        }
        // ---- hand the next queries to the lane pairs that have none in reserve
        if (pool_next >= pool_end && !drained) {
            unsigned long long c = 0;
            if (lane == 0u) c = atomicAdd(next_query, (unsigned long long)QCHUNK);
            c = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)c);
            pool_next = c;
            pool_end = c + QCHUNK < Q ? c + QCHUNK : Q;
            if (c >= Q) { drained = true; pool_next = pool_end = 0; }
        }
        bool got_n = false;
        {
            const uint32_t want_mask = (uint32_t)__builtin_amdgcn_ballot_w64(!has_n);  // low half = pairs
            const uint32_t before = __builtin_popcount(want_mask & ((1u << (lane & 31u)) - 1u));
            const uint64_t mine = pool_next + before;
            if (!has_n && mine < pool_end) {
                nq = (size_t)mine;
                got_n = true;
            }
            const uint64_t taken = pool_next + __builtin_popcount(want_mask);
            pool_next = taken < pool_end ? taken : pool_end;
        }
        if (__builtin_amdgcn_ballot_w64(has_q || got_n) == 0ull) {
            if (drained) break;
            continue;  // pool exhausted mid-pass: refill at the top
        }
        // The two start-up loads of a query taken into reserve fly together with this pass's block
        // fetches.  (Issuing them behind the block loads, from every lane, lets hipcc wait for the
        // blocks with a counted vmcnt -- measured 4 % slower: more registers, two more loads.)
        ulonglong2 rec = {0, 0};
        uint64_t first_word = 0;
        if (got_n) {
            rec = init[nq];
            first_word = packed[nq * wpq + w_table];
        }
        const bool alive = has_q;
        const bool stepping = alive && !done;

        // ---- this lane's lookup: symbol, position, directory entry -> block id
        uint32_t b = 1, blk = 0, pin = 0;
        uint64_t p = 0, pb = 0;
        bool skip = false;
        if (stepping) {
            // traced search (1-mismatch): the interval this query has when about to take symbol j
            if (trace && side == 0u && (uint32_t)j < trace_n) trace[q * trace_n + (uint32_t)j] = make_ulonglong2(lo, hi);
            if (LONGK) {
                if ((j & 31) == 31) word = packed[q * wpq + ((uint32_t)j >> 5)];
            }
            b = (uint32_t)((word >> (2u * ((uint32_t)j & 31u))) & 3u) + 1u;
            // Occ(b, -1) = 0: lower - 1 at lower == 0, and upper itself after a step that found no b
            // at the top of the BWT (upper = 0 + 0 - 1 wraps; the reference carries on the same way
            // and reports the empty interval one step later: query.cpp:11-15,35, rlebwt.cpp:269)
            p = side ? hi : lo - 1ull;
            skip = p == ~0ull;
            if (skip) p = 0;
            if (SLOTS) {
                blk = __umulhi((uint32_t)(p >> sv.p.a), sv.p.magic) >> sv.p.shift;  // p / S
                pin = (uint32_t)p - blk * sv.p.S;
                if (blk >= nblk_total) blk = 0;  // never for p < n; keeps a bad position from faulting
            } else {
                const uint2 e = ix.dir[p >> 8];
                blk = dir_decode<true>(ix, e, p);
            }
        }
        // C[b], with every lane active: a ds_bpermute returns 0 from a masked-off source lane
        pb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)((b - 1u) << 2), (int)ctab_hi) << 32) |
             (uint32_t)__builtin_amdgcn_ds_bpermute((int)((b - 1u) << 2), (int)ctab_lo);
        const uint32_t first_blk = blk;
        STAMP(0)  // pass set-up: symbol, position, slot

        // ---- fetch: one request per distinct block.  Lanes with nothing to look up ask for
        // nothing, and the upper side of a query whose two positions fall in the same block (more
        // than 4 in 10 lookups on the bench batch) reads the lower side's row instead of fetching
        // the block again: the L1's outstanding-request slots are what this kernel runs out of.
        uint32_t want = (stepping && !skip) ? blk : ~0u;
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(want, want, false, false);  // full exec
            const uint32_t other_want = side ? sw[0] : sw[1];
            if (side != 0u && want == other_want) want = ~0u;
        }
        const bool shared_row = stepping && !skip && want == ~0u;
        glds_fetch(blocks_bytes, want, lane, stage_lds);
        STAMP(1)  // issue of the block loads
        glds_wait();
        STAMP(2)  // wait for the blocks
        // dword d of this lane's block: chunk d >> 2 at (d >> 2) ^ swz(lane) of the row
        // (lane & 7) * 1 KB + (lane >> 3) * 128 B; lanes l and l + 32 share k and the swizzle
        const uint32_t swz = lane & 7u;
        const lds_u32 *own_row = reinterpret_cast<const lds_u32 *>(stage + (lane & 7u) * 64u + (lane >> 3) * SLOT_U4);
        const lds_u32 *mine0 = shared_row ? own_row - 4 * 32 : own_row;  // row of lane - 32
#define MINE(d) (mine0 + (((((uint32_t)(d)) >> 2) ^ swz) << 2) + (((uint32_t)(d)) & 3u))
        uint32_t hops = 0;
        // Occ(b, p) out of this lane's staged block, `off` = p's offset in it (slots) / derived
        // from the block's P0 (classic).  RLEBWT::getOcc, src/bwt/rlebwt.cpp:268-301.
        auto rank_staged = [&](uint32_t off) -> uint64_t {
            // the count word of symbol b (block_format.h / slots.hip)
            const uint2 cw = *reinterpret_cast<const lds_u2 *>(MINE(8u * (b - 1u)));
            const uint64_t cnt = ((uint64_t)(cw.y & 0xFFu) << 32) | cw.x;
            if (SLOTS) {
                // slot: 4 quarters of { word0, word1, 16 runs }; word1 of quarter t = what quarters
                // 0..t hold of A,C,G,T -- the scan of one quarter is all there is to add
                const uint32_t m0 = *MINE(1) >> 8, m1 = *MINE(9) >> 8;
                const uint32_t s1 = m0 >> 12, s2 = m1 & 0xFFFu, s3 = m1 >> 12;
                const uint32_t o = off + 1u;
                const uint32_t cq = (o > s1 ? 1u : 0u) + (o > s2 ? 1u : 0u) + (o > s3 ? 1u : 0u);
                const uint32_t start = cq == 0u ? 0u : cq == 1u ? s1 : cq == 2u ? s2 : s3;
                const uint32_t pq = cq ? cq - 1u : 0u;
                const uint2 held = *reinterpret_cast<const lds_u2 *>(MINE(8u * pq + 2u));
                const uint64_t h64 = ((uint64_t)held.y << 32) | held.x;
                const uint32_t before = cq ? (uint32_t)(h64 >> (11u * (b - 1u))) & 0x7FFu : 0u;
                const uint4 x = *reinterpret_cast<const lds_u4 *>(MINE(8u * cq + 4u));
                const uint32_t r[4] = {x.x, x.y, x.z, x.w};
                // run by run (RLEBWT::getOcc's scan, src/bwt/rlebwt.cpp:281-298)
                return cnt + before + runs_scan<4>(r, b, o - start);
            }
            // classic block: 4 x { header word, 24 runs }; the quarters before the one holding the
            // position are added up 4 runs per dot4
            const uint32_t m2 = *MINE(17) >> 8, m3 = *MINE(25) >> 8;
            const uint32_t s1 = m2 >> 12, s2 = m3 & 0xFFFu, s3 = m3 >> 12;
            off = ((uint32_t)p - (*MINE(1) >> 8)) & 0xFFFFFFu;  // exact directory: inside the block
            const uint32_t o = off + 1u;
            const uint32_t cq = (o > s1 ? 1u : 0u) + (o > s2 ? 1u : 0u) + (o > s3 ? 1u : 0u);
            const uint32_t start = cq == 0u ? 0u : cq == 1u ? s1 : cq == 2u ? s2 : s3;
            const uint32_t bb = __umul24(b, 0x010101u) | (b << 24);  // b in every byte (full-rate ops)
            uint32_t before = 0;
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) {
                const uint2 x0 = *reinterpret_cast<const lds_u2 *>(MINE(8 * qt + 2));
                const uint4 x1 = *reinterpret_cast<const lds_u4 *>(MINE(8 * qt + 4));
                uint32_t m = dword_matched(x0.x, bb, 0u);
                m = dword_matched(x0.y, bb, m);
                m = dword_matched(x1.x, bb, m);
                m = dword_matched(x1.y, bb, m);
                m = dword_matched(x1.z, bb, m);
                m = dword_matched(x1.w, bb, m);
                before += (cq > (uint32_t)qt) ? m : 0u;
            }
            lane_block lb;
            {
                const uint2 x0 = *reinterpret_cast<const lds_u2 *>(MINE(8u * cq + 2u));
                const uint4 x1 = *reinterpret_cast<const lds_u4 *>(MINE(8u * cq + 4u));
                lb.r[0] = x0.x; lb.r[1] = x0.y; lb.r[2] = x1.x; lb.r[3] = x1.y; lb.r[4] = x1.z; lb.r[5] = x1.w;
                lb.hdr_lo = 0; lb.hdr_hi = 0;
            }
            return cnt + before + lane_scan(lb, b, o - start);
        };
        uint64_t occ = 0;
        uint32_t off = pin;
        bool need = false;
        if (stepping && !skip) {
            // ranked straight away; the rare lane whose window continues in an overflow block is
            // found out below and ranked again (keeps the LDS round trip of the span test off the
            // common path)
            if (SLOTS) need = off >= ((*MINE(1) >> 8) & 0xFFFu);
            occ = rank_staged(off);
        }
        STAMP(3)  // rank out of LDS
        if (SLOTS) {
            // a window has at most S < 4096 pieces = 64 blocks: the bound only guards against a
            // corrupt chain, so that every wave drains
            for (int guard = 0; guard < 72 && __builtin_amdgcn_ballot_w64(need) != 0ull; ++guard) {
                uint32_t want = ~0u;
                const bool was = need;
                if (need) {
                    const uint32_t m2 = *MINE(17) >> 8;
                    want = *MINE(26);  // next: word1 of quarter 3
                    if (((m2 >> 12) & 1u) == 0u || want == 0u || want >= nblk_total) { want = ~0u; need = false; }  // never for p < n
                    else { blk = want; ++hops; }
                }
                glds_fetch(blocks_bytes, want, lane, stage_lds);
                glds_wait();
                if (want != ~0u) mine0 = own_row;  // an overflow block always lands in the lane's own row
                if (need) {
                    off = pin - ((*MINE(17) >> 8) & 0xFFFu);  // ostart: meta_2 of the new block
                    need = off >= ((*MINE(1) >> 8) & 0xFFFu);
                }
                if (was && !need) occ = rank_staged(off);
            }
        }
        STAMP(4)  // overflow chains
        // ---- the two sides of a query trade results; updateInterval (query.cpp:11-15)
        const auto sw_lo = __builtin_amdgcn_permlane32_swap((uint32_t)occ, (uint32_t)occ, false, false);
        const auto sw_hi = __builtin_amdgcn_permlane32_swap((uint32_t)(occ >> 32), (uint32_t)(occ >> 32), false, false);
        const uint64_t other = side ? (((uint64_t)sw_hi[0] << 32) | sw_lo[0]) : (((uint64_t)sw_hi[1] << 32) | sw_lo[1]);
        uint32_t other_blk = 0, other_hops = 0;
        if (COUNT_WORK) {  // swapped with every lane active
            const auto sb = __builtin_amdgcn_permlane32_swap(first_blk, first_blk, false, false);
            other_blk = side ? sb[0] : sb[1];
            const auto sh = __builtin_amdgcn_permlane32_swap(hops, hops, false, false);
            other_hops = side ? sh[0] : sh[1];
        }
        if (stepping) {
            const uint64_t occL = side ? other : occ;
            const uint64_t occU = side ? occ : other;
            if (COUNT_WORK && side == 0u) {
                w_steps += 1;
                w_occ += skip ? 1 : 2;
                w_blocks += ((skip || other_blk == first_blk) ? 1 : 2) + hops + other_hops;
            }
            lo = pb + occL;
            hi = pb + occU - 1ull;
            --j;
            done = (lo > hi) || (j < 0);  // query.cpp:35-37
        }
        if (got_n) {
            nrec = rec;
            nword = first_word;
            has_n = true;
        }
        if (alive && done) {
            if (trace && side == 0u) {
                // the positions it never reached: a search resumed there ends where this one did
                for (int jj = j < (int)trace_n ? j : (int)trace_n - 1; jj >= 0; --jj)
                    trace[q * trace_n + (uint32_t)jj] = make_ulonglong2(lo, hi);
            }
            if (side == 0u) {
                if (COUNTS_ONLY) {
                    out_lower[q] = hi >= lo ? hi - lo + 1ull : 0ull;  // service.cpp:304
                } else {
                    out_lower[q] = lo;
                    out_upper[q] = hi;
                }
            }
            has_q = false;
        }
        STAMP(5)  // exchange, update, start-up decode, result stores
        if (COUNT_WORK) ++passes;
    }
    if (COUNT_WORK) {
        if (lane == 0u) {
            for (int i = 0; i < 6; ++i) atomicAdd(&work[4 + i], ph[i]);
            atomicAdd(&work[10], passes);
        }
        if (side == 0u && (w_steps || w_ktab)) {
            atomicAdd(&work[0], w_steps);
            atomicAdd(&work[1], w_occ);
            atomicAdd(&work[2], w_blocks);
            atomicAdd(&work[3], w_ktab);
        }
    }
}

template <bool CW, bool CO, bool KT>
static void launch_w2(const slot_view *sv, int grid, hipStream_t stream, const rsbwt_view &ix,
                      const uint64_t *pk, const ulonglong2 *init, size_t Q, uint32_t k, uint32_t wpq,
                      uint64_t *lo, uint64_t *up, unsigned long long *work, ulonglong2 *trace, uint32_t trace_n) {
    unsigned long long *ctr = (unsigned long long *)(init + Q);  // zeroed counter behind the records
    slot_view none = {};
    // queries per draw from the pool: at least ~4 draws per wave, so that a batch of a few
    // thousand queries (a service micro-batch, the k-mers of a 1-mismatch slice) still occupies
    // every wave launched instead of the first few
    uint32_t qchunk = 1024;
    while (qchunk > 32u && (size_t)qchunk * (size_t)grid * WG_WAVES * 4u > Q) qchunk >>= 1;
#define RSB_LAUNCH_W(SL, LK)                                                                              \
    hipLaunchKernelGGL((search_wave_kernel<CW, CO, KT, SL, LK>), dim3(grid), dim3(64 * WG_WAVES), 0, stream, \
                       ix, (SL ? *sv : none), pk, init, ctr, Q, k, wpq, lo, up, work, trace, trace_n, qchunk)
    if (sv) {
        if (wpq > 1) RSB_LAUNCH_W(true, true);
        else RSB_LAUNCH_W(true, false);
    } else {
        if (wpq > 1) RSB_LAUNCH_W(false, true);
        else RSB_LAUNCH_W(false, false);
    }
#undef RSB_LAUNCH_W
}

template <bool CW, bool CO>
static void launch_w(bool ktab, const slot_view *sv, int grid, hipStream_t stream, const rsbwt_view &ix,
                     const uint64_t *pk, const ulonglong2 *init, size_t Q, uint32_t k, uint32_t wpq,
                     uint64_t *lo, uint64_t *up, unsigned long long *work, ulonglong2 *trace, uint32_t trace_n) {
    if (ktab) launch_w2<CW, CO, true>(sv, grid, stream, ix, pk, init, Q, k, wpq, lo, up, work, trace, trace_n);
    else launch_w2<CW, CO, false>(sv, grid, stream, ix, pk, init, Q, k, wpq, lo, up, work, trace, trace_n);
}

hipError_t launch_search_wave(const rsbwt_view &ix, const slot_view *sv, const void *d_packed,
                              const void *d_valid, size_t Q, uint32_t k, void *d_lower, void *d_upper,
                              bool counts_only, unsigned long long *d_work, int num_cus, hipStream_t stream,
                              hipEvent_t ev0, hipEvent_t ev1, const wave_search_extra *extra) {
    if (Q == 0) return hipSuccess;
    ulonglong2 *trace = extra ? (ulonglong2 *)extra->d_trace_out : nullptr;
    const uint32_t trace_n = extra ? extra->trace_n : 0u;
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    // 32 queries per wave, 4 waves per workgroup
    const size_t per_wg = 32u * WG_WAVES;
    size_t g = (Q + per_wg - 1) / per_wg;
    // Workgroups per CU: LDS admits 5 (20 waves), but 4 are as fast (2.39-2.41 against 2.40-2.50 ms:
    // the request path, not the number of lookups in flight, is what saturates) and leave 32 KB of
    // LDS and wave slots per CU to kernels that run beside the search -- RCCL's, when the previous
    // batch's intervals are gathered at N > 1.  3 are 8 % slower.  RSBWT_WAVE_WGS_PER_CU overrides.
    static const int wgs_per_cu = [] {
        const char *e = getenv("RSBWT_WAVE_WGS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 4;
    }();
    const size_t cap = (size_t)num_cus * (size_t)wgs_per_cu;
    if (g > cap) g = cap;
    const int grid = (int)g;
    const uint64_t *pk = (const uint64_t *)d_packed;
    const uint8_t *vd = (const uint8_t *)d_valid;
    uint64_t *lo = (uint64_t *)d_lower, *up = (uint64_t *)d_upper;
    const bool ktab = ix.ktab != nullptr && ix.ktab_depth >= 2 && k >= ix.ktab_depth;
    // start records of this batch: stream-ordered scratch, so concurrent calls do not share state
    ulonglong2 *init = nullptr;
    hipError_t e = hipMallocAsync((void **)&init, (Q + 1) * sizeof(ulonglong2), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(init + Q, 0, sizeof(ulonglong2), stream);  // the query counter
    if (e != hipSuccess) return e;
    const int ig = (int)((Q + 255) / 256);
    if (extra && extra->d_trace_in) {
        const ulonglong2 *tin = (const ulonglong2 *)extra->d_trace_in;
        if (ktab) hipLaunchKernelGGL(search_init_1mm_kernel<true>, dim3(ig), dim3(256), 0, stream, ix, pk, vd, Q, k, wpq, extra->variants, tin, trace_n, init);
        else hipLaunchKernelGGL(search_init_1mm_kernel<false>, dim3(ig), dim3(256), 0, stream, ix, pk, vd, Q, k, wpq, extra->variants, tin, trace_n, init);
    } else if (ktab) hipLaunchKernelGGL(search_init_kernel<true>, dim3(ig), dim3(256), 0, stream, ix, pk, vd, Q, k, wpq, init);
    else hipLaunchKernelGGL(search_init_kernel<false>, dim3(ig), dim3(256), 0, stream, ix, pk, vd, Q, k, wpq, init);
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (d_work) {
        if (counts_only) launch_w<true, true>(ktab, sv, grid, stream, ix, pk, init, Q, k, wpq, lo, up, d_work, trace, trace_n);
        else launch_w<true, false>(ktab, sv, grid, stream, ix, pk, init, Q, k, wpq, lo, up, d_work, trace, trace_n);
    } else {
        if (counts_only) launch_w<false, true>(ktab, sv, grid, stream, ix, pk, init, Q, k, wpq, lo, up, d_work, trace, trace_n);
        else launch_w<false, false>(ktab, sv, grid, stream, ix, pk, init, Q, k, wpq, lo, up, d_work, trace, trace_n);
    }
    e = hipGetLastError();
    if (ev1) (void)hipEventRecord(ev1, stream);
    const hipError_t e2 = hipFreeAsync(init, stream);
    return e != hipSuccess ? e : e2;
}



// Entries per k-mer of a traced search = the positions left of the k-mer table's reach (0: the
// 1-mismatch search has nothing to share: no table, or k within it)
uint32_t wave_trace_entries(const rsbwt_view &ix, uint32_t k) {
    const bool ktab = ix.ktab != nullptr && ix.ktab_depth >= 2 && k >= ix.ktab_depth;
    if (!ktab || k <= ix.ktab_depth || k - ix.ktab_depth > 0xFFFFu) return 0;
    return k - ix.ktab_depth;
}

}  // namespace rsb
