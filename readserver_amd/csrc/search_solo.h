// search_solo.h -- the search kernel with ONE LANE PER SEARCH (included by search_lines.hip only).
//
// search_lines_kernel gives a search two lanes, one per side of updateInterval (query.cpp:11-15), so
// a step is one pass and a wave has 64 lookups in flight for 32 searches.  Deep in a search the
// interval is narrower than a window: both positions then lie in the same line, the pair fetches it
// once and half of the wave's fetch slots stay empty -- on one shard with a deep k-mer table that is
// nearly every pass, and the launch falls short of the request rate the memory system can take.
// Here a lane owns a whole search: it looks up Occ(b, lower - 1); when upper lies within the symbols
// the same line holds (the usual case once the interval is narrow) Occ(b, upper) is ranked out of the
// same staged line in the same pass, otherwise upper is a lookup of its own in the next pass.  64
// searches per wave keep 64 lookups in flight whatever the intervals' width; a wide interval costs
// two passes per step instead of one, which a request-bound launch does not feel.
// Same start records, results, trace and counters as search_lines_kernel.
//
// STAGED RESULTS (plain searches: not COUNTS_ONLY / WL / WALK, no hit list; `pairs` bit 2 and 8 KB of dynamic LDS per
// workgroup): a lane's result is 16 bytes at its query's place, and lanes finish at passes of their own -- 8e7 scattered
// 16-byte stores per headline launch, 6.6e7 EA write requests (profiles/r05_pmc_search_kernel.json).  A wave draws its
// queries in order, so the 8 results of one 128-byte line of the result array all come from ONE wave, a few passes
// apart: they are collected in LDS -- 15 line buffers per wave, a group of 8 consecutive queries (g = q >> 3) in buffer
// g mod 15 if that was free when the group's first query was drawn -- and the lane that brings the last one in (an LDS
// counter per buffer) has lanes 0..7 store the line whole.  A group whose buffer was still taken is not staged: its
// lanes store as before.  The buffers' tags and counters are the wave's 16th line (2 KB per wave in all: with the line
// slots exactly the 160 KB of a CU at four workgroups).
//
// FUSED (a single shard behind a k-mer table, k <= 32, no trace): the kernel computes the start records itself -- no
// start-record launch before it, no 16 bytes written and read back per search.  The record of a search is its k-mer
// table entry (findInterval's answer for its last T symbols, src/bwt/query.cpp:18-21,24-41): a lane takes its NEXT
// search into reserve in two stages, one per pass -- the packed word and the validity byte, then the table entry the
// word names -- both loads flying with those passes' line fetches, so the dependent pair is never in front of a
// fetch; the search is taken up once the entry has landed.  On one shard the start-record kernel made three
// requests per search (table entry, record written, record read): this makes one.
//
// WL (the 1-mismatch search of a set, sets.hip / mm1_worklist.hip): a shard's searches are (1) wl_implicit items that
// need no record at all -- item i = variant r = i % 3T of k-mer q = i / 3T, substituted inside the k-mer table's reach:
// the lane spells the variant out of the k-mer's packed word and takes it into reserve in the two stages of FUSED
// (packed word and validity byte, then the variant's own table entry) -- and (2) the records of a WORKLIST behind them,
// 32 bytes each: {lower | next symbol << 40, upper, result index, packed word}, of a length only the device knows
// (wl_counts[s], written by the kernel that appended them): one 32-byte read per take-up.  Results go to the
// variant's canonical index (sparse results + hit map, as `pairs == 2`); a record flagged WL_DEAD is an empty slot.
//
// WALK (with FUSED; the 1-mismatch search of a set, sets.hip): the k-mers' own searches AND the step of the three
// substitutions at every position left of the k-mer table's reach, in one walk -- what a traced launch of the pair
// kernel and wl_branch_kernel (mm1_worklist.hip) did between them with a trace of 16 bytes per (k-mer, position, shard)
// written, read back, and every line fetched twice.  A lane walks its k-mer; where a lookup finds its position among
// the staged line's own pieces the three other bases are ranked off the same line (staged_occ_alts, wave_lines.h), and
// when the step completes the variants that survive it are appended to the shard's worklist (their interval after the
// step), the final ones stored as hits; a position that took a continuation line passes its three variants on
// UNSTEPPED, as the branch kernel did.  The k-mer's own interval is variant 0's hit.
#ifndef RSBWT_SEARCH_SOLO_H
#define RSBWT_SEARCH_SOLO_H

namespace rsb {

constexpr uint64_t WL_DEAD = 1ull << 63;  // worklist record: an empty slot (bit 63 of its first word)
constexpr uint32_t SOLO_STAGED_RESULTS = 4u;  // `pairs` bit 2: the launch brought SOLO_RESULTS_LDS bytes of dynamic LDS for staged results
constexpr uint32_t SOLO_RESULTS_LDS = WG_WAVES * 2048u;
constexpr uint32_t RES_BUFS = 15u;  // line buffers per wave (8 results each); the 16th line holds {tag, results still out} per buffer

#ifndef RSB_WALK1MM_WGS_PER_CU  // tuning knob (tools/build_variant.sh): the walk keeps three bases' counts across passes
#define RSB_WALK1MM_WGS_PER_CU 3
#endif
template <bool COUNT_WORK, bool COUNTS_ONLY, bool LONGK, bool FUSED = false, bool WL = false, bool WALK = false>
__global__ void __launch_bounds__(64 * WG_WAVES, WALK ? RSB_WALK1MM_WGS_PER_CU : RSB_MIN_WGS_PER_CU)
search_solo_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed,
                   const ulonglong2 *__restrict__ init, const uint8_t *__restrict__ valid,
                   unsigned long long *__restrict__ next_query,
                   size_t Q, uint32_t k, uint32_t wpq,
                   uint64_t *__restrict__ out_lower, uint64_t *__restrict__ out_upper,
                   unsigned long long *__restrict__ work,
                   ulonglong2 *__restrict__ trace, uint32_t trace_n, uint32_t qchunk, uint32_t pairs_arg,
                   const unsigned long long *__restrict__ wl_counts = nullptr, size_t wl_cap = 0, size_t wl_implicit = 0) {
    __shared__ uint4 s_stage[WG_WAVES][64 * SLOT_U4];
    extern __shared__ uint4 s_results[];  // staged results: [WG_WAVES][2][64] x 16 B when `pairs_arg` bit 2 is set (else none)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint4 *stage = s_stage[wave];
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    const uint32_t swz = lane & 7u;
    const uint32_t pairs = pairs_arg & 3u;
    constexpr bool CAN_STAGE = !COUNTS_ONLY && !WL && !WALK;
    // (group numbers are kept in 32 bits)
    const bool stage_out = CAN_STAGE && (pairs_arg & SOLO_STAGED_RESULTS) != 0u && pairs != 2u && Q < (1ull << 34);
    uint4 *results = s_results + wave * 128u;
    uint32_t *res_state = reinterpret_cast<uint32_t *>(results + RES_BUFS * 8u);  // [RES_BUFS] x {tag, out}
    const lds_u32 *mine0 = reinterpret_cast<const lds_u32 *>(stage + (lane & 7u) * 64u + (lane >> 3) * SLOT_U4);
#define SOLO_MINE(d) (mine0 + (((((uint32_t)(d)) >> 2) ^ swz) << 2) + (((uint32_t)(d)) & 3u))

    unsigned long long w_steps = 0, w_occ = 0, w_lines = 0, w_hops = 0, w_ktab = 0, passes = 0;
    unsigned long long w_surv = 0, w_unstepped = 0;  // WALK (wave-uniform: popcounts of ballots)

    uint32_t sid = blockIdx.x % nshards;
    for (uint32_t visited = 0; visited < nshards; ++visited, sid = (sid + 1u == nshards) ? 0u : sid + 1u) {
        const shard_view *sv = shards + sid;
        const char *lines_bytes = reinterpret_cast<const char *>(sv->lines);
        const uint32_t S = sv->sp.S;
        const double inv = sv->sp.inv;
        const uint32_t nlines = (uint32_t)sv->nlines;
        const bool ktab = view_uses_ktab(*sv, k);
        const int j_table = ktab ? (int)(k - sv->ktab_depth) - 1 : (int)k - 2;
        const uint32_t w_table = j_table > 0 ? (uint32_t)j_table >> 5 : 0u;
        // WL: `init` is the worklists, [nshards][wl_cap] records of two ulonglong2; wl_counts their lengths
        const ulonglong2 *init_s = WL ? init + (size_t)sid * wl_cap * 2u : init + (size_t)sid * Q;
        const size_t Qs = WALK ? Q : WL ? wl_implicit + (size_t)(wl_counts[(size_t)sid * WL_COUNT_STRIDE] < wl_cap ? wl_counts[(size_t)sid * WL_COUNT_STRIDE] : wl_cap) : Q;  // searches of this shard
        const uint32_t wl_tn = trace_n, wl_per = WL ? 3u * (k - trace_n) : 1u;  // (WL: trace_n carries tn; 3T implicit items per k-mer)
        // WALK: Q k-mers, results at their canonical indices in [nshards][Q * (3k+1)] sparse pairs / hit maps; `init` is
        // the worklists the surviving variants are appended to, wl_counts their lengths (device-side atomics)
        const size_t walk_V = 3u * (size_t)k + 1u, walk_mv = Q * walk_V;
        ulonglong2 *walk_wl = WALK ? const_cast<ulonglong2 *>(init) + (size_t)sid * wl_cap * 2u : nullptr;
        unsigned long long *walk_count = WALK ? const_cast<unsigned long long *>(wl_counts) + (size_t)sid * WL_COUNT_STRIDE : nullptr;
        uint64_t *out_lo = WALK ? out_lower + (size_t)sid * walk_mv * 2u : out_lower + (size_t)sid * Q * (pairs ? 2u : 1u);
        uint64_t *out_up = (COUNTS_ONLY || pairs) ? nullptr : out_upper + (size_t)sid * Q;
        unsigned long long *pool = next_query + (size_t)sid * POOL_STRIDE;
        if (visited != 0u && __hip_atomic_load(pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)Qs) continue;  // (search_lines_kernel: drained already)
        // per shard, as in search_lines_kernel: traces [s][Q][trace_n], hit maps [s][hit_map_words(Q)]
        ulonglong2 *trace_s = (trace && !WL) ? trace + (size_t)sid * Q * trace_n : nullptr;
        // WL: `trace` carries the implicit items' table entries read ahead for this shard (mm1_worklist.hip,
        // wl_table_entries_kernel: u64 [nshards][wl_implicit]), or nullptr = the lane reads the table itself
        const uint64_t *wl_pre = (WL && trace) ? reinterpret_cast<const uint64_t *>(trace) + (size_t)sid * wl_implicit : nullptr;
        unsigned long long *hit_map = pairs == 2u ? reinterpret_cast<unsigned long long *>(out_upper) + (size_t)sid * hit_map_words(WALK ? walk_mv : Q) : nullptr;
        // C[1..4] in lanes 0..3, picked with ds_bpermute (scalar loads: see search_lines_kernel)
        uint32_t ctab_lo, ctab_hi;
        {
            const uint64_t c1 = sv->C[1], c2 = sv->C[2], c3 = sv->C[3], c4 = sv->C[4];
            const uint32_t l3 = lane & 3u;
            const uint64_t cv = l3 == 0u ? c1 : l3 == 1u ? c2 : l3 == 2u ? c3 : c4;
            ctab_lo = (uint32_t)cv;
            ctab_hi = (uint32_t)(cv >> 32);
        }

        // FUSED: what start_record() reads of the shard, in scalar registers
        const uint64_t *__restrict__ ktab_p = sv->ktab;
        const uint32_t ktab_T = sv->ktab_depth, ktab_stride = sv->ktab_stride, ktab_fmt = sv->ktab_fmt;
        const uint64_t ix_n = sv->n;
        const uint64_t sc1 = sv->C[1], sc2 = sv->C[2], sc3 = sv->C[3], sc4 = sv->C[4];
        const uint64_t st1 = sv->total[1], st2 = sv->total[2], st3 = sv->total[3], st4 = sv->total[4];

        const uint32_t QCHUNK = qchunk;
        if (CAN_STAGE && stage_out && lane < RES_BUFS) {  // every buffer free (the previous shard's groups have all left)
            res_state[2u * lane] = ~0u;
            res_state[2u * lane + 1u] = 0u;
        }
        uint64_t pool_next = 0, pool_end = 0;  // wave-uniform
        bool drained = false;
        size_t q = 0;  // the query this lane is stepping
        bool has_q = false;
        size_t nq = 0;  // the one it runs next, start record already prefetched
        bool has_n = false;
        ulonglong2 nrec = {0, 0};
        uint64_t nword = 0;
        uint32_t nstage = 0;  // FUSED / WL implicit items: 1 = the reserve's word and validity byte are in, 2 = its table entry too (in nrec.x)
        bool nimp = false;    // WL: the reserve is an implicit item (nrec.y = validity | r << 8 | q << 32)
        int j = 0;
        uint64_t word = 0, lo = 0, hi = 0;
        // the step under way: sub 0 = looking up Occ(b, lower - 1), 1 = Occ(b, upper) with occL held;
        // cont != 0 = the lookup goes on in the group's spill line / a far line (as in the pair kernel)
        uint32_t sub = 0;
        uint64_t occL = 0, cacc = 0;
        uint32_t cont = 0, cblk = 0, cdw = 0, co = 0, tries = 0, w = 0;
        // WALK: Occ of the three other bases at lower - 1 (kept until the step's upper lookup is in too) and whether
        // both lookups of the step found their position among a staged line's own pieces
        uint64_t altL0 = 0, altL1 = 0, altL2 = 0, altU0 = 0, altU1 = 0, altU2 = 0;
        bool alt_ok = true;

        for (;;) {
            // ---- a lane whose query ended in the last pass takes up the one it had prefetched
            bool done = false;
            bool last_in = false;  // staged results: this lane's result completed its group's line
            if (FUSED && !has_q && has_n && nstage == 2u) {
                // start_record() (search_lines.hip) on the entry the reserve brought: an entry that is not an interval
                // of this BWT's rows (a damaged table) is not believed -- initInterval instead (query.cpp:18-21)
                const uint64_t e = nrec.x;
                const uint32_t width = (uint32_t)(e >> COUNT_BITS);
                if (nrec.y == 0ull) {
                    nrec.x = INIT_INVALID;
                } else if (width != KTAB_WIDE && (e & COUNT_MASK) + width <= ix_n) {
                    nrec.x = e & COUNT_MASK;
                    nrec.y = nrec.x + width - 1ull;
                } else {
                    const uint32_t bl = (uint32_t)((nword >> (2u * ((k - 1u) & 31u))) & 3u);
                    const uint64_t cb = bl == 0u ? sc1 : bl == 1u ? sc2 : bl == 2u ? sc3 : sc4;
                    const uint64_t tb = bl == 0u ? st1 : bl == 1u ? st2 : bl == 2u ? st3 : st4;
                    nrec.x = cb | INIT_FALLBACK;
                    nrec.y = cb + tb - 1ull;
                }
            }
            if (WL && !has_q && has_n && nstage == 2u) {
                has_q = true;
                has_n = false;
                sub = 0;
                cont = 0;
                word = nword;
                if (nimp) {
                    // an implicit item: start_record() on the variant's own table entry (its word is spelled out already)
                    const uint32_t r_ = (uint32_t)(nrec.y >> 8) & 0xFFu, q_ = (uint32_t)(nrec.y >> 32);
                    const uint32_t pr = (r_ * 171u) >> 9;  // r / 3 (r < 96)
                    q = (size_t)q_ * (3u * k + 1u) + 1u + 3u * (wl_tn + pr) + (r_ - 3u * pr);  // the variant's canonical index
                    const uint64_t e = nrec.x;
                    const uint32_t width = (uint32_t)(e >> COUNT_BITS);
                    bool tabulated = false;
                    if (width != KTAB_WIDE && (e & COUNT_MASK) + width <= ix_n) {
                        lo = e & COUNT_MASK;
                        hi = lo + width - 1ull;
                        j = (int)wl_tn - 1;
                        tabulated = true;
                    } else {  // not tabulated (or not an interval of this BWT): initInterval, query.cpp:18-21
                        const uint32_t bl = (uint32_t)((nword >> (2u * ((k - 1u) & 31u))) & 3u);
                        const uint64_t cb = bl == 0u ? sc1 : bl == 1u ? sc2 : bl == 2u ? sc3 : sc4;
                        const uint64_t tb = bl == 0u ? st1 : bl == 1u ? st2 : bl == 2u ? st3 : st4;
                        lo = cb;
                        hi = cb + tb - 1ull;
                        j = (int)k - 2;
                    }
                    // (the reference's unsigned compare, query.cpp:35: an empty interval at row 0, (0, 2^64 - 1), lives.
                    // And it looks at the interval only after an updateInterval, query.cpp:33-37: an initInterval that is
                    // empty -- no such symbol in the BWT -- is stepped on, as in the plain launches' start records: on a
                    // BWT whose first rows hold none of the next symbol either it becomes that wrapped interval)
                    done = (nrec.y & 0xFFull) == 0ull || (tabulated && lo > hi) || j < 0;
                } else {
                    q = nq;  // (the record's result index)
                    lo = nrec.x & COUNT_MASK;
                    hi = nrec.y;
                    j = (int)((nrec.x >> COUNT_BITS) & 0xFFFFull);
                    done = (nrec.x & WL_DEAD) != 0ull || lo > hi;
                }
                if (done) { lo = 1; hi = 0; }
            }
            if (!WL && !has_q && has_n && (!FUSED || nstage == 2u)) {
                has_q = true;
                has_n = false;
                q = nq;
                sub = 0;
                cont = 0;
                if (nrec.x & INIT_INVALID) {
                    lo = 1;
                    hi = 0;
                    j = -1;
                    done = true;
                } else if (nrec.x & INIT_EXPLICIT) {  // a 1-mismatch variant resuming its k-mer's search, or a query of a length of its own (INIT_VAR: search_lines.hip)
                    lo = nrec.x & COUNT_MASK;
                    hi = nrec.y;
                    j = (int)((nrec.x >> COUNT_BITS) & 0xFFFFull) - ((nrec.x & INIT_VAR) ? 1 : 0);
                    word = nword;
                    done = j < 0 || ((nrec.x & INIT_NOCHECK) == 0ull && lo > hi);
                    if (LONGK) {
                        if (!done && ((uint32_t)j >> 5) != w_table) word = packed[q * wpq + ((uint32_t)j >> 5)];
                    }
                } else {
                    const bool fallback = !ktab || (nrec.x & INIT_FALLBACK) != 0ull;
                    lo = nrec.x & COUNT_MASK;
                    hi = nrec.y;
                    j = fallback ? (int)k - 2 : j_table;
                    word = nword;
                    if (COUNT_WORK && !fallback) w_ktab += 1;
                    done = (j < 0) || (!fallback && lo > hi);  // query.cpp:35-37
                    if (LONGK) {
                        if (!done && ((uint32_t)j >> 5) != w_table) word = packed[q * wpq + ((uint32_t)j >> 5)];
                    }
                }
            }
            // ---- hand the next queries to the lanes that have none in reserve
            if (pool_next >= pool_end && !drained) {
                unsigned long long c = 0;
                if (lane == 0u) c = atomicAdd(pool, (unsigned long long)QCHUNK);
                c = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) |
                    __builtin_amdgcn_readfirstlane((uint32_t)c);
                pool_next = c;
                pool_end = c + QCHUNK < Qs ? c + QCHUNK : Qs;
                if (c >= Qs) { drained = true; pool_next = pool_end = 0; }
            }
            bool got_n = false;
            {
                const uint64_t want_mask = __builtin_amdgcn_ballot_w64(!has_n);
                const uint32_t before = (uint32_t)__builtin_popcountll(want_mask & ((1ull << lane) - 1ull));
                const uint64_t mine = pool_next + before;
                if (!has_n && mine < pool_end) {
                    nq = (size_t)mine;
                    got_n = true;
                }
                const uint64_t taken = pool_next + (uint32_t)__builtin_popcountll(want_mask);
                const uint64_t r0 = pool_next;
                pool_next = taken < pool_end ? taken : pool_end;
                if (CAN_STAGE && stage_out && pool_next > r0) {
                    // the groups of 8 queries that BEGIN among this pass's draws (at most 9): each gets its buffer,
                    // g mod 15, if that is free (chunks are multiples of 64 and this wave's alone: a group is drawn here whole)
                    const uint64_t first = (r0 + 7ull) & ~7ull;
                    if (first < pool_next) {
                        const uint32_t ga = (uint32_t)(first >> 3), gb = (uint32_t)((pool_next - 1ull) >> 3) + 1u;
                        const uint32_t ga15 = ga % RES_BUFS;
                        if (lane < RES_BUFS) {
                            const uint32_t d = lane >= ga15 ? lane - ga15 : lane + RES_BUFS - ga15;
                            const uint32_t g = ga + d;
                            if (g < gb && res_state[2u * lane + 1u] == 0u) {
                                const uint64_t q0 = (uint64_t)g << 3;
                                res_state[2u * lane] = g;
                                res_state[2u * lane + 1u] = (uint32_t)((q0 + 8ull < pool_end ? q0 + 8ull : pool_end) - q0);
                            }
                        }
                    }
                }
            }
            // (FUSED: a reserve whose table entry is still to come keeps the pass going -- it is taken up two passes
            // after it was drawn, and a wave whose lanes hold nothing else would otherwise spin here, or leave with it)
            if (__builtin_amdgcn_ballot_w64(has_q || got_n || ((FUSED || WL) && has_n)) == 0ull) {
                if (drained) break;
                continue;  // pool exhausted mid-pass: refill at the top
            }
            // the two start-up loads of a query taken into reserve fly with this pass's line fetches
            ulonglong2 rec = {0, 0};
            uint64_t first_word = 0;
            const bool stage_b = (FUSED || WL) && has_n && nstage == 1u;  // the reserve's word is in: its table entry now
            uint64_t entry = 0;
            uint64_t wl_index = 0, vword = 0;
            bool imp = false;
            if (WL) {
                if (got_n) {
                    imp = nq < wl_implicit;
                    if (imp) {
                        const uint32_t i32 = (uint32_t)nq, q_ = i32 / wl_per, r_ = i32 - q_ * wl_per;
                        first_word = packed[q_];
                        rec.y = (uint64_t)valid[q_] | ((uint64_t)r_ << 8) | ((uint64_t)q_ << 32);
                    } else {
                        const size_t slot = nq - wl_implicit;
                        rec = init_s[2u * slot];
                        const ulonglong2 r2 = init_s[2u * slot + 1u];
                        wl_index = r2.x;
                        first_word = r2.y;
                    }
                }
                if (stage_b) {  // spell the variant out, then its own table entry
                    const uint32_t r_ = (uint32_t)(nrec.y >> 8) & 0xFFu;
                    const uint32_t pr = (r_ * 171u) >> 9, d_ = r_ - 3u * pr, p_ = wl_tn + pr;
                    const uint32_t orig = (uint32_t)((nword >> (2u * p_)) & 3u);
                    const uint32_t alt = d_ < orig ? d_ : d_ + 1u;
                    vword = nword ^ ((uint64_t)(orig ^ alt) << (2u * p_));
                    const uint64_t code = (vword >> (2u * wl_tn)) & ((1ull << (2u * ktab_T)) - 1ull);
                    entry = wl_pre ? wl_pre[nq] : ktab_entry(ktab_p, ktab_fmt, ktab_T, ktab_stride, code);
                }
            } else if (FUSED) {
                if (got_n) {
                    rec.y = valid[nq];
                    first_word = packed[nq];
                }
                if (stage_b) {
                    const uint64_t code = (nword >> (2u * (k - ktab_T))) & ((1ull << (2u * ktab_T)) - 1ull);
                    entry = ktab_entry(ktab_p, ktab_fmt, ktab_T, ktab_stride, code);
                }
            } else if (got_n) {
                rec = init_s[nq];
                first_word = packed[nq * wpq + w_table];
            }
            const bool alive = has_q;
            const bool stepping = alive && !done;
            const bool fresh = stepping && cont == 0u;  // starts a lookup (of either position)

            // ---- this lane's lookup: symbol, position -> window line
            uint32_t b = 1, line = 0, o = 0;
            if (stepping) {
                if (LONGK) {
                    if (fresh && sub == 0u && (j & 31) == 31) word = packed[q * wpq + ((uint32_t)j >> 5)];
                }
                b = (uint32_t)((word >> (2u * ((uint32_t)j & 31u))) & 3u) + 1u;
            }
            bool no_fetch = false;  // Occ(b, -1) = 0 on the upper side too: the step completes without a line
            if (fresh) {
                if (sub == 0u) {
                    if (WALK) {  // a step begins: nothing known of the other bases yet (Occ(., -1) = 0 is what lower = 0 leaves)
                        alt_ok = true;
                        altL0 = altL1 = altL2 = altU0 = altU1 = altU2 = 0;
                    }
                    // traced search (1-mismatch): the interval this query has when about to take symbol j
                    if (trace_s && (uint32_t)j < trace_n) trace_s[q * trace_n + (uint32_t)j] = make_ulonglong2(lo, hi);
                    if (lo == 0ull) {  // Occ(b, -1) = 0 (rlebwt.cpp:269)
                        occL = 0;
                        sub = 1u;
                    }
                }
                // upper = 0 + 0 - 1 wraps after a step that found no b at the top of the BWT; the reference
                // carries on the same way and reports the empty interval one step later (query.cpp:11-15,35)
                const uint64_t p = sub ? hi : lo - 1ull;
                if (p == ~0ull) {
                    no_fetch = true;
                } else {
                    uint32_t pin;
                    w = fast_window(p, S, inv, pin);
                    line = w + (w >> GROUP_SHIFT);
                    o = pin + 1u;
                    if (line >= nlines) line = 0;  // never for p < n; keeps a bad position from faulting
                    if (COUNT_WORK) w_occ += 1;
                }
            }
            const bool looking = stepping && !no_fetch;
            // C[b], with every lane active: a ds_bpermute returns 0 from a masked-off source lane
            const uint64_t pb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)((b - 1u) << 2), (int)ctab_hi) << 32) |
                                (uint32_t)__builtin_amdgcn_ds_bpermute((int)((b - 1u) << 2), (int)ctab_lo);

            // ---- fetch: every looking lane its line
            const uint32_t want = looking ? (cont ? cblk : line) : ~0u;
            if (COUNT_WORK && want != ~0u) {
                if (cont) w_hops += 1;
                else w_lines += 1;
            }
            glds_fetch(lines_bytes, want, lane, stage_lds);
            glds_wait();

            // ---- Occ(b, p) out of this lane's staged line (RLEBWT::getOcc, rlebwt.cpp:268-301)
            bool do_scan = false, own_line = false;
            uint64_t base = 0, cnt_b = 0;
            uint32_t dw = HDR_DWORDS, rem = 0, oe_here = 0;
            uint32_t s1 = 0, s2 = 0, s3 = 0, span = 0, hb = 0;
            const sym_tab stab = make_sym_tab(b);  // v_perm_b32 table of the symbol (rank_device.h)
            if (looking) {
                if (cont != KIND_CHUNK) {
                    const uint32_t oe = cont ? co : o;
                    const uint2 cw = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(2u * (b - 1u)));
                    const uint4 h0 = *reinterpret_cast<const lds_u4 *>(SOLO_MINE(0));
                    const uint64_t cnt = ((uint64_t)(cw.y & 0xFFu) << 32) | cw.x;
                    const uint32_t m0 = h0.y >> 8, m1 = h0.w >> 8;
                    s1 = m0 & 0x3FFu;
                    s2 = (m0 >> 10) & 0x7FFu;
                    s3 = s2 + (m1 & 0x3FFu);
                    span = s3 + ((m1 >> 10) & 0x3FFu);
                    const uint32_t kind = (m1 >> 20) & 3u;
                    if (oe <= span) {
                        const uint32_t cq = (oe > s1 ? 1u : 0u) + (oe > s2 ? 1u : 0u) + (oe > s3 ? 1u : 0u);
                        const uint32_t start = cq == 0u ? 0u : cq == 1u ? s1 : cq == 2u ? s2 : s3;
                        const uint32_t hm = *SOLO_MINE(5u + 2u * ((b - 1u) >> 1)) >> 8;
                        hb = (hm >> (11u * ((b - 1u) & 1u))) & 0x7FFu;  // what quarters 0 and 1 hold of b
                        const uint32_t qd = HDR_DWORDS + 6u * (cq & 2u);
                        const uint2 x0 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(qd));
                        const uint2 x1 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(qd + 2u));
                        const uint2 x2 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(qd + 4u));
                        const uint32_t e6[6] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y};
                        const uint32_t m = matched24_tab(e6, stab);
                        base = cnt + (cq >= 2u ? hb : 0u) + ((cq & 1u) ? m : 0u);
                        dw = HDR_DWORDS + 6u * cq;
                        rem = oe - start;
                        do_scan = true;
                        own_line = true;  // the staged line's own pieces hold the position: they may hold upper too
                        oe_here = oe;
                        cnt_b = cnt;
                    } else if (kind == KIND_FAR) {
                        cblk = *SOLO_MINE(LINE_DWORDS - 1u);
                        if (cblk >= nlines) cblk = 0;  // never for a built index
                        cont = KIND_FAR;
                        co = oe - span;
                    } else if (kind == KIND_CHUNK && cont == 0u) {
                        const uint32_t m2 = *SOLO_MINE(5) >> 8, m3 = *SOLO_MINE(7) >> 8;
                        cacc = cnt;
                        cdw = 2u * (((m2 >> 22) & 3u) | (((m3 >> 22) & 3u) << 2));
                        cblk = (w >> GROUP_SHIFT) * (GROUP + 1u) + GROUP;
                        if (cblk >= nlines) cblk = 0;  // never for p < n
                        cont = KIND_CHUNK;
                        co = oe - span;
                    } else {  // a position beyond what the index holds: never for p < n
                        base = cnt;
                        do_scan = true;
                    }
                } else {
                    // the spill chunk: what the window's own 96 pieces hold of b, then the excess pieces
                    const uint2 hd = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(cdw));
                    const uint32_t hw = (b <= 2u) ? hd.x : hd.y;
                    const uint32_t tot = (hw >> (12u * ((b - 1u) & 1u))) & 0xFFFu;
                    base = cacc + tot;
                    dw = cdw + 2u;
                    rem = co;
                    do_scan = true;
                }
                // a window has at most 33 lines: the bound only guards against a corrupt chain
                if (!do_scan && ++tries > 72u) do_scan = true;
            }
            uint64_t occ = 0;
            {
                const uint2 y0 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(dw & 31u));
                const uint2 y1 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE((dw + 2u) & 31u));
                const uint2 y2 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE((dw + 4u) & 31u));
                const uint32_t r6[6] = {y0.x, y0.y, y1.x, y1.y, y2.x, y2.y};
                occ = base + rank24(r6, stab, b, rem);
#ifdef RSB_FAULT_INJECT_WILD_OCC  // fault-injection build (search_lines.hip): wild counts, answers WRONG by design
                if ((((uint32_t)q + (uint32_t)j) * 2654435761u >> 28) == 0u)
                    occ = (occ + (((uint64_t)q * 0x9E3779B97F4A7C15ull) >> (23u + (((uint32_t)q + (uint32_t)j) & 31u)))) & ((1ull << 41) - 1ull);
#endif
            }
            // ---- upper out of the same line: lower - 1 was found among the line's own pieces, and
            // upper lies d symbols further on, still among them
            bool step_done = false;
            uint64_t occU = 0;
            bool second = false;
            uint32_t oh = 0;
            // WALK: this pass's lookup found its position among the staged line's own pieces (the three other bases can be
            // ranked off the same line: for lower - 1 or for upper), or came out of a continuation line (they cannot)
            bool ev1 = false, ev1_is_L = false;
            if (WALK && do_scan && (uint32_t)j < trace_n) {
                if (own_line) {
                    ev1 = true;
                    ev1_is_L = sub == 0u;
                } else {
                    alt_ok = false;
                }
            }
            if (do_scan) {
                cont = 0;
                tries = 0;
                if (sub == 0u) {
                    occL = occ;
                    sub = 1u;
                    const uint64_t d = hi - (lo - 1ull);  // >= 1 for a live interval
                    if (own_line && d <= (uint64_t)(span - oe_here)) {
                        second = true;
                        oh = oe_here + (uint32_t)d;
                    }
                } else {
                    occU = occ;
                    step_done = true;
                }
            }
            {
                // (every lane takes part; only `second` lanes use the result)
                const uint32_t cq = (oh > s1 ? 1u : 0u) + (oh > s2 ? 1u : 0u) + (oh > s3 ? 1u : 0u);
                const uint32_t start = cq == 0u ? 0u : cq == 1u ? s1 : cq == 2u ? s2 : s3;
                const uint32_t qd = HDR_DWORDS + 6u * (cq & 2u);
                const uint2 x0 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(qd));
                const uint2 x1 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(qd + 2u));
                const uint2 x2 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(qd + 4u));
                const uint32_t e6[6] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y};
                const uint32_t m = matched24_tab(e6, stab);
                const uint32_t dw2 = HDR_DWORDS + 6u * cq;
                const uint2 y0 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(dw2));
                const uint2 y1 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(dw2 + 2u));
                const uint2 y2 = *reinterpret_cast<const lds_u2 *>(SOLO_MINE(dw2 + 4u));
                const uint32_t r6[6] = {y0.x, y0.y, y1.x, y1.y, y2.x, y2.y};
                const uint32_t sc = rank24(r6, stab, b, second ? oh - start : 0u);
                if (second) {
                    occU = cnt_b + (cq >= 2u ? hb : 0u) + ((cq & 1u) ? m : 0u) + sc;
                    step_done = true;
                    if (COUNT_WORK) w_occ += 1;
                }
            }
            if (stepping && no_fetch) {  // upper == 2^64 - 1: Occ = 0 without a line (only on the upper side)
                occU = 0;
                step_done = true;
            }
            if (WALK) {
                // ---- the three other bases at this pass's position(s): one copy of the rank code, run for the lookup
                // of the pass and again for upper where it was ranked off the same line (wave-uniform skips)
                const staged_line Lrow = {mine0, swz};
                const line_head hh = {s1, s2, s3, span, 0u};
#pragma unroll 1
                for (uint32_t ev = 0; ev < 2u; ++ev) {
                    const bool want = ev == 0u ? ev1 : (second && (uint32_t)j < trace_n);
                    if (__builtin_amdgcn_ballot_w64(want) == 0ull) continue;
                    uint64_t t3[3];
                    staged_occ_alts(Lrow, hh, want ? (ev == 0u ? oe_here : oh) : 1u, b - 1u, t3);
                    if (want) {
                        if (ev == 0u && ev1_is_L) { altL0 = t3[0]; altL1 = t3[1]; altL2 = t3[2]; }
                        else { altU0 = t3[0]; altU1 = t3[1]; altU2 = t3[2]; }
                    }
                }
                // ---- a step of a position left of the tables' reach completes: updateInterval (query.cpp:11-15) with each
                // of the three substituted symbols; the variants that survive become searches (mm1_worklist.hip's records)
                const bool emit = step_done && (uint32_t)j < trace_n;
                if (__builtin_amdgcn_ballot_w64(emit) != 0ull) {
                    const uint32_t orig = b - 1u;
                    uint64_t nlo[3], nhi[3], masks[3];
                    bool enq[3];
#pragma unroll
                    for (uint32_t d = 0; d < 3u; ++d) {
                        const uint32_t alt = d < orig ? d : d + 1u;  // the d-th base of ACGT without the original one
                        const uint64_t cb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(alt << 2), (int)ctab_hi) << 32) |
                                            (uint32_t)__builtin_amdgcn_ds_bpermute((int)(alt << 2), (int)ctab_lo);
                        nlo[d] = cb + (d == 0u ? altL0 : d == 1u ? altL1 : altL2);
                        nhi[d] = cb + (d == 0u ? altU0 : d == 1u ? altU1 : altU2) - 1ull;
                        const bool stepped_live = emit && alt_ok && nlo[d] <= nhi[d];
                        const bool final_hit = stepped_live && j == 0;
                        enq[d] = (stepped_live && j != 0) || (emit && !alt_ok);
                        if (final_hit) {
                            const size_t canon = q * walk_V + 1u + 3u * (uint32_t)j + d;
                            reinterpret_cast<ulonglong2 *>(out_lo)[canon] = make_ulonglong2(nlo[d], nhi[d]);
                            atomicOr(hit_map + (canon >> 6), 1ull << (canon & 63u));
                        }
                        masks[d] = __builtin_amdgcn_ballot_w64(enq[d]);
                        if (COUNT_WORK) {
                            w_surv += __builtin_popcountll(__builtin_amdgcn_ballot_w64((enq[d] && alt_ok) || final_hit));
                            w_unstepped += __builtin_popcountll(__builtin_amdgcn_ballot_w64(enq[d] && !alt_ok));
                        }
                    }
                    if (COUNT_WORK && emit) w_steps += 3;
                    // ONE append per wave and pass for the three rounds together
                    const uint32_t n0 = (uint32_t)__builtin_popcountll(masks[0]), n1 = (uint32_t)__builtin_popcountll(masks[1]);
                    const uint32_t total = n0 + n1 + (uint32_t)__builtin_popcountll(masks[2]);
                    if (total != 0u) {
                        unsigned long long wbase = 0;
                        if (lane == 0u) wbase = atomicAdd(walk_count, (unsigned long long)total);
                        wbase = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wbase >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wbase);
                        const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
                        for (uint32_t d = 0; d < 3u; ++d) {
                            if (!enq[d]) continue;
                            const size_t slot = (size_t)wbase + (d >= 1u ? n0 : 0u) + (d >= 2u ? n1 : 0u) + (size_t)__builtin_popcountll(masks[d] & lt);
                            if (slot >= wl_cap) continue;  // (room for every variant: the caller's sizing)
                            const uint32_t alt = d < orig ? d : d + 1u;
                            const uint64_t vword = word ^ ((uint64_t)(orig ^ alt) << (2u * (uint32_t)j));
                            const uint64_t canon = (uint64_t)q * walk_V + 1ull + 3ull * (uint32_t)j + d;
                            if (!alt_ok) wl_store(walk_wl, slot, lo, (uint32_t)j, hi, canon, vword);            // the step is the search kernel's
                            else wl_store(walk_wl, slot, nlo[d], (uint32_t)j - 1u, nhi[d], canon, vword);     // already taken
                        }
                    }
                }
            }
            // ---- updateInterval (query.cpp:11-15)
            if (step_done) {
                if (COUNT_WORK) w_steps += 1;
                lo = pb + occL;
                hi = pb + occU - 1ull;
                --j;
                done = (lo > hi) || (j < 0);  // query.cpp:35-37
                sub = 0;
            }
            if ((FUSED || WL) && stage_b) {
                nrec.x = entry;
                nstage = 2u;
                if (WL) nword = vword;
            }
            if (got_n) {
                nrec = rec;
                nword = first_word;
                has_n = true;
                nstage = (WL && !imp) ? 2u : 1u;
                nimp = imp;
                if (WL && !imp) nq = (size_t)wl_index;  // from here on the search is known by its result index
            }
            if (alive && done) {
                if (trace_s) {  // (WL: `trace` carries the table entries read ahead, not a trace)
                    // the positions it never reached: a search resumed there ends where this one did
                    for (int jj = j < (int)trace_n ? j : (int)trace_n - 1; jj >= 0; --jj)
                        trace_s[q * trace_n + (uint32_t)jj] = make_ulonglong2(lo, hi);
                }
                if (COUNTS_ONLY) {
                    if (hi >= lo) out_lo[q] = hi - lo + 1ull;  // service.cpp:304; the array is zeroed before the launch: most searches end empty and store nothing
                } else if (pairs == 2u) {
                    // sparse results (1-mismatch hit list): only a search that ends on an interval leaves
                    // anything -- its {lower, upper} at its own place and its bit in the map at out_upper (an
                    // atomic nobody waits for; a counter handing out list positions would stall the wave for a
                    // round trip per hit and serialise on one address).  compact_hits orders them afterwards.
                    if (lo <= hi) {
                        const size_t qi = WALK ? q * walk_V : q;  // (WALK: the k-mer itself is variant 0 of its 3k + 1)
                        reinterpret_cast<ulonglong2 *>(out_lo)[qi] = make_ulonglong2(lo, hi);
                        atomicOr(hit_map + (qi >> 6), 1ull << (qi & 63u));
                    }
                } else {
                    const uint32_t g32 = (uint32_t)(q >> 3), rb = g32 % RES_BUFS;
                    if (CAN_STAGE && stage_out && res_state[2u * rb] == g32) {
                        results[rb * 8u + ((uint32_t)q & 7u)] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                        last_in = atomicSub(&res_state[2u * rb + 1u], 1u) == 1u;  // (LDS: ds_add_rtn_u32)
                    } else if (pairs) {
                        reinterpret_cast<ulonglong2 *>(out_lo)[q] = make_ulonglong2(lo, hi);
                    } else {
                        out_lo[q] = lo;
                        out_up[q] = hi;
                    }
                }
                has_q = false;
            }
            if (CAN_STAGE && stage_out) {
                // the buffers whose last result has just come in: lanes 0..7 store the line whole (pairs), or its 64
                // bytes of lower and of upper
                uint64_t fm = __builtin_amdgcn_ballot_w64(last_in);
                while (fm != 0ull) {
                    const int src = __builtin_ctzll(fm);
                    fm &= fm - 1ull;
                    // (q still names the query that has just ended)
                    const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(q >> 3), src), bsel = g % RES_BUFS;
                    const size_t qq = ((size_t)g << 3) + lane;
                    if (lane < 8u && qq < Qs) {
                        const uint4 v = results[bsel * 8u + lane];
                        const uint64_t vlo = ((uint64_t)v.y << 32) | v.x, vhi = ((uint64_t)v.w << 32) | v.z;
                        if (pairs) {
                            reinterpret_cast<ulonglong2 *>(out_lo)[qq] = make_ulonglong2(vlo, vhi);
                        } else {
                            out_lo[qq] = vlo;
                            out_up[qq] = vhi;
                        }
                    }
                }
            }
            if (COUNT_WORK) ++passes;
        }
    }
    if (COUNT_WORK) {
        if (lane == 0u) atomicAdd(&work[WORK_PASSES], passes);
        if (threadIdx.x == 0u && blockIdx.x == 0u) work[WORK_SOLO] = 1ull;  // which kernel ran
        if (w_steps) atomicAdd(&work[WORK_STEPS], w_steps);
        if (w_occ) atomicAdd(&work[WORK_OCC], w_occ);
        if (w_lines) atomicAdd(&work[WORK_LINES], w_lines);
        if (w_ktab) atomicAdd(&work[WORK_KTAB], w_ktab);
        if (w_hops) atomicAdd(&work[WORK_HOPS], w_hops);
        if (WALK && lane == 0u) {
            atomicAdd(&work[13], w_surv);       // variants alive after the step of their position (mm1_worklist.hip's words)
            atomicAdd(&work[14], w_unstepped);  // variants passed on unstepped
        }
    }
#undef SOLO_MINE
}

}  // namespace rsb
#endif
