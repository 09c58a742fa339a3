// search_lines.hip -- batched findInterval over window lines, wave-cooperative (gfx950).
//
// findInterval, src/bwt/query.cpp:24-41, for a batch of k-mers against one or several shards
// resident on this GPU (every query goes to every shard: src/service/server.cpp:124,578).
//   * a wavefront carries 32 (query, shard) searches: lane i resolves Occ(b, lower-1) of search i,
//     lane i+32 Occ(b, upper) (updateInterval, query.cpp:11-15); a wave works on one shard at a
//     time, so everything that depends on the shard sits in scalar registers;
//   * one Occ lookup = ONE 128-byte request: the window line of position p is at a computable
//     address (line_format.h).  Lines are fetched the way the memory system likes them
//     (tools/gather_bench.hip): a full line per octet of lanes, by LDS-DMA straight into the wave's
//     LDS stage, each distinct line of a query once;
//   * every lane then ranks ITS line out of LDS: the header names the quarter holding the
//     position and what the first half holds of every symbol; at most one earlier quarter is added
//     up 4 runs at a time (v_dot4_u32_u8 against a 0/1 match mask) and the quarter itself is
//     scanned run by run (SDWA, 4.5 VALU per run);
//   * the ~1.5 % of lookups whose position lies past the pieces its window's line holds (spill
//     chunk / far line) are NOT chased inside the pass: the lane keeps what it learned (base count,
//     where to go, symbols left) and fetches the continuation in its next pass, while the other
//     lanes step on -- no wave ever waits a second round trip for one lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"
#include "line_format.h"
#include "rank_device.h"
#include "wave_lines.h"

namespace rsb {

// Start state of every (query, shard) search, computed ahead of the search so that a search
// entering the wave costs one independent 16-byte load instead of a chain (validity byte + packed
// word -> k-mer table entry) in front of every pass.  Record = { lower | flags, upper }.
constexpr uint64_t INIT_INVALID = 1ull << 63;   // symbol outside ACGT: result (1, 0)
constexpr uint64_t INIT_FALLBACK = 1ull << 62;  // not from the k-mer table: continue at symbol k-2
constexpr uint64_t INIT_EXPLICIT = 1ull << 61;  // continue at the symbol named in bits 40..55 (1-mismatch variants)
constexpr uint64_t INIT_NOCHECK = 1ull << 60;   // (with INIT_EXPLICIT) an initInterval: the reference looks at it only after its first update (query.cpp:33-37)
constexpr uint64_t INIT_VAR = 1ull << 59;       // (with INIT_EXPLICIT) a query of a length of its own: bits 40..55 hold the next symbol's index + 1 (0: none left)

__device__ __forceinline__ ulonglong2 start_record(const shard_view &ix, const uint64_t *pq, uint32_t k) {
    ulonglong2 rec;
    const uint64_t last = pq[(k - 1u) >> 5];
    if (view_uses_ktab(ix, k)) {
        const uint32_t T = ix.ktab_depth;
        const uint32_t off = 2u * (k - T);
        const uint32_t w0 = off >> 6, sh = off & 63u;
        uint64_t bits = (w0 == ((k - 1u) >> 5) ? last : pq[w0]) >> sh;
        if (sh + 2u * T > 64u) bits |= last << (64u - sh);
        const uint64_t e = ktab_entry(ix.ktab, ix.ktab_fmt, T, ix.ktab_stride, bits & ((1ull << (2u * T)) - 1ull));
        const uint32_t width = (uint32_t)(e >> COUNT_BITS);
        // an entry is an interval of this BWT's rows: one that is not (a damaged table) is not believed --
        // the search then starts from initInterval like an untabulated one and still ends on the right rows
        if (width != KTAB_WIDE && (e & COUNT_MASK) + width <= ix.n) {
            rec.x = e & COUNT_MASK;
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

// one thread per (query, shard): init[s * Q + q].  Adjacent lanes = the shards of one query, so that
// interleaved k-mer tables are read one stretch per query.  (A thread per query that goes through the shards itself
// writes whole 1 KB stretches instead of 16-byte records Q apart, and is slower: eight table reads one behind the
// other per thread -- 0.19 against 0.15 ms per launch, profiles/r03g_kernel_stats.csv.)
__global__ void __launch_bounds__(256)
search_init_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed,
                   const uint8_t *__restrict__ valid, size_t Q, uint32_t k, uint32_t wpq,
                   ulonglong2 *__restrict__ init) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q * nshards) return;
    const size_t q = i / nshards, s = i - q * nshards;
    ulonglong2 rec;
    if (valid[q] == 0) {
        rec.x = INIT_INVALID;
        rec.y = 0;
    } else {
        rec = start_record(shards[s], packed + q * wpq, k);
    }
    init[s * Q + q] = rec;
}

// Start records of a batch whose queries have LENGTHS OF THEIR OWN (len[q] symbols, packed from word q * wpq on): what
// the service loop's windows hold -- the reference answers a request of any length up to the front-end's limit, and a
// launch per distinct length is a launch sequence and a copy back per length (75 of them in a window of 4,096 reads
// requests).  A record says where ITS search goes on (INIT_EXPLICIT | INIT_VAR: bits 40..55 = index of
// the next symbol + 1): from the k-mer table entry of the query's last T symbols when it has that many, else from
// initInterval (INIT_NOCHECK: not looked at before its first update, query.cpp:33-37); the search kernels take such
// records up as they take a resumed 1-mismatch variant's.
__global__ void __launch_bounds__(256)
search_init_var_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed,
                       const uint8_t *__restrict__ valid, const uint32_t *__restrict__ len, size_t Q, uint32_t wpq,
                       ulonglong2 *__restrict__ init) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q * nshards) return;
    const size_t q = i / nshards, s = i - q * nshards;
    const uint32_t L = len[q];
    ulonglong2 rec;
    if (valid[q] == 0 || L == 0u) {
        rec.x = INIT_INVALID;
        rec.y = 0;
    } else {
        const shard_view &ix = shards[s];
        const uint64_t *pq = packed + q * wpq;
        bool tabulated = false;
        if (view_uses_ktab(ix, L)) {
            const uint32_t T = ix.ktab_depth;
            const uint32_t off = 2u * (L - T);
            const uint32_t w0 = off >> 6, sh = off & 63u;
            uint64_t bits = pq[w0] >> sh;
            if (sh + 2u * T > 64u) bits |= pq[w0 + 1u] << (64u - sh);
            const uint64_t e = ktab_entry(ix.ktab, ix.ktab_fmt, T, ix.ktab_stride, bits & ((1ull << (2u * T)) - 1ull));
            const uint32_t width = (uint32_t)(e >> COUNT_BITS);
            if (width != KTAB_WIDE && (e & COUNT_MASK) + width <= ix.n) {  // (start_record's rule)
                rec.x = (e & COUNT_MASK) | ((uint64_t)(L - T) << COUNT_BITS) | INIT_EXPLICIT | INIT_VAR;  // next symbol: L - T - 1
                rec.y = (e & COUNT_MASK) + width - 1ull;
                tabulated = true;
            }
        }
        if (!tabulated) {  // initInterval, query.cpp:18-21; next symbol: L - 2
            const uint32_t b = (uint32_t)((pq[(L - 1u) >> 5] >> (2u * ((L - 1u) & 31u))) & 3u) + 1u;
            rec.x = ix.C[b] | ((uint64_t)(L - 1u) << COUNT_BITS) | INIT_EXPLICIT | INIT_VAR | INIT_NOCHECK;
            rec.y = ix.C[b] + ix.total[b] - 1ull;
        }
    }
    init[s * Q + q] = rec;
}

// The same through LDS for 2..256 shards: a block computes V = 256 / S queries x S shards (adjacent lanes = the shards
// of one query: one stretch of the interleaved tables per query), turns the records round in LDS and writes V
// consecutive records per shard -- 512 B per shard and block on 8 shards where the plain kernel writes 256 records of
// 16 B each into 256 different lines (it ran at a request ceiling of its own: 0.66 ms of the 21 ms headline step).
__global__ void __launch_bounds__(256)
search_init_tiled_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed,
                         const uint8_t *__restrict__ valid, size_t Q, uint32_t k, uint32_t wpq, uint32_t V,
                         ulonglong2 *__restrict__ init) {
    __shared__ ulonglong2 tile[256];
    const size_t q0 = (size_t)blockIdx.x * V;
    const uint32_t t = threadIdx.x;
    {
        const uint32_t v = t / nshards, s = t - v * nshards;
        if (v < V && q0 + v < Q) {
            const size_t q = q0 + v;
            ulonglong2 rec;
            if (valid[q] == 0) {
                rec.x = INIT_INVALID;
                rec.y = 0;
            } else {
                rec = start_record(shards[s], packed + q * wpq, k);
            }
            tile[v * nshards + s] = rec;
        }
    }
    __syncthreads();
    {
        const uint32_t s = t / V, v = t - s * V;
        if (s < nshards && q0 + v < Q) init[(size_t)s * Q + q0 + v] = tile[v * nshards + s];
    }
}

// The same with FOUR queries per thread (round 5).  The tiled kernel is a chain of three dependent memory round trips
// per thread -- validity byte and packed word, then the table record they name, then the store -- and ran at the rate
// that chain allows at full occupancy: 0.86 ms per 10^7 queries x 8 shards where its requests (10^7 table stretches,
// 10^7 record lines) would take 0.4.  Four independent chains per thread: the four word loads are issued together,
// then the four table loads.
constexpr uint32_t INIT_U = 4;
__global__ void __launch_bounds__(256)
search_init_tiled4_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed,
                          const uint8_t *__restrict__ valid, size_t Q, uint32_t k, uint32_t V, ulonglong2 *__restrict__ init) {
    __shared__ ulonglong2 tile[INIT_U][256];
    const size_t q0 = (size_t)blockIdx.x * V * INIT_U;
    const uint32_t t = threadIdx.x;
    {
        const uint32_t v = t / nshards, s = t - v * nshards;
        if (v < V) {
            const shard_view &ix = shards[s];
            const bool tab = view_uses_ktab(ix, k);
            const uint32_t T = ix.ktab_depth, off = tab ? 2u * (k - T) : 0u;  // (k <= 32: one packed word per k-mer)
            const uint64_t *ktab = ix.ktab;
            const uint32_t fmt = ix.ktab_fmt, stride = ix.ktab_stride;
            const uint64_t n = ix.n;
            uint8_t ok[INIT_U];
            uint64_t word[INIT_U], e[INIT_U];
#pragma unroll
            for (uint32_t u = 0; u < INIT_U; ++u) {
                const size_t q = q0 + (size_t)u * V + v;
                const bool in = q < Q;
                ok[u] = in ? valid[q] : (uint8_t)2;  // (2: no such query)
                word[u] = in ? packed[q] : 0ull;
            }
#pragma unroll
            for (uint32_t u = 0; u < INIT_U; ++u)
                e[u] = (tab && ok[u] == 1) ? ktab_entry(ktab, fmt, T, stride, (word[u] >> off) & ((1ull << (2u * T)) - 1ull)) : ~0ull;
#pragma unroll
            for (uint32_t u = 0; u < INIT_U; ++u) {
                if (ok[u] == 2) continue;
                ulonglong2 rec;
                const uint32_t width = (uint32_t)(e[u] >> COUNT_BITS);
                if (ok[u] == 0) {
                    rec.x = INIT_INVALID;
                    rec.y = 0;
                } else if (tab && width != KTAB_WIDE && (e[u] & COUNT_MASK) + width <= n) {  // (start_record: an entry that is an interval of this BWT)
                    rec.x = e[u] & COUNT_MASK;
                    rec.y = rec.x + width - 1ull;
                } else {  // initInterval, query.cpp:18-21
                    const uint32_t b = (uint32_t)((word[u] >> (2u * ((k - 1u) & 31u))) & 3u) + 1u;
                    rec.x = ix.C[b] | INIT_FALLBACK;
                    rec.y = ix.C[b] + ix.total[b] - 1ull;
                }
                tile[u][v * nshards + s] = rec;
            }
        }
    }
    __syncthreads();
    {
        const uint32_t s = t / V, v = t - s * V;
        if (s < nshards) {
#pragma unroll
            for (uint32_t u = 0; u < INIT_U; ++u) {
                const size_t q = q0 + (size_t)u * V + v;
                if (q < Q) init[(size_t)s * Q + q] = tile[u][v * nshards + s];
            }
        }
    }
}

// Start records of the 3k+1 variants of m k-mers (1-mismatch search, variants_kernel's order).  A
// variant whose substituted position is left of the k-mer table's reach shares its whole suffix
// with the k-mer itself: it starts from the interval the k-mer's own (traced) search had when it
// was about to take that position -- trace[q][pos] -- and takes the substituted symbol first.
// Several shards per launch (a set's shards of one device): one thread per (variant, shard), adjacent lanes = the
// shards of one variant (interleaved k-mer tables, as search_init_kernel); shard s's traces are trace[s][m][trace_n],
// its records init[s * mv + i].
__global__ void __launch_bounds__(256)
search_init_1mm_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ vpacked,
                       const uint8_t *__restrict__ vvalid, size_t mv, uint32_t k, uint32_t wpq, uint32_t V,
                       const ulonglong2 *__restrict__ trace, uint32_t trace_n,
                       ulonglong2 *__restrict__ init) {
    const size_t t_ = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t_ >= mv * nshards) return;
    const size_t i = t_ / nshards, s = t_ - i * nshards;
    ulonglong2 rec;
    if (vvalid[i] == 0) {
        rec.x = INIT_INVALID;
        rec.y = 0;
    } else {
        const size_t q = i / V;
        const uint32_t v = (uint32_t)(i - q * V);
        const uint32_t pos = v ? (v - 1u) / 3u : ~0u;
        if (pos < trace_n) {
            const ulonglong2 t = trace[(s * (mv / V) + q) * trace_n + pos];
            rec.x = (t.x & COUNT_MASK) | ((uint64_t)pos << COUNT_BITS) | INIT_EXPLICIT;
            rec.y = t.y;
        } else {
            rec = start_record(shards[s], vpacked + i * wpq, k);
        }
    }
    init[s * mv + i] = rec;
}

// work[] of a counting launch
enum { WORK_STEPS = 0, WORK_OCC = 1, WORK_LINES = 2, WORK_KTAB = 3, WORK_PHASE0 = 4, WORK_PASSES = 10, WORK_HOPS = 11, WORK_SOLO = 12 };

// LONGK: k > 32, i.e. a query spans several packed words.  A template parameter because with the
// reload on the path -- however it is guarded at run time -- hipcc waits for vmcnt(0) at the top of
// every pass, which also waits for the start-up loads just issued by entering lanes.
// ROLE changes nothing but the kernel's name: 1 = the launches that fill a k-mer table at open time,
// so that a profile's per-kernel statistics of the query launches (ROLE 0) are not mixed with them.
template <bool COUNT_WORK, bool COUNTS_ONLY, bool LONGK, int ROLE>
__global__ void __launch_bounds__(64 * WG_WAVES, RSB_MIN_WGS_PER_CU)
search_lines_kernel(const shard_view *__restrict__ shards, uint32_t nshards, const uint64_t *__restrict__ packed,
                    const ulonglong2 *__restrict__ init, unsigned long long *__restrict__ next_query,
                    size_t Q, uint32_t k, uint32_t wpq,
                    uint64_t *__restrict__ out_lower, uint64_t *__restrict__ out_upper,
                    unsigned long long *__restrict__ work,
                    ulonglong2 *__restrict__ trace, uint32_t trace_n, uint32_t qchunk, uint32_t pairs) {
    __shared__ uint4 s_stage[WG_WAVES][64 * SLOT_U4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t side = lane >> 5;  // 0: lower-1 side, 1: upper side
    uint4 *stage = s_stage[wave];
    // LDS byte address of this wave's stage, in a scalar register (it goes to M0)
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)stage);
    // dword d of this lane's line: chunk d >> 2 at (d >> 2) ^ swz(lane) of the row
    // (lane & 7) * 1 KB + (lane >> 3) * 128 B; lanes l and l + 32 share k and the swizzle
    const uint32_t swz = lane & 7u;
    const lds_u32 *own_row = reinterpret_cast<const lds_u32 *>(stage + (lane & 7u) * 64u + (lane >> 3) * SLOT_U4);
#define MINE(d) (mine0 + (((((uint32_t)(d)) >> 2) ^ swz) << 2) + (((uint32_t)(d)) & 3u))

    unsigned long long w_steps = 0, w_occ = 0, w_lines = 0, w_hops = 0, w_ktab = 0;
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

    // A wave searches one shard at a time: it starts on shard blockIdx % nshards (workgroups are
    // dealt round-robin over the XCDs, so with 8 shards each XCD starts on its own) and draws
    // queries from that shard's pool until it is empty, then moves to the next shard that still has
    // queries; equal shards finish together and the tail is shared by all waves.
    uint32_t sid = blockIdx.x % nshards;
    for (uint32_t visited = 0; visited < nshards; ++visited, sid = (sid + 1u == nshards) ? 0u : sid + 1u) {
        const shard_view *sv = shards + sid;
        const char *lines_bytes = reinterpret_cast<const char *>(sv->lines);
        const uint32_t S = sv->sp.S;
        const double inv = sv->sp.inv;
        const uint32_t nlines = (uint32_t)sv->nlines;
        const bool ktab = view_uses_ktab(*sv, k);
        // symbol a table-started query continues with, and the packed word holding it
        const int j_table = ktab ? (int)(k - sv->ktab_depth) - 1 : (int)k - 2;
        const uint32_t w_table = j_table > 0 ? (uint32_t)j_table >> 5 : 0u;
        const ulonglong2 *init_s = init + (size_t)sid * Q;
        // results: lower[s][q] and upper[s][q], or (pairs) {lower, upper}[s][q] -- one 16-byte store
        // instead of two 8-byte ones: the stores of ended searches are scattered, each is a request of
        // its own, and requests are what this kernel is bound by
        uint64_t *out_lo = out_lower + (size_t)sid * Q * (pairs ? 2u : 1u);
        uint64_t *out_up = (COUNTS_ONLY || pairs) ? nullptr : out_upper + (size_t)sid * Q;
        unsigned long long *pool = next_query + (size_t)sid * POOL_STRIDE;
        // (a pool other waves have drained already is left with a load: an atomic on a line every wave of the launch
        // probes at the end is what the tail of a launch, and most of a small one, would wait for)
        if (visited != 0u && __hip_atomic_load(pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)Q) continue;
        // a traced search's record and a hit-list search's map are per shard: [s][Q][trace_n], [s][hit_map_words(Q)]
        ulonglong2 *trace_s = trace ? trace + (size_t)sid * Q * trace_n : nullptr;
        unsigned long long *hit_map = pairs == 2u ? reinterpret_cast<unsigned long long *>(out_upper) + (size_t)sid * hit_map_words(Q) : nullptr;
        // C[b]: lanes 0..3 of every wave keep C[1..4] and a lane picks its symbol's entry with two
        // ds_bpermute reads (selects out of scalar registers cost 15 VALU instructions a pass; a
        // dynamically indexed load would be a dependent global load in every pass)
        // (read with scalar loads and selected per lane: a vector load here would make hipcc wait for
        // vmcnt(0) at the table's first use in EVERY pass -- behind the start-up loads just issued for
        // entering queries, whose latency must overlap the line fetches instead)
        uint32_t ctab_lo, ctab_hi;
        {
            const uint64_t c1 = sv->C[1], c2 = sv->C[2], c3 = sv->C[3], c4 = sv->C[4];
            const uint32_t l3 = lane & 3u;
            const uint64_t cv = l3 == 0u ? c1 : l3 == 1u ? c2 : l3 == 2u ? c3 : c4;
            ctab_lo = (uint32_t)cv;
            ctab_hi = (uint32_t)(cv >> 32);
        }

        // Queries are handed out dynamically: a wave draws chunks of QCHUNK consecutive queries from
        // the shard's counter (one atomic per chunk) and gives the next one to whichever lane pair
        // has finished (ballot + popcount, no further atomics).
        const uint32_t QCHUNK = qchunk;
        uint64_t pool_next = 0, pool_end = 0;  // wave-uniform
        bool drained = false;                  // the shard's counter ran past Q
        size_t q = 0;          // the query this lane pair is stepping
        bool has_q = false;
        size_t nq = 0;         // the one it runs next, start record already prefetched
        bool has_n = false;
        ulonglong2 nrec = {0, 0};
        uint64_t nword = 0;
        int j = 0;
        uint64_t word = 0, lo = 0, hi = 0;
        // the lookup of the current step: `ready` = this side's Occ is in occ_hold; cont != 0 = the
        // position lies past this window's own pieces and the lookup goes on, in the next pass, in
        // the group's spill line (KIND_CHUNK: at dword cdw, base count cacc) or in far line cblk
        bool ready = false;
        uint64_t occ_hold = 0, cacc = 0;
        uint32_t cont = 0, cblk = 0, cdw = 0, co = 0, tries = 0;

        for (;;) {
            // ---- a pair whose query ended in the last pass takes up the one it had prefetched: it
            // steps in this very pass
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
                } else if (nrec.x & INIT_EXPLICIT) {  // a 1-mismatch variant resuming its k-mer's search, or a query of a length of its own (INIT_VAR)
                    lo = nrec.x & COUNT_MASK;
                    hi = nrec.y;
                    j = (int)((nrec.x >> COUNT_BITS) & 0xFFFFull) - ((nrec.x & INIT_VAR) ? 1 : 0);
                    word = nword;
                    // the shared suffix was already absent (query.cpp:35-37); an initInterval is not looked at (INIT_NOCHECK)
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
                    if (COUNT_WORK && !fallback && side == 0u) w_ktab += 1;
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
                if (lane == 0u) c = atomicAdd(pool, (unsigned long long)QCHUNK);
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
            // The two start-up loads of a query taken into reserve fly together with this pass's
            // line fetches.
            ulonglong2 rec = {0, 0};
            uint64_t first_word = 0;
            if (got_n) {
#ifdef RSB_NT_RECORDS  // tuning knob: start records and packed words are read once
                rec.x = __builtin_nontemporal_load(&init_s[nq].x);
                rec.y = __builtin_nontemporal_load(&init_s[nq].y);
                first_word = __builtin_nontemporal_load(&packed[nq * wpq + w_table]);
#else
                rec = init_s[nq];
                first_word = packed[nq * wpq + w_table];
#endif
            }
            const bool alive = has_q;
            const bool stepping = alive && !done;
            const bool fresh = stepping && !ready && cont == 0u;  // starts the lookup of a new LF step

            // ---- this lane's lookup: symbol, position -> window line
            uint32_t b = 1, line = 0, o = 0, w = 0;
            if (stepping) {
                if (LONGK) {
                    if (fresh && (j & 31) == 31) word = packed[q * wpq + ((uint32_t)j >> 5)];
                }
                b = (uint32_t)((word >> (2u * ((uint32_t)j & 31u))) & 3u) + 1u;
            }
            if (fresh) {
                // traced search (1-mismatch): the interval this query has when about to take symbol j
                if (trace && side == 0u && (uint32_t)j < trace_n) trace_s[q * trace_n + (uint32_t)j] = make_ulonglong2(lo, hi);
                // Occ(b, -1) = 0: lower - 1 at lower == 0, and upper itself after a step that found no b
                // at the top of the BWT (upper = 0 + 0 - 1 wraps; the reference carries on the same way
                // and reports the empty interval one step later: query.cpp:11-15,35, rlebwt.cpp:269)
                const uint64_t p = side ? hi : lo - 1ull;
                if (p == ~0ull) {
                    occ_hold = 0;
                    ready = true;
                } else {
                    uint32_t pin;
                    w = fast_window(p, S, inv, pin);
                    line = w + (w >> GROUP_SHIFT);
                    o = pin + 1u;
                    if (line >= nlines) line = 0;  // never for p < n; keeps a bad position from faulting
                    if (COUNT_WORK) w_occ += 1;
                }
            }
            const bool looking = stepping && !ready;  // fetches this pass: a new lookup or a continuation
            // C[b], with every lane active: a ds_bpermute returns 0 from a masked-off source lane
            const uint64_t pb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)((b - 1u) << 2), (int)ctab_hi) << 32) |
                                (uint32_t)__builtin_amdgcn_ds_bpermute((int)((b - 1u) << 2), (int)ctab_lo);
            STAMP(0)  // pass set-up: symbol, position, line

            // ---- fetch: one request per distinct line.  Lanes with nothing to look up ask for
            // nothing, and the upper side of a query whose two positions fall in the same line reads
            // the lower side's row instead of fetching the line again.
            uint32_t want = looking ? (cont ? cblk : line) : ~0u;
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(want, want, false, false);  // full exec
                const uint32_t other_want = side ? sw[0] : sw[1];
                if (side != 0u && want == other_want) want = ~0u;
            }
            const bool shared_row = looking && want == ~0u;
            if (COUNT_WORK && want != ~0u) {
                if (cont) w_hops += 1;
                else w_lines += 1;
            }
            glds_fetch(lines_bytes, want, lane, stage_lds);
            STAMP(1)  // issue of the line loads
            glds_wait();
            STAMP(2)  // wait for the lines
            const lds_u32 *mine0 = shared_row ? own_row - 4 * 32 : own_row;  // row of lane - 32

            // ---- Occ(b, p) out of this lane's staged line.  RLEBWT::getOcc, src/bwt/rlebwt.cpp:268-301.
            const sym_tab stab = make_sym_tab(b);  // v_perm_b32 table of the symbol (rank_device.h)
            bool do_scan = false;
            uint64_t base = 0;
            uint32_t dw = HDR_DWORDS, rem = 0;
            if (looking) {
                if (cont != KIND_CHUNK) {
                    // a window line (or the far line that continues it): the count word of symbol b and
                    // the quarter starts
                    const uint32_t oe = cont ? co : o;
                    const uint2 cw = *reinterpret_cast<const lds_u2 *>(MINE(2u * (b - 1u)));
                    const uint4 h0 = *reinterpret_cast<const lds_u4 *>(MINE(0));
                    const uint64_t cnt = ((uint64_t)(cw.y & 0xFFu) << 32) | cw.x;
                    const uint32_t m0 = h0.y >> 8, m1 = h0.w >> 8;
                    const uint32_t s1 = m0 & 0x3FFu, s2 = (m0 >> 10) & 0x7FFu;
                    const uint32_t s3 = s2 + (m1 & 0x3FFu), span = s3 + ((m1 >> 10) & 0x3FFu);
                    const uint32_t kind = (m1 >> 20) & 3u;
                    if (oe <= span) {
                        const uint32_t cq = (oe > s1 ? 1u : 0u) + (oe > s2 ? 1u : 0u) + (oe > s3 ? 1u : 0u);
                        const uint32_t start = cq == 0u ? 0u : cq == 1u ? s1 : cq == 2u ? s2 : s3;
                        // what quarters 0 and 1 hold of b
                        const uint32_t hm = *MINE(5u + 2u * ((b - 1u) >> 1)) >> 8;
                        const uint32_t hb = (hm >> (11u * ((b - 1u) & 1u))) & 0x7FFu;
                        // an odd quarter also needs the quarter before it: added up 4 runs per dot4
                        const uint32_t qd = HDR_DWORDS + 6u * (cq & 2u);
                        const uint2 x0 = *reinterpret_cast<const lds_u2 *>(MINE(qd));
                        const uint2 x1 = *reinterpret_cast<const lds_u2 *>(MINE(qd + 2u));
                        const uint2 x2 = *reinterpret_cast<const lds_u2 *>(MINE(qd + 4u));
                        const uint32_t e6[6] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y};
                        const uint32_t m = matched24_tab(e6, stab);
                        const uint32_t bb = __builtin_amdgcn_perm(0u, b, 0u);  // (the timing experiment below)
                        (void)bb;
#ifdef RSB_EXPERIMENT_EXTRA_DOT4  // timing experiment (answers unchanged): the quarter sum done twice
                        {
                            uint32_t m2 = dword_matched(x0.x ^ 1u, bb, 1u);
                            m2 = dword_matched(x0.y ^ 1u, bb, m2);
                            m2 = dword_matched(x1.x ^ 1u, bb, m2);
                            m2 = dword_matched(x1.y ^ 1u, bb, m2);
                            m2 = dword_matched(x2.x ^ 1u, bb, m2);
                            m2 = dword_matched(x2.y ^ 1u, bb, m2);
                            asm volatile("" ::"v"(m2));
                        }
#endif
                        base = cnt + (cq >= 2u ? hb : 0u) + ((cq & 1u) ? m : 0u);
                        dw = HDR_DWORDS + 6u * cq;
                        rem = oe - start;
                        do_scan = true;
                    } else if (kind == KIND_FAR) {
                        cblk = *MINE(LINE_DWORDS - 1u);
                        if (cblk >= nlines) cblk = 0;  // never for a built index
                        cont = KIND_FAR;
                        co = oe - span;
                    } else if (kind == KIND_CHUNK && cont == 0u) {
                        const uint32_t m2 = *MINE(5) >> 8, m3 = *MINE(7) >> 8;
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
                    const uint2 hd = *reinterpret_cast<const lds_u2 *>(MINE(cdw));
                    const uint32_t hw = (b <= 2u) ? hd.x : hd.y;
                    const uint32_t tot = (hw >> (12u * ((b - 1u) & 1u))) & 0xFFFu;
                    base = cacc + tot;
                    dw = cdw + 2u;
                    rem = co;
                    do_scan = true;
                }
                // a window has at most 33 lines: the bound only guards against a corrupt chain, so that
                // every wave drains
                if (!do_scan && ++tries > 72u) do_scan = true;
            }
            // the rank within the (at most) 24 pieces at dword dw (rlebwt.cpp:281-298; rank_device.h, rank24: dword totals
            // by v_dot4, the dword holding the position run by run); lanes with nothing to scan take part with rem = 0
            {
                const uint2 y0 = *reinterpret_cast<const lds_u2 *>(MINE(dw & 31u));
                const uint2 y1 = *reinterpret_cast<const lds_u2 *>(MINE((dw + 2u) & 31u));
                const uint2 y2 = *reinterpret_cast<const lds_u2 *>(MINE((dw + 4u) & 31u));
                const uint32_t r6[6] = {y0.x, y0.y, y1.x, y1.y, y2.x, y2.y};
                const uint32_t sc = rank24(r6, stab, b, rem);
#ifdef RSB_EXPERIMENT_EXTRA_SCAN2  // timing experiment (answers unchanged): 8 more pieces scanned
                {
                    const uint32_t sc2 = runs_scan<2>(r6 + 2, b, rem + 1u);
                    asm volatile("" ::"v"(sc2));
                }
#endif
                if (do_scan) {
                    occ_hold = base + sc;
#ifdef RSB_FAULT_INJECT_WILD_OCC  // fault-injection build (tools/README.md): answers are WRONG by design -- one
                    // lookup in 16 returns a wild count, so lower / upper leave [0, n) and every fetch index
                    // has to survive a position the index does not hold.  Wild within what ANY line contents
                    // can give: a count is a 40-bit word plus a few 12-bit fields, < 2^41 -- which is what
                    // leaves bit 63 of occ_hold free to carry `ready` in the exchange below (a first version of
                    // this injection added 64-bit values, the partner lane read the flag out of a stale count,
                    // the two sides of a pair fell out of step and their wave never drained)
                    if ((((uint32_t)q + (uint32_t)j) * 2654435761u >> 28) == 0u)
                        occ_hold = (occ_hold + (((uint64_t)q * 0x9E3779B97F4A7C15ull) >> (23u + (((uint32_t)q + (uint32_t)j) & 31u)))) & ((1ull << 41) - 1ull);
#endif
                    ready = true;
                    cont = 0;
                    tries = 0;
                }
            }
            STAMP(3)  // rank out of LDS
            // ---- the two sides of a query trade results; updateInterval (query.cpp:11-15)
            // (bit 63 travels as the ready flag: a count is < 2^41 whatever the line held)
            const uint32_t occ_hi = (uint32_t)(occ_hold >> 32) | (ready ? 0x80000000u : 0u);
            const auto sw_lo = __builtin_amdgcn_permlane32_swap((uint32_t)occ_hold, (uint32_t)occ_hold, false, false);
            const auto sw_hi = __builtin_amdgcn_permlane32_swap(occ_hi, occ_hi, false, false);
            const uint32_t other_hi = side ? sw_hi[0] : sw_hi[1];
            const uint64_t other = ((uint64_t)(other_hi & 0x7FFFFFFFu) << 32) | (side ? sw_lo[0] : sw_lo[1]);
            STAMP(4)  // exchange
            if (stepping && ready && (other_hi & 0x80000000u)) {
                const uint64_t occL = side ? other : occ_hold;
                const uint64_t occU = side ? occ_hold : other;
                if (COUNT_WORK && side == 0u) w_steps += 1;
                lo = pb + occL;
                hi = pb + occU - 1ull;
                --j;
                done = (lo > hi) || (j < 0);  // query.cpp:35-37
                ready = false;
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
                        trace_s[q * trace_n + (uint32_t)jj] = make_ulonglong2(lo, hi);
                }
                if (side == 0u) {
                    if (COUNTS_ONLY) {
                        if (hi >= lo) out_lo[q] = hi - lo + 1ull;  // service.cpp:304; the array is zeroed before the launch: most searches end empty and store nothing
                    } else if (pairs == 2u) {
                        // sparse results (1-mismatch hit list): only a search that ends on an interval leaves
                        // anything -- its {lower, upper} at its own place and its bit in the map at out_upper (an
                        // atomic nobody waits for; a counter handing out list positions would stall the wave for a
                        // round trip per hit and serialise on one address).  compact_hits orders them afterwards.
                        if (lo <= hi) {
                            reinterpret_cast<ulonglong2 *>(out_lo)[q] = make_ulonglong2(lo, hi);
                            atomicOr(hit_map + (q >> 6), 1ull << (q & 63u));
                        }
                    } else if (pairs) {
#ifdef RSB_NT_RESULTS  // tuning knob: results are written once and read by another kernel / the host
                        __builtin_nontemporal_store(lo, &out_lo[2 * q]);
                        __builtin_nontemporal_store(hi, &out_lo[2 * q + 1]);
#else
                        reinterpret_cast<ulonglong2 *>(out_lo)[q] = make_ulonglong2(lo, hi);
#endif
                    } else {
                        out_lo[q] = lo;
                        out_up[q] = hi;
                    }
                }
                has_q = false;
            }
            STAMP(5)  // update, start-up decode, result stores
            if (COUNT_WORK) ++passes;
        }
    }
    if (COUNT_WORK) {
        if (lane == 0u) {
            for (int i = 0; i < 6; ++i) atomicAdd(&work[WORK_PHASE0 + i], ph[i]);
            atomicAdd(&work[WORK_PASSES], passes);
        }
        if (w_steps) atomicAdd(&work[WORK_STEPS], w_steps);
        if (w_occ) atomicAdd(&work[WORK_OCC], w_occ);
        if (w_lines) atomicAdd(&work[WORK_LINES], w_lines);
        if (w_ktab) atomicAdd(&work[WORK_KTAB], w_ktab);
        if (w_hops) atomicAdd(&work[WORK_HOPS], w_hops);
    }
#undef MINE
#undef STAMP
}

}  // namespace rsb
#include "search_solo.h"
namespace rsb {

// Which kernel a plain search runs on.  RSBWT_SEARCH_KERNEL = pair | solo | auto (default).
static int search_kernel_choice() {
    static const int choice = [] {
        const char *e = getenv("RSBWT_SEARCH_KERNEL");
        if (e && !strcmp(e, "pair")) return 0;
        if (e && !strcmp(e, "solo")) return 1;
        return 2;
    }();
    return choice;
}

template <bool CW, bool CO>
static void launch_solo(int grid, hipStream_t stream, const shard_view *shards, uint32_t nshards, const uint64_t *pk,
                        const ulonglong2 *init, const uint8_t *valid, unsigned long long *ctr, size_t Q, uint32_t k, uint32_t wpq,
                        uint64_t *lo, uint64_t *up, unsigned long long *work, ulonglong2 *trace, uint32_t trace_n,
                        uint32_t pairs, bool fused) {
    uint32_t qchunk = 1024;
    while (qchunk > 64u && (size_t)qchunk * (size_t)grid * WG_WAVES * 4u > Q * nshards) qchunk >>= 1;
    // results staged in LDS and stored a whole 1 KB group at a time (search_solo.h, STAGED RESULTS): 8 KB of dynamic
    // LDS per workgroup on top of the 32 KB of line slots -- four workgroups then fill a CU's 160 KB.
    // RSBWT_NO_STAGED_RESULTS: A/B knob (tools/README.md); also what leaves LDS to kernels running beside the search.
    static const bool no_staged = getenv("RSBWT_NO_STAGED_RESULTS") != nullptr;
    const bool staged = !CO && pairs != 2u && !no_staged;
    const uint32_t pa = pairs | (staged ? SOLO_STAGED_RESULTS : 0u);
    const size_t dyn = staged ? SOLO_RESULTS_LDS : 0u;
    if (fused)  // (one shard, k <= 32, a k-mer table, no trace: the kernel makes its own start records)
        hipLaunchKernelGGL((search_solo_kernel<CW, CO, false, true>), dim3(grid), dim3(64 * WG_WAVES), dyn, stream, shards, nshards,
                           pk, init, valid, ctr, Q, k, wpq, lo, up, work, trace, trace_n, qchunk, pa);
    else if (wpq > 1)
        hipLaunchKernelGGL((search_solo_kernel<CW, CO, true>), dim3(grid), dim3(64 * WG_WAVES), dyn, stream, shards, nshards,
                           pk, init, valid, ctr, Q, k, wpq, lo, up, work, trace, trace_n, qchunk, pa);
    else
        hipLaunchKernelGGL((search_solo_kernel<CW, CO, false>), dim3(grid), dim3(64 * WG_WAVES), dyn, stream, shards, nshards,
                           pk, init, valid, ctr, Q, k, wpq, lo, up, work, trace, trace_n, qchunk, pa);
}

template <bool CW, bool CO>
static void launch_k(int grid, hipStream_t stream, const shard_view *shards, uint32_t nshards, const uint64_t *pk,
                     const ulonglong2 *init, unsigned long long *ctr, size_t Q, uint32_t k, uint32_t wpq,
                     uint64_t *lo, uint64_t *up, unsigned long long *work, ulonglong2 *trace, uint32_t trace_n,
                     uint32_t pairs) {
    // queries per draw from the pool: at least ~4 draws per wave, so that a batch of a few
    // thousand queries (a service micro-batch, the k-mers of a 1-mismatch slice) still occupies
    // every wave launched instead of the first few
    uint32_t qchunk = 1024;
    while (qchunk > 32u && (size_t)qchunk * (size_t)grid * WG_WAVES * 4u > Q * nshards) qchunk >>= 1;
    if (wpq > 1)
        hipLaunchKernelGGL((search_lines_kernel<CW, CO, true, 0>), dim3(grid), dim3(64 * WG_WAVES), 0, stream, shards, nshards,
                           pk, init, ctr, Q, k, wpq, lo, up, work, trace, trace_n, qchunk, pairs);
    else
        hipLaunchKernelGGL((search_lines_kernel<CW, CO, false, 0>), dim3(grid), dim3(64 * WG_WAVES), 0, stream, shards, nshards,
                           pk, init, ctr, Q, k, wpq, lo, up, work, trace, trace_n, qchunk, pairs);
}

static void launch_init(const shard_view *d_shards, uint32_t nshards, const uint64_t *pk, const uint8_t *vd, size_t Q, uint32_t k,
                        uint32_t wpq, ulonglong2 *init, hipStream_t stream) {
    static const bool untiled = getenv("RSBWT_INIT_UNTILED") != nullptr;  // A/B knob (tools/README.md)
    static const bool one_per_thread = getenv("RSBWT_INIT_ONE_PER_THREAD") != nullptr;  // A/B knob (tools/README.md)
    if (nshards >= 2u && nshards <= 256u && !untiled && wpq == 1u && !one_per_thread) {
        const uint32_t V = 256u / nshards;
        hipLaunchKernelGGL(search_init_tiled4_kernel, dim3((unsigned)((Q + (size_t)V * INIT_U - 1) / ((size_t)V * INIT_U))), dim3(256), 0, stream,
                           d_shards, nshards, pk, vd, Q, k, V, init);
    } else if (nshards >= 2u && nshards <= 256u && !untiled) {
        const uint32_t V = 256u / nshards;
        hipLaunchKernelGGL(search_init_tiled_kernel, dim3((unsigned)((Q + V - 1) / V)), dim3(256), 0, stream, d_shards, nshards, pk, vd, Q,
                           k, wpq, V, init);
    } else {
        const size_t nrec = Q * nshards;
        hipLaunchKernelGGL(search_init_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, stream, d_shards, nshards, pk, vd, Q, k,
                           wpq, init);
    }
}

hipError_t launch_search_init_var(const shard_view *d_shards, uint32_t nshards, const void *d_packed, const void *d_valid,
                                  const void *d_len, size_t Q, uint32_t wpq, void *d_init, hipStream_t stream) {
    if (Q == 0 || nshards == 0) return hipSuccess;
    const size_t nrec = Q * nshards;
    hipLaunchKernelGGL(search_init_var_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, stream, d_shards, nshards,
                       (const uint64_t *)d_packed, (const uint8_t *)d_valid, (const uint32_t *)d_len, Q, wpq, (ulonglong2 *)d_init);
    return hipGetLastError();
}

hipError_t launch_search_init(const shard_view *d_shards, uint32_t nshards, const void *d_packed, const void *d_valid, size_t Q,
                              uint32_t k, void *d_init, hipStream_t stream) {
    if (Q == 0 || nshards == 0) return hipSuccess;
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    launch_init(d_shards, nshards, (const uint64_t *)d_packed, (const uint8_t *)d_valid, Q, k, wpq, (ulonglong2 *)d_init, stream);
    return hipGetLastError();
}

hipError_t launch_search(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_packed,
                         const void *d_valid, size_t Q, uint32_t k, void *d_lower, void *d_upper, bool counts_only,
                         unsigned long long *d_work, int num_cus, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1,
                         const search_extra *extra) {
    if (Q == 0 || nshards == 0) return hipSuccess;
    if (extra && extra->table_build && nshards != 1) return hipErrorInvalidValue;  // a table is one shard's
    // traced / resumed / hit-list searches of several shards: traces [s][Q][trace_n], sparse results [s][Q], hit maps
    // [s][hit_map_words(Q)]
    const bool hit_list = extra && extra->d_hit_bits && !counts_only;
    const uint32_t pairs = hit_list ? 2u : (extra && extra->pairs && !counts_only) ? 1u : 0u;
    ulonglong2 *trace = extra ? (ulonglong2 *)extra->d_trace_out : nullptr;
    const uint32_t trace_n = extra ? extra->trace_n : 0u;
    const uint32_t wpq = (k + 31u) / 32u ? (k + 31u) / 32u : 1u;
    // one lane per search (search_solo.h) where intervals are narrow for most of a search: shards whose
    // k-mer tables are deep (extra->narrow: what is left of a hit are steps inside one window); behind
    // a shallow table the first steps are wide, where pairs take one pass and a lone lane two
    // (and only when the batch fills every lane of the launch: below that nothing is saturated and the
    // pairs answer sooner -- a lone request of the service loop takes half the passes).
    // Until round 5 several shards per launch stayed on lane pairs ("at the request ceiling already"): the
    // launch is bound by the instructions a SIMD issues, not by requests (+ 12.6 % VALU = + 5.7 %
    // time, the same build), a pair spends a whole lane on `upper` where 97 % of the steps find it in the line
    // `lower - 1` staged, and the lone lanes run the headline's 8 x 20 GB shards in 19.05 ms against 20.04
    const bool table_build = extra && extra->table_build;
    const int choice = search_kernel_choice();
    size_t g = 0;
    // Workgroups per CU: LDS admits 5 (20 waves), but 4 are as fast (the request path, not the
    // number of lookups in flight, is what saturates) and leave 32 KB of LDS and wave slots per CU
    // to kernels that run beside the search -- RCCL's, when the previous batch's intervals are
    // gathered at N > 1.  RSBWT_WAVE_WGS_PER_CU overrides.
    static const int wgs_per_cu = [] {
        const char *e = getenv("RSBWT_WAVE_WGS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : RSB_MIN_WGS_PER_CU;
    }();
    // RSBWT_SEARCH_SPARE_WGS = n: n workgroups fewer than the chip holds.  A search launch is persistent (its workgroups
    // stay until the batch is done) and fills every CU's registers (4 waves x 128 VGPRs per SIMD) and LDS: a kernel
    // that should run BESIDE it -- RCCL's, gathering the previous batch at N > 1 -- finds room only on CUs a workgroup
    // short.  bench.py sets it for its N > 1 ranks (32: a workgroup slot on 32 CUs for the collective's channels).
    static const size_t spare_wgs = [] {
        const char *e = getenv("RSBWT_SEARCH_SPARE_WGS");
        const long v = e ? atol(e) : 0;
        return (size_t)(v > 0 ? v : 0);
    }();
    const size_t cap_all = (size_t)num_cus * (size_t)wgs_per_cu;
    const size_t cap = cap_all > 2 * spare_wgs ? cap_all - spare_wgs : cap_all;
    // (the variants of a 1-mismatch search that resume from a trace start on their k-mer's narrow interval and live
    // 2.4 steps: what bounds their launch is how fast searches are taken up, and a wave of lone lanes takes up 64 per
    // pass where pairs take 32 -- any number of shards: 25.0 -> 23.5 ms per batch of 4e5 31-mers x 8 shards)
    const bool resumed = extra && extra->d_trace_in;
    const bool solo = !table_build && (choice == 1 || (choice == 2 && Q * nshards >= cap * WG_WAVES * 64u &&
                                                       (resumed || (extra && extra->narrow))));
    // 32 (pairs) or 64 (solo) searches per wave, 4 waves per workgroup
    const size_t per_wg = (solo ? 64u : 32u) * WG_WAVES;
    g = (Q * nshards + per_wg - 1) / per_wg;
    if (g > cap) g = cap;
    const int grid = (int)g;
    const uint64_t *pk = (const uint64_t *)d_packed;
    const uint8_t *vd = (const uint8_t *)d_valid;
    uint64_t *lo = (uint64_t *)d_lower, *up = hit_list ? (uint64_t *)extra->d_hit_bits : (uint64_t *)d_upper;
    // start records of this batch + the shards' query counters: scratch of this launch sequence
    // alone, so concurrent calls do not share state
    const size_t nrec = Q * nshards;
    // a full batch on one shard behind a deep k-mer table (extra->narrow says there is one and k reaches it): the
    // one-lane kernel makes its own start records (search_solo.h, FUSED) -- no start-record launch, no records
    static const bool no_fused_start = getenv("RSBWT_NO_FUSED_START") != nullptr;  // A/B knob (tools/README.md)
    const bool fused = solo && nshards == 1 && extra && extra->narrow && !extra->d_init && !resumed && !trace && wpq == 1 && !no_fused_start;
    const bool prepared = (extra && extra->d_init) || fused;  // the start records exist already / are not needed: only the counters are scratch
    scratch_cache::lease mem;
    hipError_t e = scratch.take((prepared ? 0 : nrec * sizeof(ulonglong2)) + nshards * POOL_STRIDE * sizeof(unsigned long long), stream, &mem);
    if (e != hipSuccess) return e;
    if (counts_only) {  // counts: only the searches that find something store theirs
        e = hipMemsetAsync(d_lower, 0, nrec * sizeof(uint64_t), stream);
        if (e != hipSuccess) {
            scratch.give(mem, stream);
            return e;
        }
    }
    ulonglong2 *init = fused ? nullptr : prepared ? (ulonglong2 *)extra->d_init : (ulonglong2 *)mem.p;
    unsigned long long *ctr = prepared ? (unsigned long long *)mem.p : (unsigned long long *)(init + nrec);
    e = hipMemsetAsync(ctr, 0, nshards * POOL_STRIDE * sizeof(unsigned long long), stream);
    if (e != hipSuccess) {
        scratch.give(mem, stream);
        return e;
    }
    const unsigned ig = (unsigned)((nrec + 255) / 256);
    if (prepared) {
        // nothing to compute
    } else if (extra && extra->d_trace_in)
        hipLaunchKernelGGL(search_init_1mm_kernel, dim3(ig), dim3(256), 0, stream, d_shards, nshards, pk, vd, Q, k, wpq,
                           extra->variants, (const ulonglong2 *)extra->d_trace_in, trace_n, init);
    else
        launch_init(d_shards, nshards, pk, vd, Q, k, wpq, init, stream);
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (extra && extra->table_build && !d_work && !counts_only && wpq == 1) {
        uint32_t qchunk = 1024;
        while (qchunk > 32u && (size_t)qchunk * (size_t)grid * WG_WAVES * 4u > Q * nshards) qchunk >>= 1;
        hipLaunchKernelGGL((search_lines_kernel<false, false, false, 1>), dim3(grid), dim3(64 * WG_WAVES), 0, stream, d_shards,
                           nshards, pk, init, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, qchunk, 0u);
    } else if (solo) {
        if (d_work) {
            if (counts_only) launch_solo<true, true>(grid, stream, d_shards, nshards, pk, init, vd, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs, fused);
            else launch_solo<true, false>(grid, stream, d_shards, nshards, pk, init, vd, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs, fused);
        } else {
            if (counts_only) launch_solo<false, true>(grid, stream, d_shards, nshards, pk, init, vd, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs, fused);
            else launch_solo<false, false>(grid, stream, d_shards, nshards, pk, init, vd, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs, fused);
        }
    } else if (d_work) {
        if (counts_only) launch_k<true, true>(grid, stream, d_shards, nshards, pk, init, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs);
        else launch_k<true, false>(grid, stream, d_shards, nshards, pk, init, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs);
    } else {
        if (counts_only) launch_k<false, true>(grid, stream, d_shards, nshards, pk, init, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs);
        else launch_k<false, false>(grid, stream, d_shards, nshards, pk, init, ctr, Q, k, wpq, lo, up, d_work, trace, trace_n, pairs);
    }
    e = hipGetLastError();
    if (ev1) (void)hipEventRecord(ev1, stream);
    scratch.give(mem, stream);
    return e;
}

// The worklists of a set's 1-mismatch search (mm1_worklist.hip): every record a live search, the lists' lengths known
// to the device only -- the grid is what the GPU holds at once, the pools end where the counts say.
hipError_t launch_search_worklist(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_packed,
                                  const void *d_valid, size_t m, uint32_t tn, const void *d_worklists, const void *d_counts, size_t wl_cap,
                                  uint32_t k, void *d_sparse, void *d_hit_bits, unsigned long long *d_work, int num_cus,
                                  hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, const void *d_pre) {
    if (nshards == 0 || m == 0) return hipSuccess;
    if (k > 32u || tn == 0 || tn >= k) return hipErrorInvalidValue;
    static const int wgs_per_cu = [] {
        const char *e = getenv("RSBWT_WAVE_WGS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : RSB_MIN_WGS_PER_CU;
    }();
    const size_t implicit = m * 3u * (size_t)(k - tn), mv = m * (3u * (size_t)k + 1u);
    size_t g = ((implicit + wl_cap) * nshards + 64u * WG_WAVES - 1) / (64u * WG_WAVES);
    // RSBWT_SEARCH_SPARE_WGS = n: n workgroups fewer than the chip holds.  A search launch is persistent (its workgroups
    // stay until the batch is done) and fills every CU's registers (4 waves x 128 VGPRs per SIMD) and LDS: a kernel
    // that should run BESIDE it -- RCCL's, gathering the previous batch at N > 1 -- finds room only on CUs a workgroup
    // short.  bench.py sets it for its N > 1 ranks (32: a workgroup slot on 32 CUs for the collective's channels).
    static const size_t spare_wgs = [] {
        const char *e = getenv("RSBWT_SEARCH_SPARE_WGS");
        const long v = e ? atol(e) : 0;
        return (size_t)(v > 0 ? v : 0);
    }();
    const size_t cap_all = (size_t)num_cus * (size_t)wgs_per_cu;
    const size_t cap = cap_all > 2 * spare_wgs ? cap_all - spare_wgs : cap_all;
    if (g > cap) g = cap;
    scratch_cache::lease mem;
    hipError_t e = scratch.take(nshards * POOL_STRIDE * sizeof(unsigned long long), stream, &mem);
    if (e != hipSuccess) return e;
    unsigned long long *ctr = (unsigned long long *)mem.p;
    e = hipMemsetAsync(ctr, 0, nshards * POOL_STRIDE * sizeof(unsigned long long), stream);
    if (e != hipSuccess) {
        scratch.give(mem, stream);
        return e;
    }
    if (ev0) (void)hipEventRecord(ev0, stream);
    // (the table entries of the implicit items read ahead for all shards: inside this launch's event pair, so that the
    // time the library reports for the worklist search includes it)
    if (d_pre) {
        e = launch_wl_table_entries(d_shards, nshards, d_packed, m, k, tn, const_cast<void *>(d_pre), stream);
        if (e != hipSuccess) {
            scratch.give(mem, stream);
            return e;
        }
    }
    uint32_t qchunk = 1024;
    while (qchunk > 64u && (size_t)qchunk * g * WG_WAVES * 4u > implicit * nshards) qchunk >>= 1;
    if (d_work)
        hipLaunchKernelGGL((search_solo_kernel<true, false, false, false, true>), dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, d_shards,
                           nshards, (const uint64_t *)d_packed, (const ulonglong2 *)d_worklists, (const uint8_t *)d_valid, ctr, mv, k, 1u,
                           (uint64_t *)d_sparse, (uint64_t *)d_hit_bits, d_work, (ulonglong2 *)const_cast<void *>(d_pre), tn, qchunk, 2u,
                           (const unsigned long long *)d_counts, wl_cap, implicit);
    else
        hipLaunchKernelGGL((search_solo_kernel<false, false, false, false, true>), dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream, d_shards,
                           nshards, (const uint64_t *)d_packed, (const ulonglong2 *)d_worklists, (const uint8_t *)d_valid, ctr, mv, k, 1u,
                           (uint64_t *)d_sparse, (uint64_t *)d_hit_bits, d_work, (ulonglong2 *)const_cast<void *>(d_pre), tn, qchunk, 2u,
                           (const unsigned long long *)d_counts, wl_cap, implicit);
    e = hipGetLastError();
    if (ev1) (void)hipEventRecord(ev1, stream);
    scratch.give(mem, stream);
    return e;
}

// The k-mers' own searches and the step of the three substitutions of every position left of the tables' reach, in one
// walk (search_solo.h, WALK): m k-mers per shard, worklists / counts / sparse results as mm1_worklist.hip lays them out.
hipError_t launch_search_walk(scratch_cache &scratch, const shard_view *d_shards, uint32_t nshards, const void *d_packed,
                              const void *d_valid, size_t m, uint32_t tn, void *d_worklists, void *d_counts, size_t wl_cap, uint32_t k,
                              void *d_sparse, void *d_hit_bits, unsigned long long *d_work, int num_cus, hipStream_t stream,
                              hipEvent_t ev0, hipEvent_t ev1) {
    if (nshards == 0 || m == 0) return hipSuccess;
    if (k > 32u || tn == 0 || tn >= k) return hipErrorInvalidValue;
    static const int wgs_per_cu = [] {
        const char *e = getenv("RSBWT_WALK1MM_WGS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return v > 0 && v <= RSB_WALK1MM_WGS_PER_CU ? v : RSB_WALK1MM_WGS_PER_CU;
    }();
    size_t g = (m * nshards + 64u * WG_WAVES - 1) / (64u * WG_WAVES);
    // RSBWT_SEARCH_SPARE_WGS = n: n workgroups fewer than the chip holds.  A search launch is persistent (its workgroups
    // stay until the batch is done) and fills every CU's registers (4 waves x 128 VGPRs per SIMD) and LDS: a kernel
    // that should run BESIDE it -- RCCL's, gathering the previous batch at N > 1 -- finds room only on CUs a workgroup
    // short.  bench.py sets it for its N > 1 ranks (32: a workgroup slot on 32 CUs for the collective's channels).
    static const size_t spare_wgs = [] {
        const char *e = getenv("RSBWT_SEARCH_SPARE_WGS");
        const long v = e ? atol(e) : 0;
        return (size_t)(v > 0 ? v : 0);
    }();
    const size_t cap_all = (size_t)num_cus * (size_t)wgs_per_cu;
    const size_t cap = cap_all > 2 * spare_wgs ? cap_all - spare_wgs : cap_all;
    if (g > cap) g = cap;
    scratch_cache::lease mem;
    hipError_t e = scratch.take(nshards * POOL_STRIDE * sizeof(unsigned long long), stream, &mem);
    if (e != hipSuccess) return e;
    unsigned long long *ctr = (unsigned long long *)mem.p;
    e = hipMemsetAsync(ctr, 0, nshards * POOL_STRIDE * sizeof(unsigned long long), stream);
    if (e != hipSuccess) {
        scratch.give(mem, stream);
        return e;
    }
    if (ev0) (void)hipEventRecord(ev0, stream);
    uint32_t qchunk = 1024;
    while (qchunk > 64u && (size_t)qchunk * g * WG_WAVES * 4u > m * nshards) qchunk >>= 1;
    if (d_work)
        hipLaunchKernelGGL((search_solo_kernel<true, false, false, true, false, true>), dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream,
                           d_shards, nshards, (const uint64_t *)d_packed, (const ulonglong2 *)d_worklists, (const uint8_t *)d_valid, ctr, m, k, 1u,
                           (uint64_t *)d_sparse, (uint64_t *)d_hit_bits, d_work, (ulonglong2 *)nullptr, tn, qchunk, 2u,
                           (const unsigned long long *)d_counts, wl_cap, (size_t)0);
    else
        hipLaunchKernelGGL((search_solo_kernel<false, false, false, true, false, true>), dim3((unsigned)g), dim3(64 * WG_WAVES), 0, stream,
                           d_shards, nshards, (const uint64_t *)d_packed, (const ulonglong2 *)d_worklists, (const uint8_t *)d_valid, ctr, m, k, 1u,
                           (uint64_t *)d_sparse, (uint64_t *)d_hit_bits, d_work, (ulonglong2 *)nullptr, tn, qchunk, 2u,
                           (const unsigned long long *)d_counts, wl_cap, (size_t)0);
    e = hipGetLastError();
    if (ev1) (void)hipEventRecord(ev1, stream);
    scratch.give(mem, stream);
    return e;
}

// Entries per k-mer of a traced search = the positions left of the k-mer table's reach (0: the
// 1-mismatch search has nothing to share: no table, or k within it)
uint32_t trace_entries(const shard_view &ix, uint32_t k) {
    const bool ktab = ix.ktab != nullptr && ix.ktab_depth >= 2 && k >= ix.ktab_depth;
    if (!ktab || k <= ix.ktab_depth || k - ix.ktab_depth > 0xFFFFu) return 0;
    return k - ix.ktab_depth;
}

}  // namespace rsb
